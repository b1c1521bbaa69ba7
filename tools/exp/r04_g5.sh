python tools/exp/r04_scene_setup_cost.py > gpurun_out/r04_setup_cost.txt 2>&1
python tools/exp/r04_gi_counters.py 640 360 > gpurun_out/r04_gi_counters.txt 2>&1
python tools/exp/r04_gi_counters.py 1920 1080 >> gpurun_out/r04_gi_counters.txt 2>&1
tools/exp/r04_perwave_waves.sh run > gpurun_out/r04_perwave.txt 2>&1
