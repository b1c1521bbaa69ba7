set -e
python -m pytest tests/test_gpu_overlap.py tests/test_gpu_gather.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_t1.txt
python bench.py --steps 200 --warmup 20 > gpurun_out/r04_c2_a.json 2> gpurun_out/r04_c2_a.err
python bench.py --steps 200 --warmup 20 --pretend-ranks 8 --no-cpu-baseline > gpurun_out/r04_c2_pr8.json 2> gpurun_out/r04_c2_pr8.err
python bench.py --steps 200 --warmup 20 --pretend-ranks 8 --no-cpu-baseline --option overlap_frames=0 > gpurun_out/r04_c2_pr8_off.json 2> gpurun_out/r04_c2_pr8_off.err
python bench.py --steps 100 --warmup 10 --subdiv 7 --floor-grid 256 --no-cpu-baseline --no-parity > gpurun_out/r04_stress_a.json 2> gpurun_out/r04_stress_a.err
python bench.py --steps 100 --warmup 10 --subdiv 7 --floor-grid 256 --no-cpu-baseline --no-parity --option overlap_frames=0 > gpurun_out/r04_stress_off.json 2> gpurun_out/r04_stress_off.err
