python -m pytest tests -q -m gpu --durations=40 > gpurun_out/r04_full1.txt 2>&1
tail -60 gpurun_out/r04_full1.txt
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04_c2_b.json 2> gpurun_out/r04_c2_b.err
python bench.py --config C5 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r04_c5_b.json 2> gpurun_out/r04_c5_b.err
python bench.py --steps 100 --warmup 10 --subdiv 7 --floor-grid 256 --no-cpu-baseline --no-parity > gpurun_out/r04_stress_b.json 2> gpurun_out/r04_stress_b.err
