# K render streams: share of an 8-way partition, enqueued C2 / stress frames, host cost per enqueued frame
python -m pytest tests/test_gpu_overlap.py -x -q -m gpu > gpurun_out/r04_t3.txt 2>&1
python tools/host_overhead.py > gpurun_out/r04_host_overhead.txt 2>&1
for k in 1 2 3 4; do
  RT64_RENDER_STREAMS=$k python bench.py --steps 300 --warmup 20 --pretend-ranks 8 --no-cpu-baseline > gpurun_out/r04_pr8_k$k.json 2> gpurun_out/r04_pr8_k$k.err
  RT64_RENDER_STREAMS=$k python bench.py --steps 300 --warmup 20 --pretend-ranks 4 --no-cpu-baseline > gpurun_out/r04_pr4_k$k.json 2> gpurun_out/r04_pr4_k$k.err
done
for k in 3 4; do
  RT64_RENDER_STREAMS=$k python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity > gpurun_out/r04_c2_k$k.json 2> gpurun_out/r04_c2_k$k.err
  RT64_RENDER_STREAMS=$k python bench.py --steps 100 --warmup 10 --subdiv 7 --floor-grid 256 --no-cpu-baseline --no-parity > gpurun_out/r04_stress_k$k.json 2> gpurun_out/r04_stress_k$k.err
done
