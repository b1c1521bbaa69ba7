# multi-GPU projection inputs, measured on one GPU: what every rank's band costs (GI configurations) / what a strip share costs enqueued (C2)
mkdir -p gpurun_out
for c in C5 C5-literal C4-literal C3; do
  python tools/band_costs.py --config $c --ranks 2,4,8 --frames 30 > gpurun_out/r04_band_costs_$c.jsonl 2> gpurun_out/r04_band_costs_$c.err
done
for pr in 2 4 8; do
  python bench.py --steps 300 --warmup 20 --pretend-ranks $pr --no-cpu-baseline > gpurun_out/r04_share_C2_pr$pr.json 2> gpurun_out/r04_share_C2_pr$pr.err
done
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity > gpurun_out/r04_share_C2_pr1.json 2>> gpurun_out/r04_share_C2_pr1.err
python tools/host_overhead.py > gpurun_out/r04_host_overhead2.txt 2>&1
