#!/bin/bash
# All five workloads of the round through tools/profile_round.sh (rocprofv3 stats + three PMC passes each) and the bench lines of the same build.
set -e
tools/profile_round.sh r03 C2
tools/profile_round.sh r03 C3 --config C3
tools/profile_round.sh r03 C4 --config C4
tools/profile_round.sh r03 C5 --config C5
tools/profile_round.sh r03 stress_7_256 --subdiv 7 --floor-grid 256
cp gpurun_out/kernel_counters.json profiles/kernel_counters.json
python bench.py > gpurun_out/r03_bench_C2.json
for c in C3 C4 C5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r03_bench_$c.json; done
python bench.py --subdiv 7 --floor-grid 256 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench_stress_7_256.json
