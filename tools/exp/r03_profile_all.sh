#!/bin/bash
# All five workloads of the round through tools/profile_round.sh (rocprofv3 stats + three PMC passes each) and the bench lines of the same build.
#   tools/exp/r03_profile_all.sh a   -> C2, C3, C4          tools/exp/r03_profile_all.sh b   -> C5, stress, bench lines (incl. the literal C4 / C5)
# (two gpurun calls: one call is limited to 20 minutes)
set -e
if [ "${1:-a}" = a ]; then
  tools/profile_round.sh r03 C2
  tools/profile_round.sh r03 C3 --config C3
  tools/profile_round.sh r03 C4 --config C4
  cp gpurun_out/kernel_counters.json profiles/kernel_counters.json      # (gpurun_out/ does not travel to the next box; profiles/ does: copy the merged file back before part b)
else
  mkdir -p gpurun_out && cp profiles/kernel_counters.json gpurun_out/kernel_counters.json        # part a's workloads (same source hash, or the merge starts over)
  tools/profile_round.sh r03 C5 --config C5
  tools/profile_round.sh r03 stress_7_256 --subdiv 7 --floor-grid 256
  cp gpurun_out/kernel_counters.json profiles/kernel_counters.json
  python bench.py > gpurun_out/r03_bench_C2.json
  for c in C3 C4 C5 C4-literal C5-literal; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r03_bench_$c.json; done
  python bench.py --subdiv 7 --floor-grid 256 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench_stress_7_256.json
fi
