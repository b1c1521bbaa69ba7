#!/bin/bash
# All workloads of the round through tools/profile_round.sh (rocprofv3 stats + three PMC passes each) and the bench lines of the same build.
#   tools/exp/r04_profile_all.sh a   -> C2, C3, C4, C5          tools/exp/r04_profile_all.sh b   -> stress, C4-literal, C5-literal, bench lines
# (two gpurun calls: one call is limited to 20 minutes)
set -e
if [ "${1:-a}" = a ]; then
  tools/profile_round.sh r04 C2
  tools/profile_round.sh r04 C3 --config C3
  tools/profile_round.sh r04 C4 --config C4
  tools/profile_round.sh r04 C5 --config C5
  cp gpurun_out/kernel_counters.json profiles/kernel_counters.json      # (gpurun_out/ does not travel to the next box; profiles/ does: copy the merged file back before part b)
else
  mkdir -p gpurun_out && cp profiles/kernel_counters.json gpurun_out/kernel_counters.json        # part a's workloads (same source hash, or the merge starts over)
  tools/profile_round.sh r04 stress_7_256 --subdiv 7 --floor-grid 256
  tools/profile_round.sh r04 C4-literal --config C4-literal
  tools/profile_round.sh r04 C5-literal --config C5-literal
  cp gpurun_out/kernel_counters.json profiles/kernel_counters.json
  python bench.py > gpurun_out/r04_bench_C2.json
  for c in C3 C4 C5 C4-literal C5-literal; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r04_bench_$c.json; done
  python bench.py --subdiv 7 --floor-grid 256 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r04_bench_stress_7_256.json
fi
