#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box:  tools/profile_round.sh r01 [extra bench flags]
#   1. --kernel-trace --stats of the default bench workload (C2, 1080p)      -> gpurun_out/<tag>_stats/
#   2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE, each in its own pass with --kernel-trace only (MI355X_MICROARCH.md, HBM section)
# The caller copies the summaries into profiles/ (tools/collect_traffic.py turns 2+3 into profiles/hbm_traffic.json).
set -e
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- python3 "$REPO/bench.py" --steps 100 --warmup 10 --no-cpu-baseline "$@" > "$OUT/${TAG}_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_pmc_fetch" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/${TAG}_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_pmc_write" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/${TAG}_pmc_write.log" 2>&1
find "$OUT/${TAG}_stats" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_kernel_stats.csv"
find "$OUT/${TAG}_pmc_fetch" -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_pmc_fetch_size.csv"
find "$OUT/${TAG}_pmc_write" -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_pmc_write_size.csv"
# keep the merged-back payload small: the raw traces are not needed once the summaries exist
rm -rf "$OUT/${TAG}_stats" "$OUT/${TAG}_pmc_fetch" "$OUT/${TAG}_pmc_write"
head -12 "$OUT/${TAG}_kernel_stats.csv"
