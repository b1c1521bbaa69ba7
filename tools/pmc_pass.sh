#!/bin/bash
# One rocprofv3 counter pass over a short bench run:  tools/pmc_pass.sh <tag> "<counters>" [bench flags]
# (--kernel-trace + --pmc only, as gpurun requires).  Writes gpurun_out/<tag>.csv (per-dispatch counter rows).
set -e
TAG=$1; CTRS=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/$TAG" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$OUT/$TAG.log" 2>&1
find "$OUT/$TAG" -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} "$OUT/$TAG.csv"
rm -rf "$OUT/$TAG"
