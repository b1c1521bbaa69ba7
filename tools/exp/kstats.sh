#!/bin/bash
# Per-kernel average durations of one bench workload:  tools/exp/kstats.sh <tag> [bench flags]     (RT64_LIBRARY_PATH selects an experimental build)
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$TAG -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity "$@" > $OUT/ks_$TAG.log 2>&1
f=$(find $OUT/ks_$TAG -name '*kernel_stats.csv' | head -1)
cp $f $OUT/ks_$TAG.csv; rm -rf $OUT/ks_$TAG
cd $REPO
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/ks_$TAG.csv")))
print("== $TAG")
for r in rows[:12]:
    print("%-70s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
