#!/bin/bash
# Collect the rocprofv3 evidence of one bench workload on the GPU box:  tools/profile_round.sh <tag> <workload> [bench flags]
#   e.g.  tools/profile_round.sh r02 C2        |  tools/profile_round.sh r02 C5 --config C5  |  tools/profile_round.sh r02 stress_7_256 --subdiv 7 --floor-grid 256
#   1. --kernel-trace --stats                                   -> gpurun_out/<tag>_<workload>_kernel_stats.csv
#   2. --pmc FETCH_SIZE   3. --pmc WRITE_SIZE   4. --pmc SQ_*   -> gpurun_out/<tag>_<workload>_pmc_{fetch,write,sq}.csv (trimmed to the first dispatches)
# each PMC pass on its own with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 sections; gpurun refuses other mixes).
# tools/collect_counters.py then merges the four into profiles/kernel_counters.json (run it here AND commit the result: bench.py
# refuses a file whose source hash differs from the kernels it runs).
set -e
TAG=${1:-r04}; WL=${2:-C2}; shift 2 || true
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
P=$OUT/${TAG}_${WL}
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "${P}_stats" -- python3 "$REPO/bench.py" --steps 60 --warmup 8 --timed-loop-only "$@" > "${P}_stats.log" 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "${P}_fetch" -- python3 "$REPO/bench.py" --steps 6 --warmup 2 --timed-loop-only "$@" > "${P}_pmc_fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "${P}_write" -- python3 "$REPO/bench.py" --steps 6 --warmup 2 --timed-loop-only "$@" > "${P}_pmc_write.log" 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "${P}_sq" -- python3 "$REPO/bench.py" --steps 6 --warmup 2 --timed-loop-only "$@" > "${P}_pmc_sq.log" 2>&1
echo "sq pass done"
find "${P}_stats" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "${P}_kernel_stats.csv"
for k in fetch write sq; do
  f=$(find "${P}_$k" -name '*counter_collection.csv' | head -1)
  # keep the header + the rows of the library's kernels from the LAST 3000 lines (steady state): the summaries are what is judged
  (head -1 "$f"; tail -n 3000 "$f" | grep -v "at::\|rocclr\|elementwise" || true) > "${P}_pmc_$k.csv"
done
rm -rf "${P}_stats" "${P}_fetch" "${P}_write" "${P}_sq"
cd "$REPO"
python3 -c "import sys; sys.path.insert(0, 'tools'); import collect_counters as c; print(c.source_hash())" > "${P}_source_hash.txt"
python3 tools/collect_counters.py --source-hash "$(cat ${P}_source_hash.txt)" --workload "$WL" --stats "${P}_kernel_stats.csv" --fetch "${P}_pmc_fetch.csv" --write "${P}_pmc_write.csv" --sq "${P}_pmc_sq.csv" \
    --command "tools/profile_round.sh $TAG $WL $*" --out "$OUT/kernel_counters.json"
head -14 "${P}_kernel_stats.csv"
