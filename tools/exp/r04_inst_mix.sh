#!/bin/bash
# Dynamic instruction mix of the C2 frame kernel (wave-instructions per launch by class).
set -e
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d "$OUT/inst_mix" -- python3 "$REPO/bench.py" --steps 6 --warmup 2 --timed-loop-only ${1:+--config $1} > "$OUT/inst_mix.log" 2>&1
cd "$REPO"
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/inst_mix/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "lean_frame" not in k and "bounce" not in k and "atrous" not in k: continue
    acc[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": n[k[:70]] += 1
for k, v in acc.items():
    d = max(n[k], 1)
    print(k, "launches", d, {c: round(x / d / 1e6, 3) for c, x in sorted(v.items())}, "(millions of wave-instructions per launch)")
PY
rm -rf gpurun_out/inst_mix
