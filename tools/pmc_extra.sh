#!/bin/bash
# Extra SQ counter passes for some kernels (diagnosis, not the judged profile):  tools/pmc_extra.sh <tag> <kernel-name-regex> [bench flags]
# One rocprofv3 --kernel-trace --pmc pass per counter set; prints the per-launch averages of the kernels whose name matches.
set -e
TAG=${1:-x}; KERN=${2:-lean_frame}; shift 2 || true
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
n=0
for SET in \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_LDS_LOAD" \
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" \
  "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_INSTS SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES" \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" \
  "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
do
  n=$((n+1))
  D=$OUT/${TAG}_x$n
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$D" -- python3 "$REPO/bench.py" --steps 6 --warmup 2 --no-cpu-baseline "$@" > "$OUT/${TAG}_x$n.log" 2>&1
  f=$(find "$D" -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$KERN" <<'PY'
import csv, sys, collections
import re
tot, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Kernel_Name"]):
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]
        tot[(name, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
for k in sorted(tot):
    print("%-44s %-28s %16.1f  (%d launches)" % (k[0][:44], k[1], tot[k] / cnt[k], cnt[k]))
PY
  rm -rf "$D"
done
