#!/bin/bash
# Register / scratch / LDS use of every kernel of one translation unit, from the compiler's own remarks:
#   tools/kernel_resources.sh passes.hip [extra hipcc flags]
F=${1:-passes.hip}; shift || true
cd "$(dirname "$0")/../sm64rt-legacy-renderer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Rpass-analysis=kernel-resource-usage "$@" -c "$F" -o /tmp/kr_$$.o 2>&1 | python3 -c '
import re, sys
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*?):\s*(\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name" or k == "Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
import subprocess
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
    print("%-52s VGPR %4s AGPR %3s SGPR %4s scratch %5s occ %2s LDS %6s" % (n[-52:], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
rm -f /tmp/kr_$$.o
