#!/usr/bin/env python3
"""Average rocprofv3 counter rows per kernel:  tools/pmc_summary.py gpurun_out/<tag>.csv [...]"""
import csv, sys
from collections import defaultdict
for path in sys.argv[1:]:
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in sorted(tot):
        if any(x in k for x in ("at::", "rocclr", "bc7", "elementwise")):
            continue
        print(k, {c: round(tot[k][c] / cnt[k][c]) for c in sorted(tot[k])})
