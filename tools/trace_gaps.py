#!/usr/bin/env python3
"""Per-kernel durations and the idle gaps between consecutive kernels from a rocprofv3 --kernel-trace csv:  tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
rows = rows[len(rows) // 2:]                      # steady state
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for a, b in zip(rows, rows[1:]):
    k = short(a["Kernel_Name"])
    dur[k] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"]); gap[k] += int(b["Start_Timestamp"]) - int(a["End_Timestamp"]); cnt[k] += 1
if len(sys.argv) > 2:            # raw sequence of one frame
    for r in rows[-12:]:
        print("%-36s start %10.2f dur %8.2f" % (short(r["Kernel_Name"]), (int(r["Start_Timestamp"]) - int(rows[-12]["Start_Timestamp"])) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for k in sorted(dur, key=lambda k: -cnt[k])[:10]:
    print("%-36s n=%5d  dur %8.2f us   gap to next %7.2f us" % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3))
