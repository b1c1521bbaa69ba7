#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md prescribes) into
profiles/hbm_traffic.json: measured HBM bytes per launch for each kernel of the frame.

Corrections applied (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
of a wide coalesced read, so fetch bytes are reported both raw and doubled ("fetch_x2" is the upper estimate; the access pattern
here is a mix of 16-byte node loads and 4-8 byte image/texel loads, so the truth lies between).  bench.py's `traffic` uses
raw_fetch*2 + write.
usage: tools/collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import csv
import json
import sys
from collections import defaultdict

SHORT = {"lean_frame_kernel": "lean_frame", "raster_draw_kernel": "raster_draw", "primary_trace_kernel": "primary_trace", "primary_shade_kernel": "primary_shade", "direct_kernel": "direct",
         "compose_post_kernel": "compose_post", "indirect_constant_kernel": "indirect_constant", "indirect_kernel": "indirect",
         "lbvh_small_kernel": "lbvh_small", "svgf_atrous_kernel": "svgf_atrous", "svgf_variance_kernel": "svgf_variance"}


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            key = next((v for k, v in SHORT.items() if k in name), None)
            if key is None:
                continue
            tot[key] += float(row["Counter_Value"]); cnt[key] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_units": "bytes per launch; fetch counters KiB->bytes; traffic = 2*fetch_raw + write (gfx950 FETCH_SIZE half-count correction)"}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
        out[k] = int(2 * f + w)
        out[k + "_detail"] = {"fetch_raw": int(f), "fetch_x2": int(2 * f), "write": int(w)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_") and not k.endswith("_detail")}, indent=1))


if __name__ == "__main__":
    main()
