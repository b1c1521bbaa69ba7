#!/usr/bin/env python3
"""Merge the rocprofv3 passes of one bench workload into profiles/kernel_counters.json, the file bench.py's `roofline` object reads.

Passes (tools/profile_round.sh runs them; each PMC pass on its own with --kernel-trace only, as MI355X_MICROARCH.md prescribes):
  --stats  *_kernel_stats.csv          rocprofv3 --kernel-trace --stats: average duration per kernel
  --fetch  *_counter_collection.csv    --pmc FETCH_SIZE
  --write  *_counter_collection.csv    --pmc WRITE_SIZE
  --sq     *_counter_collection.csv    --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
Corrections (guide, HBM section): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
reads, so hbm_bytes = 2 x fetch + write (an upper estimate for the narrow image / texel loads in the mix).
The file carries the hash of the kernel sources it was measured on; bench.py refuses it when the sources changed since.
"""
import argparse
import csv
import hashlib
import json
import os
import subprocess
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHORT = [("lean_frame_kernel<true, true,", "full_frame"), ("lean_frame_kernel<false, true,", "full_frame"), ("lean_frame_kernel", "lean_frame"),
         ("raster_draw_kernel", "raster_draw"), ("raster_setup_kernel", "raster_setup"), ("primary_trace_kernel", "primary_trace"),
         ("primary_shade_kernel", "primary_shade"), ("direct_kernel", "direct"), ("compose_post_kernel", "compose_post"), ("post_process_kernel", "post_process"),
         ("indirect_constant_kernel", "indirect_constant"), ("indirect_kernel", "indirect_klist"), ("bounce_trace_plain_kernel", "bounce_trace"),
         ("bounce_trace_refill_kernel", "bounce_trace"), ("bounce_trace_split_kernel", "bounce_trace"), ("bounce_hit_kernel", "bounce_hit"), ("bounce_miss_kernel", "bounce_miss"),
         ("bounce_resolve_kernel", "bounce_resolve"), ("reflection_kernel", "reflection"), ("refraction_kernel", "refraction"),
         ("svgf_atrous_kernel", "svgf_atrous"), ("svgf_variance_kernel", "svgf_variance"), ("svgf_guide_kernel", "svgf_guide"), ("gaussian_kernel", "gaussian"),
         ("lbvh_small_batch_kernel", "lbvh_small_batch"), ("lbvh_small_kernel", "lbvh_small"), ("lg_", "lbvh_large"), ("taa_", "taa_upsample")]


def short(name):
    return next((v for k, v in SHORT if k in name), None)


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sm64rt-legacy-renderer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp", ".inc")) or name == "Makefile":
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def counter_rows(path):
    tot, cnt, meta = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int)), {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
            meta[k] = {m: int(float(r[m])) for m in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size", "Workgroup_Size") if m in r and r[m] != ""}
    return {k: {c: tot[k][c] / cnt[k][c] for c in tot[k]} for k in tot}, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--stats"); ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--sq")
    ap.add_argument("--command", default="")
    ap.add_argument("--source-hash", default="", help="hash of the kernel sources the passes ran on (the profile script records it on the GPU box); default: the tree's current one")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "kernel_counters.json"))
    a = ap.parse_args()
    sh = a.source_hash or source_hash()
    doc = {}
    if os.path.exists(a.out):
        try:
            doc = json.load(open(a.out))
        except Exception:
            doc = {}
    if doc.get("source_hash") != sh:
        doc = {}
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = ""
    doc.update({"source_hash": sh, "_units": "per launch (averages over the profiled dispatches); bytes: FETCH_SIZE / WRITE_SIZE KiB -> bytes, hbm_bytes = 2 x fetch + write (gfx950 FETCH_SIZE half-count correction)"})
    if commit:
        doc["collected_at_or_after_commit"] = commit
    kernels = defaultdict(dict)
    if a.stats:
        with open(a.stats) as f:
            for r in csv.DictReader(f):
                k = short(r["Name"])
                if k is None:
                    continue
                e = kernels[k]
                calls, avg = int(r["Calls"]), float(r["AverageNs"])
                if "avg_ns" in e:            # several template instances share a short name: call-weighted
                    n0 = e["calls"]; e["avg_ns"] = (e["avg_ns"] * n0 + avg * calls) / (n0 + calls); e["calls"] = n0 + calls
                else:
                    e["avg_ns"], e["calls"] = avg, calls
    for path, names in ((a.fetch, ("FETCH_SIZE",)), (a.write, ("WRITE_SIZE",)), (a.sq, None)):
        if not path:
            continue
        vals, meta = counter_rows(path)
        for k, cs in vals.items():
            for c, v in cs.items():
                if names is None or c in names:
                    kernels[k][c] = v
            kernels[k].update(meta.get(k, {}))
    for k, e in kernels.items():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["fetch_bytes_x2"] = 2.0 * e["FETCH_SIZE"] * 1024.0; e["write_bytes"] = e["WRITE_SIZE"] * 1024.0
            e["hbm_bytes"] = e["fetch_bytes_x2"] + e["write_bytes"]
    usable = {k: {c: (round(v, 1) if isinstance(v, float) else v) for c, v in e.items()} for k, e in kernels.items() if "hbm_bytes" in e and "SQ_INSTS_VALU" in e}
    partial = {k: {c: (round(v, 1) if isinstance(v, float) else v) for c, v in e.items()} for k, e in kernels.items() if k not in usable}
    doc.setdefault("workloads", {})[a.workload] = {"command": a.command, "kernels": usable, "incomplete": partial}
    json.dump(doc, open(a.out, "w"), indent=1, sort_keys=True)
    for k, e in sorted(usable.items(), key=lambda kv: -kv[1].get("avg_ns", 0) * kv[1].get("calls", 1)):
        print("%-18s avg %9.1f us  hbm %8.1f MB (fetch x2 %7.1f + write %7.1f)  VALU insts %10.0f  scratch %s  vgpr %s+%s" % (
            k, e.get("avg_ns", 0) / 1e3, e["hbm_bytes"] / 1e6, e["fetch_bytes_x2"] / 1e6, e["write_bytes"] / 1e6, e["SQ_INSTS_VALU"],
            e.get("Scratch_Size"), e.get("VGPR_Count"), e.get("Accum_VGPR_Count")))


if __name__ == "__main__":
    main()
