#!/usr/bin/env python3
"""When did every wave of the one-kernel frame start and end?  tools/tile_timing.py [--subdiv 7 --floor-grid 256] [--option key=value ...] [--out prefix]

Runs a few frames with the device option tile_timing = 1 (the frame kernel stamps the chip-wide 100 MHz clock at each wave's start and end),
reads the records back (RT64_ReadbackTileTiming) and prints what the launch looked like from the inside:
  * how long the launch was, the sum of all wave lifetimes, and the mean number of waves resident per SIMD that follows from it,
  * the distribution of wave lifetimes (a few long waves among many short ones is a schedule problem, not a bandwidth problem),
  * the residency curve: waves alive in each tenth of the launch,
  * the waves that ended last, with their position in the frame.
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene

ap = argparse.ArgumentParser()
ap.add_argument("--subdiv", type=int, default=0); ap.add_argument("--floor-grid", type=int, default=1)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--option", action="append", default=[]); ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--out", default=""); ap.add_argument("--max-lights", type=int, default=12)
a = ap.parse_args()
W, H = a.width, a.height
lib = rt64.Library()
data = sample_scene.make_sample_scene(subdiv=a.subdiv, floor_grid=a.floor_grid)
s = sample_scene.Rt64Scene(lib, data, W, H, hip_device=0)
for kv in a.option:
    k, _, v = kv.partition("="); assert s.option(k, float(v)), k
if a.max_lights != 12:
    s.set_view_description(max_lights=a.max_lights)
per_wave = not any(kv.replace(" ", "") == "per_wave_frame=0" for kv in a.option) and a.subdiv > 0
for _ in range(a.frames):
    s.draw()
s.option("tile_timing", 1)
s.draw(); s.draw()
st = s.stats()
NW = (8192 + 8) * 4
rec = np.zeros((NW * 3, 4), dtype=np.uint32)
n = lib.ReadbackTileTiming(s.device, rec.ctypes.data, rec.nbytes)
assert n == rec.nbytes, lib.last_error()
s.close()
start, end, diag = rec[0:2 * NW:2], rec[1:2 * NW:2], rec[2 * NW:]
ran = (start[:, 3] == 1) & ((end[:, 3] & 0xFF) == 1)
t0 = start[ran, 0].astype(np.int64); t1 = end[ran, 0].astype(np.int64)
cyc = (end[ran, 1].astype(np.int64) - start[ran, 1].astype(np.int64)) & 0xFFFFFFFF
life = (t1 - t0) * 0.01                      # us
launch0, launch1 = t0.min(), t1.max()
span = (launch1 - launch0) * 0.01
waves = int(ran.sum())
print("frame %dx%d, %d triangles: fused frame %d, kernel %.1f us by HIP events; %d waves recorded" % (W, H, st.triangleCount, st.fusedFrame, st.msPrimaryTrace * 1e3, waves))
print("launch span (first wave start -> last wave end) %.1f us; sum of wave lifetimes %.0f us -> %.2f waves resident per SIMD on average (1024 SIMDs)" % (span, life.sum(), life.sum() / span / 1024.0))
print("mean shader clock while a wave lives: %.2f GHz" % (cyc.sum() / (life.sum() * 1e3)))
q = np.percentile(life, [1, 10, 25, 50, 75, 90, 99, 100])
print("wave lifetime us: p1 %.1f p10 %.1f p25 %.1f median %.1f p75 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(q))
order = np.argsort(life)[::-1]
top = life[order]
for share in (0.01, 0.05, 0.10, 0.25):
    k = max(1, int(waves * share))
    print("  the longest %4.0f %% of the waves (%5d) hold %4.1f %% of all wave time" % (100 * share, k, 100 * top[:k].sum() / life.sum()))
edges = np.linspace(launch0, launch1, 11)
alive = [int(((t0 < edges[i + 1]) & (t1 > edges[i])).sum()) for i in range(10)]
mid = [float((np.minimum(t1, edges[i + 1]) - np.maximum(t0, edges[i])).clip(min=0).sum() / max(edges[i + 1] - edges[i], 1)) for i in range(10)]
print("waves alive per tenth of the launch (time-averaged): " + " ".join("%.0f" % m for m in mid) + "   (capacity at 3 waves/SIMD: 3072)")
started = [int(((t0 >= edges[i]) & (t0 < edges[i + 1])).sum()) for i in range(10)]
print("waves started per tenth: " + " ".join(str(x) for x in started))
idx = np.flatnonzero(ran)
last = np.argsort(t1)[::-1][:8]
tilesX, strips = (W + 15) // 16, (H + 15) // 16
tiles = tilesX * strips
def where(record):
    if per_wave:
        xcd, j = record & 7, record >> 3
        quad, tile_seq = j & 3, (j >> 2) * 8 + xcd
    else:
        tile_seq, quad = record >> 2, record & 3
    tile = tiles - 1 - tile_seq
    return (tile % tilesX) * 16 + (quad & 1) * 8, (tile // tilesX) * 16 + (quad >> 1) * 8
worst_lane = end[ran, 2].astype(np.int64); total_visits = (end[ran, 3] >> 8).astype(np.int64)
print("visits (nodes + triangles): all waves %d; per wave the busiest lane makes %.1f x the wave's mean lane (median over waves %.1f x)" % (
    total_visits.sum(), (worst_lane.sum() * 64.0) / max(total_visits.sum(), 1), float(np.median(worst_lane * 64.0 / np.maximum(total_visits, 1)))))
long_ = order[:max(1, waves // 100)]
print("the longest 1 %% of the waves: busiest lane %.0f visits on average (max %d), wave mean lane %.0f; %.3f us of wave lifetime per visit of the busiest lane" % (
    worst_lane[long_].mean(), worst_lane[long_].max(), total_visits[long_].mean() / 64.0, life[long_].sum() / max(worst_lane[long_].sum(), 1)))
short_ = order[waves // 2:]
print("the shorter half of the waves: busiest lane %.1f visits on average; %.3f us per visit of the busiest lane" % (worst_lane[short_].mean(), life[short_].sum() / max(worst_lane[short_].sum(), 1)))
if diag[ran].any():          # diagnostic build (-DRT_PROFILE_TRIPS): wave-level trips of the node loop and of the leaf step, pops that went to the HBM spill slab
    tn, tl, sm, ss = (diag[ran, k].astype(np.int64) for k in range(4))
    print("DIAG all waves: node-loop trips %d, leaf trips %d, spill pops %d" % (tn.sum(), tl.sum(), ss.sum()))
    for name, sel in (("longest 1 %", long_), ("shorter half", short_)):
        print("DIAG %s: node trips %.0f, leaf trips %.0f per wave; busiest lane visits %.0f; spill pops per wave %.0f (worst lane %.0f); %.3f us per trip" % (
            name, tn[sel].mean(), tl[sel].mean(), worst_lane[sel].mean(), ss[sel].mean(), sm[sel].mean(), life[sel].sum() / max((tn[sel] + tl[sel]).sum(), 1)))
print("longest waves at pixel blocks (x, y): " + ", ".join("%s %.0f us / busiest lane %d visits" % (where(int(idx[j])), life[j], worst_lane[j]) for j in order[:12]))
print("last waves to end: " + ", ".join("record %d (lived %.0f us, ended at %.0f us)" % (idx[j], life[j], (t1[j] - launch0) * 0.01) for j in last))
if a.out:
    np.savez_compressed(a.out + ".npz", start=start[ran], end=end[ran], index=idx)
    json.dump({"span_us": span, "sum_life_us": float(life.sum()), "resident_per_simd": float(life.sum() / span / 1024.0), "waves": waves,
               "percentiles_us": dict(zip(["p1", "p10", "p25", "p50", "p75", "p90", "p99", "max"], [float(x) for x in q])), "alive_per_tenth": mid}, open(a.out + ".json", "w"))
