#!/usr/bin/env python3
"""Which images of one random scene differ from the oracle's, and where:  python tools/exp/r04_fuzz_detail.py seed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
import test_gpu_fuzz as T
from test_gpu_features import _render_pair
seed = int(sys.argv[1])
lib = rt64.Library(); data = sample_scene.make_sample_scene()
d, view, chosen, per_frame = T.random_scene(data, seed)
print(chosen)
got, ref, st = _render_pair(lib, d, frames=chosen["frames"], view_desc=view, options={"denoiser_mode": 1}, per_frame=per_frame)
print(T.compare(got, ref, st, chosen))
pairs = [("OUTPUT_RGBA32F", "output"), ("DIFFUSE", "diffuse"), ("DIRECT_LIGHT_RAW", "directLight"), ("INDIRECT_LIGHT_RAW", "indirectLight"), ("REFLECTION", "reflection"), ("REFRACTION", "refraction"), ("TRANSPARENT", "transparent")]
out = np.abs(got["OUTPUT_RGBA32F"][..., :3] - ref["output"][..., :3]).max(axis=-1)
ys, xs = np.nonzero(out > 2e-2)
print("pixels beyond 0.02:", len(ys), "rows", np.unique(ys)[:20], "cols", np.unique(xs)[:20])
for g, r in pairs:
    if r not in ref or ref[r] is None: continue
    a, b = got[g].astype(np.float64), ref[r].astype(np.float64)
    if a.shape != b.shape: print(g, "shapes", a.shape, b.shape); continue
    dd = np.abs(a - b).reshape(a.shape[0], a.shape[1], -1).max(axis=-1)
    print("%-20s max %.4f, pixels > 0.02: %d, of them among the output's: %d" % (g, dd.max(), int((dd > 2e-2).sum()), int(((dd > 2e-2) & (out > 2e-2)).sum())))
inst = got["INSTANCE_ID"]
for y, x in list(zip(ys, xs))[:12]:
    print("(%d,%d) inst %d/%d out %s vs %s direct %s vs %s" % (y, x, inst[y, x], ref["instanceId"][y, x], np.round(got["OUTPUT_RGBA32F"][y, x, :3], 3), np.round(ref["output"][y, x, :3], 3),
          np.round(got["DIRECT_LIGHT_RAW"][y, x, :3], 3), np.round(ref["directLight"][y, x, :3], 3)))
