#!/usr/bin/env python3
"""One-off soak: random HUD (raster) instances against the oracle's rasteriser (tests/test_gpu_raster.py::random_hud).  python tools/exp/r04_fuzz_hud.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
import test_gpu_raster as T
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
lib = rt64.Library(); data = sample_scene.make_sample_scene()
nbad = 0; t0 = time.time()
for seed in range(first, first + count):
    try:
        bad = T.compare_hud(lib, T.random_hud(data, seed))
    except Exception as e:
        bad = ["exception %r, last error %r" % (e, lib.last_error())]
    if bad:
        nbad += 1; print("seed %d: %s" % (seed, bad), flush=True)
    if (seed - first) % 50 == 49:
        print("seed %d done, %.0f s" % (seed, time.time() - t0), flush=True)
print("HUDs %d, disagreeing %d" % (count, nbad))
