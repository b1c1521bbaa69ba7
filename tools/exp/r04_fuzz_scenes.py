#!/usr/bin/env python3
"""One-off soak: random scenes (tests/test_gpu_fuzz.py::random_scene) against the oracle.  python tools/exp/r04_fuzz_scenes.py [first] [count] [without,features] [option=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
import test_gpu_fuzz as T
from test_gpu_features import _render_pair
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
without = tuple(w for w in sys.argv[3].split(",") if w) if len(sys.argv) > 3 else ()
options = {"denoiser_mode": 1}
import test_gpu_features as F
for kv in [a for a in sys.argv[4:] if a.startswith("size=")]:       # size=WxH: frame size of the comparison (default: the tests' 320 x 180)
    F.W, F.H = (int(v) for v in kv[5:].split("x"))
for kv in [a for a in sys.argv[4:] if not a.startswith("size=")]:                      # further device options, e.g. lds_cache=0 (every walk fetches its nodes from HBM / L2: trace_ray_stepwise), simple_kernels=0 (the general kernels)
    k, _, v = kv.partition("="); options[k] = float(v)
lib = rt64.Library(); data = sample_scene.make_sample_scene()
nbad = 0; t0 = time.time()
for seed in range(first, first + count):
    d, view, chosen, per_frame = T.random_scene(data, seed, without)
    try:
        got, ref, st = _render_pair(lib, d, frames=chosen["frames"], view_desc=view, options=options, per_frame=per_frame)
        bad = T.compare(got, ref, st, chosen)
    except Exception as e:
        bad = ["exception %r, last error %r" % (e, lib.last_error())]
    if bad:
        nbad += 1; print("seed %d %s: %s" % (seed, chosen, bad), flush=True)
    if (seed - first) % 10 == 9:
        print("seed %d done, %.0f s" % (seed, time.time() - t0), flush=True)
print("scenes %d, disagreeing %d" % (count, nbad))
