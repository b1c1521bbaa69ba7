#!/usr/bin/env python3
"""The ops of one random session up to its first differing read:  python tools/exp/r04_soak_debug.py seed [ops]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
import test_gpu_overlap as T
seed, ops = int(sys.argv[1]), (int(sys.argv[2]) if len(sys.argv) > 2 else 150)
lib = rt64.Library(); data = sample_scene.make_sample_scene()
la, lb = [], []
a = T._random_session(lib, data, 1, seed, ops=ops, log=la)
b = T._random_session(lib, data, 0, seed, ops=ops, log=lb)
assert la == lb
first = None
for (ka, fa, xa), (kb, fb, xb) in zip(a, b):
    if not np.array_equal(xa.view(np.uint8), xb.view(np.uint8)):
        d = (xa.view(np.uint8) != xb.view(np.uint8))
        rows = np.nonzero(d.reshape(xa.shape[0], -1).any(axis=1))[0]
        print("first differing read:", ka, fa, "shape", xa.shape, "rows", rows[:5], "...", rows[-5:], "count", int(d.sum()))
        first = fa; break
print("ops (op, frame before):")
for op, f in la:
    if first is not None and f > first: break
    print(" ", op, f)
