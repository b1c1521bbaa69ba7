#!/usr/bin/env python3
"""One-off soak of the render streams: many long random host sessions (tests/test_gpu_overlap.py::_random_session), three streams against one, byte for byte.
   python tools/exp/r04_soak_sessions.py [first_seed] [count] [ops] [width height]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
import test_gpu_overlap as T
first, count, ops = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 40), (int(sys.argv[3]) if len(sys.argv) > 3 else 150)
if len(sys.argv) > 5:            # frame size of the sessions (default: the tests' 320 x 180; at 1920 x 1080 a frame is long enough for three of them to really run side by side)
    T.W, T.H = int(sys.argv[4]), int(sys.argv[5])
lib = rt64.Library()
data = sample_scene.make_sample_scene()
bad = 0; t0 = time.time()
for seed in range(first, first + count):
    a = T._random_session(lib, data, 1, seed, ops=ops)
    b = T._random_session(lib, data, 0, seed, ops=ops)
    same = len(a) == len(b) and all(ka == kb and fa == fb and np.array_equal(xa.view(np.uint8), xb.view(np.uint8)) for (ka, fa, xa), (kb, fb, xb) in zip(a, b))
    if not same:
        bad += 1
        print("seed %d DIFFERS" % seed, [(ka, fa) for (ka, fa, xa), (kb, fb, xb) in zip(a, b) if not np.array_equal(xa.view(np.uint8), xb.view(np.uint8))][:5], flush=True)
    if (seed - first) % 10 == 9:
        print("seed %d done, %d reads in the last session, %.0f s, last error %r" % (seed, len(a), time.time() - t0, lib.last_error()), flush=True)
print("sessions %d, differing %d" % (count, bad))
sys.exit(1 if bad else 0)
