#!/usr/bin/env python3
"""One-off soak: many random gather sessions (tests/test_gpu_gather.py::test_random_gather_sessions_return_the_frames_the_device_drew).  python tools/exp/r04_soak_gather.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
import test_gpu_gather as T
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
lib = rt64.Library(); data = sample_scene.make_sample_scene()
bad = 0; t0 = time.time()
for seed in range(first, first + count):
    try:
        T.test_random_gather_sessions_return_the_frames_the_device_drew(lib, data, seed)
    except AssertionError as e:
        bad += 1; print("seed %d FAILED: %r" % (seed, str(e)[:300]), flush=True)
    if (seed - first) % 20 == 19:
        print("seed %d done, %.0f s" % (seed, time.time() - t0), flush=True)
print("sessions %d, failing %d" % (count, bad))
sys.exit(1 if bad else 0)
