#!/usr/bin/env python3
"""One-off soak: random / pathological meshes through the BLAS builders against the oracle (tests/test_gpu_mesh_fuzz.py).  python tools/exp/r04_fuzz_meshes.py [first] [count] [sizes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
import test_gpu_mesh_fuzz as T
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
if len(sys.argv) > 3:            # triangle counts to draw from instead of the tests' (e.g. 131072,131073,200000: the large-tree builder's second group level)
    T.SIZES = [int(v) for v in sys.argv[3].split(",")]
lib = rt64.Library(); data = sample_scene.make_sample_scene()
nbad = 0; t0 = time.time(); seen = {}
for seed in range(first, first + count):
    _, kind, n = T.random_mesh(seed)
    seen[(kind, n)] = seen.get((kind, n), 0) + 1
    try:
        bad = T.compare_mesh(lib, data, seed)
    except Exception as e:
        bad = ["%s, %d triangles: exception %r, last error %r" % (kind, n, e, lib.last_error())]
    if bad:
        nbad += 1; print("seed %d: %s" % (seed, bad), flush=True)
    if (seed - first) % 25 == 24:
        print("seed %d done, %.0f s" % (seed, time.time() - t0), flush=True)
print("meshes %d (%d distinct kind x size pairs), disagreeing %d" % (count, len(seen), nbad))
