"""Which image moves when fold_guide / fold_compose are switched off (diagnosis aid of round 3)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import __graft_entry__ as g
g.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
import test_gpu_configs as T
lib = rt64.Library(); data = sample_scene.make_sample_scene()
base, _, _ = T._bench_pair(lib, data, "C3", frames=3)
for opts in ({"fold_guide": 0}, {"fold_compose": 0}):
    b, _, _ = T._bench_pair(lib, data, "C3", frames=3, options=opts)
    for k in base:
        d = (base[k].view(np.uint8) != b[k].view(np.uint8))
        if d.any():
            idx = np.argwhere(d.reshape(base[k].shape[0], base[k].shape[1], -1).any(axis=-1))
            print(opts, k, int(d.sum()), "pixels", len(idx), idx[:5].tolist(), base[k][tuple(idx[0])], b[k][tuple(idx[0])])
print("done")
