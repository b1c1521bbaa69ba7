"""Frame time of the default C2 frame with and without the per-frame HIP event records (option profile_passes): what the two barrier packets cost."""
import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as g; g.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
lib = rt64.Library()
data = sample_scene.make_sample_scene()
s = sample_scene.Rt64Scene(lib, data, 1920, 1080, hip_device=0)
for pp in (1, 0, 1, 0):
    s.option("profile_passes", pp)
    for _ in range(30): s.draw()
    t0 = time.perf_counter()
    for _ in range(400): s.draw()
    print("profile_passes", pp, (time.perf_counter() - t0) / 400 * 1e3, "ms/frame")
s.close()
