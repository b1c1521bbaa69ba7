"""What the per-pass HIP events cost a synchronous C2 frame: tools/profile_passes_cost.py (MI355X box).  Measured: 0.3-0.5 us of 195 us."""
import sys, time
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
g.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
lib = rt64.Library()
data = sample_scene.make_sample_scene()
s = sample_scene.Rt64Scene(lib, data, 1920, 1080, hip_device=0)
for pp in (1, 0, 1, 0):
    s.option("profile_passes", pp)
    for _ in range(30): s.draw()
    t0 = time.perf_counter()
    for _ in range(400): s.draw()
    dt = (time.perf_counter() - t0) / 400 * 1e3
    print("profile_passes", pp, "ms/frame %.5f" % dt)
s.close()
