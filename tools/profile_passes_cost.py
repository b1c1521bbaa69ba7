"""What the per-pass HIP events cost a synchronous frame: tools/profile_passes_cost.py [C2|C3|C4|C5] (MI355X box).  Measured: C2 0.3-0.5 us of 195 us."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = sample_scene.BENCH_CONFIGS[cfgname]
lib = rt64.Library()
data = sample_scene.make_sample_scene()
sample_scene.apply_bench_config(data, cfgname)
s = sample_scene.Rt64Scene(lib, data, cfg["width"], cfg["height"], hip_device=0)
if cfg["gi_samples"] or cfg["denoiser"]:
    s.set_view_description(gi_samples=cfg["gi_samples"], denoiser=cfg["denoiser"])
for pp in (1, 0, 1, 0):
    s.option("profile_passes", pp)
    for _ in range(30): s.draw()
    t0 = time.perf_counter()
    for _ in range(300): s.draw()
    dt = (time.perf_counter() - t0) / 300 * 1e3
    print(cfgname, "profile_passes", pp, "ms/frame %.5f" % dt)
s.close()
