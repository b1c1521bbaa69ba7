"""One band of an N-way partition of a config, alone on the device, for a kernel trace: tools/exp/r03_one_band.py C5 1624 1719 [exchange]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as g
g.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
cfgname, y0, y1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = sample_scene.BENCH_CONFIGS[cfgname]
lib = rt64.Library()
data = sample_scene.make_sample_scene(); sample_scene.apply_bench_config(data, cfgname)
s = sample_scene.Rt64Scene(lib, data, cfg["width"], cfg["height"], hip_device=0)
s.set_view_description(gi_samples=cfg["gi_samples"], denoiser=cfg["denoiser"])
s.option("profile_passes", 0)
s.set_tile(y0, y1)
if len(sys.argv) > 4:
    H = cfg["height"]
    starts = (C.c_int * 4)(0, y0, y1, H)
    noop = rt64.HALO_EXCHANGE(lambda user, regions, count: None)
    assert lib.SetDeviceHaloExchange(s.device, C.cast(noop, C.c_void_p), None, starts, 1, 3) == 1
    s.option("halo_dry_run", 1)
for _ in range(60):
    s.draw()
s.close()
