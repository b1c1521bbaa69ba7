#!/usr/bin/env python3
"""How close are the GI rays' visit counters to the oracle's?  (round 4: bounce directions by the shared sincos of direction spec D1)
   python tools/exp/r04_gi_counters.py [W H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
from oracle import oracle_py

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 360)
lib = rt64.Library()
for config, frames in (("C3", 2), ("C5", 2)):
    data = sample_scene.make_sample_scene()
    cfg = sample_scene.BENCH_CONFIGS[config]
    sample_scene.apply_bench_config(data, config)
    s = sample_scene.Rt64Scene(lib, data, W, H, hip_device=0)
    o = oracle_py.OracleScene(data)
    s.set_view_description(gi_samples=cfg["gi_samples"], denoiser=cfg["denoiser"])
    s.option("count_traversal", 1)
    for f in range(frames):
        s.draw()
        ref = o.render(W, H, giSamples=cfg["gi_samples"], denoiserEnabled=int(cfg["denoiser"]), denoiserMode=1, images=(f == frames - 1))
    st = s.stats(); c = ref["counters"]
    oi_nodes = c["nodesVisited"] - c["nodesVisitedPrimary"] - c["nodesVisitedShadow"]
    fin = s.readback(rt64.IMAGE_FINAL_RGBA8); raw = s.readback(rt64.IMAGE_INDIRECT_LIGHT_RAW); nrm = s.readback(rt64.IMAGE_SHADING_NORMAL)
    d = np.abs(fin.astype(np.int32) - ref["final"].astype(np.int32))
    print(config, W, H, "indirect rays", st.indirectRays, c["indirectRays"], "| GPU nodesIndirect", st.nodesIndirect, "trisIndirect", st.trianglesIndirect,
          "| GPU nodes total", st.nodesVisited, "oracle total", c["nodesVisited"], "diff", int(st.nodesVisited) - int(c["nodesVisited"]),
          "| tris total", st.trianglesTested, c["trianglesTested"], "diff", int(st.trianglesTested) - int(c["trianglesTested"]),
          "| max rgba8 diff", int(d.max()), "px>1:", int((d > 1).any(axis=-1).sum()),
          "| raw GI max diff", float(np.abs(raw[..., :3] - ref["indirectLight"][..., :3]).max()),
          "| shading normal px differing", int((nrm != ref["shadingNormal"]).any(axis=-1).sum()), "of", W * H)
    s.close(); o.close()
