"""The multi-GPU gather behind the C ABI (RT64_GetGatherUniqueId / RT64_CreateGather / RT64_SubmitGather / RT64_ReadbackGather) on the
one GPU of the test box: a world of one goes through the same calls (RCCL communicator of one rank, two slots, send buffer written by the
frame kernel, reassembly kernel on the library's comm stream).  Layout with more ranks: tests/test_tiles_gloo.py (CPU) checks the
library's layout exports; N > 1 over xGMI is the driver's scaling run (bench.py --gpus N uses these same exports)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("bands,gi", [(0, 0), (1, 1)])
def test_world_of_one_gather_returns_the_devices_frame(rt64_lib, sample_data, bands, gi):
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    W, H = 320, 180
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    try:
        if gi:
            s.set_view_description(gi_samples=1, denoiser=True)
        uid = (C.c_uint8 * rt64.GATHER_ID_BYTES)()
        assert rt64_lib.GetGatherUniqueId(uid, len(uid)) == 1, rt64_lib.last_error()
        g = rt64_lib.CreateGather(s.device, uid, len(uid), 0, 1, bands)
        assert g, rt64_lib.last_error()
        s.option("sync_present", 0)                          # frames are enqueued; the exchange of frame k runs beside frame k + 1
        slots = []
        for _ in range(5):
            s.draw()
            slots.append(rt64_lib.SubmitGather(g))
        assert slots == [0, 1, 0, 1, 0]
        assert bool(s.stats().packedFinal) == (gi == 0)      # the one-kernel frame writes the send buffer itself; GI frames are packed by a copy
        out = np.zeros((H, W, 4), dtype=np.uint8)
        assert rt64_lib.ReadbackGather(g, -1, out.ctypes.data, out.nbytes, 0) == out.nbytes, rt64_lib.last_error()
        assert rt64_lib.GetGatherFrame(g, slots[-1])
        s.option("sync_present", 1)
        own = s.readback(rt64.IMAGE_FINAL_RGBA8)
        assert np.array_equal(out, own)
        rt64_lib.DestroyGather(g)
        s.draw()                                             # the device keeps working after the gather is gone
        assert np.array_equal(s.readback(rt64.IMAGE_FINAL_RGBA8)[..., 3], own[..., 3])
    finally:
        s.close()


def test_c_host_gathers_through_the_c_abi(rt64_lib):
    """tools/sample_host.c --ranks 1: fork-before-GPU launcher, id over a pipe, RT64_CreateGather + RT64_SubmitGather per frame from C; the
    gathered frame has the checksum of the frame the same host renders without a gather."""
    from test_c_host import build_host
    host = build_host()
    env = dict(os.environ, RT64_LIBRARY_PATH=rt64_lib.path)
    base = [host, "--width", "320", "--height", "180", "--frames", "3", "--assets", os.path.join(ROOT, "assets", "sample")]
    a = json.loads(next(l for l in subprocess.run(base, env=env, capture_output=True, text=True, timeout=300, check=True).stdout.splitlines() if l.startswith("{")))
    b = json.loads(next(l for l in subprocess.run(base + ["--ranks", "1"], env=env, capture_output=True, text=True, timeout=300, check=True).stdout.splitlines() if l.startswith("{")))
    assert b["ranks"] == 1 and (a["checksum"], a["fnv1a"]) == (b["checksum"], b["fnv1a"])
