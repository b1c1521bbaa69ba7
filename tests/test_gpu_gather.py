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


def test_set_gather_bands_on_a_world_of_one_and_the_dry_run_option(rt64_lib, sample_data):
    """RT64_SetGatherBands on the one rank a test box has: a gather of cost-balanced bands takes new boundaries (here the only valid ones, [0, H]) and goes on
    gathering the same frame; a gather of equal bands takes boundaries as well; boundaries that do not span the frame and gathers of interleaved strips are refused with an error; device option halo_dry_run is
    accepted (the timing aid of tools/band_costs.py: it only has an effect on a band with an exchange set up)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    W, H = 320, 180
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    try:
        s.set_view_description(gi_samples=1, denoiser=True)
        s.draw()                                             # cost-balanced bands are cut from a whole frame
        uid = (C.c_uint8 * rt64.GATHER_ID_BYTES)()
        assert rt64_lib.GetGatherUniqueId(uid, len(uid)) == 1
        g = rt64_lib.CreateGather(s.device, uid, len(uid), 0, 1, 2)
        assert g, rt64_lib.last_error()
        s.draw(); rt64_lib.SubmitGather(g)
        a = np.zeros((H, W, 4), dtype=np.uint8)
        assert rt64_lib.ReadbackGather(g, -1, a.ctypes.data, a.nbytes, 0) == a.nbytes
        assert rt64_lib.SetGatherBands(g, (C.c_int * 2)(0, H)) == 1, rt64_lib.last_error()
        assert rt64_lib.SetGatherBands(g, (C.c_int * 2)(0, H - 1)) == 0 and "starts" in rt64_lib.last_error()
        assert s.option("halo_dry_run", 1) and s.option("halo_dry_run", 0)
        s.draw(); rt64_lib.SubmitGather(g)
        b = np.zeros((H, W, 4), dtype=np.uint8)
        assert rt64_lib.ReadbackGather(g, -1, b.ctypes.data, b.nbytes, 0) == b.nbytes
        assert np.array_equal(b, s.readback(rt64.IMAGE_FINAL_RGBA8))
        assert np.abs(a.astype(np.int32) - b.astype(np.int32)).mean() < 2.0          # consecutive frames of the same scene (GI noise only)
        rt64_lib.DestroyGather(g)
        assert rt64_lib.GetGatherUniqueId(uid, len(uid)) == 1    # (an id makes one communicator)
        g1 = rt64_lib.CreateGather(s.device, uid, len(uid), 0, 1, 1)          # equal bands take boundaries too (no whole frame needed for a first cut) ...
        assert g1 and rt64_lib.SetGatherBands(g1, (C.c_int * 2)(0, H)) == 1, rt64_lib.last_error()
        got = (C.c_int * 2)()
        assert rt64_lib.GetGatherBands(g1, got, 2) == 1 and list(got) == [0, H]
        rt64_lib.DestroyGather(g1)
        assert rt64_lib.GetGatherUniqueId(uid, len(uid)) == 1
        g0 = rt64_lib.CreateGather(s.device, uid, len(uid), 0, 1, 0)          # ... interleaved strips do not
        assert g0 and rt64_lib.SetGatherBands(g0, (C.c_int * 2)(0, H)) == 0 and "bands = 0" in rt64_lib.last_error()
        rt64_lib.DestroyGather(g0)
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


def test_cost_balanced_bands_reassemble_the_whole_frame(rt64_lib, sample_data):
    """bands = 2: RT64_CreateGather cuts contiguous bands of about equal modelled cost from the device's last whole frame.  With a world of one
    the gather itself is trivial, so the cut is checked through RT64_BalanceGatherBands on the same hit counts: three devices rendering those
    bands (GI + SVGF, with the denoiser halo) reassemble the single-device frame bit for bit, and the geometry-heavy bottom bands are thinner."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    W, H, N = 320, 180, 3
    whole = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    parts = []
    try:
        whole.set_view_description(gi_samples=1, denoiser=True)
        whole.draw()
        hits = (whole.readback(rt64.IMAGE_FIRST_INSTANCE_ID) >= 0).sum(axis=1).astype(np.uint32)
        starts = (C.c_int * (N + 1))()
        rt64_lib.BalanceGatherBands(hits.ctypes.data_as(C.POINTER(C.c_uint)), W, H, N, starts)
        s = list(starts)
        assert s[0] == 0 and s[-1] == H and (s[1] - s[0]) > (s[3] - s[2])                      # sky band on top is taller than the geometry band at the bottom
        # a gather created on the device after that whole frame derives the same boundaries (world of one: [0, H])
        uid = (C.c_uint8 * rt64.GATHER_ID_BYTES)()
        assert rt64_lib.GetGatherUniqueId(uid, len(uid)) == 1
        g = rt64_lib.CreateGather(whole.device, uid, len(uid), 0, 1, 2)
        assert g, rt64_lib.last_error()
        got = (C.c_int * 2)()
        assert rt64_lib.GetGatherBands(g, got, 2) == 1 and list(got) == [0, H]
        rt64_lib.DestroyGather(g)
        parts = [sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0) for _ in range(N)]
        for p, (a, b) in zip(parts, zip(s, s[1:])):
            p.set_view_description(gi_samples=1, denoiser=True)
            p.set_tile(a, b)
        for _ in range(3):
            for p in parts:
                p.draw()
        for _ in range(2):
            whole.draw()
        for image in (rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_INDIRECT_LIGHT_FILTERED):
            tiled = np.concatenate([p.readback(image) for p in parts], axis=0)
            assert np.array_equal(tiled, whole.readback(image))
    finally:
        whole.close()
        for p in parts:
            p.close()
