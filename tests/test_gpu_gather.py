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
    # ... and with the direct gather (RT64_GetGatherDirectHandle / RT64_SetGatherDirect from C): the same frame again
    c = json.loads(next(l for l in subprocess.run(base + ["--ranks", "1", "--direct"], env=env, capture_output=True, text=True, timeout=300, check=True).stdout.splitlines() if l.startswith("{")))
    assert (a["checksum"], a["fnv1a"]) == (c["checksum"], c["fnv1a"])


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


def test_direct_gather_on_a_world_of_one(rt64_lib, sample_data):
    """RT64_SetGatherDirect: the frames store their rows straight into the gather's frame slots (six of them, fine-grained device memory rank 0 exports over IPC) and
    RT64_SubmitGather exchanges a token instead of the rows.  On the one rank a test box has: slots come round 0 .. 5, every gathered frame is the frame the device
    drew (moving camera: no two are equal), frames overlap on the render streams as without a gather, no packed copy is made, and the gather goes back to the RCCL
    exchange of the rows (enable = 0) and on again."""
    import copy
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    W, H = 320, 180

    def camera(k):
        d = copy.copy(sample_data); v = np.array(sample_data.view, dtype=np.float32).copy(); v[3][0] += 0.2 * k; d.view = v
        return d
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    try:
        uid = (C.c_uint8 * rt64.GATHER_ID_BYTES)()
        assert rt64_lib.GetGatherUniqueId(uid, len(uid)) == 1, rt64_lib.last_error()
        g = rt64_lib.CreateGather(s.device, uid, len(uid), 0, 1, 0)
        assert g, rt64_lib.last_error()
        handle = (C.c_uint8 * 64)()
        assert rt64_lib.GetGatherDirectHandle(g, handle, 64) == 64, rt64_lib.last_error()
        assert rt64_lib.SetGatherDirect(g, handle, 64, 1) == 1, rt64_lib.last_error()
        s.option("sync_present", 0)
        slots, own = [], {}
        for k in range(9):
            s.data = camera(k)
            s.draw()
            slots.append(rt64_lib.SubmitGather(g))
            if k >= 6:
                own[k] = s.readback(rt64.IMAGE_FINAL_RGBA8).copy()
        assert slots == [0, 1, 2, 3, 4, 5, 0, 1, 2]
        st = s.stats()
        assert not st.packedFinal                                   # the rows are stored once, into the frame slot itself
        for k in (6, 7, 8):                                         # on rank 0 the frames of the last three submits are intact
            buf = np.zeros((H, W, 4), dtype=np.uint8)
            assert rt64_lib.ReadbackGather(g, slots[k], buf.ctypes.data, buf.nbytes, 0) == buf.nbytes, rt64_lib.last_error()
            assert np.array_equal(buf, own[k]), k
        assert not np.array_equal(own[7], own[8])
        assert rt64_lib.GetGatherFrame(g, slots[-1])
        for k in range(9, 13):                                      # a burst with no host-side wait in between: the frames alternate over the render streams
            s.data = camera(k); s.draw(); last = rt64_lib.SubmitGather(g)
        assert int(s.stats().overlappedFrame) == 1
        frame = s.readback(rt64.IMAGE_FINAL_RGBA8)
        buf = np.zeros((H, W, 4), dtype=np.uint8)
        assert rt64_lib.ReadbackGather(g, last, buf.ctypes.data, buf.nbytes, 0) == buf.nbytes and np.array_equal(buf, frame)
        # back to the RCCL exchange of the rows: two slots, send buffer written by the frame kernel
        assert rt64_lib.SetGatherDirect(g, None, 0, 0) == 1, rt64_lib.last_error()
        got = []
        for k in range(13, 16):
            s.data = camera(k); s.draw(); got.append(rt64_lib.SubmitGather(g))
        assert got == [0, 1, 0] and bool(s.stats().packedFinal)
        frame = s.readback(rt64.IMAGE_FINAL_RGBA8)
        assert rt64_lib.ReadbackGather(g, -1, buf.ctypes.data, buf.nbytes, 0) == buf.nbytes and np.array_equal(buf, frame)
        assert rt64_lib.SetGatherDirect(g, handle, 64, 1) == 1      # ... and on again
        s.data = camera(16); s.draw(); assert rt64_lib.SubmitGather(g) == 0
        rt64_lib.DestroyGather(g)
    finally:
        s.close()


_OWNER = r'''
import ctypes as C, sys, os
sys.path.insert(0, os.environ["RT64_REPO"])
import __graft_entry__ as graft
graft.load_package()
import numpy as np
from sm64rt_legacy_renderer_amd import rt64
lib = rt64.Library()
W, H = 320, 180
dev = lib.CreateDeviceHeadless(W, H, 0)
uid = (C.c_uint8 * rt64.GATHER_ID_BYTES)(); assert lib.GetGatherUniqueId(uid, len(uid)) == 1
g = lib.CreateGather(dev, uid, len(uid), 0, 1, 0); assert g, lib.last_error()
handle = (C.c_uint8 * 64)()
assert lib.GetGatherDirectHandle(g, handle, 64) == 64, lib.last_error()
assert lib.SetGatherDirect(g, handle, 64, 1) == 1, lib.last_error()
print("HANDLE " + bytes(handle).hex(), flush=True)
slot = int(sys.stdin.readline().split()[1])            # the renderer says which slot its last frame went to
for _ in range(slot + 1):
    assert lib.SubmitGather(g) >= 0                     # (a world of one: records this slot's events; nothing is drawn here)
buf = np.zeros((H, W, 4), dtype=np.uint8)
assert lib.ReadbackGather(g, slot, buf.ctypes.data, buf.nbytes, 0) == buf.nbytes, lib.last_error()
print("SUM %d %d" % (int(buf.astype(np.int64).sum()), int(buf[..., :3].max())), flush=True)
sys.stdin.readline()
lib.DestroyGather(g); lib.DestroyDevice(dev)
'''


def test_direct_gather_stores_into_another_process_through_ipc(rt64_lib, sample_data):
    """The peer half of the direct gather on one GPU: the frame slots live in ANOTHER process (an owner that exports them with RT64_GetGatherDirectHandle), this process maps
    them (RT64_SetGatherDirect(..., 2): hipIpcOpenMemHandle) and its frame kernels store the back buffer there.  The owner reads the slot it was told and reports the frame's byte
    sum: it is the frame this process drew.  (Between two GPUs the same stores travel over xGMI; that leg needs a multi-GPU node: the driver's scaling run.)"""
    import copy, sys
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    W, H = 320, 180
    env = dict(os.environ, RT64_REPO=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    owner = subprocess.Popen([sys.executable, "-c", _OWNER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env)
    s = None
    try:
        line = ""
        while not line.startswith("HANDLE "):
            line = owner.stdout.readline()
            assert line, "the owner process ended: " + str(owner.poll())
        raw = bytes.fromhex(line.split()[1])
        handle = (C.c_uint8 * 64)(*raw)
        s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
        uid = (C.c_uint8 * rt64.GATHER_ID_BYTES)()
        assert rt64_lib.GetGatherUniqueId(uid, len(uid)) == 1
        g = rt64_lib.CreateGather(s.device, uid, len(uid), 0, 1, 0)
        assert g, rt64_lib.last_error()
        assert rt64_lib.SetGatherDirect(g, handle, 64, 2) == 1, rt64_lib.last_error()      # 2: this rank 0 maps the handle too -- the slots are the owner's
        s.option("sync_present", 0)
        slot = -1
        for k in range(4):
            d = copy.copy(sample_data); v = np.array(sample_data.view, dtype=np.float32).copy(); v[3][0] += 0.3 * k; d.view = v; s.data = d
            s.draw(); slot = rt64_lib.SubmitGather(g)
        assert slot == 3
        mine = np.zeros((H, W, 4), dtype=np.uint8)
        assert rt64_lib.ReadbackGather(g, slot, mine.ctypes.data, mine.nbytes, 0) == mine.nbytes, rt64_lib.last_error()     # (waits for the frame; reads the mapped slot)
        assert mine[..., :3].max() > 0
        owner.stdin.write("SLOT %d\n" % slot); owner.stdin.flush()
        line = owner.stdout.readline()
        assert line.startswith("SUM "), line
        total, peak = int(line.split()[1]), int(line.split()[2])
        assert total == int(mine.astype(np.int64).sum()) and peak == int(mine[..., :3].max())
        rt64_lib.DestroyGather(g)
        owner.stdin.write("BYE\n"); owner.stdin.flush()
        assert owner.wait(timeout=60) == 0
    finally:
        if s:
            s.close()
        if owner.poll() is None:
            owner.kill()


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_gather_sessions_return_the_frames_the_device_drew(rt64_lib, sample_data, seed):
    """A random session of what a rank does around the gather -- enqueued frames + submits, bursts, camera and instance moves, switching between the exchange of the rows
    and the direct gather, sync_present flips, tables re-staged every frame -- on the one rank a test box has: every gathered frame that is read back (the latest slot, and in
    direct mode one of the last three) is byte for byte the frame the device drew for that submit."""
    import copy, random
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    W, H = 320, 180
    rng = random.Random(seed)
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    try:
        uid = (C.c_uint8 * rt64.GATHER_ID_BYTES)()
        assert rt64_lib.GetGatherUniqueId(uid, len(uid)) == 1, rt64_lib.last_error()
        g = rt64_lib.CreateGather(s.device, uid, len(uid), 0, 1, 0)
        assert g, rt64_lib.last_error()
        handle = (C.c_uint8 * 64)()
        assert rt64_lib.GetGatherDirectHandle(g, handle, 64) == 64, rt64_lib.last_error()
        direct = False
        history = []                                   # (slot, frame the device drew) of the submits, newest last
        checked = 0

        def submit():
            s.draw()
            slot = rt64_lib.SubmitGather(g)
            assert slot >= 0, rt64_lib.last_error()
            history.append((slot, s.readback(rt64.IMAGE_FINAL_RGBA8).copy()))
            del history[:-3]
        for _ in range(60):
            op = rng.choice(["submit", "submit", "submit", "burst", "camera", "move", "direct", "sync", "rebuild", "check", "check"])
            if op == "submit":
                submit()
            elif op == "burst":
                for _ in range(rng.randint(2, 7)):
                    d = copy.copy(s.data); v = np.array(d.view, dtype=np.float32).copy(); v[3][0] += 0.07; d.view = v; s.data = d
                    s.draw(); slot = rt64_lib.SubmitGather(g); assert slot >= 0
                history.clear(); history.append((slot, s.readback(rt64.IMAGE_FINAL_RGBA8).copy()))
            elif op == "camera":
                d = copy.copy(s.data); v = np.array(d.view, dtype=np.float32).copy(); v[3][0] += rng.uniform(-0.3, 0.3); v[3][1] += rng.uniform(-0.2, 0.2); d.view = v; s.data = d
            elif op == "move":
                d = copy.copy(s.data); d.instances = [copy.copy(i) for i in d.instances]
                k = rng.choice([i for i, inst in enumerate(d.instances) if inst.name in ("sphere", "floor")])
                t = np.array(d.instances[k].transform, dtype=np.float32).copy(); t[3][1] += rng.uniform(-0.2, 0.2)
                d.instances[k].previous_transform = d.instances[k].transform; d.instances[k].transform = t
                s.data = d; s.set_instance(k, d.instances[k])
            elif op == "direct":
                direct = not direct
                assert rt64_lib.SetGatherDirect(g, handle if direct else None, 64 if direct else 0, 1 if direct else 0) == 1, rt64_lib.last_error()
                history.clear()                        # (the slots of the other mode are gone)
            elif op == "sync":
                s.option("sync_present", rng.choice([0, 1]))
            elif op == "rebuild":
                s.option("always_rebuild", rng.choice([0, 1]))
            elif op == "check" and history:
                slot, frame = history[-1] if not direct else rng.choice(history)
                buf = np.zeros((H, W, 4), dtype=np.uint8)
                assert rt64_lib.ReadbackGather(g, slot, buf.ctypes.data, buf.nbytes, 0) == buf.nbytes, rt64_lib.last_error()
                assert np.array_equal(buf, frame), (seed, slot, direct)
                checked += 1
        submit()
        slot, frame = history[-1]
        buf = np.zeros((H, W, 4), dtype=np.uint8)
        assert rt64_lib.ReadbackGather(g, slot, buf.ctypes.data, buf.nbytes, 0) == buf.nbytes and np.array_equal(buf, frame)
        assert checked >= 1
        rt64_lib.DestroyGather(g)
    finally:
        s.close()
