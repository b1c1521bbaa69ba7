"""Two render streams (device option overlap_frames): enqueued pixel-local frames alternate between two HIP streams so that frame
k + 1 starts beside the tail of frame k.  Every frame must be the frame the one-stream path renders, byte for byte -- whichever stream and
whichever of the two back buffers it landed in -- and anything that is not pixel-local (an upload, a frame with history, a readback) must
fall back into order."""
import copy
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 320, 180


def _camera(data, k):
    """Frame k's view matrix: the sample camera, moved a little every frame (so that no two frames are the same image)."""
    d = copy.copy(data)
    v = np.array(data.view, dtype=np.float32).copy()
    v[3][0] += 0.25 * k
    v[3][1] -= 0.1 * k
    d.view = v
    return d


def _run(rt64_lib, sample_data, overlap, lds_cache, frames, change_at=None, groups=0, interleave=None):
    """Draw `frames` enqueued frames with a moving camera; returns the back buffer after every frame listed in `check` plus the per-frame stats flags."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    finals, flags = {}, []
    try:
        assert s.option("lds_cache", lds_cache) and s.option("overlap_frames", overlap)
        if groups:
            assert s.option("max_frame_groups", groups)
        if interleave:
            s.set_interleave(*interleave)
        s.draw()                       # first frame: uploads + builds (synchronous)
        s.option("sync_present", 0)
        for k in range(frames):
            s.data = _camera(sample_data, k)
            if change_at is not None and k == change_at:      # a table change in the middle of the run: that frame uploads, so it has to run behind its predecessor
                d2 = copy.copy(s.data)
                d2.instances = [copy.copy(i) for i in d2.instances]
                t = np.array(d2.instances[0].transform, dtype=np.float32).copy(); t[3][1] += 0.5
                d2.instances[0].transform = t; d2.instances[0].previous_transform = t
                s.data = d2
                s.set_instance(0, d2.instances[0])
                sample_data_moved = d2
            elif change_at is not None and k > change_at:
                d2 = _camera(sample_data_moved, k); s.data = d2
            s.draw()
            if k in (frames - 1, frames - 2, 1):
                flags.append(int(s.stats().overlappedFrame))           # (reading the stats waits for the frame)
                finals[k] = s.readback(rt64.IMAGE_FINAL_RGBA8).copy()
        # several frames back to back with no host-side wait in between, then only the last one is looked at
        for k in range(frames, frames + 6):
            s.data = _camera(sample_data, k) if change_at is None else _camera(sample_data_moved, k)
            s.draw()
        st = s.stats()
        flags.append(int(st.overlappedFrame))
        finals["burst"] = s.readback(rt64.IMAGE_FINAL_RGBA8).copy()
        hit = s.readback(rt64.IMAGE_PRIMARY_HIT).copy()               # materialise of an overlapped frame: the FULL variant re-traces the LAST frame
        return finals, flags, hit
    finally:
        s.close()


@pytest.mark.parametrize("lds_cache,groups", [(1, 0), (0, 0), (0, 20)])
def test_overlapped_frames_are_the_one_stream_frames(rt64_lib, sample_data, lds_cache, groups):
    """C2-style frames (one-kernel lean frame; lds_cache = 0 puts the sample scene on the per-wave form with the cost-ordered tiles, whose order arrays are per
    stream) with the camera moving every frame: enqueued on two alternating streams or on one, every back buffer that is read is the same."""
    a, fa, ha = _run(rt64_lib, sample_data, 1, lds_cache, 7, groups=groups)
    b, fb, hb = _run(rt64_lib, sample_data, 0, lds_cache, 7, groups=groups)
    assert fa[-1] == 1 and any(fa), fa          # the overlapped run really alternated streams ...
    assert not any(fb), fb                      # ... and the reference run never did
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(ha.view(np.uint8), hb.view(np.uint8))


def test_a_table_change_between_overlapped_frames_runs_in_order(rt64_lib, sample_data):
    """An instance moves in the middle of an overlapped run: that frame uploads new tables and a new TLAS, so it is joined behind the frame before it and the one
    after it does not start before the upload -- the images equal the one-stream run's."""
    a, fa, ha = _run(rt64_lib, sample_data, 1, 1, 8, change_at=4)
    b, fb, hb = _run(rt64_lib, sample_data, 0, 1, 8, change_at=4)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(ha.view(np.uint8), hb.view(np.uint8))


def test_overlapped_strips_fill_both_gather_slots(rt64_lib, sample_data):
    """The N > 1 loop on a world of one: RT64_DrawDevice + RT64_SubmitGather with frames enqueued on alternating streams, each frame writing its packed rows into its own
    slot's send buffer.  The assembled frames of the last two slots are the last two frames."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    out = {}
    for overlap in (1, 0):
        s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
        try:
            assert s.option("overlap_frames", overlap)
            uid = (C.c_uint8 * rt64.GATHER_ID_BYTES)()
            assert rt64_lib.GetGatherUniqueId(uid, len(uid)) == 1, rt64_lib.last_error()
            g = rt64_lib.CreateGather(s.device, uid, len(uid), 0, 1, 0)
            assert g, rt64_lib.last_error()
            s.option("sync_present", 0)
            slots = []
            for k in range(9):
                s.data = _camera(sample_data, k)
                s.draw()
                slots.append(rt64_lib.SubmitGather(g))
            st = s.stats()
            assert bool(st.packedFinal) and int(st.overlappedFrame) == overlap
            frames = []
            for slot in (slots[-2], slots[-1]):
                buf = np.zeros((H, W, 4), dtype=np.uint8)
                assert rt64_lib.ReadbackGather(g, slot, buf.ctypes.data, buf.nbytes, 0) == buf.nbytes, rt64_lib.last_error()
                frames.append(buf)
            assert not np.array_equal(frames[0], frames[1])          # the camera moved between them
            out[overlap] = frames
            rt64_lib.DestroyGather(g)
        finally:
            s.close()
    for x, y in zip(out[1], out[0]):
        assert np.array_equal(x, y)


def _random_session(rt64_lib, sample_data, overlap, seed, ops=45, log=None):
    """A random session of the calls a host makes between and around frames -- enqueued and waited-for frames, camera moves, an instance that moves (table upload + TLAS),
    a mesh that is re-sent (BLAS refit), GI switched on and off (frames with history), raster instances that change, everything re-staged every frame (always_rebuild), a
    strip partition that changes, readbacks of several images, picking, option changes -- returning everything read."""
    import random
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    rng = random.Random(seed)
    data = copy.copy(sample_data)
    data.instances = [copy.copy(i) for i in sample_data.instances]
    data.meshes = [copy.copy(m) for m in sample_data.meshes]
    data.meshes[0] = sample_scene.MeshData(data.meshes[0].name, data.meshes[0].flags | rt64.MESH_RAYTRACE_UPDATABLE, data.meshes[0].vertices.copy(), data.meshes[0].indices)
    s = sample_scene.Rt64Scene(rt64_lib, data, W, H, hip_device=0)
    out = []
    try:
        assert s.option("overlap_frames", overlap)
        frame = 0; gi = 0; scale = 1.0
        for _ in range(ops):
            op = rng.choice(["draw", "draw", "draw", "draw", "burst", "camera", "move", "mesh", "gi", "sync", "read", "read_gbuffer", "pick", "lds", "rebuild", "hud", "strips", "prologue", "light", "resize", "scale", "texture"])
            if log is not None:
                log.append((op, frame))
            if op == "draw":
                s.draw(); frame += 1
            elif op == "burst":
                for _ in range(rng.randint(2, 5)):
                    d = copy.copy(s.data); v = np.array(d.view, dtype=np.float32).copy(); v[3][0] += 0.1; d.view = v; s.data = d
                    s.draw(); frame += 1
            elif op == "camera":
                d = copy.copy(s.data); v = np.array(d.view, dtype=np.float32).copy(); v[3][0] += rng.uniform(-0.3, 0.3); v[3][1] += rng.uniform(-0.2, 0.2); d.view = v; s.data = d
            elif op == "move":
                d = copy.copy(s.data); d.instances = [copy.copy(i) for i in d.instances]
                k = rng.choice([i for i, inst in enumerate(d.instances) if inst.name in ("sphere", "floor")])
                t = np.array(d.instances[k].transform, dtype=np.float32).copy(); t[3][1] += rng.uniform(-0.2, 0.2)
                d.instances[k].previous_transform = d.instances[k].transform; d.instances[k].transform = t
                s.data = d; s.set_instance(k, d.instances[k])
            elif op == "mesh":
                v = data.meshes[0].vertices.copy(); v["position"][:, :3] *= np.float32(1.0 + rng.uniform(-0.02, 0.02))
                s.set_mesh(s.meshes[0], v, data.meshes[0].indices)
            elif op == "gi":
                gi = rng.choice([0, 1]); s.set_view_description(gi_samples=gi, denoiser=bool(gi), resolution_scale=scale)
                if gi: s.set_interleave(0, 1)                                  # (a frame that filters across rows is not cut into strips: the library refuses it)
            elif op == "sync":
                s.option("sync_present", rng.choice([0, 1]))
            elif op == "lds":
                s.option("lds_cache", rng.choice([0, 1]))
            elif op == "rebuild":
                s.option("always_rebuild", rng.choice([0, 1]))                 # tables, TLAS and raster lists re-staged every frame (the reference's behaviour)
            elif op == "prologue":
                s.option("frame_prologue", rng.choice([0, 1]))
            elif op == "hud":                                                  # a raster instance changes: its list is set up again (and gBackground redrawn, for a background instance)
                d = copy.copy(s.data); d.instances = [copy.copy(i) for i in d.instances]
                k = rng.choice([i for i, inst in enumerate(d.instances) if inst.name.startswith("hud")])
                d.instances[k].scissor = rng.choice([None, (rng.randint(0, W // 2), rng.randint(0, H // 2), rng.randint(8, W // 2), rng.randint(8, H // 2))])
                s.data = d; s.set_instance(k, d.instances[k])
            elif op == "light":                                                # the light table changes (same bytes for both sessions)
                s._lights[0].position.y = s._lights[0].position.y + rng.uniform(-0.3, 0.3); s._lights[0].diffuseColor.x = rng.uniform(0.4, 1.0)
            elif op == "resize":                                               # RT64_SetDeviceSize: takes effect at the next frame, every image is created again
                w, h = rng.choice([(W, H), (272, 150), (200, 112)]); rt64_lib.SetDeviceSize(s.device, w, h); s.set_interleave(0, 1)
            elif op == "scale":
                scale = rng.choice([1.0, 1.0, 0.5]); s.set_view_description(gi_samples=gi, denoiser=bool(gi), resolution_scale=scale)
                if scale != 1.0: s.set_interleave(0, 1)
            elif op == "texture":                                              # the sphere / floor takes another diffuse texture of the scene: the texture slots of the frame change
                d = copy.copy(s.data); d.instances = [copy.copy(i) for i in d.instances]
                k = rng.choice([i for i, inst in enumerate(d.instances) if inst.name in ("sphere", "floor")])
                d.instances[k].diffuse = rng.choice([t for t in range(len(d.textures)) if t != d.sky])
                s.data = d; s.set_instance(k, d.instances[k])
            elif op == "strips" and not gi and scale == 1.0:
                n = rng.choice([1, 2, 3]); s.set_interleave(rng.randrange(n), n)
            elif frame and op == "read":
                out.append(("final", frame, s.readback(rt64.IMAGE_FINAL_RGBA8).copy()))
            elif frame and op == "read_gbuffer":
                out.append(("hit", frame, s.readback(rt64.IMAGE_PRIMARY_HIT).copy())); out.append(("out", frame, s.readback(rt64.IMAGE_OUTPUT_RGBA32F).copy()))
            elif frame and op == "pick":
                out.append(("pick", frame, np.array([int(bool(rt64_lib.GetViewRaytracedInstanceAt(s.view, W // 2, H // 2)))])))
        s.draw()
        out.append(("final", -1, s.readback(rt64.IMAGE_FINAL_RGBA8).copy())); out.append(("hit", -1, s.readback(rt64.IMAGE_PRIMARY_HIT).copy()))
        return out
    finally:
        s.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_call_sequences_render_the_same_on_one_stream_and_on_three(rt64_lib, sample_data, seed):
    """Whatever a host does between frames, the render streams must not show: a random session of draws, bursts of enqueued frames, camera and instance moves, mesh refits, GI on and
    off, sync_present flips, readbacks and picking returns byte for byte what the same session returns with overlap_frames = 0."""
    a = _random_session(rt64_lib, sample_data, 1, seed)
    b = _random_session(rt64_lib, sample_data, 0, seed)
    assert len(a) == len(b) > 1
    for (ka, fa, xa), (kb, fb, xb) in zip(a, b):
        assert (ka, fa) == (kb, fb)
        assert np.array_equal(xa.view(np.uint8), xb.view(np.uint8)), (seed, ka, fa)


def test_an_instance_that_moves_every_frame_keeps_the_frames_lean_and_side_by_side(rt64_lib, sample_data):
    """What a game does: some instance moves in every frame, so the frame tables (and the TLAS, and the head of the scene-cache image) change in every frame.  Changed tables
    go into the next of the view's table slots (one per render stream) instead of over the ones the frame before was rendered with: that frame -- possibly still running on
    another stream, and kept implicit as a lean frame -- needs neither a wait nor a materialise, the frames stay one-kernel lean frames and stay overlapped.  Byte for byte the
    frames of the same session drawn synchronously on one stream; a G-buffer image of the LAST frame read afterwards is that frame's (its slot was not overwritten)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    out = {}
    for overlap, sync in ((1, 0), (0, 1)):
        data = copy.copy(sample_data)
        data.instances = [copy.copy(i) for i in sample_data.instances]
        s = sample_scene.Rt64Scene(rt64_lib, data, W, H, hip_device=0)
        try:
            assert s.option("overlap_frames", overlap)
            s.draw()
            s.option("sync_present", sync)
            k = next(i for i, inst in enumerate(data.instances) if inst.name == "sphere")
            finals, flags = [], []
            for f in range(9):
                d = copy.copy(s.data); d.instances = [copy.copy(i) for i in d.instances]
                t = np.array(d.instances[k].transform, dtype=np.float32).copy(); t[3][0] = 0.15 * f; t[3][1] = 0.05 * f
                d.instances[k].previous_transform = d.instances[k].transform; d.instances[k].transform = t
                s.data = d                                   # (Rt64Scene.draw re-sends the sphere's descriptor from s.data like the sample host, main.cpp:129)
                s.draw()
                if f >= 6:
                    st = s.stats()
                    flags.append((int(st.leanFrame), int(st.fusedFrame), int(st.overlappedFrame)))
                    finals.append(s.readback(rt64.IMAGE_FINAL_RGBA8).copy())
            for f in range(9, 14):                           # a burst with no wait in between
                d = copy.copy(s.data); d.instances = [copy.copy(i) for i in d.instances]
                t = np.array(d.instances[k].transform, dtype=np.float32).copy(); t[3][0] = 0.15 * f
                d.instances[k].previous_transform = d.instances[k].transform; d.instances[k].transform = t
                s.data = d; s.draw()
            st = s.stats()
            flags.append((int(st.leanFrame), int(st.fusedFrame), int(st.overlappedFrame)))
            finals.append(s.readback(rt64.IMAGE_FINAL_RGBA8).copy())
            hit = s.readback(rt64.IMAGE_PRIMARY_HIT).copy(); pos = s.readback(rt64.IMAGE_SHADING_POSITION).copy()
            out[overlap] = (finals, flags, hit, pos)
        finally:
            s.close()
    a, b = out[1], out[0]
    assert all(f[0] == 1 and f[1] == 1 for f in a[1] + b[1]), (a[1], b[1])        # every frame a one-kernel lean frame although its tables changed
    assert a[1][-1][2] == 1 and not any(f[2] for f in b[1])                        # ... and the enqueued ones ran side by side
    for x, y in zip(a[0], b[0]):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[0][0], a[0][1])
    assert np.array_equal(a[2].view(np.uint8), b[2].view(np.uint8)) and np.array_equal(a[3].view(np.uint8), b[3].view(np.uint8))


@pytest.mark.parametrize("overlap", [1, 0])
def test_a_refused_frame_leaves_the_last_complete_frame_current(rt64_lib, sample_data, overlap):
    """RT64_DrawDevice refuses a frame with GI + denoiser on interleaved strips (the filter reads across rows).  The refusal comes after the frame has taken the next
    render stream and its back-buffer slot: the device must go back to the stream and slot of the last complete frame, so that a readback returns that frame -- and
    the frames after the refusal render as if it had not happened."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    try:
        assert s.option("overlap_frames", overlap)
        s.option("sync_present", 0)
        frames = []
        for k in range(3):
            d = copy.copy(s.data); v = np.array(d.view, dtype=np.float32).copy(); v[3][0] += 0.15; d.view = v; s.data = d
            s.draw(); frames.append(s.readback(rt64.IMAGE_FINAL_RGBA8).copy())
        for k in range(2):                       # two more enqueued frames, so that the refused one follows a pure frame on another stream
            s.draw()
        last = s.readback(rt64.IMAGE_FINAL_RGBA8).copy()
        s.draw(); s.draw()
        s.set_view_description(gi_samples=1, denoiser=True); s.set_interleave(1, 3)
        s.draw()
        assert "interleaved strips" in rt64_lib.last_error()
        s.set_interleave(0, 1)
        assert np.array_equal(s.readback(rt64.IMAGE_FINAL_RGBA8), last)          # the whole last complete frame, not the slot the refused frame had taken
        s.set_view_description(gi_samples=0, denoiser=False)
        s.draw()
        assert np.array_equal(s.readback(rt64.IMAGE_FINAL_RGBA8), last)          # same camera, same scene: the same frame again
    finally:
        s.close()
