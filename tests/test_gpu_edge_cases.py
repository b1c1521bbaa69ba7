"""Edge cases of the render path through the C ABI: ragged and tiny frame sizes, resizes, empty scenes, degenerate geometry,
light-count limits, object lifetime.  Parity bars as in test_gpu_parity.py (geometry bit-exact, shading within tolerance)."""
import copy
import ctypes as C

import numpy as np
import pytest

from test_gpu_features import _variant, _rmse

pytestmark = pytest.mark.gpu


def _pair(rt64_lib, data, w, h, frames=1, **kw):
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    s = sample_scene.Rt64Scene(rt64_lib, data, w, h, hip_device=0)
    o = oracle_py.OracleScene(data)
    try:
        s.option("count_traversal", 1)
        for f in range(frames):
            s.draw()
            ref = o.render(w, h, images=(f == frames - 1), **kw)
        got = {k: s.readback(getattr(rt64, "IMAGE_" + k)) for k in ("OUTPUT_RGBA32F", "FINAL_RGBA8", "PRIMARY_HIT", "INSTANCE_ID")}
        return got, ref, s.stats()
    finally:
        s.close(); o.close()


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (17, 16), (123, 77), (33, 200)])
def test_ragged_frame_sizes(rt64_lib, sample_data, w, h):
    """Sizes that are not multiples of the 16x16 / 32x8 tiles, down to a single pixel."""
    got, ref, st = _pair(rt64_lib, sample_data, w, h)
    assert got["PRIMARY_HIT"].shape == (h, w, 4)
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"]) and np.array_equal(got["INSTANCE_ID"], ref["instanceId"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    assert np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32)).max() <= 1
    assert st.primaryRays == w * h == ref["counters"]["primaryRays"] and st.shadowRays == ref["counters"]["shadowRays"]
    assert (st.nodesVisited, st.trianglesTested) == (ref["counters"]["nodesVisited"], ref["counters"]["trianglesTested"])


def test_resize_between_frames_recreates_the_images(rt64_lib, sample_data):
    """RT64_SetDeviceSize stands in for a window resize (rt64_device.cpp:1039): the next frame has the new size and no stale history."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, 160, 90, hip_device=0)
    try:
        s.draw()
        assert s.readback(rt64.IMAGE_FINAL_RGBA8).shape == (90, 160, 4)
        rt64_lib.SetDeviceSize(s.device, 208, 120)
        s.draw()
        out = s.readback(rt64.IMAGE_OUTPUT_RGBA32F)
        hit = s.readback(rt64.IMAGE_PRIMARY_HIT)
    finally:
        s.close()
    o = oracle_py.OracleScene(sample_data)
    try:
        ref = o.render(208, 120)
    finally:
        o.close()
    assert out.shape == (120, 208, 4) and np.array_equal(hit, ref["primaryHit"])
    assert _rmse(out[..., :3], ref["output"][..., :3]) <= 1e-3


def test_scene_without_instances_and_without_lights(rt64_lib, sample_data):
    """No instance at all: the cleared back buffer (rt64_device.cpp:996-997).  No light: only ambient + eye light remain."""
    def empty(d):
        d.instances = []
    got, ref, st = _pair(rt64_lib, _variant(sample_data, empty), 64, 48)
    assert st.primaryRays == 0 and (got["FINAL_RGBA8"] == np.array([0, 0, 0, 255], dtype=np.uint8)).all()
    assert np.array_equal(got["FINAL_RGBA8"], ref["final"])

    def dark(d):
        d.lights = []
    got, ref, st = _pair(rt64_lib, _variant(sample_data, dark), 96, 54)
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"]) and st.shadowRays == 0 == ref["counters"]["shadowRays"]
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3


def test_single_triangle_and_degenerate_triangles(rt64_lib, sample_data):
    """A one-triangle ray-traced mesh (LBVH with n = 1: root with one leaf and RT64_NO_CHILD) and a mesh with zero-area and
    duplicate triangles next to valid ones."""
    from sm64rt_legacy_renderer_amd import sample_scene

    def mod(d):
        one = np.zeros(3, dtype=sample_scene.VERTEX_DTYPE)
        one["position"] = [(-2.0, 0.2, 3.0, 1.0), (2.0, 0.2, 3.0, 1.0), (0.0, 3.0, 3.0, 1.0)]
        one["normal"] = (0.0, 0.0, 1.0); one["uv"] = [(0, 0), (1, 0), (0, 1)]; one["input1"] = 1.0
        d.meshes.append(sample_scene.MeshData("one", d.meshes[0].flags, one, np.array([0, 1, 2], dtype=np.uint32)))
        deg = np.zeros(6, dtype=sample_scene.VERTEX_DTYPE)
        deg["position"] = [(3.0, 0.1, 2.0, 1.0), (5.0, 0.1, 2.0, 1.0), (4.0, 2.0, 2.0, 1.0), (3.0, 0.1, 2.0, 1.0), (3.0, 0.1, 2.0, 1.0), (4.0, 1.0, 2.0, 1.0)]
        deg["normal"] = (0.0, 0.0, 1.0); deg["input1"] = 1.0
        idx = np.array([0, 1, 2, 3, 4, 5, 0, 0, 0, 0, 1, 2, 0, 1, 1], dtype=np.uint32)     # valid, zero-area (two equal corners), point, duplicate, line
        d.meshes.append(sample_scene.MeshData("deg", d.meshes[0].flags, deg, idx))
        for m in (len(d.meshes) - 2, len(d.meshes) - 1):
            i = copy.copy(d.instances[1]); i.mesh = m; i.material = sample_scene.copy_material(d.instances[1].material); i.name = "extra%d" % m
            i.flags = 2                                                   # both faces
            d.instances.append(i)
    got, ref, st = _pair(rt64_lib, _variant(sample_data, mod), 160, 90)
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert set(np.unique(got["INSTANCE_ID"])) >= {2, 3}                   # both extra instances are visible
    assert (st.nodesVisited, st.trianglesTested) == (ref["counters"]["nodesVisited"], ref["counters"]["trianglesTested"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3


def test_more_lights_than_the_shader_keeps(rt64_lib, sample_data):
    """RT64_SetSceneLights with more than MAX_LIGHTS (16, Lights.hlsli:25) entries and maxLights = 4: the random selection without
    replacement (Lights.hlsli:115-168) picks the same lights on both sides."""
    from sm64rt_legacy_renderer_amd import rt64

    def mod(d):
        base = d.lights[0]
        ls = []
        for k in range(24):
            l = rt64.LIGHT(); C.memmove(C.byref(l), C.byref(base), C.sizeof(rt64.LIGHT))
            l.position = rt64.VECTOR3(-12.0 + k, 6.0 + (k % 3), 4.0 - (k % 5)); l.diffuseColor = rt64.VECTOR3(0.2 + 0.03 * k, 0.5, 0.9 - 0.03 * k)
            ls.append(l)
        d.lights = ls
    from test_gpu_features import _render_pair
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod), view_desc=dict(max_lights=4))
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert st.shadowRays == ref["counters"]["shadowRays"] > 0
    dd = np.abs(got["DIRECT_LIGHT_RAW"][..., :3] - ref["directLight"][..., :3]).max(axis=2)
    assert (dd > 1e-2).mean() < 2e-3


def test_destroying_an_instance_and_a_view_between_frames(rt64_lib, sample_data):
    """Object lifetime through the ABI: DestroyInstance removes it from the next frame; a second view / scene can be created and
    destroyed without disturbing the first (rt64_scene.cpp:43-51)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, 128, 72, hip_device=0)
    try:
        s.draw()
        extra_scene = rt64_lib.CreateScene(s.device); extra_view = rt64_lib.CreateView(extra_scene)
        rt64_lib.DestroyView(extra_view); rt64_lib.DestroyScene(extra_scene)
        k = next(i for i, inst in enumerate(sample_data.instances) if inst.name == "sphere")
        rt64_lib.DestroyInstance(s.instances[k]); s.instances[k] = None; s._sphere_k = -1
        s.draw()
        ids = s.readback(rt64.IMAGE_INSTANCE_ID)
        hit = s.readback(rt64.IMAGE_PRIMARY_HIT)
    finally:
        s.instances = [h for h in s.instances if h]
        s.close()
    d = _variant(sample_data, lambda dd: None); d.instances = [i for i in d.instances if i.name != "sphere"]
    o = oracle_py.OracleScene(d)
    try:
        ref = o.render(128, 72)
    finally:
        o.close()
    assert np.array_equal(hit, ref["primaryHit"]) and set(np.unique(ids)) == {-1, 0}


def test_many_meshes_set_between_frames_build_in_one_batch(rt64_lib, sample_data):
    """RT64_SetMesh records the BLAS work, RT64_DrawDevice runs it for all meshes together (one workgroup per small tree, large trees on
    the multi-kernel path); a mesh set twice before a frame, refits of UPDATABLE meshes and rebuilds after a size change all end in the
    tree the oracle builds."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    rng = np.random.default_rng(3)

    def blob(cx, cz, n_tri, r=0.35):
        v = np.zeros(3 * n_tri, dtype=sample_scene.VERTEX_DTYPE)
        c = np.array([cx, 0.6, cz]) + rng.normal(0, r, size=(n_tri, 1, 3))
        p = c + rng.normal(0, 0.12, size=(n_tri, 3, 3))
        v["position"][:, :3] = p.reshape(-1, 3).astype(np.float32); v["position"][:, 3] = 1.0
        v["normal"] = (0.0, 1.0, 0.0); v["input1"] = 1.0; v["uv"] = rng.random((3 * n_tri, 2)).astype(np.float32)
        return v, np.arange(3 * n_tri, dtype=np.uint32)

    def mod(d):
        # index 11 (6000 triangles) is UPDATABLE and refitted in frame 1: the multi-kernel path's refit (leaf boxes + the chunked bottom-up fit on the kept topology)
        sizes = [1, 2, 7, 64, 300, 1024, 1500, 4096, 5000, 33, 77, 6000] + [int(x) for x in rng.integers(3, 200, size=21)]
        for k, n in enumerate(sizes):
            v, i = blob(-6.0 + 1.5 * (k % 9), -2.0 + 1.5 * (k // 9), n)
            d.meshes.append(sample_scene.MeshData("blob%d" % k, rt64.MESH_RAYTRACE_ENABLED | (rt64.MESH_RAYTRACE_UPDATABLE if k % 2 else 0), v, i))
            inst = copy.copy(d.instances[1]); inst.mesh = len(d.meshes) - 1; inst.material = sample_scene.copy_material(d.instances[1].material)
            inst.name = "blob%d" % k; inst.flags = 2
            d.instances.append(inst)
    data = _variant(sample_data, mod)
    s = sample_scene.Rt64Scene(rt64_lib, data, 160, 90, hip_device=0)
    o = oracle_py.OracleScene(data)
    try:
        s.option("count_traversal", 1)
        first = len(sample_data.meshes)
        for frame in range(3):
            for k in range(first, len(data.meshes)):
                m = data.meshes[k]
                if frame == 0:
                    continue
                v = m.vertices.copy()
                if (k + frame) % 3 == 0:                                   # same shape: refit when UPDATABLE, rebuild otherwise
                    v["position"][:, 1] += np.float32(0.05 * frame)
                    s.set_mesh(s.meshes[k], v, m.indices); o.set_mesh(o.meshes[k], v, m.indices)
                elif (k + frame) % 3 == 1 and len(m.indices) >= 6:        # fewer triangles: rebuild; set twice before the frame
                    keep = (len(m.indices) // 3 - 1) * 3
                    s.set_mesh(s.meshes[k], v, m.indices); o.set_mesh(o.meshes[k], v, m.indices)
                    s.set_mesh(s.meshes[k], v[:keep], m.indices[:keep]); o.set_mesh(o.meshes[k], v[:keep], m.indices[:keep])
            s.draw()
            ref = o.render(160, 90)
        hit = s.readback(rt64.IMAGE_PRIMARY_HIT)
        st = s.stats()
        assert np.array_equal(hit, ref["primaryHit"])
        assert (st.nodesVisited, st.trianglesTested) == (ref["counters"]["nodesVisited"], ref["counters"]["trianglesTested"])
        assert len(np.unique(s.readback(rt64.IMAGE_INSTANCE_ID))) > 12
    finally:
        s.close(); o.close()


def test_five_thousand_instances_use_the_large_tlas_builder(rt64_lib, sample_data):
    """More than 4096 ray-traced instances: the TLAS goes through the multi-kernel LBVH path (radix sort over instance boxes) and is
    bit-identical to the oracle's; the frame's hits and traversal counters follow."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    rng = np.random.default_rng(9)

    def mod(d):
        tri = np.zeros(3, dtype=sample_scene.VERTEX_DTYPE)
        tri["position"] = [(-0.06, 0.0, 0.0, 1.0), (0.06, 0.0, 0.0, 1.0), (0.0, 0.1, 0.0, 1.0)]
        tri["normal"] = (0.0, 0.0, 1.0); tri["input1"] = 1.0; tri["uv"] = [(0, 0), (1, 0), (0, 1)]
        d.meshes.append(sample_scene.MeshData("chip", rt64.MESH_RAYTRACE_ENABLED, tri, np.array([0, 1, 2], dtype=np.uint32)))
        base = d.instances[1]
        for k in range(5000):
            inst = copy.copy(base); inst.mesh = len(d.meshes) - 1; inst.material = sample_scene.copy_material(base.material); inst.name = "chip%d" % k
            t = np.eye(4, dtype=np.float32)
            t[3, :3] = (rng.uniform(-7, 7), rng.uniform(0.1, 4.5), rng.uniform(-3, 5))
            inst.transform = t; inst.previous_transform = t; inst.flags = 2
            d.instances.append(inst)
    data = _variant(sample_data, mod)
    s = sample_scene.Rt64Scene(rt64_lib, data, 160, 90, hip_device=0)
    o = oracle_py.OracleScene(data)
    try:
        s.option("count_traversal", 1)
        s.draw()
        ref = o.render(160, 90)
        assert s.stats().instanceCount == 5002
        hit = s.readback(rt64.IMAGE_PRIMARY_HIT)
        assert np.array_equal(hit, ref["primaryHit"])
        st = s.stats()
        assert (st.nodesVisited, st.trianglesTested) == (ref["counters"]["nodesVisited"], ref["counters"]["trianglesTested"])
        nodes = np.empty(5001 * 64, dtype=np.uint8)
        n = rt64_lib.ReadbackViewAccel(s.view, rt64.ACCEL_NODES, nodes.ctypes.data, nodes.nbytes)
        assert n == nodes.nbytes
        tl = oracle_py.bvh_to_numpy(o.L.oracle_scene_tlas(o.scene))
        assert np.array_equal(nodes.view(np.uint8), tl["nodes"].view(np.uint8).reshape(-1))
    finally:
        s.close(); o.close()


def test_soak_random_frame_sequence_ends_like_a_fresh_render(rt64_lib, sample_data):
    """120 frames of seeded random host activity on ONE device -- instance transforms and materials changing, the view description
    hopping between pixel-local frames, GI + SVGF, GI without a filter and soft shadows, mesh re-uploads, image
    readbacks in between (G-buffer rebuilds on demand), enqueued and synchronous frames -- then the scene is put into a known state
    and one pixel-local frame is rendered.  That frame must equal, byte for byte, what a fresh device renders from the same state:
    none of the caches in between (frame tables, raster lists, LDS scene cache layout, lean / fused frame bookkeeping, upload ring,
    temporal history) may leak into it."""
    import copy
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    Wd, Hd = 224, 128
    rng = np.random.default_rng(20261004)
    data = copy.copy(sample_data)
    data.instances = [copy.copy(i) for i in sample_data.instances]
    for i in data.instances:
        i.material = sample_scene.copy_material(i.material)
    data.meshes = [copy.copy(m) for m in sample_data.meshes]
    k_sphere = next(i for i, inst in enumerate(data.instances) if inst.name == "sphere")
    base_vertices = data.meshes[data.instances[k_sphere].mesh].vertices.copy()
    images = [rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_INSTANCE_ID, rt64.IMAGE_DIFFUSE, rt64.IMAGE_DIRECT_LIGHT_RAW, rt64.IMAGE_SHADING_NORMAL,
              rt64.IMAGE_DEPTH, rt64.IMAGE_FLOW, rt64.IMAGE_PRIMARY_HIT]
    views = [dict(), dict(gi_samples=1, denoiser=True), dict(gi_samples=2, denoiser=False), dict(di_samples=2), dict(gi_samples=1, denoiser=True, di_samples=1)]

    def translate(dx, dy, dz):
        m = np.eye(4, dtype=np.float32); m[3, :3] = (dx, dy, dz); return m

    s = sample_scene.Rt64Scene(rt64_lib, data, Wd, Hd, hip_device=0)
    try:
        for frame in range(120):
            op = rng.integers(0, 10)
            if op <= 2:                                  # the sphere moves (tables change: upload + TLAS rebuild)
                inst = copy.copy(data.instances[k_sphere])
                inst.previous_transform = inst.transform
                inst.transform = translate(*(rng.uniform(-1.5, 1.5, 3) * (1, 0.3, 1)))
                data.instances[k_sphere] = inst
                s.set_instance(k_sphere, inst)
            elif op == 3:                                # material change on a random ray-traced instance
                k = int(rng.integers(0, len(data.instances)))
                inst = copy.copy(data.instances[k]); inst.material = sample_scene.copy_material(inst.material)
                inst.material.specularExponent = float(rng.uniform(1, 40)); inst.material.selfLight.x = float(rng.uniform(0, 0.2))
                data.instances[k] = inst
                s.set_instance(k, inst)
            elif op == 4:                                # another kind of frame
                s.set_view_description(**views[int(rng.integers(0, len(views)))])
            elif op == 5:                                # re-upload of the sphere mesh (same counts)
                v = base_vertices.copy(); v["position"][:, :3] *= np.float32(rng.uniform(0.9, 1.1))
                s.set_mesh(s.meshes[data.instances[k_sphere].mesh], v, data.meshes[data.instances[k_sphere].mesh].indices)
            elif op == 6:
                s.option("sync_present", int(rng.integers(0, 2)))
            elif op == 7:
                s.readback(images[int(rng.integers(0, len(images)))])
            elif op == 8:
                s.option("fused_lean", int(rng.integers(0, 2)))
            s.draw(can_reproject=bool(rng.integers(0, 4)))
        # known state
        s.option("sync_present", 1); s.option("fused_lean", 1)
        s.set_view_description()
        s.set_mesh(s.meshes[data.instances[k_sphere].mesh], base_vertices, data.meshes[data.instances[k_sphere].mesh].indices)
        final_inst = copy.copy(data.instances[k_sphere]); final_inst.transform = translate(0.25, 0.0, -0.5); final_inst.previous_transform = final_inst.transform
        data.instances[k_sphere] = final_inst
        s.set_instance(k_sphere, final_inst)
        s.draw()
        after_soak = [s.readback(i) for i in images]
    finally:
        s.close()
    f = sample_scene.Rt64Scene(rt64_lib, data, Wd, Hd, hip_device=0)
    try:
        f.draw()
        fresh = [f.readback(i) for i in images]
    finally:
        f.close()
    for name, a, b in zip(images, after_soak, fresh):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), "image %d differs after the soak" % name


def _device_free_bytes():
    """hipMemGetInfo through the runtime the library itself is linked against (no torch in this process)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return int(free.value)


def test_device_memory_comes_back_and_does_not_creep(rt64_lib, sample_data):
    """Nothing accumulates on the device: (1) two thousand frames of a host that re-sends everything every frame (always_rebuild), moves an instance, re-sends a mesh,
    flips between pixel-local and GI + SVGF frames, changes the partition and reads images back leave the device's free memory where it was after the first hundred
    (table slots, the upload ring, raster lists, scene-cache images, spill slabs and per-stream back buffers are all reused, not re-allocated); (2) creating and
    destroying the whole scene twenty times returns every byte (RT64_Destroy* free what RT64_Create* / RT64_SetMesh / RT64_CreateTexture allocated)."""
    import copy
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    rng = np.random.default_rng(7)
    data = copy.copy(sample_data)
    data.instances = [copy.copy(i) for i in sample_data.instances]
    data.meshes = [copy.copy(m) for m in sample_data.meshes]
    data.meshes[0] = sample_scene.MeshData(data.meshes[0].name, data.meshes[0].flags | rt64.MESH_RAYTRACE_UPDATABLE, data.meshes[0].vertices.copy(), data.meshes[0].indices)
    k_sphere = next(i for i, inst in enumerate(data.instances) if inst.name == "sphere")
    before_all = _device_free_bytes()
    s = sample_scene.Rt64Scene(rt64_lib, data, 320, 180, hip_device=0)
    try:
        s.option("always_rebuild", 1)
        mark = None
        for frame in range(2000):
            if frame % 3 == 0:
                inst = copy.copy(data.instances[k_sphere]); t = np.array(inst.transform, dtype=np.float32).copy(); t[3][0] = np.float32(rng.uniform(-1, 1))
                inst.previous_transform = inst.transform; inst.transform = t; data.instances[k_sphere] = inst; s.set_instance(k_sphere, inst)
            if frame % 50 == 7:
                v = data.meshes[0].vertices.copy(); v["position"][:, :3] *= np.float32(rng.uniform(0.98, 1.02)); s.set_mesh(s.meshes[0], v, data.meshes[0].indices)
            if frame % 100 == 40:
                s.set_interleave(0, 1); s.set_view_description(gi_samples=1, denoiser=True)
            if frame % 100 == 60:
                s.set_view_description(gi_samples=0, denoiser=False); s.set_interleave(int(rng.integers(0, 3)), 3)
            if frame % 37 == 0:
                s.option("sync_present", int(rng.integers(0, 2)))
            s.draw()
            if frame % 97 == 0:
                s.readback(rt64.IMAGE_PRIMARY_HIT); s.readback(rt64.IMAGE_FINAL_RGBA8)
            if frame == 300:                                  # every kind of frame of the cycle has run three times: the working set exists
                s.option("sync_present", 1); s.draw(); mark = _device_free_bytes()
        s.option("sync_present", 1); s.draw()
        end = _device_free_bytes()
        assert mark is not None and mark - end <= (2 << 20), (mark, end)          # (the allocator's own granules)
    finally:
        s.close()
    # What stays after the first device of a process is the HIP runtime's own (code objects, the hardware queues' scratch: ~0.85 GB here); a second, third ... device adds nothing.
    base = _device_free_bytes()
    for _ in range(20):
        t = sample_scene.Rt64Scene(rt64_lib, sample_data, 320, 180, hip_device=0)
        t.draw(); t.close()
    after_all = _device_free_bytes()
    assert base - after_all <= (8 << 20), (before_all, base, after_all)
