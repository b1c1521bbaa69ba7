"""GPU parity on variants of the sample scene that exercise the paths the stock scene leaves idle: the sorted per-pixel hit
list (non-opaque instances), transparent-geometry lighting, alpha shadows, texture-edge any-hit, back-face cull disable,
depth bias, fog, reflection / refraction chains, GI bounce + temporal accumulation + the reference's Gaussian filter,
BLAS refit of UPDATABLE meshes.  Tolerances: composed image RMSE <= 1e-3 (BASELINE.json gate); ids exact."""
import copy
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 320, 180


def _variant(sample_data, fn):
    from sm64rt_legacy_renderer_amd import sample_scene
    d = copy.copy(sample_data)
    d.instances = [copy.copy(i) for i in sample_data.instances]
    for i in d.instances:
        i.material = sample_scene.copy_material(i.material)
    d.meshes = [copy.copy(m) for m in sample_data.meshes]
    fn(d)
    return d


def _render_pair(rt64_lib, data, frames=1, view_desc=None, options=None, per_frame=None, extra_images=(), images=None):
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    s = sample_scene.Rt64Scene(rt64_lib, data, W, H, hip_device=0)
    o = oracle_py.OracleScene(data)
    try:
        kw = {}
        if view_desc:
            s.set_view_description(**view_desc)
            kw = dict(diSamples=view_desc.get("di_samples", 0), giSamples=view_desc.get("gi_samples", 0), maxLights=view_desc.get("max_lights", 12),
                      denoiserEnabled=int(view_desc.get("denoiser", False)), resolutionScale=float(view_desc.get("resolution_scale", 1.0)),
                      motionBlurStrength=float(view_desc.get("motion_blur", 0.0)))
        for k, v in (options or {}).items():
            assert s.option(k, v)
            if k == "denoiser_mode":
                kw["denoiserMode"] = int(v)
        s.option("count_traversal", 1)
        for f in range(frames):
            if per_frame:
                per_frame(f, s, o)
            s.draw()
            ref = o.render(W, H, images=(f == frames - 1), **kw)
        names = images or (("OUTPUT_RGBA32F", "FINAL_RGBA8", "INSTANCE_ID", "PRIMARY_HIT", "DIFFUSE", "DIRECT_LIGHT_RAW", "INDIRECT_LIGHT_RAW",
                            "INDIRECT_LIGHT_FILTERED", "REFLECTION", "REFRACTION", "TRANSPARENT") + tuple(extra_images))
        got = {k: s.readback(getattr(rt64, "IMAGE_" + k)) for k in names}
        st = s.stats()
        return got, ref, st
    finally:
        s.close(); o.close()


def _rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def _check(got, ref, st=None, rmse=1e-3, exact_hits=True):
    if exact_hits:
        assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
        assert np.array_equal(got["INSTANCE_ID"], ref["instanceId"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= rmse
    assert _rmse(got["FINAL_RGBA8"][..., :3] / 255.0, ref["final"][..., :3] / 255.0) <= rmse
    if st is not None:
        c = ref["counters"]
        assert st.primaryRays == c["primaryRays"] and st.shadowRays == c["shadowRays"]
        assert st.nodesVisited == c["nodesVisited"] and st.trianglesTested == c["trianglesTested"]


def test_translucent_sphere_uses_hit_list_and_transparent_lighting(rt64_lib, sample_data):
    """alpha 0.5 <= APPLY_LIGHTS_MINIMUM_ALPHA: the sphere goes through the 'transparent geometry that needs lighting' path
    (PrimaryRayGen.hlsl:136-148) and the floor behind it shows through; shadows of the sphere are alpha-accumulated."""
    def mod(d):
        d.instances[1].material.solidAlphaMultiplier = 0.5
        d.instances[1].material.shadowAlphaMultiplier = 0.3
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod))
    _check(got, ref, st)
    assert np.abs(got["TRANSPARENT"][..., :3] - ref["transparent"][..., :3]).max() < 4e-3
    assert (ref["transparent"][..., :3] > 0).any() and (got["INSTANCE_ID"] == 1).mean() > 0.3095      # the floor is the lit primary hit behind the sphere (stock share: 30.9 %)
    partial = (ref["directLight"][..., 0] > 0.2) & (ref["directLight"][..., 0] < 0.7)
    assert partial.any()


def test_multi_layer_alpha_cull_disable_and_depth_bias(rt64_lib, sample_data):
    def mod(d):
        d.instances[1].material.solidAlphaMultiplier = 0.7
        d.instances[1].flags = 2                                 # RT64_INSTANCE_DISABLE_BACKFACE_CULLING: back faces join the list
        d.instances[1].material.depthBias = 0.05
        d.instances[3].material.solidAlphaMultiplier = 0.8
        d.instances[3].material.depthBias = -0.02
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod))
    _check(got, ref, st)
    assert np.abs(got["DIFFUSE"] - ref["diffuse"]).max() <= 1.0 / 255.0 + 1e-6


def test_texture_edge_anyhit_ignores_hits(rt64_lib, sample_data):
    """SHADER_OPT_TEXTURE_EDGE (bit 26): alpha > 0.3 -> 1 else IgnoreHit (rt64_shader.cpp:502-511).  Vertex alpha varies over
    the sphere, so part of it disappears for primary rays and for shadow rays alike."""
    def mod(d):
        d.shader_id = 0x01200a00 | (1 << 26)
        m = copy.copy(d.meshes[0]); v = m.vertices.copy()
        v["input1"][:, 3] = np.clip(0.5 + 0.5 * np.sin(3.0 * v["position"][:, 0]), 0.0, 1.0).astype(np.float32)
        m.vertices = v; d.meshes[0] = m
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod))
    _check(got, ref, st)
    sphere_px = (ref["instanceId"] == 0).mean()
    assert 0.02 < sphere_px < 0.12                                # stock scene: 12.8 %


def test_fog_reflection_refraction(rt64_lib, sample_data):
    def mod(d):
        d.instances[3].material.reflectionFactor = 0.3; d.instances[3].material.reflectionShineFactor = 0.2     # floor mirrors
        d.instances[3].material.fogEnabled = 1; d.instances[3].material.fogMul = 2000.0; d.instances[3].material.fogOffset = -1800.0
        d.instances[1].material.refractionFactor = 0.9; d.instances[1].material.solidAlphaMultiplier = 0.6           # glassy sphere
        d.instances[1].material.reflectionFactor = 0.1
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod))
    # reflection rewrites gInstanceId (ReflectionRayGen.hlsl:120): compare the first-hit records only
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    assert np.abs(got["REFLECTION"] - ref["reflection"]).max() < 8e-3 and np.abs(got["REFRACTION"] - ref["refraction"]).max() < 8e-3
    assert (ref["reflection"][..., :3] > 0.01).mean() > 0.1 and (ref["refraction"][..., :3] > 0.01).mean() > 0.03
    c = ref["counters"]
    assert st.reflectionRays == c["reflectionRays"] > 0 and st.refractionRays == c["refractionRays"] > 0


def test_soft_shadows_many_lights(rt64_lib, sample_data):
    """diSamples > 0 (disc-sampled lights, blue noise) and several lights with importance selection (Lights.hlsli:115-168)."""
    from sm64rt_legacy_renderer_amd import rt64

    def mod(d):
        ls = []
        for k, (pos, col) in enumerate([((15000.0, 30000.0, 15000.0), (0.8, 0.75, 0.65)), ((-6.0, 4.0, 3.0), (0.9, 0.2, 0.1)),
                                         ((5.0, 3.0, 6.0), (0.1, 0.3, 0.9)), ((0.0, 8.0, -4.0), (0.2, 0.7, 0.2))]):
            l = rt64.LIGHT(); C.memmove(C.byref(l), C.byref(d.lights[0]), C.sizeof(rt64.LIGHT))
            l.position = rt64.VECTOR3(*pos); l.diffuseColor = rt64.VECTOR3(*col); l.specularColor = rt64.VECTOR3(*col)
            if k:
                l.attenuationRadius = 40.0; l.pointRadius = 0.5; l.attenuationExponent = 2.0
            ls.append(l)
        d.lights = ls
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod), view_desc=dict(di_samples=2, max_lights=3))
    # light selection thresholds and disc-sample positions go through pow/rsqrt (1-ulp device ops): ray counts match,
    # individual shadow rays may differ in the last bit, so node counts are compared loosely
    _check(got, ref, None)
    c = ref["counters"]
    assert st.shadowRays == c["shadowRays"] and abs(st.nodesVisited - c["nodesVisited"]) < 1e-3 * c["nodesVisited"]
    dd = np.abs(got["DIRECT_LIGHT_RAW"][..., :3] - ref["directLight"][..., :3]).max(axis=-1)
    assert (dd > 1e-2).mean() < 2e-3, float((dd > 1e-2).mean())     # a handful of pixels sit on a light-selection / penumbra threshold


def test_c3_gi_bounce_temporal_and_gaussian_filter(rt64_lib, sample_data):
    """BASELINE config C3 at reduced size with the reference's own denoiser: 1 GI sample, temporal reprojection from frame 1 on,
    five 3x3 Gaussian passes (rt64_view.cpp:1512-1530).  Bounce directions use device sin/cos, so a few pixels pick another
    triangle: compare with an image tolerance."""
    got, ref, st = _render_pair(rt64_lib, sample_data, frames=4, view_desc=dict(gi_samples=1, denoiser=True), options={"denoiser_mode": 0})
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    assert _rmse(got["INDIRECT_LIGHT_FILTERED"][..., :3], ref["filteredIndirect"][..., :3]) <= 2e-3
    hist = ref["indirectLight"][..., 3]
    assert 4.0 <= hist.max() < 8.0       # grows by ~1 per frame on static pixels (the normal weight pow(n.n', 128) can exceed 1 with fp16 normals)
    assert st.indirectRays == ref["counters"]["indirectRays"] > 0


def test_c3_gi_svgf_denoiser(rt64_lib, sample_data):
    """BASELINE config C3 at reduced size: 1 GI sample + SVGF (temporal moments, variance estimate, 5 a-trous iterations)."""
    got, ref, st = _render_pair(rt64_lib, sample_data, frames=6, view_desc=dict(gi_samples=1, denoiser=True), options={"denoiser_mode": 1})
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    assert _rmse(got["INDIRECT_LIGHT_FILTERED"][..., :3], ref["filteredIndirect"][..., :3]) <= 2e-3
    # the filter does something: filtered GI is smoother than the raw accumulation on lit pixels
    lit = ref["instanceId"] >= 0
    raw, flt = ref["indirectLight"][..., 1], ref["filteredIndirect"][..., 1]
    def roughness(a):
        d = np.abs(np.diff(a, axis=1)); return d[lit[:, 1:] & lit[:, :-1]].mean()
    assert roughness(flt) < 0.6 * roughness(raw)


def test_c3_bounce_traversal_with_wave_refill_matches_plain_walk(rt64_lib, sample_data):
    """Option bounce_refill=1: the bounce rays are traced by the persistent walk that refills finished lanes by wave ballot
    (passes.hip bounce_trace_refill_kernel).  Per-ray arithmetic is unchanged, so the frame equals the oracle's like the plain walk,
    and with 2 samples per pixel images and traversal counters are bit-identical to the plain walk's."""
    got, ref, st = _render_pair(rt64_lib, sample_data, frames=4, view_desc=dict(gi_samples=1, denoiser=True), options={"bounce_refill": 1, "denoiser_mode": 1})
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    assert _rmse(got["INDIRECT_LIGHT_FILTERED"][..., :3], ref["filteredIndirect"][..., :3]) <= 2e-3
    assert st.indirectRays == ref["counters"]["indirectRays"] > 0
    got1, _, st1 = _render_pair(rt64_lib, sample_data, frames=3, view_desc=dict(gi_samples=2, denoiser=True), options={"bounce_refill": 1, "denoiser_mode": 1})
    got0, _, st0 = _render_pair(rt64_lib, sample_data, frames=3, view_desc=dict(gi_samples=2, denoiser=True), options={"bounce_refill": 0, "denoiser_mode": 1})
    assert np.array_equal(got1["INDIRECT_LIGHT_RAW"], got0["INDIRECT_LIGHT_RAW"]) and np.array_equal(got1["OUTPUT_RGBA32F"], got0["OUTPUT_RGBA32F"])
    assert (st1.nodesVisited, st1.trianglesTested, st1.indirectRays) == (st0.nodesVisited, st0.trianglesTested, st0.indirectRays)


def test_resolution_scale_resamples_the_render_target_to_the_screen(rt64_lib, sample_data):
    """RT64_VIEW_DESC.resolutionScale (rt64_view.cpp:138-139): every image is lround(screen x scale), PostProcessPS.hlsl resamples the
    composed output to the screen-size back buffer with the LINEAR/WRAP static sampler.  0.75 (upsample) and 1.5 (supersample)."""
    for scale, (rw, rh) in ((0.75, (240, 135)), (1.5, (480, 270))):
        got, ref, st = _render_pair(rt64_lib, sample_data, view_desc=dict(resolution_scale=scale))
        assert (st.width, st.height, st.screenWidth, st.screenHeight) == (rw, rh, W, H)
        assert got["OUTPUT_RGBA32F"].shape == (rh, rw, 4) and got["FINAL_RGBA8"].shape == (H, W, 4) == ref["final"].shape
        assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"]) and np.array_equal(got["INSTANCE_ID"], ref["instanceId"])
        assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
        assert np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32)).max() <= 1
        assert st.primaryRays == rw * rh == ref["counters"]["primaryRays"]


def test_motion_blur_gathers_along_the_flow(rt64_lib, sample_data):
    """PostProcessPS.hlsl:14-33: with motionBlurStrength > 0 the back buffer averages motionBlurSamples taps of the output along the
    screen-space motion vector.  The camera strafes between frames so that the flow is not zero."""
    data = _variant(sample_data, lambda d: None)
    base = data.view.copy()

    def per_frame(f, s, o):
        v = base.copy(); v[3, 0] = base[3, 0] - 0.35 * f        # view matrix translation row: camera moves +x
        data.view = v
    got, ref, st = _render_pair(rt64_lib, data, frames=3, view_desc=dict(motion_blur=1.0), per_frame=per_frame)
    assert np.abs(ref["flow"]).max() > 2.0                       # pixels of motion
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    d = np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32))
    assert d.max() <= 2 and (d > 1).mean() < 1e-3
    sharp = np.clip(np.round(ref["output"][..., :3] * 255.0), 0, 255)
    assert np.abs(sharp - ref["final"][..., :3]).mean() > 0.5      # the blur changed the picture


def test_full_frame_after_lean_frames_sees_a_complete_previous_frame(rt64_lib, sample_data):
    """Lean frames (giSamples = 0) skip the history guides and the GI buffer.  When the host then turns GI + denoiser on, the first
    full frame reprojects from the previous one (IndirectRayGen.hlsl:43-56): the library produces what the lean frame skipped before
    it renders, so the sequence matches the oracle, which writes every image every frame."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    o = oracle_py.OracleScene(sample_data)
    try:
        kw = {}
        for f in range(5):
            if f == 2:
                s.set_view_description(gi_samples=1, denoiser=True)
                kw = dict(giSamples=1, denoiserEnabled=1, denoiserMode=1)
            s.draw()
            assert bool(s.stats().leanFrame) == (f < 2)
            ref = o.render(W, H, **kw)
        got_out, got_gi = s.readback(rt64.IMAGE_OUTPUT_RGBA32F), s.readback(rt64.IMAGE_INDIRECT_LIGHT_FILTERED)
        raw = s.readback(rt64.IMAGE_INDIRECT_LIGHT_RAW)
        assert _rmse(got_out[..., :3], ref["output"][..., :3]) <= 1e-3
        assert _rmse(got_gi[..., :3], ref["filteredIndirect"][..., :3]) <= 2e-3
        hit = ref["instanceId"] >= 0
        assert np.abs(raw[..., 3][hit] - ref["indirectLight"][..., 3][hit]).max() < 0.51      # history length: 3 GI frames on static pixels
    finally:
        s.close(); o.close()


def test_band_partition_with_denoiser_halo_equals_the_whole_frame(rt64_lib, sample_data):
    """Image-tile partition of a GI + SVGF frame: every device renders its band plus the filter's halo and the bands, put
    together, are bit-identical to the frame one device renders alone -- for the SVGF denoiser and for the reference's Gaussian passes."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    bands = [(0, 70), (70, 131), (131, H)]                       # ragged on purpose: not multiples of the 16-row tiles
    for mode in (1, 0):
        whole = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
        parts = [sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0) for _ in bands]
        try:
            for s in [whole] + parts:
                s.set_view_description(gi_samples=1, denoiser=True)
                assert s.option("denoiser_mode", mode)
            for s, (a, b) in zip(parts, bands):
                s.set_tile(a, b)
            parts[1].option("count_traversal", 1)
            for f in range(4):
                for s in [whole] + parts:
                    s.draw()
            for image in (rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_INDIRECT_LIGHT_FILTERED):
                full = whole.readback(image)
                tiled = np.concatenate([s.readback(image) for s in parts], axis=0)
                assert tiled.shape == full.shape and np.array_equal(tiled, full), (mode, image)
            assert parts[1].stats().primaryRays == (min(H, 131 + (66 if mode else 5)) - max(0, 70 - (66 if mode else 5))) * W      # band + halo
        finally:
            for s in [whole] + parts:
                s.close()


def test_c4_refit_of_updatable_mesh(rt64_lib, sample_data):
    """Per-frame vertex animation of an UPDATABLE mesh (rt64_mesh.cpp:129,149-157): SetMesh with unchanged counts refits the BLAS;
    hits stay bit-identical to the oracle's refit."""
    from sm64rt_legacy_renderer_amd import rt64

    def mod(d):
        m = copy.copy(d.meshes[0]); m.flags = m.flags | rt64.MESH_RAYTRACE_UPDATABLE; d.meshes[0] = m
    data = _variant(sample_data, mod)
    base = data.meshes[0].vertices.copy()

    def per_frame(f, s, o):
        if f == 0:
            return
        v = base.copy()
        v["position"][:, :3] += (0.1 * np.sin(f * 0.1 + base["position"][:, 1]))[:, None].astype(np.float32) * base["normal"]     # SURVEY 8d C4 displacement
        s.set_mesh(s.meshes[0], v, data.meshes[0].indices)
        o.set_mesh(o.meshes[0], v, data.meshes[0].indices)
    got, ref, st = _render_pair(rt64_lib, data, frames=3, per_frame=per_frame)
    _check(got, ref, st)


@pytest.mark.gpu
@pytest.mark.parametrize("lds_cache", [1, 0])
def test_one_kernel_lean_frame_matches_the_three_kernel_path(rt64_lib, sample_data, lds_cache):
    """A lean frame runs as lean_frame_kernel (device option fused_lean, default 1) or as primary_trace + primary_shade + direct
    (fused_lean = 0).  Same arithmetic, same rounding points: every image is bit-identical, including the ones the fused frame only
    produces on readback (View::materialise), and so are the ray / node / triangle counters."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    images = [rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_PRIMARY_HIT, rt64.IMAGE_INSTANCE_ID, rt64.IMAGE_DIFFUSE, rt64.IMAGE_DIRECT_LIGHT_RAW,
              rt64.IMAGE_DIRECT_LIGHT_FILTERED, rt64.IMAGE_INDIRECT_LIGHT_FILTERED, rt64.IMAGE_SHADING_POSITION, rt64.IMAGE_SHADING_NORMAL, rt64.IMAGE_SHADING_SPECULAR,
              rt64.IMAGE_FLOW, rt64.IMAGE_DEPTH, rt64.IMAGE_VIEW_DIRECTION, rt64.IMAGE_FIRST_INSTANCE_ID]
    got = {}
    for fused in (1, 0):
        s = sample_scene.Rt64Scene(rt64_lib, sample_data, 333, 187, hip_device=0)
        try:
            s.option("fused_lean", fused)
            s.option("lds_cache", lds_cache)                 # 0: BVH nodes and instance records from HBM / L2 (the path of scenes too big for the LDS scene cache)
            s.option("count_traversal", 1)
            s.draw(); s.draw()                               # second frame: the steady state (cached frame tables)
            st = s.stats()
            assert st.leanFrame == 1 and st.fusedFrame == fused
            final_first = s.readback(rt64.IMAGE_FINAL_RGBA8)     # before anything materialises the other images
            got[fused] = ([final_first] + [s.readback(i) for i in images],
                          (st.primaryRays, st.shadowRays, st.nodesVisited, st.trianglesTested, st.nodesPrimary, st.trianglesPrimary, st.nodesDirect, st.trianglesDirect))
        finally:
            s.close()
    assert got[1][1] == got[0][1]
    for a, b in zip(got[1][0], got[0][0]):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.gpu
def test_lean_frame_without_stored_records_materialises_the_same_images(rt64_lib, sample_data):
    """The one-kernel lean frame stores the back buffer only (option lean_records, default 0); every other image comes from
    View::materialise re-running the FULL frame kernel.  With lean_records = 1 the frame stores its hit records and direct light itself
    (round-1 behaviour).  Both must give the same bytes in every image, the same counters, and the counters of the frame must not move
    when an image is read back (the re-traced rays are not counted) -- synchronous and enqueued frames."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    images = [rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_PRIMARY_HIT, rt64.IMAGE_INSTANCE_ID, rt64.IMAGE_DIFFUSE, rt64.IMAGE_DIRECT_LIGHT_RAW,
              rt64.IMAGE_DIRECT_LIGHT_FILTERED, rt64.IMAGE_SHADING_POSITION, rt64.IMAGE_SHADING_NORMAL, rt64.IMAGE_FIRST_INSTANCE_ID]
    got = {}
    for records in (0, 1):
        for sync in (1, 0):
            s = sample_scene.Rt64Scene(rt64_lib, sample_data, 333, 187, hip_device=0)
            try:
                s.option("lean_records", records); s.option("count_traversal", 1); s.option("sync_present", sync)
                s.draw(); s.draw()
                imgs = [s.readback(i) for i in images]          # the first readback of a non-final image materialises
                st = s.stats()                                    # enqueued frames: counters are read now, after the re-trace
                assert st.leanFrame == 1 and st.fusedFrame == 1
                got[(records, sync)] = (imgs, (st.primaryRays, st.shadowRays, st.nodesVisited, st.trianglesTested))
                picked = rt64_lib.GetViewRaytracedInstanceAt(s.view, 166, 110)
                assert picked == s.instances[1]                   # the sphere, through firstInstanceId of the materialised G-buffer
            finally:
                s.close()
    ref_imgs, ref_ctr = got[(1, 1)]
    assert ref_ctr[0] == 333 * 187
    for key, (imgs, ctr) in got.items():
        assert ctr == ref_ctr, key
        for a, b in zip(imgs, ref_imgs):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), key


@pytest.mark.gpu
@pytest.mark.parametrize("band", [None, (48, 128)])
def test_one_kernel_full_frame_matches_the_separate_kernels(rt64_lib, sample_data, band):
    """Full (non-lean) frames whose instances are all opaque -- here 1 GI sample + SVGF -- run primary visibility, the G-buffer and
    DirectRayGen as lean_frame_kernel<.., FULL> (fusedFrame == 2) unless fused_lean = 0.  Both forms give the same bytes in every image
    over three frames of temporal history, also on a band of the frame with the denoiser halo around it."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    images = [rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_PRIMARY_HIT, rt64.IMAGE_INSTANCE_ID, rt64.IMAGE_DIFFUSE, rt64.IMAGE_DIRECT_LIGHT_RAW,
              rt64.IMAGE_DIRECT_LIGHT_FILTERED, rt64.IMAGE_INDIRECT_LIGHT_RAW, rt64.IMAGE_INDIRECT_LIGHT_FILTERED, rt64.IMAGE_SHADING_POSITION, rt64.IMAGE_SHADING_NORMAL,
              rt64.IMAGE_SHADING_SPECULAR, rt64.IMAGE_FLOW, rt64.IMAGE_DEPTH, rt64.IMAGE_VIEW_DIRECTION, rt64.IMAGE_FIRST_INSTANCE_ID, rt64.IMAGE_REACTIVE_MASK, rt64.IMAGE_LOCK_MASK]
    got = {}
    for fused in (1, 0):
        s = sample_scene.Rt64Scene(rt64_lib, sample_data, 240, 176, hip_device=0)
        try:
            s.option("fused_lean", fused)
            s.option("count_traversal", 1)
            s.set_view_description(gi_samples=1, denoiser=True)
            if band:
                s.set_tile(*band)
            for _ in range(3):
                s.draw()
            st = s.stats()
            assert st.leanFrame == 0 and st.fusedFrame == (2 if fused else 0)
            y0, y1 = band if band else (0, 176)
            got[fused] = ([s.readback(i)[y0:y1] for i in images],
                          (st.primaryRays, st.shadowRays, st.indirectRays, st.nodesVisited, st.trianglesTested))
        finally:
            s.close()
    assert got[1][1] == got[0][1]
    for a, b in zip(got[1][0], got[0][0]):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,fused", [(2, 1), (3, 1), (8, 1), (8, 0)])
def test_interleaved_strip_partition_reassembles_the_whole_frame(rt64_lib, sample_data, ranks, fused):
    """bench.py's N > 1 partition of a pixel-local frame: rank r renders the 16-row strips r, r + N, ... (RT64_SetDeviceInterleave) and
    RT64_CopyDeviceImage packs them back to back for the gather.  Every rank's strips, rendered here one rank after the other on one
    device, reassemble (tiles.assemble, the function rank 0 runs on the gathered buckets) into exactly the frame a single device renders;
    a height that is not a multiple of the strip height leaves a ragged last strip."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene, tiles
    hip = C.CDLL("libamdhip64.so")              # the runtime librt64.so is linked against: a device buffer for RT64_CopyDeviceImage
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]; hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    W, H = 208, 150
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    try:
        s.option("fused_lean", fused)
        s.draw()
        whole = s.readback(rt64.IMAGE_FINAL_RGBA8).copy()
        mx = tiles.max_owned_rows(H, ranks) * W * 4
        dev_buf = C.c_void_p()
        assert hip.hipMalloc(C.byref(dev_buf), ranks * mx) == 0 and hip.hipMemset(dev_buf, 0, ranks * mx) == 0
        rows_seen = np.zeros(H, dtype=np.int32)
        for r in range(ranks):
            s.set_interleave(r, ranks)
            s.draw()
            st = s.stats()
            assert st.stripRank == r and st.stripCount == ranks and st.rowsRendered == tiles.owned_rows(H, r, ranks)
            n = rt64_lib.CopyDeviceImage(s.device, rt64.IMAGE_FINAL_RGBA8, dev_buf.value + r * mx, mx)
            assert n == tiles.owned_rows(H, r, ranks) * W * 4
            if fused:            # RT64_SetDeviceGatherTarget: the frame kernel itself leaves the same packed rows in a send buffer
                target = C.c_void_p()
                assert hip.hipMalloc(C.byref(target), mx) == 0 and hip.hipMemset(target, 0xAB, mx) == 0
                rt64_lib.SetDeviceGatherTarget(s.device, target, mx)
                s.draw()
                assert s.stats().packedFinal == 1
                a = np.zeros(n, dtype=np.uint8); b = np.zeros(n, dtype=np.uint8)
                assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), target, n, 2) == 0 and hip.hipMemcpy(b.ctypes.data_as(C.c_void_p), C.c_void_p(dev_buf.value + r * mx), n, 2) == 0
                assert np.array_equal(a, b)
                rt64_lib.SetDeviceGatherTarget(s.device, None, 0)
                s.draw()
                assert s.stats().packedFinal == 0
                hip.hipFree(target)
            mine = s.readback(rt64.IMAGE_FINAL_RGBA8)                # the host readback packs the owned rows the same way
            assert np.array_equal(mine, np.concatenate([whole[a:b] for a, b in tiles.strip_ranges(H, r, ranks)]))
            for a, b in tiles.strip_ranges(H, r, ranks):
                rows_seen[a:b] += 1
        packed = np.zeros((ranks, mx), dtype=np.uint8)
        assert hip.hipMemcpy(packed.ctypes.data_as(C.c_void_p), dev_buf, ranks * mx, 2) == 0          # hipMemcpyDeviceToHost (synchronises)
        hip.hipFree(dev_buf)
        assert (rows_seen == 1).all()
        assert np.array_equal(tiles.assemble(packed, H, W, ranks), whole)
    finally:
        s.close()


def _mutation_sequences():
    """Scene changes a host can make between a lean frame and the next frame (name -> function(scene, data, lib) applied after frame 1)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene

    def add_instance_and_gi(s, data, lib):
        h = lib.CreateInstance(s.scene)
        s.instances.append(h)
        inst = copy.copy(data.instances[1]); inst.material = sample_scene.copy_material(inst.material); inst.name = "sphere2"
        t = np.eye(4, dtype=np.float32); t[3, 0] = 3.5; t[3, 2] = -2.0
        inst.transform = inst.previous_transform = t
        s.set_instance(len(s.instances) - 1, inst)
        s.set_view_description(gi_samples=1, denoiser=True)

    def rebuild_mesh_and_translucent(s, data, lib):
        v = data.meshes[0].vertices.copy(); v["position"][:, 0] += np.float32(1.25); v["position"][:, 1] *= np.float32(0.8)
        s.set_mesh(s.meshes[0], v, data.meshes[0].indices)                  # not UPDATABLE: a rebuild with another topology
        inst = copy.copy(data.instances[1]); inst.material = sample_scene.copy_material(inst.material)
        inst.material.solidAlphaMultiplier = 0.6                              # the sphere leaves rule O1: k-buffer frame
        data.instances[1] = inst

    def remove_instance_and_reflection(s, data, lib):
        lib.DestroyInstance(s.instances[3]); s.instances[3] = None          # the floor goes
        inst = copy.copy(data.instances[1]); inst.material = sample_scene.copy_material(inst.material)
        inst.material.reflectionFactor = 0.4
        data.instances[1] = inst

    def swap_texture(s, data, lib):
        t = data.textures[4]                                                   # tiles_dif: destroyed and replaced by its negative
        lib.DestroyTexture(s.textures[4])
        d = rt64.TEXTURE_DESC(); buf = np.ascontiguousarray(255 - t.data); buf[..., 3] = 255
        d.bytes = buf.ctypes.data; d.byteCount = buf.nbytes; d.format = t.format; d.width, d.height, d.rowPitch = t.width, t.height, t.width * 4
        s.textures[4] = lib.CreateTexture(s.device, d)
        for k, inst in enumerate(data.instances):
            if s.instances[k] is not None and inst.diffuse == 4:
                s.set_instance(k, inst)
        s.set_view_description(gi_samples=1, denoiser=True)
    return dict(add_instance_and_gi=add_instance_and_gi, rebuild_mesh_and_translucent=rebuild_mesh_and_translucent,
                remove_instance_and_reflection=remove_instance_and_reflection, swap_texture=swap_texture)


@pytest.mark.parametrize("change", sorted(_mutation_sequences()))
def test_scene_change_after_a_lean_frame_keeps_that_frame_intact(rt64_lib, sample_data, change):
    """A lean frame stores its back buffer only; its other images are re-traced on demand from the frame's meshes, textures, tables, TLAS and
    LDS cache image (View::materialise).  When the host then changes the scene -- adds or removes an instance, rebuilds a mesh, replaces a
    texture -- and the next frame is not lean, the library must produce the lean frame's images BEFORE the new contents overwrite what they
    are made from (round-2 ADVICE, high).  Same call sequence with lean_frames = 0 (every frame stores its whole G-buffer): every image of
    the frame after the change -- history-dependent ones included -- and of the frame before it (read after the change) is the same bytes."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    images = [rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_PRIMARY_HIT, rt64.IMAGE_INSTANCE_ID, rt64.IMAGE_DIFFUSE, rt64.IMAGE_DIRECT_LIGHT_RAW,
              rt64.IMAGE_INDIRECT_LIGHT_RAW, rt64.IMAGE_INDIRECT_LIGHT_FILTERED, rt64.IMAGE_SHADING_POSITION, rt64.IMAGE_SHADING_NORMAL, rt64.IMAGE_DEPTH,
              rt64.IMAGE_FLOW, rt64.IMAGE_REFLECTION, rt64.IMAGE_TRANSPARENT, rt64.IMAGE_FIRST_INSTANCE_ID]
    got = {}
    for lean in (1, 0):
        data = _variant(sample_data, lambda d: None)
        s = sample_scene.Rt64Scene(rt64_lib, data, 333, 187, hip_device=0)
        try:
            s.option("lean_frames", lean)
            s.draw(); s.draw()
            assert s.stats().leanFrame == lean
            _mutation_sequences()[change](s, data, rt64_lib)
            before = [s.readback(i) for i in (rt64.IMAGE_INSTANCE_ID, rt64.IMAGE_PRIMARY_HIT, rt64.IMAGE_SHADING_NORMAL, rt64.IMAGE_DIFFUSE)]      # frame 1, read after the change
            # Rt64Scene.draw re-sends the sphere's descriptor like the sample host (main.cpp:129): from `data`, which the change updated
            s._desc_cache.clear()
            s.draw(); s.draw()
            assert s.stats().leanFrame == 0
            got[lean] = before + [s.readback(i) for i in images]
        finally:
            s.instances = [h for h in s.instances if h is not None]
            s.close()
    for k, (a, b) in enumerate(zip(got[1], got[0])):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), (change, k)


def test_frames_after_a_scene_change_hold_off_lean_rendering_then_return_to_it(rt64_lib, sample_data):
    """A host that changes its scene every frame would pay for a materialise per frame; after a change that had to materialise a lean frame the
    next RT64_LEAN_HOLDOFF_FRAMES frames store their G-buffer themselves, then lean frames resume.  Back buffers are the same bytes either way."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    data = _variant(sample_data, lambda d: None)
    s = sample_scene.Rt64Scene(rt64_lib, data, 320, 180, hip_device=0)
    try:
        s.draw(); s.draw()
        ref = s.readback(rt64.IMAGE_FINAL_RGBA8).copy()
        assert s.stats().leanFrame == 1
        s.set_mesh(s.meshes[0], data.meshes[0].vertices, data.meshes[0].indices)       # same arrays: a change as far as the library can tell
        lean = []
        for f in range(8):
            s.draw()
            lean.append(int(s.stats().leanFrame))
            assert np.array_equal(s.readback(rt64.IMAGE_FINAL_RGBA8), ref), f
        assert lean[0] == 0 and lean[-1] == 1 and sorted(lean) == lean, lean
    finally:
        s.close()


def test_host_side_tree_depth_is_the_depth_the_device_builder_reports(rt64_lib, sample_data):
    """RT64_ACCEL_HOST_DEPTH (computed at RT64_SetMesh, no wait) against BlasHeader::depth written by the device builder, for the sample
    meshes, a random soup, duplicates, and a refit that keeps the topology."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, 64, 64, hip_device=0)
    rng = np.random.default_rng(3)
    try:
        def depths(mesh):
            hdr = np.zeros(8, dtype=np.uint32); host = np.zeros(1, dtype=np.uint32)
            assert rt64_lib.ReadbackMeshAccel(mesh, rt64.ACCEL_HEADER, hdr.ctypes.data, hdr.nbytes) == 32
            assert rt64_lib.ReadbackMeshAccel(mesh, rt64.ACCEL_HOST_DEPTH, host.ctypes.data, 4) == 4
            return int(hdr[7]), int(host[0])
        for k, m in enumerate(sample_data.meshes):
            if m.flags & rt64.MESH_RAYTRACE_ENABLED:
                dev, host = depths(s.meshes[k])
                assert dev == host and 1 <= dev < 64, (m.name, dev, host)
        h = rt64_lib.CreateMesh(s.device, rt64.MESH_RAYTRACE_ENABLED | rt64.MESH_RAYTRACE_UPDATABLE)
        s.meshes.append(h)
        for n in (1, 2, 77, 1024, 4096):
            p = np.zeros(3 * n, dtype=sample_scene.VERTEX_DTYPE)
            p["position"][:, :3] = rng.normal(size=(3 * n, 3)).astype(np.float32); p["position"][:, 3] = 1.0
            if n == 77:
                p["position"][3:] = p["position"][np.arange(3 * n - 3) % 3]        # 77 copies of one triangle
            s.set_mesh(h, p, np.arange(3 * n, dtype=np.uint32))
            dev, host = depths(h)
            assert dev == host, (n, dev, host)
            q = p.copy(); q["position"][:, :3] = rng.normal(size=(3 * n, 3)).astype(np.float32)
            s.set_mesh(h, q, np.arange(3 * n, dtype=np.uint32))                    # same shape on an UPDATABLE mesh: refit, topology (and depth) kept
            dev2, host2 = depths(h)
            assert dev2 == dev and host2 == host, (n, dev2, host2)
    finally:
        s.close()


@pytest.mark.parametrize("gi_samples,groups", [(1, 0), (4, 0), (2, 7), (8, 3)])
def test_bounce_walk_in_two_phases_matches_the_plain_walk(rt64_lib, sample_data, gi_samples, groups):
    """Device option bounce_split: the bounce rays take the TLAS part of their walk first, the ones that reach an instance are compacted through LDS
    and walked in full by dense waves (passes.hip bounce_trace_split_kernel).  Per ray the operations are those of the plain walk: the GI buffers, the
    composed image and the traversal counters are identical -- with one, several and more than four samples per pixel (rounds of four (tile, sample)
    items), a frame size that leaves partial tiles (320 x 180) and grids that give a workgroup several tiles.  (GPU against GPU: the oracle is not run.)"""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    runs = []
    for split in (0, 1):
        s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
        try:
            s.set_view_description(gi_samples=gi_samples, denoiser=True)
            for k, v in (("bounce_split", split), ("bounce_refill", 0), ("denoiser_mode", 1), ("count_traversal", 1)) + ((("bounce_groups", groups),) if groups else ()):
                assert s.option(k, v)
            for _ in range(3):
                s.draw()
            runs.append(({im: s.readback(im) for im in (rt64.IMAGE_INDIRECT_LIGHT_RAW, rt64.IMAGE_INDIRECT_LIGHT_FILTERED, rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_FINAL_RGBA8)}, s.stats()))
        finally:
            s.close()
    (a, sa), (b, sb) = runs
    assert sa.indirectRays == sb.indirectRays > 0 and sa.shadowRays == sb.shadowRays
    assert (sa.nodesVisited, sa.trianglesTested) == (sb.nodesVisited, sb.trianglesTested)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("gi", [0, 1])
def test_tiles_started_in_cost_order_render_the_same_frame(rt64_lib, sample_data, gi):
    """Scenes that walk from HBM run the one-kernel frame as one-wave workgroups; device option tile_order (on by default) starts their tiles in the order of the cost
    the frame before recorded, most expensive first (tile_order_kernel).  lds_cache = 0 puts the sample scene on that path: over four frames -- the first one in the
    bottom-up order, the others in the recorded one -- every image and every counter equals the run with the option off, for the lean kernel (gi = 0) and the
    G-buffer-writing one (gi = 1), with several trips per workgroup as well (max_frame_groups = 20)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    W, H = 320, 180
    names = ("OUTPUT_RGBA32F", "FINAL_RGBA8", "PRIMARY_HIT", "DIRECT_LIGHT_RAW", "SHADING_NORMAL")
    out = {}
    for groups in (0, 20):
        for order in (1, 0):
            s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
            try:
                if gi:
                    s.set_view_description(gi_samples=1, denoiser=True)
                assert s.option("lds_cache", 0) and s.option("tile_order", order) and s.option("count_traversal", 1)
                if groups:
                    assert s.option("max_frame_groups", groups)
                for _ in range(4):
                    s.draw()
                st = s.stats()
                assert st.fusedFrame == (2 if gi else 1)
                out[groups, order] = ({k: s.readback(getattr(rt64, "IMAGE_" + k)) for k in names},
                                      (st.primaryRays, st.shadowRays, st.nodesVisited, st.trianglesTested, st.nodesPrimary, st.trianglesPrimary))
            finally:
                s.close()
        a, b = out[groups, 1], out[groups, 0]
        assert a[1] == b[1]
        for k in names:
            assert np.array_equal(a[0][k].view(np.uint8), b[0][k].view(np.uint8)), (groups, k)
