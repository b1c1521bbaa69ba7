"""GPU parity of the raster (HUD / background) instance pass against the oracle's rasteriser (oracle/oracle_raster.c).

Reference behaviour restated: rt64_shader.cpp:312-442 (raster vertex + pixel shader, blend state), rt64_view.cpp:1225-1254 (drawInstances),
:1292-1319 (background pass and gBackground), :1657-1661 (foreground pass).  Coverage is integer arithmetic on both sides, so which pixels
a triangle touches is compared exactly; colours go through fp32 interpolation + the texture sampler and may differ by one RGBA8 step."""
import copy

import numpy as np
import pytest

from test_gpu_features import _render_pair, _variant, _rmse, W, H

pytestmark = pytest.mark.gpu


def _hud_mesh(sample_scene, rt64, tris, alpha=1.0, w=1.0):
    """Clip-space triangles [(x, y) x 3] -> MeshData with the sample's vertex layout (position float4, normal, uv, input1 rgba)."""
    v = np.zeros(3 * len(tris), dtype=sample_scene.VERTEX_DTYPE)
    k = 0
    for tri in tris:
        for j, (x, y) in enumerate(tri):
            v["position"][k] = (x * w, y * w, 0.0, w)
            v["normal"][k] = (0.0, 1.0, 0.0)
            v["uv"][k] = ((0.0, 0.0), (3.0, 0.0), (0.0, 3.0))[j]
            v["input1"][k] = ((1.0, 0.2, 0.2, alpha), (0.2, 1.0, 0.2, alpha), (0.2, 0.2, 1.0, alpha * 0.5))[j]
            k += 1
    return sample_scene.MeshData("hud", 0, v, np.arange(len(v), dtype=np.uint32))


def _final_close(got, ref, max_step=1, frac=2e-3):
    d = np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32))
    assert d.max() <= max_step + 1 and (d > max_step).mean() < frac, (int(d.max()), float((d > max_step).mean()))


def test_sample_hud_triangles_foreground_and_background(rt64_lib, sample_data):
    """The stock scene: HUD B (foreground) is visible over the ray-traced frame, HUD A (background) only lands in gBackground."""
    from sm64rt_legacy_renderer_amd import rt64
    got, ref, st = _render_pair(rt64_lib, sample_data, extra_images=("BACKGROUND",))
    assert ref["background"] is not None and (ref["background"][..., 3] > 0).sum() > 100
    assert np.array_equal(got["BACKGROUND"][..., 3] > 0, ref["background"][..., 3] > 0)          # coverage: bit-exact
    assert np.abs(got["BACKGROUND"].astype(np.int32) - ref["background"].astype(np.int32)).max() <= 1
    _final_close(got, ref)
    sharp = np.clip(np.floor(ref["output"][..., :3] * 255.0 + 0.5), 0, 255)
    assert (np.abs(sharp - ref["final"][..., :3]).max(axis=2) > 8).sum() > 100                     # the HUD triangle changed the back buffer


def test_translucent_overlapping_layers_blend_in_draw_order(rt64_lib, sample_data):
    """Vertex alpha < 1, three overlapping triangles in one instance + a second instance on top: every layer blends with the
    RGBA8 value the previous one stored (D3D12 blend state of rt64_shader.cpp:400-413), in index / instance order."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene

    def mod(d):
        d.meshes.append(_hud_mesh(sample_scene, rt64, [[(-0.9, -0.8), (0.6, -0.6), (-0.2, 0.9)], [(-0.5, -0.9), (0.9, 0.1), (-0.7, 0.5)],
                                                        [(0.1, -0.7), (0.8, 0.8), (-0.6, 0.2)]], alpha=0.6))
        d.meshes.append(_hud_mesh(sample_scene, rt64, [[(-0.3, -0.3), (0.3, -0.3), (0.0, 0.4)]], alpha=0.35))
        base = d.instances[0]                                    # hudB: foreground, tiles texture
        for m in (len(d.meshes) - 2, len(d.meshes) - 1):
            i = copy.copy(base); i.mesh = m; i.material = sample_scene.copy_material(base.material); i.name = "hud%d" % m
            d.instances.append(i)
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod))
    _final_close(got, ref)
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3                    # the ray-traced frame underneath is untouched


def test_scissor_viewport_and_perspective_w(rt64_lib, sample_data):
    """Per-instance scissor and viewport rectangles (RT64_RECT, origin bottom-left, rt64_view.cpp:1114-1136) and w != 1
    (perspective-correct attribute interpolation)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene

    def mod(d):
        d.meshes.append(_hud_mesh(sample_scene, rt64, [[(-1.0, -1.0), (1.0, -1.0), (0.0, 1.0)]], alpha=0.8, w=2.5))
        i = copy.copy(d.instances[0]); i.mesh = len(d.meshes) - 1; i.material = sample_scene.copy_material(d.instances[0].material)
        i.scissor = (40, 30, 170, 90); i.viewport = (20, 10, 200, 140)
        d.instances.append(i)
        v = d.meshes[-1].vertices; v["position"][1] = (0.8 * 0.7, -1.0 * 0.7, 0.0, 0.7)                # per-vertex w
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod))
    _final_close(got, ref)
    d = np.abs(ref["final"].astype(np.int32) - np.clip(np.floor(ref["output"] * 255.0 + 0.5), 0, 255)[..., :4].astype(np.int32))[..., :3].max(axis=2) > 4
    ys, xs = np.nonzero(d[:, 70:])                          # away from the stock HUD triangle: only inside the scissor (x 40..210, y 60..150 from the top)
    assert xs.size > 500 and xs.min() + 70 >= 40 and xs.max() + 70 < 210 and ys.min() >= H - 30 - 90 and ys.max() < H - 30


def test_background_shows_through_a_translucent_sky_and_feeds_the_env_map(rt64_lib, sample_data):
    """gBackground is read by the ray-gen shaders (BgSky.hlsli:89-95): behind a sky plane with alpha < 1 in the primary pass and as an
    environment map by bounce rays that miss."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene

    def mod(d):
        d.textures = list(d.textures)                                 # _variant copies the scene shallowly
        sky = copy.copy(d.textures[d.sky]); sky.data = sky.data.copy(); sky.data[..., 3] = 96; d.textures[d.sky] = sky
        d.meshes.append(_hud_mesh(sample_scene, rt64, [[(-1.0, -0.2), (1.0, -0.1), (0.0, 1.0)], [(-1.0, 1.0), (-1.0, 0.2), (0.2, 1.0)]], alpha=1.0))
        i = copy.copy(d.instances[0]); i.mesh = len(d.meshes) - 1; i.material = sample_scene.copy_material(d.instances[0].material)
        i.flags = rt64.INSTANCE_RASTER_BACKGROUND
        d.instances.append(i)
    data = _variant(sample_data, mod)
    got, ref, st = _render_pair(rt64_lib, data, frames=2, view_desc=dict(gi_samples=1, denoiser=True), options={"denoiser_mode": 1}, extra_images=("BACKGROUND",))
    assert np.array_equal(got["BACKGROUND"][..., 3] > 0, ref["background"][..., 3] > 0)
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    assert _rmse(got["INDIRECT_LIGHT_FILTERED"][..., :3], ref["filteredIndirect"][..., :3]) <= 2e-3
    _final_close(got, ref)


def test_raster_only_scene_draws_background_then_foreground_on_the_cleared_buffer(rt64_lib, sample_data):
    """No ray-traced instance: the frame is the cleared back buffer + background instances + foreground instances (rt64_view.cpp:1292-1296,1652-1661)."""
    def mod(d):
        d.instances = [i for i in d.instances if i.name.startswith("hud")]
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod), images=("FINAL_RGBA8", "BACKGROUND"))
    assert st.primaryRays == 0
    _final_close(got, ref, max_step=1, frac=1e-3)
    assert (ref["final"][..., :3].max(axis=2) > 0).sum() > 200                  # both HUD triangles are on screen
    assert np.array_equal(got["BACKGROUND"][..., 3] > 0, ref["background"][..., 3] > 0)


def test_ray_traced_picture_in_the_first_instances_viewport_and_scissor(rt64_lib, sample_data):
    """The first ray-traced instance's rectangles confine the ray-traced picture (rt64_view.cpp:1258-1271,1624-1626): it is drawn with
    that viewport (squeezed into it) and scissor; around it the cleared back buffer and the background instances stay visible."""
    def mod(d):
        d.instances[1].viewport = (30, 20, 260, 140)             # the sphere: first ray-traced instance in scene order
        d.instances[1].scissor = (50, 30, 200, 100)
    got, ref, st = _render_pair(rt64_lib, _variant(sample_data, mod), extra_images=("BACKGROUND",))
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3          # sky UV uses the viewport's aspect (gParams.viewport.zw)
    _final_close(got, ref)
    f = ref["final"]
    inside = np.zeros((H, W), dtype=bool); inside[H - 30 - 100:H - 30, 50:250] = True
    assert (f[~inside][:, :3].max(axis=1) == 0).mean() > 0.9                             # outside the scissor: cleared buffer (+ the HUD triangles)
    assert (f[inside][:, :3].max(axis=1) > 0).mean() > 0.99                              # inside: the squeezed picture


def test_hud_blended_inside_the_frame_kernel_equals_its_own_launch(rt64_lib, sample_data):
    """fold_foreground (default 1): on a one-kernel lean frame the foreground list is blended over each pixel before the kernel stores
    it.  Translucent overlapping layers in two instances, drawn that way and as a separate raster_draw launch (fold_foreground = 0),
    give the same back buffer byte for byte; so does a 3-way interleaved strip partition of the folded frame."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene, tiles

    def mod(d):
        d.meshes.append(_hud_mesh(sample_scene, rt64, [[(-0.9, -0.8), (0.6, -0.6), (-0.2, 0.9)], [(-0.5, -0.9), (0.9, 0.1), (-0.7, 0.5)],
                                                        [(0.1, -0.7), (0.8, 0.8), (-0.6, 0.2)]], alpha=0.6))
        d.meshes.append(_hud_mesh(sample_scene, rt64, [[(-0.3, -0.3), (0.3, -0.3), (0.0, 0.4)]], alpha=0.35))
        base = d.instances[0]
        for m in (len(d.meshes) - 2, len(d.meshes) - 1):
            i = copy.copy(base); i.mesh = m; i.material = sample_scene.copy_material(base.material); i.name = "hud%d" % m
            d.instances.append(i)
    data = _variant(sample_data, mod)
    finals = {}
    for fold in (1, 0):
        s = sample_scene.Rt64Scene(rt64_lib, data, 272, 150, hip_device=0)
        try:
            s.option("fold_foreground", fold)
            s.draw(); s.draw()
            assert s.stats().fusedFrame == 1
            finals[fold] = s.readback(rt64.IMAGE_FINAL_RGBA8).copy()
            if fold:
                parts = []
                for r in range(3):
                    s.set_interleave(r, 3); s.draw()
                    parts.append(s.readback(rt64.IMAGE_FINAL_RGBA8).copy())
                mx = tiles.max_owned_rows(150, 3) * 272 * 4
                packed = np.zeros((3, mx), dtype=np.uint8)
                for r in range(3):
                    packed[r, :parts[r].size] = parts[r].reshape(-1)
                assert np.array_equal(tiles.assemble(packed, 150, 272, 3), finals[1])
        finally:
            s.close()
    assert np.array_equal(finals[1], finals[0])
    assert (finals[1] != 0).any()


def test_homogeneous_clipping_of_triangles_that_cross_w_zero_the_depth_range_and_the_guard_band(rt64_lib, sample_data):
    """Raster spec S0 (the fixed-function clipper in front of the rasteriser, rt64_shader.cpp:312-442 / rt64_view.cpp:1225-1254): HUD
    triangles with corners behind the eye (w <= 0), outside 0 <= z <= w or far outside the viewport are clipped into pieces, not
    skipped.  The pieces' coverage is integer arithmetic on identically computed clip vertices: bit-exact against the oracle; colours
    within one RGBA8 step.  Drawn as the foreground list folded into the frame kernel and as its own launch."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    tris4 = [[(-0.8, -0.6, 0.2, 1.0), (0.9, -0.7, 0.2, 1.0), (0.1, 0.4, 0.5, -0.5)],
             [(-0.5, -0.9, 0.1, 0.6), (0.4, 0.8, 0.3, -0.2), (0.9, -0.2, 0.3, -0.4)],
             [(-0.9, -0.8, -0.4, 1.0), (0.8, -0.6, 0.5, 1.0), (0.0, 0.9, 1.6, 1.0)],
             [(-30.0, -0.5, 0.5, 1.0), (0.9, -0.6, 0.5, 1.0), (0.2, 25.0, 0.5, 1.0)],
             [(-0.2, -0.2, 0.5, 1.0), (0.3, -0.2, 0.5, 1.0), (0.0, 0.3, 0.5, 1.0)]]          # an ordinary one on top: draw order across clipped pieces

    def mod(d):
        m = _hud_mesh(sample_scene, rt64, [[(0, 0)] * 3] * len(tris4), alpha=0.7)
        k = 0
        for tri in tris4:
            for p in tri:
                m.vertices["position"][k] = p; k += 1
        d.meshes.append(m)
        i = copy.copy(d.instances[0]); i.mesh = len(d.meshes) - 1; i.material = sample_scene.copy_material(d.instances[0].material); i.name = "clipped"
        d.instances.append(i)
    data = _variant(sample_data, mod)
    finals = []
    for fold in (1, 0):
        got, ref, st = _render_pair(rt64_lib, data, options={"fold_foreground": fold})
        base = np.clip(np.floor(ref["output"][..., :3] * 255.0 + 0.5), 0, 255).astype(np.int32)
        touched_ref = np.abs(ref["final"][..., :3].astype(np.int32) - base).max(axis=2) > 0
        assert touched_ref.mean() > 0.25                                                   # the pieces cover a good part of the frame
        d = np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32))
        assert d.max() <= 2 and (d > 1).mean() < 1e-3, (int(d.max()), float((d > 1).mean()))
        finals.append(got["FINAL_RGBA8"])
    assert np.array_equal(finals[0], finals[1])


@pytest.mark.parametrize("many", [0, 1])
def test_frame_prologue_in_one_launch_equals_the_separate_copy_setups_and_clear(rt64_lib, sample_data, many):
    """frame_prologue (default 1): the table upload and the setup of the short raster lists leave as one launch, and gBackground is cleared by the draw
    that fills it.  The same host calls with frame_prologue = 0 (copy, one setup launch per list, memset, draw) give the same back buffer and the
    same gBackground byte for byte -- with the lists re-staged every frame (always_rebuild, the reference's behaviour), with a background list, a
    foreground list that is clipped, and (many = 1) a foreground list too long for the launch's arguments, which keeps its own setup launch."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene

    def mod(d):
        d.meshes.append(_hud_mesh(sample_scene, rt64, [[(-1.0, -0.2), (1.0, -0.1), (0.0, 1.0)], [(-1.0, 1.0), (-1.0, 0.2), (0.2, 1.0)]], alpha=0.9))
        i = copy.copy(d.instances[0]); i.mesh = len(d.meshes) - 1; i.material = sample_scene.copy_material(d.instances[0].material); i.name = "bg2"
        i.flags = rt64.INSTANCE_RASTER_BACKGROUND
        d.instances.append(i)
        m = _hud_mesh(sample_scene, rt64, [[(0, 0)] * 3] * 2, alpha=0.7)
        for k, p in enumerate([(-0.8, -0.6, 0.2, 1.0), (0.9, -0.7, 0.2, 1.0), (0.1, 0.4, 0.5, -0.5), (-0.2, -0.2, 0.5, 1.0), (0.3, -0.2, 0.5, 1.0), (0.0, 0.3, 0.5, 1.0)]):
            m.vertices["position"][k] = p
        d.meshes.append(m)
        for n in range(10 if many else 1):
            i = copy.copy(d.instances[0]); i.mesh = len(d.meshes) - 1; i.material = sample_scene.copy_material(d.instances[0].material); i.name = "fg%d" % n
            d.instances.append(i)
    data = _variant(sample_data, mod)
    out = {}
    for pro in (1, 0):
        s = sample_scene.Rt64Scene(rt64_lib, data, 272, 150, hip_device=0)
        try:
            assert s.option("frame_prologue", pro)
            s.option("always_rebuild", 1)
            frames = []
            for f in range(4):
                s.draw()
                frames.append((s.readback(rt64.IMAGE_FINAL_RGBA8).copy(), s.readback(rt64.IMAGE_BACKGROUND).copy()))
            out[pro] = frames
        finally:
            s.close()
    for (fa, ba), (fb, bb) in zip(out[1], out[0]):
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)
    assert (out[1][-1][1][..., 3] > 0).sum() > 100 and (out[1][-1][1][..., 3] == 0).sum() > 100        # gBackground: drawn where the list covers it, cleared elsewhere


def random_hud(sample_data, seed):
    """A seeded HUD: one to three raster instances (foreground or background) of one to six random clip-space triangles each -- corners inside, outside, far outside
    the viewport, with w from 0.3 to 3 and now and then behind the eye (w < 0) or outside 0 <= z <= w -- with random vertex alpha, scissor and viewport rectangles."""
    import random
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    rng = random.Random(seed)

    def mod(d):
        for n in range(rng.randint(1, 3)):
            count = rng.randint(1, 6)
            m = _hud_mesh(sample_scene, rt64, [[(0, 0)] * 3] * count, alpha=rng.choice([1.0, 0.7, 0.35]))
            for k in range(3 * count):
                w = rng.choice([1.0, 1.0, rng.uniform(0.3, 3.0), rng.uniform(0.3, 3.0), -rng.uniform(0.2, 1.0)])
                reach = rng.choice([1.0, 1.0, 1.6, 6.0])
                z = rng.choice([0.5, 0.5, 0.2, rng.uniform(-0.3, 1.4)]) * abs(w)
                m.vertices["position"][k] = (rng.uniform(-reach, reach) * abs(w), rng.uniform(-reach, reach) * abs(w), z, w)
            d.meshes.append(m)
            i = copy.copy(d.instances[0]); i.mesh = len(d.meshes) - 1; i.material = sample_scene.copy_material(d.instances[0].material); i.name = "fuzz%d" % n
            if rng.random() < 0.3:
                i.flags = rt64.INSTANCE_RASTER_BACKGROUND
            if rng.random() < 0.3:
                i.scissor = (rng.randint(0, W // 2), rng.randint(0, H // 2), rng.randint(8, W // 2), rng.randint(8, H // 2))
            if rng.random() < 0.3:
                i.viewport = (rng.randint(0, W // 3), rng.randint(0, H // 3), rng.randint(W // 4, W), rng.randint(H // 4, H))
            d.instances.append(i)
    return _variant(sample_data, mod)


def compare_hud(rt64_lib, data):
    got, ref, st = _render_pair(rt64_lib, data, images=("FINAL_RGBA8", "BACKGROUND"))
    bad = []
    if not np.array_equal(got["BACKGROUND"][..., 3] > 0, ref["background"][..., 3] > 0):
        bad.append("gBackground coverage differs on %d pixels" % int(((got["BACKGROUND"][..., 3] > 0) != (ref["background"][..., 3] > 0)).sum()))
    db = np.abs(got["BACKGROUND"].astype(np.int32) - ref["background"].astype(np.int32))
    if db.max() > 2 or (db > 1).mean() > 1e-3:
        bad.append("gBackground colours: max %d, %.5f beyond one step" % (int(db.max()), float((db > 1).mean())))
    d = np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32))
    if d.max() > 2 or (d > 1).mean() > 1e-3:
        bad.append("back buffer: max %d, %.5f beyond one step" % (int(d.max()), float((d > 1).mean())))
    return bad


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_random_hud_triangles_against_the_oracles_rasteriser(rt64_lib, sample_data, seed):
    """Coverage is integer arithmetic on identically clipped vertices: bit-exact.  Colours: one RGBA8 step (two where layers stack)."""
    bad = compare_hud(rt64_lib, random_hud(sample_data, seed))
    assert not bad, (seed, bad)
