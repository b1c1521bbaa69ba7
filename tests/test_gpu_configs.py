"""GPU parity of BASELINE.json's configurations C4 and C5 exactly as bench.py defines them (sample_scene.BENCH_CONFIGS /
apply_bench_config -- the same code bench.py calls), at a reduced frame size, against the CPU oracle; the several-tiles-per-workgroup
walk of the one-kernel frame (every frame above 1080p runs on it); the sky modifiers of RT64_SCENE_DESC (Color.hlsli:9-43,
BgSky.hlsli:20-93); and the raster background cache key (a texture swap in the same slot).
Tolerances: hit records / ids exact, composed image RMSE <= 1e-3 (BASELINE.json gate), filtered GI RMSE <= 2e-3."""
import copy
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 320, 180


def _variant(sample_data, fn=None):
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    d = copy.copy(sample_data)
    d.instances = [copy.copy(i) for i in sample_data.instances]
    for i in d.instances:
        i.material = sample_scene.copy_material(i.material)
    d.meshes = [copy.copy(m) for m in sample_data.meshes]
    desc = rt64.SCENE_DESC(); C.memmove(C.byref(desc), C.byref(sample_data.desc), C.sizeof(rt64.SCENE_DESC)); d.desc = desc
    if fn:
        fn(d)
    return d


def _rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


_ORACLE_FRAMES = {}          # (config, frames, width, height) -> the oracle's last frame: the A/B tests below render the same configuration several times


def _oracle_frames(sample_data, config, frames, width, height):
    """Frame `frames` of bench.py's `--config` on the oracle (the call sequence of _bench_pair); small frames are kept for the session."""
    from sm64rt_legacy_renderer_amd import sample_scene
    from oracle import oracle_py
    key = (config, frames, width, height)
    if key in _ORACLE_FRAMES:
        return _ORACLE_FRAMES[key]
    data = _variant(sample_data)
    cfg = sample_scene.BENCH_CONFIGS[config]
    anim = sample_scene.apply_bench_config(data, config)
    o = oracle_py.OracleScene(data)
    try:
        kw = dict(giSamples=cfg["gi_samples"], denoiserEnabled=int(cfg["denoiser"]), denoiserMode=1, primarySpp=cfg.get("primary_spp", 1), giBounces=cfg.get("gi_bounces", 1))
        for f in range(frames):
            if anim is not None:
                o.set_mesh(o.meshes[0], anim[(f + 1) % len(anim)], data.meshes[0].indices)
            ref = o.render(width, height, images=(f == frames - 1), **kw)
    finally:
        o.close()
    if width * height <= W * H:
        _ORACLE_FRAMES[key] = ref
    return ref


def _bench_pair(rt64_lib, sample_data, config, frames, width=W, height=H, bands=None, options=None, need_ref=True):
    """Render `frames` steps of bench.py's `--config` on the HIP library (whole frame, or one device per band) and on the oracle (need_ref = False: the
    caller compares two runs of the library with each other and wants no oracle frame)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    data = _variant(sample_data)
    cfg = sample_scene.BENCH_CONFIGS[config]
    anim = sample_scene.apply_bench_config(data, config)
    parts = [sample_scene.Rt64Scene(rt64_lib, data, width, height, hip_device=0) for _ in (bands or [None])]
    try:
        for s, band in zip(parts, bands or [None]):
            s.set_view_description(gi_samples=cfg["gi_samples"], denoiser=cfg["denoiser"])
            for k, v in (options or {}).items():
                assert s.option(k, v)
            if band:
                s.set_tile(*band)
            s.option("count_traversal", 1)
            assert s.option("primary_spp", cfg.get("primary_spp", 1)) and s.option("gi_bounces", cfg.get("gi_bounces", 1))
        for f in range(frames):
            if anim is not None:          # bench.py step(): frame_no += 1; SetMesh(anim[frame_no % len])
                v = anim[(f + 1) % len(anim)]
                for s in parts:
                    s.set_mesh(s.meshes[0], v, data.meshes[0].indices)
            for s in parts:
                s.draw()
        names = ("OUTPUT_RGBA32F", "FINAL_RGBA8", "PRIMARY_HIT", "INDIRECT_LIGHT_RAW", "INDIRECT_LIGHT_FILTERED", "REFLECTION", "DIFFUSE", "INSTANCE_ID")
        got = {k: np.concatenate([s.readback(getattr(rt64, "IMAGE_" + k)) for s in parts], axis=0) for k in names}
        stats = [s.stats() for s in parts]
    finally:
        for s in parts:
            s.close()
    return got, (_oracle_frames(sample_data, config, frames, width, height) if need_ref else None), stats


def _check_gi_frame(got, ref):
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    assert _rmse(got["FINAL_RGBA8"][..., :3] / 255.0, ref["final"][..., :3] / 255.0) <= 1e-3
    assert _rmse(got["INDIRECT_LIGHT_FILTERED"][..., :3], ref["filteredIndirect"][..., :3]) <= 2e-3
    assert np.abs(got["DIFFUSE"] - ref["diffuse"]).max() <= 1.0 / 255.0 + 1e-6


def test_c4_refit_every_frame_with_two_gi_samples_and_svgf(rt64_lib, sample_data):
    """C4 as `bench.py --config C4` sets it: the sphere is UPDATABLE and re-sent (refit) every frame, 2 GI samples, SVGF -- 5 frames of
    temporal history over moving geometry."""
    got, ref, st = _bench_pair(rt64_lib, sample_data, "C4", frames=5)
    _check_gi_frame(got, ref)
    c = ref["counters"]
    assert st[0].primaryRays == c["primaryRays"] == W * H and st[0].indirectRays == c["indirectRays"] > 0
    assert st[0].leanFrame == 0 and st[0].fusedFrame == 2
    hist = ref["indirectLight"][..., 3]
    assert hist.max() >= 6.0                      # 2 samples per frame: history grows by 2 per frame on pixels that stay valid
    hit = ref["instanceId"] >= 0
    assert np.abs(got["INDIRECT_LIGHT_RAW"][..., 3][hit] - hist[hit]).max() < 1.01


@pytest.mark.parametrize("config,width,height,frames", [("C2", 1920, 1080, 1), ("C3", 1920, 1080, 2), ("C4", 2560, 1440, 2), ("C5", 3840, 2160, 2), ("C4-literal", 2560, 1440, 1), ("C5-literal", 3840, 2160, 1)])
def test_baseline_size_frames_against_the_oracle(rt64_lib, sample_data, config, width, height, frames):
    """BASELINE.json's configurations at their OWN sizes (C2, C3: 1920 x 1080; C4: 2560 x 1440; C5: 3840 x 2160; and C4 / C5 as BASELINE words them, with the
    primary_spp / gi_bounces extensions), as bench.py sets them up: hit records bit-exact, composed image within the BASELINE gate (RMSE <= 1e-3), back
    buffer within one RGBA8 step almost everywhere, every ray's visit counters equal to the oracle's.  These sizes run the code the small frames of the other
    tests do not: 8 160 workgroups of the one-kernel frame with several tiles each above 1080p, the per-tile grids of the bounce kernels, full-size SVGF.
    (Round 3 ran C4 / C5 at half size per axis for the oracle's time; the oracle took 1.2 s per small frame then because OpenMP started one thread per
    logical CPU of the box: oracle_py.host_threads.)"""
    got, ref, st = _bench_pair(rt64_lib, sample_data, config, frames=frames, width=width, height=height)
    assert got["PRIMARY_HIT"].shape == (height, width, 4)
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    assert _rmse(got["FINAL_RGBA8"][..., :3] / 255.0, ref["final"][..., :3] / 255.0) <= 1e-3
    d = np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32))
    assert (d > 1).mean() < 1e-5 and d.max() <= 16, (d.max(), (d > 1).mean())     # (round 3 allowed 2e-3 of a GI frame's pixels beyond one step: the bounce directions went through different sin / cos then; measured now: 9 of 8.3 M pixels of C5 at 4K)
    assert np.abs(got["DIFFUSE"] - ref["diffuse"]).max() <= 1.0 / 255.0 + 1e-6
    c = ref["counters"]
    from sm64rt_legacy_renderer_amd import sample_scene
    assert st[0].primaryRays == c["primaryRays"] == width * height * sample_scene.BENCH_CONFIGS.get(config, {}).get("primary_spp", 1)
    assert st[0].nodesPrimary == c["nodesVisitedPrimary"] and st[0].trianglesPrimary == c["trianglesTestedPrimary"]
    # every ray of the frame -- primary, shadow, GI bounce (direction spec D1: the same sine / cosine bits on both sides), reflection -- walks the oracle's
    # nodes and tests the oracle's triangles: the geometry contract no longer stops at the first hit
    # -- exactly on frames whose rays all start from exact values (C2: primary + shadow rays), and to a few visits in a hundred million where a ray starts from a
    # tolerance-level shading result: a bounce ray's frame is the RGBA16F shading normal (about 150 of 2 M pixels round to a neighbouring half on the two sides) and a
    # second bounce starts from the first hit's un-rounded normal.  Measured: C3 / C5 at 1080p 0 of 19.7 M / 41.4 M node visits, C4-literal 1 of 14.1 M triangle tests,
    # C5-literal 5 of 385.6 M node visits.
    slack = (lambda total: 0) if config == "C2" else (lambda total: max(16, int(2e-7 * total)))
    assert abs(int(st[0].nodesVisited) - c["nodesVisited"]) <= slack(c["nodesVisited"]), (st[0].nodesVisited, c["nodesVisited"])
    assert abs(int(st[0].trianglesTested) - c["trianglesTested"]) <= slack(c["trianglesTested"]), (st[0].trianglesTested, c["trianglesTested"])
    if config == "C2":           # (the oracle counts the shadow rays of every pass together; the library by pass: only a frame without GI / reflection has the same split)
        assert st[0].nodesDirect == c["nodesVisitedShadow"] and st[0].trianglesDirect == c["trianglesTestedShadow"]
    if config != "C2":
        assert _rmse(got["INDIRECT_LIGHT_FILTERED"][..., :3], ref["filteredIndirect"][..., :3]) <= 2e-3
        assert st[0].indirectRays == c["indirectRays"] > 0
        assert st[0].nodesIndirect > 0 and st[0].reflectionRays == c["reflectionRays"]


@pytest.mark.parametrize("bands", [None, [(0, 64), (64, 121), (121, H)]])
def test_c5_four_gi_samples_reflective_floor_svgf(rt64_lib, sample_data, bands):
    """C5 as `bench.py --config C5` sets it: 4 GI samples, floor reflectionFactor 0.3 (two reflection bounces), SVGF -- once whole, once
    as the contiguous bands + denoiser halo of the multi-GPU partition (ragged on purpose)."""
    got, ref, st = _bench_pair(rt64_lib, sample_data, "C5", frames=4, bands=bands)
    _check_gi_frame(got, ref)
    assert np.abs(got["REFLECTION"] - ref["reflection"]).max() < 8e-3
    assert (ref["reflection"][..., :3] > 0.01).mean() > 0.1
    c = ref["counters"]
    if not bands:
        assert st[0].reflectionRays == c["reflectionRays"] > 0 and st[0].indirectRays == c["indirectRays"] > 0
    else:                                          # pixel-local passes stay on the owned rows: their rays add up to the whole frame's
        assert sum(s.reflectionRays for s in st) == c["reflectionRays"] > 0


@pytest.mark.parametrize("config,frames,bands", [("C4-literal", 3, None), ("C5-literal", 2, None), ("C5-literal", 2, [(0, 64), (64, 121), (121, H)])])
def test_c4_c5_as_baseline_words_them_primary_spp_and_two_bounces(rt64_lib, sample_data, config, frames, bands):
    """BASELINE.json words C4 "2-bounce GI 1440p 2spp" and C5 "4K 4spp full path trace".  The reference has neither knob; the library carries both as
    extensions (device options primary_spp / gi_bounces) and the oracle implements the same rules (P1-P4, B1-B3 in oracle/oracle_render.c): N jittered
    sub-frames -- every pass up to Compose, history advancing after each -- averaged before PostProcess, and a second cosine-weighted bounce whose
    radiance stands where the constant ambient term stands at the first hit.  Hit records of the last sub-frame bit-exact, the averaged image within the
    BASELINE gate, rays counted over all sub-frames."""
    from sm64rt_legacy_renderer_amd import sample_scene
    cfg = sample_scene.BENCH_CONFIGS[config]
    got, ref, st = _bench_pair(rt64_lib, sample_data, config, frames=frames, bands=bands)
    _check_gi_frame(got, ref)
    c = ref["counters"]
    if bands:            # the multi-GPU partition (ragged bands + denoiser halo, every sub-frame): the bands put together are the oracle's frame; halo rows are traced again
        assert sum(s.primaryRays for s in st) > c["primaryRays"] and sum(s.reflectionRays for s in st) == c["reflectionRays"] > 0
        return
    assert st[0].primaryRays == c["primaryRays"] == cfg["primary_spp"] * W * H
    assert st[0].indirectRays == c["indirectRays"]
    hit = int((ref["instanceId"] >= 0).sum())
    assert c["indirectRays"] > cfg["primary_spp"] * hit * 1.02          # first-bounce rays of every sub-frame + the second bounces of those that hit a surface
    if config == "C5-literal":
        assert np.abs(got["REFLECTION"] - ref["reflection"]).max() < 8e-3 and st[0].reflectionRays == c["reflectionRays"] > 0


def test_second_bounce_on_the_wavefront_and_the_k_buffer_paths(rt64_lib, sample_data):
    """gi_bounces = 2 against the oracle and against itself switched off: more indirect rays (one more per GI ray that resolved to a surface), a raw GI image
    that moved, both on the wavefront GI kernels of an opaque frame (bounce_hit_kernel<.., true>) and on the one-kernel k-buffer form a translucent
    instance forces (indirect_kernel<true, true>); values other than 1 and 2 are refused."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    out = {}
    for name, translucent in (("opaque", False), ("klist", True)):
        def mod(d, translucent=translucent):
            if translucent:
                d.instances[0].material.solidAlphaMultiplier = 0.6          # not provably opaque any more: k-buffer kernels (indirect_kernel<true>)
        data = _variant(sample_data, mod)
        for bounces in (1, 2):
            s = sample_scene.Rt64Scene(rt64_lib, data, W, H, hip_device=0)
            o = oracle_py.OracleScene(data)
            try:
                s.set_view_description(gi_samples=2, denoiser=False)
                assert s.option("gi_bounces", bounces) and s.option("count_traversal", 1)
                for _ in range(2):
                    s.draw()
                    ref = o.render(W, H, giSamples=2, giBounces=bounces)
                got = {k: s.readback(getattr(rt64, "IMAGE_" + k)) for k in ("OUTPUT_RGBA32F", "PRIMARY_HIT", "INDIRECT_LIGHT_RAW")}
                st = s.stats()
            finally:
                s.close(); o.close()
            assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
            assert _rmse(got["INDIRECT_LIGHT_RAW"][..., :3], ref["indirectLight"][..., :3]) <= 2e-3, (name, bounces)
            assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
            assert st.indirectRays == ref["counters"]["indirectRays"]
            out[name, bounces] = (ref, st)
        one, two = out[name, 1], out[name, 2]
        assert two[1].indirectRays > one[1].indirectRays
        assert np.abs(two[0]["indirectLight"][..., :3] - one[0]["indirectLight"][..., :3]).mean() > 1e-4
    s = sample_scene.Rt64Scene(rt64_lib, _variant(sample_data), W, H, hip_device=0)
    try:
        assert not s.option("gi_bounces", 3) and not s.option("gi_bounces", 0) and not s.option("primary_spp", 0)
    finally:
        s.close()


def test_primary_spp_on_a_frame_without_gi(rt64_lib, sample_data):
    """primary_spp = 4 on the plain sample frame (no GI: otherwise a lean one-kernel frame): four jittered full frames averaged.  Parity with the oracle, rays
    counted four times, silhouette pixels take values the one-sample frame does not have, the HUD is drawn once over the mean, and a frame after the
    option is switched off is the plain frame again (byte for byte)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    data = _variant(sample_data)
    s = sample_scene.Rt64Scene(rt64_lib, data, W, H, hip_device=0)
    o = oracle_py.OracleScene(data)
    try:
        assert s.option("count_traversal", 1)
        s.draw()
        plain = {k: s.readback(getattr(rt64, "IMAGE_" + k)) for k in ("OUTPUT_RGBA32F", "FINAL_RGBA8")}
        ref1 = o.render(W, H)
        assert s.option("primary_spp", 4)
        for _ in range(2):
            s.draw()
            ref = o.render(W, H, primarySpp=4)
        got = {k: s.readback(getattr(rt64, "IMAGE_" + k)) for k in ("OUTPUT_RGBA32F", "FINAL_RGBA8", "PRIMARY_HIT")}
        st = s.stats()
        assert st.primaryRays == ref["counters"]["primaryRays"] == 4 * W * H and st.shadowRays == ref["counters"]["shadowRays"]
        assert st.nodesPrimary == ref["counters"]["nodesVisitedPrimary"] and st.trianglesPrimary == ref["counters"]["trianglesTestedPrimary"]
        assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])                # the last sub-frame's records
        assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-4
        assert np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32)).max() <= 1
        moved = np.abs(ref["output"][..., :3] - ref1["output"][..., :3]).max(axis=-1) > 0.02
        assert 0.002 < moved.mean() < 0.2                                           # edges and texture detail, not the whole picture
        assert np.abs(got["OUTPUT_RGBA32F"][..., :3] - plain["OUTPUT_RGBA32F"][..., :3]).max(axis=-1)[moved].min() > 0.01
        assert s.option("primary_spp", 1)
        s.draw()
        again = {k: s.readback(getattr(rt64, "IMAGE_" + k)) for k in ("OUTPUT_RGBA32F", "FINAL_RGBA8")}
        for k in plain:
            assert np.array_equal(plain[k].view(np.uint8), again[k].view(np.uint8)), k
    finally:
        s.close(); o.close()


@pytest.mark.parametrize("config", ["C2", "C3"])
def test_several_tiles_per_workgroup_walk_of_the_one_kernel_frame(rt64_lib, sample_data, config):
    """launch_lean_frame gives every workgroup ceil(tiles / max groups) tiles once a frame has more than RT_MAX_FRAME_GROUPS tiles (every
    frame above 1080p: C4 and C5).  max_frame_groups = 50 puts a 320 x 180 frame (240 tiles) on that walk (5 tiles per workgroup) for
    the lean kernel (C2, FULL = false) and the full one (C3, FULL = true): same bytes as one workgroup per tile, and parity with the oracle."""
    got, ref, st = _bench_pair(rt64_lib, sample_data, config, frames=3, options={"max_frame_groups": 50})
    base, _, st0 = _bench_pair(rt64_lib, sample_data, config, frames=3, need_ref=False)
    assert st[0].fusedFrame == st0[0].fusedFrame == (1 if config == "C2" else 2)
    for k in got:
        assert np.array_equal(got[k].view(np.uint8), base[k].view(np.uint8)), k
    assert (st[0].primaryRays, st[0].shadowRays, st[0].nodesVisited, st[0].trianglesTested) == (st0[0].primaryRays, st0[0].shadowRays, st0[0].nodesVisited, st0[0].trianglesTested)
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    c = ref["counters"]          # (GI frames included: bounce directions are bit-exact since round 4, direction spec D1)
    assert st[0].nodesPrimary == c["nodesVisitedPrimary"] and st[0].trianglesPrimary == c["trianglesTestedPrimary"]
    assert st[0].nodesVisited == c["nodesVisited"] and st[0].trianglesTested == c["trianglesTested"]


def test_sky_modifiers_hsl_yaw_and_diffuse_multiplier(rt64_lib, sample_data):
    """RT64_SCENE_DESC.skyHSLModifier / skyYawOffset / skyDiffuseMultiplier away from their defaults: ModRGBWithHSL (Color.hlsli:9-43)
    on the sky plane of the primary rays (SampleSky2D) and of the GI / reflection rays (SampleSkyPlane, BgSky.hlsli:54-87)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py

    def mod(d):
        d.desc.skyHSLModifier = rt64.VECTOR3(0.13, -0.21, 0.06)
        d.desc.skyYawOffset = 0.83
        d.desc.skyDiffuseMultiplier = rt64.VECTOR3(0.9, 1.1, 0.7)
        d.instances[3].material.reflectionFactor = 0.4
    data = _variant(sample_data, mod)
    plain = _variant(sample_data, lambda d: setattr(d.instances[3].material, "reflectionFactor", 0.4))
    out = {}
    for name, dd in (("mod", data), ("plain", plain)):
        s = sample_scene.Rt64Scene(rt64_lib, dd, W, H, hip_device=0)
        o = oracle_py.OracleScene(dd)
        try:
            s.set_view_description(gi_samples=1, denoiser=False)
            for _ in range(2):
                s.draw()
                ref = o.render(W, H, giSamples=1)
            out[name] = ({k: s.readback(getattr(rt64, "IMAGE_" + k)) for k in ("OUTPUT_RGBA32F", "FINAL_RGBA8", "PRIMARY_HIT", "DIFFUSE", "INDIRECT_LIGHT_RAW", "REFLECTION")}, ref)
        finally:
            s.close(); o.close()
    got, ref = out["mod"]
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
    assert np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32)).max() <= 1
    sky = ref["primaryHit"][..., 3] == 0xFFFFFFFF
    assert np.abs(got["DIFFUSE"][sky] - ref["diffuse"][sky]).max() <= 1.0 / 255.0 + 1e-6          # the modified sky colour itself, RGBA8
    assert _rmse(got["INDIRECT_LIGHT_RAW"][..., :3], ref["indirectLight"][..., :3]) <= 2e-3
    assert np.abs(got["REFLECTION"] - ref["reflection"]).max() < 8e-3
    # ... and the modifiers did something: sky pixels, GI and the mirrored sky all moved away from the unmodified scene's
    p = out["plain"][1]
    assert np.abs(ref["diffuse"][sky][..., :3] - p["diffuse"][sky][..., :3]).mean() > 0.03
    assert np.abs(ref["indirectLight"][..., :3] - p["indirectLight"][..., :3]).mean() > 1e-3
    assert np.abs(ref["reflection"][..., :3] - p["reflection"][..., :3]).mean() > 1e-3


def test_background_texture_swapped_in_the_same_slot_redraws_gbackground(rt64_lib, sample_data):
    """The raster lists are cached on their table bytes; a texture is only a slot number there, so the key also carries the texture
    object's identity: swapping the background instance's diffuse texture for another one that lands in the same slot must redraw
    gBackground (the environment map of missed / GI / reflection rays)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    T_CLOUDS, T_CHECKER = 3, 7
    yy, xx = np.mgrid[0:64, 0:64]
    checker = np.zeros((64, 64, 4), dtype=np.uint8)
    checker[..., 0] = np.where((xx // 8 + yy // 8) % 2, 230, 30); checker[..., 1] = 90; checker[..., 2] = np.where((xx // 8 + yy // 8) % 2, 20, 200); checker[..., 3] = 255

    def make(tex):
        def mod(d):
            d.sky = None                                   # no sky plane: the background shows in every missed pixel (and clouds.png is otherwise unused)
            d.textures = list(d.textures) + [sample_scene.TextureData("checker", rt64.TEXTURE_FORMAT_RGBA8, checker, 64, 64)]
            for i in d.instances:
                if i.name == "hudA":                       # the RASTER_BACKGROUND instance: it is the only user of `tex`, which gets slot 4 of the frame either way
                    i.diffuse = tex
            m = copy.copy(d.meshes[1]); v = m.vertices.copy()
            v["position"][:, :2] = [(-1.0, -1.0), (3.0, -1.0), (-1.0, 3.0)]; v["uv"] = [(0.0, 1.0), (2.0, 1.0), (0.0, -1.0)]      # cover the screen
            m.vertices = v; d.meshes[1] = m
        return _variant(sample_data, mod)
    a, b = make(T_CLOUDS), make(T_CHECKER)
    s = sample_scene.Rt64Scene(rt64_lib, a, W, H, hip_device=0)
    oa, ob = oracle_py.OracleScene(a), oracle_py.OracleScene(b)
    try:
        s.draw(); s.draw()
        bg_a = s.readback(rt64.IMAGE_BACKGROUND)
        ra = oa.render(W, H)
        assert np.abs(bg_a.astype(np.int32) - ra["background"].astype(np.int32)).max() <= 1
        k = next(i for i, inst in enumerate(b.instances) if inst.name == "hudA")
        s.data = b
        s.set_instance(k, b.instances[k])                  # same mesh, same shader, other texture: the slot order of the frame does not change
        s.draw()
        bg_b = s.readback(rt64.IMAGE_BACKGROUND)
        rb = ob.render(W, H)
        assert np.abs(bg_b.astype(np.int32) - rb["background"].astype(np.int32)).max() <= 1
        assert np.abs(rb["background"].astype(np.int32) - ra["background"].astype(np.int32)).mean() > 2.0
        final = s.readback(rt64.IMAGE_FINAL_RGBA8)
        assert np.abs(final.astype(np.int32) - rb["final"].astype(np.int32)).max() <= 1
    finally:
        s.close(); oa.close(); ob.close()


@pytest.mark.parametrize("config", ["C2", "C3", "C5"])
def test_simple_frame_kernels_equal_the_general_kernels(rt64_lib, sample_data, config):
    """The sample scene is a "simple" frame (every texture a power of two in both sizes, every instance shadow-opaque): it runs the kernels of
    passes_simple.hip, compiled without non-power-of-two addressing and without the shadow any-hit program.  Device option simple_kernels = 0
    sends the same frames through the general kernels: every image and every counter is identical."""
    a, _, sa = _bench_pair(rt64_lib, sample_data, config, frames=3, need_ref=False)
    b, _, sb = _bench_pair(rt64_lib, sample_data, config, frames=3, options={"simple_kernels": 0}, need_ref=False)
    for k in a:
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    assert (sa[0].primaryRays, sa[0].shadowRays, sa[0].indirectRays, sa[0].reflectionRays, sa[0].nodesVisited, sa[0].trianglesTested) == \
           (sb[0].primaryRays, sb[0].shadowRays, sb[0].indirectRays, sb[0].reflectionRays, sb[0].nodesVisited, sb[0].trianglesTested)


@pytest.mark.parametrize("config,bands", [("C3", None), ("C4", None), ("C5", None), ("C5", [(0, 64), (64, 121), (121, H)])])
def test_folded_guide_and_compose_equal_their_own_launches(rt64_lib, sample_data, config, bands):
    """Frames with the wavefront GI chain + SVGF: bounce_resolve_kernel writes the filter's guide records (the same bytes as svgf_guide_kernel) and the last
    a-trous iteration composes its pixels itself (the operations of compose_post_kernel on the value it has just rounded).  Device options fold_guide = 0 /
    fold_compose = 0 bring the two launches back; fold_variance = 0 has svgf_variance_kernel make the filter input of every pixel again instead of only the young ones
    bounce_resolve_kernel marked.  The guide and variance folds change no byte.  The Compose fold is another instantiation of the a-trous kernel, whose
    arithmetic is compiled with fp-contract(fast) (tolerance-tested filter): the compiler may fuse a multiply-add differently in it, so a handful of filtered
    values may differ by one RGBA16F step -- and nothing else."""
    a, ref, sa = _bench_pair(rt64_lib, sample_data, config, frames=3, bands=bands)
    g, _, _ = _bench_pair(rt64_lib, sample_data, config, frames=3, bands=bands, options={"fold_guide": 0}, need_ref=False)
    v, _, _ = _bench_pair(rt64_lib, sample_data, config, frames=3, bands=bands, options={"fold_variance": 0}, need_ref=False)
    for k in a:
        assert np.array_equal(a[k].view(np.uint8), g[k].view(np.uint8)), k
        assert np.array_equal(a[k].view(np.uint8), v[k].view(np.uint8)), k
    b, _, sb = _bench_pair(rt64_lib, sample_data, config, frames=3, bands=bands, options={"fold_guide": 0, "fold_compose": 0}, need_ref=False)
    for k in a:
        if k in ("OUTPUT_RGBA32F", "FINAL_RGBA8", "INDIRECT_LIGHT_FILTERED"):
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            assert (d > 0).any(axis=-1).mean() < 2e-3 and d.max() <= (1.0 if k == "FINAL_RGBA8" else 1.0 / 512.0), (k, d.max(), (d > 0).mean())
        else:
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    _check_gi_frame(a, ref)
