"""The upscaler-equivalent stage: RT64_VIEW_DESC.upscaler / upscalerMode, Halton pixel jitter, and the temporal upscaler
that consumes rtOutput + flow + reactive mask + lock mask + depth where the reference calls a vendor SDK (rt64_upscaler.h:25-48,
rt64_view.cpp:1273-1281,1584-1618).
CPU: the quality -> render-size table and jitter phase count against the values rt64_fsr.cpp:98-130 / rt64_upscaler.cpp:11-36 imply, the
Halton sequence against a pure-Python restatement of rt64_common.h:347-361, and a property of the oracle's upscaler that does not share
its code: on a static scene the accumulated image approaches the native-resolution render more closely than a bilinear upsample does.
GPU: jittered primary rays hit exactly what the oracle's do, the upscaled image and the back buffer match the oracle's."""
import ctypes as C

import numpy as np
import pytest

W, H = 320, 180


def _halton(i, b):
    f, r = 1.0, 0.0
    f = np.float32(f); r = np.float32(r)
    while i > 0:
        f = np.float32(f / np.float32(b)); r = np.float32(r + f * np.float32(i % b)); i //= b
    return float(r)


def test_quality_table_and_jitter_phases(oracle_lib):
    L = oracle_lib
    def info(up, mode, dw, dh):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        ok = L.oracle_upscaler_info(up, mode, dw, dh, a, b, c)
        return ok, a.value, b.value, c.value
    assert info(0, 4, 1920, 1080)[0] == 0 and info(2, 4, 1920, 1080)[0] == 0 and info(4, 4, 1920, 1080)[0] == 0      # OFF, DLSS, XeSS: no stage here
    assert info(3, 6, 1920, 1080) == (1, 1920, 1080, 8)                      # native: 8 phases
    assert info(3, 5, 1920, 1080) == (1, 1478, 831, 13)                      # ultra quality: 77 %
    assert info(3, 4, 1920, 1080) == (1, 1280, 720, 18)                      # quality 1.5x
    assert info(3, 3, 1920, 1080) == (1, 1129, 635, 23)                      # balanced 1.7x
    assert info(3, 2, 1920, 1080) == (1, 960, 540, 32)                       # performance 2x
    assert info(3, 1, 1920, 1080) == (1, 640, 360, 72)                       # ultra performance 3x
    assert info(1, 0, 1280, 720)[1:3] == (985, 554) and info(1, 0, 1920, 1080)[1:3] == (1280, 720)      # auto: by display size
    assert info(1, 0, 2560, 1440)[1:3] == (1505, 847) and info(1, 0, 3840, 2160)[1:3] == (1920, 1080) and info(1, 0, 7680, 4320)[1:3] == (2560, 1440)


def test_jitter_follows_the_halton_2_3_sequence(sample_data):
    from oracle import oracle_py
    o = oracle_py.OracleScene(sample_data)
    try:
        seen = []
        for f in range(20):
            r = o.render(64, 36, upscaler=3, upscalerMode=4)                 # 64 / 42 -> phases = int(8 * (64/42)^2) = 18
            seen.append(r["pixelJitter"])
        for f, (jx, jy) in enumerate(seen):
            i = f % 18 + 1
            assert abs(jx - (_halton(i, 2) - 0.5)) < 1e-7 and abs(jy - (_halton(i, 3) - 0.5)) < 1e-7, f
        assert seen[18] == seen[0] and seen[1] != seen[0]
        assert o.render(64, 36)["pixelJitter"] == (0.0, 0.0)                 # no upscaler, no jitter (rt64_view.cpp:1273-1281)
    finally:
        o.close()


def test_accumulated_upsample_beats_a_bilinear_upsample_on_a_static_scene(sample_data):
    from oracle import oracle_py
    o = oracle_py.OracleScene(sample_data)
    n = oracle_py.OracleScene(sample_data)
    b = oracle_py.OracleScene(sample_data)
    try:
        for _ in range(18):
            r = o.render(W, H, upscaler=3, upscalerMode=4)
        native = n.render(W, H)
        bil = b.render(W, H, resolutionScale=213.0 / 320.0)
        assert r["output"].shape[:2] == (120, 213) and r["upscaled"].shape == (H, W, 4) and r["upscaled"][..., 3].max() == 18.0
        hud = np.zeros((H, W), dtype=bool); hud[:, :100] = True             # the HUD triangles are drawn at screen resolution in every variant
        err_taa = np.abs(r["final"].astype(np.float64) - native["final"])[~hud].mean()
        err_bil = np.abs(bil["final"].astype(np.float64) - native["final"])[~hud].mean()
        assert err_taa < 0.9 * err_bil, (err_taa, err_bil)
    finally:
        o.close(); n.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,gi", [(4, 0), (2, 1)])
def test_upscaler_parity_with_jitter_and_camera_motion(rt64_lib, sample_data, mode, gi):
    """FSR-slot upscaler, quality (1.5x) and performance (2x, with 1 GI sample + SVGF): eight frames, the camera strafes from frame 4 on so
    that the history is fetched along real motion vectors.  Jittered primary hits are bit-exact; the upscaled image (an accumulation of
    eight frames), its frame count channel and the back buffer match the oracle."""
    import copy
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    data = copy.copy(sample_data)
    base = sample_data.view.copy()
    s = sample_scene.Rt64Scene(rt64_lib, data, W, H, hip_device=0)
    o = oracle_py.OracleScene(data)
    try:
        assert rt64_lib.GetViewUpscalerSupport(s.view, rt64.UPSCALER_FSR) and not rt64_lib.GetViewUpscalerSupport(s.view, rt64.UPSCALER_DLSS)
        s.set_view_description(gi_samples=gi, denoiser=bool(gi), upscaler=rt64.UPSCALER_FSR, upscaler_mode=mode)
        for f in range(8):
            v = base.copy(); v[3, 0] = base[3, 0] - 0.05 * max(0, f - 3)
            data.view = v
            s.draw()
            ref = o.render(W, H, giSamples=gi, denoiserEnabled=int(bool(gi)), denoiserMode=1, upscaler=rt64.UPSCALER_FSR, upscalerMode=mode)
        st = s.stats()
        rw, rh = (213, 120) if mode == 4 else (160, 90)
        assert (st.width, st.height, st.screenWidth, st.screenHeight) == (rw, rh, W, H) and ref["output"].shape[:2] == (rh, rw)
        hit = s.readback(rt64.IMAGE_PRIMARY_HIT)
        assert np.array_equal(hit, ref["primaryHit"])                       # same jitter, same rays
        assert np.abs(ref["flow"]).max() > 0.5
        up = s.readback(rt64.IMAGE_UPSCALED)
        assert up.shape == (H, W, 4) and np.abs(up[..., 3] - ref["upscaled"][..., 3]).max() < 0.05
        rm = float(np.sqrt(np.mean((up[..., :3].astype(np.float64) - ref["upscaled"][..., :3]) ** 2)))
        assert rm <= 1e-3, rm
        final = s.readback(rt64.IMAGE_FINAL_RGBA8)
        d = np.abs(final.astype(np.int32) - ref["final"].astype(np.int32))
        assert d.max() <= 3 and (d > 1).mean() < 2e-3, (int(d.max()), float((d > 1).mean()))
        # turning the upscaler off again drops the jitter and goes back to the native render size
        s.set_view_description()
        s.draw()
        ref = o.render(W, H)
        assert np.array_equal(s.readback(rt64.IMAGE_PRIMARY_HIT), ref["primaryHit"]) and s.stats().width == W
    finally:
        s.close(); o.close()
