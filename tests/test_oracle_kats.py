"""Known-answer tests that pin the CPU oracle's helpers against independent restatements (pure Python / numpy) of the
published formulas the reference uses.  CPU only."""
import ctypes as C
import json
import math
import os

import numpy as np

from oracle import oracle_py

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _tea(v0, v1, backoff=16):
    """Random.hlsli:14-26 in Python integers."""
    M = 0xFFFFFFFF
    s0 = 0
    for _ in range(backoff):
        s0 = (s0 + 0x9e3779b9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xa341316c) & M ^ ((v1 + s0) & M) ^ (((v1 >> 5) + 0xc8013ea4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xad90777d) & M ^ ((v0 + s0) & M) ^ (((v0 >> 5) + 0x7e95761e) & M))) & M
    return v0


def test_init_rand_and_next_rand(oracle_lib):
    kats = json.load(open(os.path.join(GOLD, "kats.json")))
    for k in kats["init_rand"]:
        assert oracle_lib.oracle_init_rand(k["val0"], k["val1"], 16) == k["seed"] == _tea(k["val0"], k["val1"])
        s = k["seed"]
        st = C.c_uint32(s)
        for expected in k["next"]:
            s = (1664525 * s + 1013904223) & 0xFFFFFFFF          # Random.hlsli:28-30
            assert oracle_lib.oracle_next_rand(C.byref(st)) == np.float32((s & 0x00FFFFFF) / float(0x01000000)) == np.float32(expected)


def test_halton_sequence(oracle_lib):
    def halton(i, b):                                             # rt64_common.h:347-357 in float32
        f, r = np.float32(1.0), np.float32(0.0)
        while i > 0:
            f = np.float32(f / np.float32(b)); r = np.float32(r + f * np.float32(i % b)); i //= b
        return r
    for i in range(1, 64):
        for b in (2, 3):
            assert oracle_lib.oracle_halton(i, b) == halton(i, b)
    assert oracle_lib.oracle_halton(1, 2) == 0.5 and oracle_lib.oracle_halton(2, 2) == 0.25 and oracle_lib.oracle_halton(3, 2) == 0.75


def test_morton30(oracle_lib):
    def interleave(x, y, z):
        code = 0
        for b in range(10):
            code |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
        return code
    rng = np.random.default_rng(1)
    for x, y, z in rng.integers(0, 1024, size=(500, 3)):
        assert oracle_lib.oracle_morton30(int(x), int(y), int(z)) == interleave(int(x), int(y), int(z))
    assert oracle_lib.oracle_morton30(1023, 1023, 1023) == (1 << 30) - 1


def test_combiner_decode_sample_shader(oracle_lib):
    """shaderId 0x01200a00 of the sample (main.cpp:217) expands as in SURVEY appendix A3."""
    out = (C.c_int * 28)()
    oracle_lib.oracle_decode_combiner(0x01200a00, out)
    o = list(out)
    assert o[0:4] == [0, 0, 0, 5]            # colour slots: 0,0,0,TEXEL0
    assert o[4:8] == [0, 0, 0, 1]            # alpha slots: 0,0,0,INPUT_1
    assert o[8] == 1 and o[9] == 1 and o[10] == 0           # one vertex input, texture 0 used
    assert o[11] == 1 and o[12] == 1                        # do_single for colour and alpha
    assert o[17] == 0 and o[18] == 1 and o[19] == 0 and o[20] == 0   # colour != alpha, opt_alpha, no edge, no noise
    assert o[21] == 52 and o[22] == 16 and o[23] == 28 and o[24] == 36   # VERTEX of main.cpp:36-41
    # a shader without opt_alpha and with two inputs: float3 inputs, no uv
    oracle_lib.oracle_decode_combiner((1 << 0) | (2 << 3), out)
    assert out[8] == 2 and out[18] == 0 and out[21] == 16 + 12 + 12 + 12


def test_half_float_conversion_matches_numpy(oracle_lib):
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    for h in (0, 1, 0x3C00, 0x7BFF, 0x7C00, 0xFBFF, 0x8000, 0x03FF, 0x0400):
        assert oracle_lib.oracle_f16_to_f32(h) == f[h] or (np.isnan(f[h]))
    rng = np.random.default_rng(2)
    vals = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 100, rng.random(20000).astype(np.float32),
                           np.array([0.0, -0.0, 1.0, 65504.0, 65520.0, 1e-8, 6.1e-5, 5.96e-8, 2.98e-8, np.inf, -np.inf], dtype=np.float32)])
    got = np.array([oracle_lib.oracle_f32_to_f16(float(v)) for v in vals], dtype=np.uint16)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, want)
    back = np.array([oracle_lib.oracle_f16_to_f32(int(h)) for h in got], dtype=np.float32)
    assert np.array_equal(back, want.view(np.float16).astype(np.float32), equal_nan=True)


def test_perspective_and_inverse(oracle_lib):
    fov, aspect, zn, zf = math.radians(45.0), 16.0 / 9.0, 0.1, 1000.0
    P = oracle_py.M4()
    oracle_lib.oracle_perspective_fov_rh(fov, aspect, zn, zf, C.byref(P))
    p = P.to_numpy()
    h = 1.0 / math.tan(fov / 2)
    want = np.array([[h / aspect, 0, 0, 0], [0, h, 0, 0], [0, 0, zf / (zn - zf), -1], [0, 0, zn * zf / (zn - zf), 0]])
    assert np.allclose(p, want, rtol=1e-6, atol=1e-7)
    inv = oracle_py.M4()
    assert oracle_lib.oracle_matrix_inverse(C.byref(P), C.byref(inv))
    assert np.allclose(inv.to_numpy().astype(np.float64), np.linalg.inv(p.astype(np.float64)), rtol=1e-6, atol=1e-6)
    # primary-ray target of SURVEY A2: (d.x / w, -d.y / h, -1)
    d = np.array([0.3, -0.7, 1.0, 1.0]); d[1] = -d[1]
    t = d @ inv.to_numpy().astype(np.float64)
    assert np.allclose(t[:3], [0.3 / (h / aspect), 0.7 / h, -1.0], rtol=1e-5)


def test_hsl_roundtrip_and_envmap(oracle_lib):
    rng = np.random.default_rng(3)
    for rgb in rng.random((200, 3)).astype(np.float32):
        a = (C.c_float * 3)(*rgb); hsl = (C.c_float * 3)(); b = (C.c_float * 3)()
        oracle_lib.oracle_rgb_to_hsl(a, hsl)
        oracle_lib.oracle_hsl_to_rgb(hsl, b)
        assert np.allclose(list(b), rgb, atol=2e-5)
    uv = (C.c_float * 2)()
    oracle_lib.oracle_fake_envmap_uv((C.c_float * 3)(0.0, 0.0, -1.0), 0.0, uv)      # looking down -z: yaw = pi, pitch = pi
    assert abs(uv[0] - 0.5) < 1e-6 and abs(uv[1] - 0.5) < 1e-6
    oracle_lib.oracle_fake_envmap_uv((C.c_float * 3)(0.0, 1.0, 0.0), 0.0, uv)       # straight up: pitch = -pi/2 + pi
    assert abs(uv[1] - 0.25) < 1e-6


def test_shader_constants_of_oracle_and_kernels_are_the_references():
    """The handful of constants the reference's shaders are written around (Constants.hlsli, Ray.hlsli, Lights.hlsli, GlobalHitBuffers.hlsli, BgSky.hlsli,
    GaussianFilterRGB3x3CS.hlsl), read out of the reference's own files where they are at hand (the build container) and out of the oracle's and the kernels' headers:
    the same numbers on all three sides.  (A value, not a behaviour -- but every one of them is a place where a restatement can silently drift.)"""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def macros(path):
        out = {}
        for m in re.finditer(r"^[ \t]*#[ \t]*define[ \t]+(\w+)[ \t]+\(?(-?[0-9][0-9.eE+-]*)f?\)?", open(path).read(), re.M):
            out[m.group(1)] = float(m.group(2))
        return out
    want = {"EPSILON": 1e-6, "M_PI": 3.14159265, "APPLY_LIGHTS_MINIMUM_ALPHA": 0.5, "MAX_HIT_QUERIES": 16, "MAX_LIGHTS": 16, "RAY_MIN_DISTANCE": 0.1, "RAY_MAX_DISTANCE": 100000.0,
            "SCREEN_WIDTH": 320, "SCREEN_HEIGHT": 240}
    shaders = "/root/reference/src/rt64lib/shaders"
    if os.path.isdir(shaders):
        ref = {}
        for f in ("Constants.hlsli", "Ray.hlsli", "Lights.hlsli", "GlobalHitBuffers.hlsli", "BgSky.hlsli"):
            ref.update(macros(os.path.join(shaders, f)))
        for k, v in want.items():
            assert ref[k] == v, (k, ref[k])
        # the 3 x 3 Gaussian's weights (GaussianFilterRGB3x3CS.hlsl) as both sides spell them
        g = open(os.path.join(shaders, "GaussianFilterRGB3x3CS.hlsl")).read()
        for lit in ("0.077847", "0.123317", "0.195346"):
            assert lit in g and lit in open(os.path.join(root, "sm64rt-legacy-renderer_amd", "csrc", "passes.hip")).read() and lit in open(os.path.join(root, "oracle", "oracle_render.c")).read(), lit
    om = macros(os.path.join(root, "oracle", "oracle_math.h")); om.update(macros(os.path.join(root, "oracle", "oracle_internal.h")))
    dm = macros(os.path.join(root, "sm64rt-legacy-renderer_amd", "csrc", "device_math.h")); dm.update(macros(os.path.join(root, "sm64rt-legacy-renderer_amd", "csrc", "rt64_gpu.h")))
    pairs = [("EPSILON", "O_EPSILON", "RT_EPSILON"), ("M_PI", "O_PI", "RT_PI"), ("APPLY_LIGHTS_MINIMUM_ALPHA", "O_APPLY_LIGHTS_MINIMUM_ALPHA", "RT_APPLY_LIGHTS_MINIMUM_ALPHA"),
             ("MAX_HIT_QUERIES", "O_MAX_HIT_QUERIES", "RT64_MAX_HIT_QUERIES"), ("MAX_LIGHTS", "O_MAX_LIGHTS", "RT64_MAX_LIGHTS"),
             ("RAY_MIN_DISTANCE", "O_RAY_MIN_DISTANCE", "RT_RAY_MIN_DISTANCE"), ("RAY_MAX_DISTANCE", "O_RAY_MAX_DISTANCE", "RT_RAY_MAX_DISTANCE")]
    for r, o, d in pairs:
        assert om[o] == want[r] and dm[d] == want[r], (r, om.get(o), dm.get(d))
