"""BC7 decoder and software sampler of the oracle, pinned against an independent decoder (Pillow) and hand-computed texels."""
import ctypes as C
import hashlib
import io
import json
import os
import struct

import numpy as np
import pytest

from oracle import oracle_py

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _decode(oracle_lib, blocks):
    out = np.zeros((len(blocks), 64), dtype=np.uint8)
    for k, b in enumerate(blocks):
        buf = (C.c_uint8 * 16)(*b)
        o = (C.c_uint8 * 64)()
        oracle_lib.oracle_decode_bc7_block(buf, o)
        out[k] = np.frombuffer(bytes(o), dtype=np.uint8)
    return out.reshape(len(blocks), 4, 4, 4)


def _pillow_decode(blocks):
    from PIL import Image
    n = len(blocks)
    raw = b"".join(bytes(b) for b in blocks)
    hdr = struct.pack("<4s7I44xII4s5I4I4x", b"DDS ", 124, 0x1007 | 0x80000, 4, 4 * n, len(raw), 0, 1, 32, 4, b"DX10", 0, 0, 0, 0, 0, 0x1000, 0, 0, 0)
    img = Image.open(io.BytesIO(hdr + struct.pack("<5I", 98, 3, 0, 1, 0) + raw)); img.load()
    a = np.asarray(img.convert("RGBA"), dtype=np.uint8)
    return a.reshape(4, n, 4, 4).transpose(1, 0, 2, 3)


def test_bc7_random_blocks_all_modes_vs_pillow(oracle_lib):
    """Random 128-bit blocks hit every mode, partition, rotation and index-selection combination."""
    pytest.importorskip("PIL")
    rng = np.random.default_rng(7)
    blocks = []
    for mode in range(8):
        for _ in range(400):
            b = rng.integers(0, 256, size=16, dtype=np.uint8)
            v = int.from_bytes(b.tobytes(), "little")
            v = (v >> (mode + 1) << (mode + 1)) | (1 << mode)         # force the mode bits
            blocks.append(np.frombuffer(v.to_bytes(16, "little"), dtype=np.uint8))
    got = _decode(oracle_lib, blocks)
    want = _pillow_decode(blocks)
    assert np.array_equal(got, want)


def test_bc7_sample_texture_hash_and_pillow(oracle_lib, sample_data):
    kats = json.load(open(os.path.join(GOLD, "kats.json")))
    raw = sample_data.textures[0].data
    t = oracle_lib.oracle_texture_create_dds(raw.ctypes.data, raw.nbytes)
    assert t
    w, h, m = C.c_int(), C.c_int(), C.c_int()
    oracle_lib.oracle_texture_info(t, C.byref(w), C.byref(h), C.byref(m))
    assert (w.value, h.value, m.value) == (512, 512, 10) and kats["bc7_mips"] == 10
    sha = hashlib.sha256()
    mips = []
    for mip in range(m.value):
        mw, mh = C.c_int(), C.c_int()
        p = oracle_lib.oracle_texture_mip(t, mip, C.byref(mw), C.byref(mh))
        a = np.ctypeslib.as_array(p, shape=(mh.value, mw.value, 4)).copy()
        mips.append(a); sha.update(a.tobytes())
    assert sha.hexdigest() == kats["bc7_grass_dif_sha256"]
    try:
        from PIL import Image
        img = Image.open(io.BytesIO(raw.tobytes())); img.load()
        assert np.array_equal(mips[0], np.asarray(img.convert("RGBA"), dtype=np.uint8))
    except ImportError:
        pass
    assert (mips[0][..., 3] == 255).all() and mips[9].shape == (1, 1, 4)
    oracle_lib.oracle_texture_destroy(t)


def test_sampler_addressing_and_bilinear(oracle_lib):
    tex = np.zeros((2, 2, 4), dtype=np.uint8)
    tex[0, 0] = (255, 0, 0, 255); tex[0, 1] = (0, 255, 0, 255); tex[1, 0] = (0, 0, 255, 255); tex[1, 1] = (255, 255, 255, 0)
    t = oracle_lib.oracle_texture_create_rgba8(tex.ctypes.data, 2, 2, 8)
    out = (C.c_float * 4)()

    def s(u, v, filt=1, ha=0, va=0):
        oracle_lib.oracle_texture_sample(t, u, v, 0, 0, 0, 0, filt, ha, va, out)
        return np.array(list(out), dtype=np.float32)
    assert np.allclose(s(0.25, 0.25), [1, 0, 0, 1])                       # texel centre
    assert np.allclose(s(0.5, 0.25), [0.5, 0.5, 0, 1])                    # halfway between texel 0 and 1
    assert np.allclose(s(0.5, 0.5), [0.5, 0.5, 0.5, 0.75])               # all four
    assert np.allclose(s(0.0, 0.25, ha=0), [0.5, 0.5, 0, 1])             # WRAP: left neighbour of texel 0 is texel 1
    assert np.allclose(s(0.0, 0.25, ha=2), [1, 0, 0, 1])                 # CLAMP
    assert np.allclose(s(0.0, 0.25, ha=1), [1, 0, 0, 1])                 # MIRROR: neighbour -1 mirrors to texel 0
    assert np.allclose(s(1.25, 0.25, ha=1), [0, 1, 0, 1])                # MIRROR: texel 2 mirrors to texel 1
    assert np.allclose(s(0.6, 0.1, filt=0), [0, 1, 0, 1])                # POINT
    assert np.allclose(s(-0.4, 0.1, filt=0, ha=0), [0, 1, 0, 1])         # POINT + WRAP of a negative coordinate
    oracle_lib.oracle_texture_destroy(t)


def test_sample_grad_mip_selection(oracle_lib, sample_data):
    raw = sample_data.textures[0].data
    t = oracle_lib.oracle_texture_create_dds(raw.ctypes.data, raw.nbytes)
    out = (C.c_float * 4)()

    def s(du):
        oracle_lib.oracle_texture_sample(t, 0.37, 0.61, du, 0.0, 0.0, du, 1, 0, 0, out)
        return np.array(list(out), dtype=np.float32)
    base = s(0.0)                       # zero gradients -> mip 0 (bounce rays, IndirectRayGen.hlsl:65-69)
    assert np.array_equal(base, s(1.0 / 512.0))          # one texel per pixel: lod = log2(1) = 0
    coarse = s(1.0)                     # whole texture per pixel: last mip (1x1): constant colour everywhere
    oracle_lib.oracle_texture_sample(t, 0.9, 0.1, 1.0, 0.0, 0.0, 1.0, 1, 0, 0, out)
    assert np.allclose(coarse, list(out))
    mid_a, mid_b = s(2.0 / 512.0), s(4.0 / 512.0)        # lod 1 and 2
    half = s(2.0 ** 1.5 / 512.0)                          # lod 1.5: average of the two levels
    assert np.allclose(half, 0.5 * (mid_a + mid_b), atol=1e-6)
    oracle_lib.oracle_texture_destroy(t)
