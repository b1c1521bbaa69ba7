"""Colour-combiner coverage: shader ids beyond the sample's 0x01200a00 -- every formula shape (single, multiply, mix,
general) for colour and alpha, 1-4 vertex inputs with and without alpha, TEXEL0A, texture edge, noise --, every sampler mode, and
randomised material attributes.  The vertex buffers are laid out as the shader's VertexLayout dictates (rt64_shader.cpp:80-96).
GPU frame vs oracle frame: hit records bit-exact, composed output within tolerance."""
import copy

import numpy as np
import pytest

from test_gpu_features import _render_pair, _rmse

pytestmark = pytest.mark.gpu

S0, IN1, IN2, IN3, IN4, TEX0, TEX0A, TEX1 = range(8)
OPT_ALPHA, OPT_FOG, OPT_EDGE, OPT_NOISE = 1 << 24, 1 << 25, 1 << 26, 1 << 27


def cc(color, alpha=None, opts=0):
    alpha = alpha or color
    v = 0
    for i, c in enumerate(color):
        v |= c << (3 * i)
    for i, c in enumerate(alpha):
        v |= c << (12 + 3 * i)
    return v | opts


def relayout(mesh, shader_id, rng):
    """Rebuild a sample mesh (position, normal, uv, input1) for the vertex layout of `shader_id`, with random colours in inputs 2..4."""
    from sm64rt_legacy_renderer_amd import sample_scene
    items = [(shader_id >> (3 * i)) & 7 for i in range(8)]
    n_inputs = max([c for c in items if 1 <= c <= 4] + [0])
    uses_tex = any(c in (5, 6, 7) for c in items)
    alpha = bool(shader_id & OPT_ALPHA)
    fields = [("position", "<f4", 4), ("normal", "<f4", 3)] + ([("uv", "<f4", 2)] if uses_tex else []) + [("input%d" % (k + 1), "<f4", 4 if alpha else 3) for k in range(n_inputs)]
    out = np.zeros(len(mesh.vertices), dtype=np.dtype(fields))
    out["position"] = mesh.vertices["position"]; out["normal"] = mesh.vertices["normal"]
    if uses_tex:
        out["uv"] = mesh.vertices["uv"]
    for k in range(n_inputs):
        col = rng.random((len(out), 4 if alpha else 3)).astype(np.float32)
        if alpha:
            col[:, 3] = 0.35 + 0.65 * col[:, 3]
        out["input%d" % (k + 1)] = col
    return sample_scene.MeshData(mesh.name, mesh.flags, out, mesh.indices)


CASES = [
    ("single texel, no alpha", cc((S0, S0, S0, TEX0)), {}),
    ("texel x input, alpha from input", cc((TEX0, S0, IN1, S0), (S0, S0, S0, IN1), OPT_ALPHA), dict(filter=0)),
    ("mix of two inputs by texel alpha", cc((IN1, IN2, TEX0A, IN2), (IN1, IN2, TEX0, IN2), OPT_ALPHA), dict(haddr=1, vaddr=2)),
    ("general (a - b) * c + d, four inputs", cc((IN1, IN2, IN3, IN4), (IN4, IN3, IN2, IN1), OPT_ALPHA), dict(haddr=2, vaddr=1)),
    ("general with texel, inputs without alpha", cc((TEX0, IN1, IN2, IN1)), dict(filter=0, haddr=1, vaddr=1)),
    ("texture edge", cc((TEX0, S0, IN1, S0), (TEX0, S0, IN1, S0), OPT_ALPHA | OPT_EDGE), {}),
    ("noise", cc((IN1, S0, IN2, S0), (S0, S0, S0, IN1), OPT_ALPHA | OPT_NOISE), {}),
    ("texel1 placeholder + fog bit", cc((TEX1, S0, TEX0, S0), (S0, S0, S0, TEX0), OPT_ALPHA | OPT_FOG), dict(haddr=2, vaddr=2)),
]


@pytest.mark.parametrize("name,shader_id,sampler", CASES, ids=[c[0] for c in CASES])
def test_combiner_and_sampler_variants(rt64_lib, sample_data, name, shader_id, sampler):
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    rng = np.random.default_rng(shader_id & 0xFFFF)
    d = copy.copy(sample_data)
    d.shader_id = shader_id
    d.shader_filter = sampler.get("filter", 1); d.shader_haddr = sampler.get("haddr", 0); d.shader_vaddr = sampler.get("vaddr", 0)
    d.meshes = [relayout(m, shader_id, rng) for m in sample_data.meshes]
    d.instances = [copy.copy(i) for i in sample_data.instances]
    for i in d.instances:
        m = sample_scene.copy_material(i.material)
        m.diffuseColorMix = rt64.VECTOR4(float(rng.random()), float(rng.random()), float(rng.random()), float(0.4 * rng.random()))
        m.selfLight = rt64.VECTOR3(float(0.2 * rng.random()), float(0.1 * rng.random()), 0.0)
        m.specularExponent = float(1.0 + 30.0 * rng.random()); m.uvDetailScale = float(0.5 + 3.0 * rng.random())
        m.ignoreNormalFactor = float(rng.random() * 0.5); m.solidAlphaMultiplier = float(0.6 + 0.4 * rng.random())
        m.shadowAlphaMultiplier = float(0.5 + 0.5 * rng.random())
        if shader_id & OPT_FOG:
            m.fogEnabled = 1; m.fogMul = 0.02; m.fogOffset = -0.05; m.fogColor = rt64.VECTOR3(0.6, 0.7, 0.9)
        i.material = m
    got, ref, st = _render_pair(rt64_lib, d, frames=2)
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"]) and np.array_equal(got["INSTANCE_ID"], ref["instanceId"])
    assert st.primaryRays == ref["counters"]["primaryRays"] and st.shadowRays == ref["counters"]["shadowRays"]
    assert (st.nodesVisited, st.trianglesTested) == (ref["counters"]["nodesVisited"], ref["counters"]["trianglesTested"])
    dd = np.abs(got["DIFFUSE"] - ref["diffuse"]).max(axis=2)
    assert (dd > 1.5 / 255.0).mean() < 2e-3, float((dd > 1.5 / 255.0).mean())          # texel-boundary flips under point filtering are isolated pixels
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 2e-3
    fd = np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32))
    assert (fd > 1).mean() < 3e-3


def _random_materials(d, rng, fog=False):
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    d.instances = [copy.copy(i) for i in d.instances]
    for i in d.instances:
        m = sample_scene.copy_material(i.material)
        m.diffuseColorMix = rt64.VECTOR4(float(rng.random()), float(rng.random()), float(rng.random()), float(0.4 * rng.random()))
        m.selfLight = rt64.VECTOR3(float(0.2 * rng.random()), float(0.1 * rng.random()), 0.0)
        m.specularExponent = float(1.0 + 30.0 * rng.random()); m.uvDetailScale = float(0.5 + 3.0 * rng.random())
        m.ignoreNormalFactor = float(rng.random() * 0.5); m.solidAlphaMultiplier = float(0.6 + 0.4 * rng.random())
        m.shadowAlphaMultiplier = float(0.5 + 0.5 * rng.random())
        if fog:
            m.fogEnabled = 1; m.fogMul = 0.02; m.fogOffset = -0.05; m.fogColor = rt64.VECTOR3(0.6, 0.7, 0.9)
        i.material = m


def _check_variant(got, ref, st):
    assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"]) and np.array_equal(got["INSTANCE_ID"], ref["instanceId"])
    assert st.primaryRays == ref["counters"]["primaryRays"] and st.shadowRays == ref["counters"]["shadowRays"]
    assert (st.nodesVisited, st.trianglesTested) == (ref["counters"]["nodesVisited"], ref["counters"]["trianglesTested"])
    dd = np.abs(got["DIFFUSE"] - ref["diffuse"]).max(axis=2)
    assert (dd > 1.5 / 255.0).mean() < 2e-3, float((dd > 1.5 / 255.0).mean())          # texel-boundary flips under point filtering are isolated pixels
    assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 2e-3
    fd = np.abs(got["FINAL_RGBA8"].astype(np.int32) - ref["final"].astype(np.int32))
    assert (fd > 1).mean() < 3e-3


def _resized(tex, w, h, name):
    """Nearest-neighbour resample of an RGBA8 sample texture to w x h (a non-power-of-two size for the general addressing path)."""
    from sm64rt_legacy_renderer_amd import sample_scene
    ys = (np.arange(h) * tex.height // h)[:, None]; xs = (np.arange(w) * tex.width // w)[None, :]
    return sample_scene.TextureData(name, tex.format, np.ascontiguousarray(tex.data[ys, xs]), w, h)


SAMPLERS = [(f, h, v) for f in (0, 1) for h in (0, 1, 2) for v in (0, 1, 2)]


@pytest.mark.parametrize("npot", [0, 1], ids=["pow2", "npot"])
@pytest.mark.parametrize("filt,haddr,vaddr", SAMPLERS, ids=["%s-%s-%s" % ("PL"[f], "WMC"[h], "WMC"[v]) for f, h, v in SAMPLERS])
def test_every_sampler_variant(rt64_lib, sample_data, filt, haddr, vaddr, npot):
    """All 18 filter x hAddr x vAddr sampler variants of rt64_view.cpp:711-721 (point / linear x wrap / mirror / clamp per axis), on
    power-of-two textures (the mask addressing of the `simple` kernels) and with non-power-of-two diffuse / normal / specular textures on
    the floor (the general addressing path); floor uvs run over [-1.3, 2.4] so that every mode wraps, mirrors or clamps inside the picture
    (the sphere's acos uvs reach pi).  The shader mixes texel colour and texel alpha with a vertex input."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    shader_id = cc((TEX0, S0, IN1, S0), (TEX0, S0, IN1, S0), OPT_ALPHA)
    rng = np.random.default_rng(100 + 9 * filt + 3 * haddr + vaddr)
    d = copy.copy(sample_data)
    d.shader_id = shader_id; d.shader_filter = filt; d.shader_haddr = haddr; d.shader_vaddr = vaddr
    d.meshes = [relayout(m, shader_id, rng) for m in sample_data.meshes]
    floor = copy.copy(d.meshes[3]); v = floor.vertices.copy()
    v["uv"] = v["uv"] * np.float32(3.7) - np.float32(1.3)
    floor.vertices = v; d.meshes[3] = floor
    if npot:
        d.textures = list(sample_data.textures)
        for k, (w, h) in ((4, (100, 60)), (5, (37, 91)), (6, (130, 77))):          # tiles_dif / tiles_nrm / tiles_spc
            d.textures[k] = _resized(sample_data.textures[k], w, h, sample_data.textures[k].name + "_npot")
    _random_materials(d, rng)
    got, ref, st = _render_pair(rt64_lib, d, frames=2)
    _check_variant(got, ref, st)
    assert (ref["primaryHit"][..., 3] != 0xFFFFFFFF).mean() > 0.4


def _random_shader_id(rng):
    """A valid colour-combiner id (rt64_shader.cpp:32-78): eight 3-bit items (0, INPUT 1-4, TEXEL0, TEXEL0A, TEXEL1) + the option bits."""
    items = rng.integers(0, 8, size=8)
    shape = rng.integers(0, 4)
    for base in (0, 4):                     # steer some ids onto the special-cased formula shapes (do_single / do_multiply / do_mix)
        if shape == 1:
            items[base + 2] = 0
        elif shape == 2:
            items[base + 1] = 0; items[base + 3] = 0
        elif shape == 3:
            items[base + 3] = items[base + 1]
    if rng.random() < 0.25:
        items[4:] = items[:4]               # color_alpha_same
    v = 0
    for i, c in enumerate(items):
        v |= int(c) << (3 * i)
    if rng.random() < 0.7:
        v |= OPT_ALPHA
    if rng.random() < 0.2:
        v |= OPT_FOG
    if rng.random() < 0.15:
        v |= OPT_EDGE
    if rng.random() < 0.15:
        v |= OPT_NOISE
    return v


@pytest.mark.parametrize("k", range(64))
def test_random_shader_ids(rt64_lib, sample_data, k):
    """A seeded sweep of 64 random valid shader ids, each with the vertex layout its id dictates (1-4 inputs with / without alpha, with /
    without uv: rt64_shader.cpp:80-96), a random sampler and random material attributes -- the game emits dozens of combiner ids."""
    rng = np.random.default_rng(7000 + k)
    shader_id = _random_shader_id(rng)
    d = copy.copy(sample_data)
    d.shader_id = shader_id
    d.shader_filter = int(rng.integers(0, 2)); d.shader_haddr = int(rng.integers(0, 3)); d.shader_vaddr = int(rng.integers(0, 3))
    d.meshes = [relayout(m, shader_id, rng) for m in sample_data.meshes]
    _random_materials(d, rng, fog=bool(shader_id & OPT_FOG))
    got, ref, st = _render_pair(rt64_lib, d, frames=2)
    _check_variant(got, ref, st)


def test_rgba8_texture_with_padded_row_pitch(rt64_lib, sample_data):
    """RT64_TEXTURE_DESC.rowPitch > 4 * width (rt64_texture.cpp:28-140 copies row by row): the floor's three RGBA8 textures are handed over
    with 64 bytes of padding per row (filled with a value that would show), one of them non-power-of-two; same picture as the oracle's, and the
    same bytes as the tightly packed upload."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    finals = {}
    for pad in (64, 0):
        d = copy.copy(sample_data)
        d.textures = list(sample_data.textures)
        d.textures[4] = _resized(sample_data.textures[4], 100, 60, "tiles_dif_npot")
        for k in (4, 5, 6):
            t = copy.copy(d.textures[k]); t.row_pitch = 4 * t.width + pad; d.textures[k] = t
        got, ref, st = _render_pair(rt64_lib, d, frames=1)
        assert np.array_equal(got["PRIMARY_HIT"], ref["primaryHit"])
        assert _rmse(got["OUTPUT_RGBA32F"][..., :3], ref["output"][..., :3]) <= 1e-3
        assert np.abs(got["DIFFUSE"] - ref["diffuse"]).max() <= 1.0 / 255.0 + 1e-6
        finals[pad] = got["FINAL_RGBA8"]
    assert np.array_equal(finals[64], finals[0])
