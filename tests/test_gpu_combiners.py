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
