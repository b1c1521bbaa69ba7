"""Pins the oracle's rasteriser (oracle/oracle_raster.c, raster spec S0-S8) with checks that do not share its code:
coverage against exact rational arithmetic on the snapped vertices, the fill rule on shared edges (every pixel of two
triangles that share an edge is drawn exactly once), blending against a closed form, scissor / viewport placement."""
import copy
from fractions import Fraction

import numpy as np
import pytest


def _scene(sample_data, tris, alpha=1.0, flags=0, scissor=None, viewport=None, keep_stock=False, color=None):
    """A raster-only scene: clip-space triangles with per-vertex colour (input1) and no texture contribution issues (tiles texture stays)."""
    from sm64rt_legacy_renderer_amd import sample_scene
    d = copy.copy(sample_data)
    d.meshes = list(sample_data.meshes)
    v = np.zeros(3 * len(tris), dtype=sample_scene.VERTEX_DTYPE)
    k = 0
    for tri in tris:
        for (x, y) in tri:
            v["position"][k] = (x, y, 0.0, 1.0); v["normal"][k] = (0, 1, 0); v["uv"][k] = (0.5, 0.5)
            v["input1"][k] = (1.0, 1.0, 1.0, alpha) if color is None else (*color, alpha)
            k += 1
    d.meshes.append(sample_scene.MeshData("t", 0, v, np.arange(len(v), dtype=np.uint32)))
    base = sample_data.instances[0]
    inst = copy.copy(base); inst.mesh = len(d.meshes) - 1; inst.material = sample_scene.copy_material(base.material); inst.flags = flags
    inst.scissor = scissor; inst.viewport = viewport
    d.instances = ([copy.copy(i) for i in sample_data.instances if i.name.startswith("hud")] if keep_stock else []) + [inst]
    return d


def _exact_coverage(tri, w, h, vp=None):
    """Coverage by the D3D11 rules with exact integers: vertices snapped to 1/256 pixel exactly as spec S1/S2 says (fp32 operations
    reproduced with numpy float32), then rational edge tests at pixel centres with the top-left rule."""
    f32 = np.float32
    vx, vy, vw, vh = (f32(0), f32(0), f32(w), f32(h)) if vp is None else map(f32, vp)
    pts = []
    for (x, y) in tri:
        rw = f32(1.0) / f32(1.0)
        xs = ((f32(x) * rw) * f32(0.5) + f32(0.5)) * vw + vx
        ys = (f32(0.5) - (f32(y) * rw) * f32(0.5)) * vh + vy
        pts.append((int(np.rint(xs * f32(256.0))), int(np.rint(ys * f32(256.0)))))
    (x0, y0), (x1, y1), (x2, y2) = pts
    area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0)
    cov = np.zeros((h, w), dtype=bool)
    if area == 0:
        return cov
    if area < 0:
        (x1, y1), (x2, y2) = (x2, y2), (x1, y1)
    P = [(x0, y0), (x1, y1), (x2, y2)]
    for py in range(h):
        for px in range(w):
            cx, cy = Fraction(px * 256 + 128), Fraction(py * 256 + 128)
            ok = True
            for a in range(3):
                ax, ay = P[a]; bx, by = P[(a + 1) % 3]
                e = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
                if e < 0 or (e == 0 and not (by - ay < 0 or (by - ay == 0 and bx - ax > 0))):
                    ok = False
                    break
            cov[py, px] = ok
    return cov


W, H = 96, 64


def _render(sample_data, data, **kw):
    from oracle import oracle_py
    o = oracle_py.OracleScene(data)
    try:
        return o.render(W, H, **kw)
    finally:
        o.close()


@pytest.mark.parametrize("tri", [[(-0.8, -0.7), (0.7, -0.5), (-0.1, 0.9)], [(-1.0, -1.0), (1.0, -1.0), (-1.0, 1.0)], [(0.25, 0.25), (0.25, -0.5), (-0.5, 0.25)],
                                 [(-0.3333, 0.1), (0.9, 0.1), (0.2, 0.1001)]])
def test_coverage_matches_exact_rational_fill_rule(sample_data, oracle_lib, tri):
    ref = _render(sample_data, _scene(sample_data, [tri]))
    drawn = ref["final"][..., :3].max(axis=2) > 0                 # cleared buffer is black, the triangle is white x texture (texel > 0)
    assert np.array_equal(drawn, _exact_coverage(tri, W, H))


def test_shared_edges_are_drawn_exactly_once(sample_data, oracle_lib):
    """Two triangles of a quad (both windings) with alpha 0.5: a pixel on the shared diagonal blended twice would be brighter."""
    quad = [[(-0.75, -0.5), (0.5, -0.5), (0.5, 0.625)], [(-0.75, -0.5), (-0.75, 0.625), (0.5, 0.625)]]
    ref = _render(sample_data, _scene(sample_data, quad, alpha=0.5))
    a = ref["final"][..., 1].astype(np.int32)
    inside = _exact_coverage(quad[0], W, H) | _exact_coverage(quad[1], W, H)
    assert not (_exact_coverage(quad[0], W, H) & _exact_coverage(quad[1], W, H)).any()
    vals = np.unique(a[inside])
    assert len(vals) == 1 and (a[~inside] == 0).all(), vals         # one blend everywhere inside, nothing outside


def test_blend_closed_form_and_order(sample_data, oracle_lib):
    """SRC_ALPHA / INV_SRC_ALPHA with the target quantised after every layer.  The sample shader 0x01200a00 takes its colour from
    TEXEL0 and its alpha from INPUT1: layer 1 has alpha 0.5 over the cleared buffer, layer 2 alpha 0.25 on top."""
    full = [(-1.0, -1.0), (3.0, -1.0), (-1.0, 3.0)]
    from sm64rt_legacy_renderer_amd import sample_scene
    d = _scene(sample_data, [full], alpha=0.5, color=(1.0, 0.0, 0.0))
    d2 = _scene(sample_data, [full], alpha=0.25, color=(0.0, 1.0, 0.0))
    d.meshes.append(d2.meshes[-1]); top = copy.copy(d2.instances[-1]); top.mesh = len(d.meshes) - 1; d.instances.append(top)
    ref = _render(sample_data, d)
    px = ref["final"][H // 2, W // 2].astype(np.int32)
    tex = None
    from oracle import oracle_py
    # texel of tiles_dif at uv (0.5, 0.5): sample through the oracle's own sampler is not independent; read the mip-0 texel average instead
    t = sample_data.textures[sample_data.instances[0].diffuse].data.astype(np.float64) / 255.0
    hh, ww = t.shape[:2]
    tex = (t[hh // 2 - 1:hh // 2 + 1, ww // 2 - 1:ww // 2 + 1].mean(axis=(0, 1)))       # bilinear at the exact centre = mean of the 4 texels
    def q(x): return np.floor(np.clip(x, 0, 1) * 255 + 0.5)
    l1 = q(np.array([tex[0] * 0.5, tex[1] * 0.5, tex[2] * 0.5, 0.5 + 1.0 * 0.5]))       # over (0, 0, 0, 1): rgb = src*a, a = a + 1*(1-a)
    l2 = q(np.array([tex[0] * 0.25 + l1[0] / 255 * 0.75, tex[1] * 0.25 + l1[1] / 255 * 0.75, tex[2] * 0.25 + l1[2] / 255 * 0.75, 0.25 + l1[3] / 255 * 0.75]))
    assert np.abs(px - l2).max() <= 1, (px, l2)


def test_scissor_and_viewport_rectangles(sample_data, oracle_lib):
    """RT64_RECT has its origin at the bottom-left (rt64_view.cpp:1114-1136): viewport (8, 4, 64, 32) maps NDC to x 8..72, y (H-4-32)..(H-4)."""
    full = [(-1.0, -1.0), (3.0, -1.0), (-1.0, 3.0)]
    ref = _render(sample_data, _scene(sample_data, [full], viewport=(8, 4, 64, 32), scissor=(16, 8, 24, 12)))
    drawn = ref["final"][..., :3].max(axis=2) > 0
    ys, xs = np.nonzero(drawn)
    assert (xs.min(), xs.max() + 1, ys.min(), ys.max() + 1) == (16, 40, H - 8 - 12, H - 8)
    ref = _render(sample_data, _scene(sample_data, [[(-1.0, -1.0), (1.0, -1.0), (-1.0, 1.0)]], viewport=(8, 4, 64, 32)))
    assert np.array_equal(ref["final"][..., :3].max(axis=2) > 0, _exact_coverage([(-1.0, -1.0), (1.0, -1.0), (-1.0, 1.0)], W, H, vp=(8, H - 4 - 32, 64, 32)))


def _scene4(sample_data, tris4, alpha=1.0):
    """Raster-only scene from clip-space triangles with full (x, y, z, w) positions and a colour per vertex."""
    from sm64rt_legacy_renderer_amd import sample_scene
    d = copy.copy(sample_data)
    d.meshes = list(sample_data.meshes)
    v = np.zeros(3 * len(tris4), dtype=sample_scene.VERTEX_DTYPE)
    cols = ((1.0, 0.1, 0.1), (0.1, 1.0, 0.1), (0.1, 0.1, 1.0))
    k = 0
    for tri in tris4:
        for j, p in enumerate(tri):
            v["position"][k] = p; v["normal"][k] = (0, 1, 0); v["uv"][k] = (0.5, 0.5); v["input1"][k] = (*cols[j], alpha)
            k += 1
    d.meshes.append(sample_scene.MeshData("t", 0, v, np.arange(len(v), dtype=np.uint32)))
    base = sample_data.instances[0]
    inst = copy.copy(base); inst.mesh = len(d.meshes) - 1; inst.material = sample_scene.copy_material(base.material); inst.flags = 0
    d.instances = [inst]
    return d


def _homogeneous_coverage(tri4, w, h):
    """What a triangle with clip-space corners (x, y, z, w) covers, WITHOUT clipping: 2-D homogeneous rasterisation in float64.
    Pixel (px, py) sees the point with barycentrics l = M^-1 (xn, yn, 1), M columns = (x, y, w) of the corners; it is inside the
    triangle in front of the eye iff every l_i >= 0 (then w = 1 > 0 by construction), and inside the depth range iff 0 <= sum l_i z_i <= 1.
    Returns (inside mask, distance-to-boundary proxy = min |l_i| scaled, perspective-correct barycentrics)."""
    P = np.array(tri4, dtype=np.float64)
    M = np.stack([P[:, 0], P[:, 1], P[:, 3]], axis=0)           # rows x, y, w ; columns corners
    Mi = np.linalg.inv(M)
    ys, xs = np.mgrid[0:h, 0:w]
    xn = (xs + 0.5) / w * 2.0 - 1.0; yn = 1.0 - (ys + 0.5) / h * 2.0
    l = np.einsum("ij,jhw->ihw", Mi, np.stack([xn, yn, np.ones_like(xn)], axis=0))
    z = (l * P[:, 2][:, None, None]).sum(axis=0)
    inside = (l >= 0).all(axis=0) & (z >= 0) & (z <= 1)
    margin = np.minimum(np.abs(l).min(axis=0) / np.abs(l).sum(axis=0).clip(1e-30), np.minimum(np.abs(z), np.abs(1 - z)))
    bary = l / l.sum(axis=0, keepdims=True).clip(1e-30)
    return inside, margin, bary


@pytest.mark.parametrize("tri4", [
    [(-0.8, -0.6, 0.2, 1.0), (0.9, -0.7, 0.2, 1.0), (0.1, 0.4, 0.5, -0.5)],          # one corner behind the eye (w < 0)
    [(-0.5, -0.9, 0.1, 0.6), (0.4, 0.8, 0.3, -0.2), (0.9, -0.2, 0.3, -0.4)],          # two corners behind the eye
    [(-0.9, -0.8, -0.4, 1.0), (0.8, -0.6, 0.5, 1.0), (0.0, 0.9, 1.6, 1.0)],           # crosses the near (z = 0) and the far (z = w) plane
    [(-30.0, -0.5, 0.5, 1.0), (0.9, -0.6, 0.5, 1.0), (0.2, 25.0, 0.5, 1.0)],          # far outside the guard band in x and in y
])
def test_homogeneous_clipping_matches_unclipped_homogeneous_rasterisation(sample_data, oracle_lib, tri4):
    """Raster spec S0: triangles that leave the w > 0 half space, the depth range or the guard band are clipped, not skipped.  Checked
    against a rasterisation that never clips (2-D homogeneous coordinates, float64): same pixels except within a hair of an edge,
    and the same perspective-correct vertex colours."""
    d = _scene4(sample_data, [tri4])
    d.shader_id = 0x01200a00                                       # colour = TEXEL0, alpha = INPUT1.a: coverage only
    ref = _render(sample_data, d)
    drawn = ref["final"][..., :3].max(axis=2) > 0
    inside, margin, bary = _homogeneous_coverage(tri4, W, H)
    clear = margin > 0.02
    assert inside.sum() > 200 and np.array_equal(drawn[clear], inside[clear])
    assert (drawn != inside).mean() < 0.03
    # colours: colour slot d = INPUT_1, alpha slot d = TEXEL0 (alpha 255; it also keeps the uv in the vertex layout), opt_alpha:
    # the pixel is the interpolated vertex colour itself
    d.shader_id = (1 << 9) | (5 << 21) | (1 << 24)
    ref = _render(sample_data, d)
    cols = np.array([(1.0, 0.1, 0.1), (0.1, 1.0, 0.1), (0.1, 0.1, 1.0)])
    want = np.einsum("ihw,ic->hwc", bary, cols)
    sel = inside & (margin > 0.05)
    got = ref["final"][..., :3].astype(np.float64) / 255.0
    assert sel.sum() > 50 and np.abs(got[sel] - want[sel]).max() < 0.02
