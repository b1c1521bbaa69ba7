/* oracle_raster.c -- raster (HUD / background) instance pass of the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h).
 *
 * Restates the raster path of the reference: instances whose mesh was not created with RT64_MESH_RAYTRACE_ENABLED are drawn
 * with a generated vertex + pixel shader (ref:private/rt64_shader.cpp:312-442): clip-space positions pass through, attributes are
 * interpolated, the pixel shader samples the diffuse texture and evaluates the colour combiner, and the result is alpha-blended
 * (SRC_ALPHA / INV_SRC_ALPHA, alpha ONE / INV_SRC_ALPHA, no depth test, no culling) into an RGBA8 target in instance order
 * (ref:private/rt64_view.cpp:1225-1254 drawInstances, :1292-1319 background pass + gBackground copy, :1657-1661 foreground pass).
 * The rasteriser itself is fixed-function hardware in the reference; this file follows the published Direct3D 11 rasterisation
 * rules.  Raster spec (every step is part of the contract with csrc/raster.hip; coverage is integer arithmetic => bit-exact):
 *   S0  homogeneous clipping (the fixed-function clipper in front of the rasteriser; D3D11 functional spec: 0 < w, 0 <= z <= w, x and y
 *       against a guard band).  A triangle whose three vertices satisfy all seven plane distances d >= 0 is drawn as it is (B = identity).
 *       Otherwise Sutherland-Hodgman over the planes in this order, d evaluated in fp32, unfused:
 *           W: w - 2^-20   NEAR: z   FAR: w - z   X-: x + 4w   X+: 4w - x   Y-: y + 4w   Y+: 4w - y
 *       For every polygon edge A -> B: A is kept iff dA >= 0; when exactly one end is inside, with P the inside and Q the outside end,
 *       t = dP / (dP - dQ) and the new vertex is P + t (Q - P) for x, y, z, w and for the three barycentric weights of the ORIGINAL
 *       triangle (each component: P.c + t * (Q.c - P.c)).  The polygon (<= 10 vertices) is fanned from its first vertex into
 *       sub-triangles (V0, Vj, Vj+1); each goes through S1-S8 with its own w.  An attribute at a sub-triangle corner k is
 *       (B[k][0] a0 + B[k][1] a1) + B[k][2] a2 of the original vertices' values (attributes are linear in clip space).
 *   S1  rw = 1/w ; xs = ((x*rw)*0.5 + 0.5)*vpW + vpX ; ys = (0.5 - (y*rw)*0.5)*vpH + vpY          (fp32, unfused)
 *   S2  X = lrintf(xs*256), Y = lrintf(ys*256) (24.8 fixed point, round-to-nearest-even); |X|,|Y| > 2^22 => triangle skipped
 *   S3  area2 = (X1-X0)(Y2-Y0) - (Y1-Y0)(X2-X0) (int64); 0 => skipped; < 0 => vertices 1 and 2 swapped (CullMode NONE)
 *   S4  edge E_ab(P) = (bx-ax)(Py-ay) - (by-ay)(Px-ax) at the pixel centre P = (256 px + 128, 256 py + 128);
 *       covered iff for all three edges E > 0 or (E == 0 and (dy < 0 or (dy == 0 and dx > 0)))   (top-left rule, y down)
 *   S5  scissor: left <= px < right, top <= py < bottom
 *   S6  l0 = E12/area2, l1 = E20/area2, l2 = E01/area2 (fp32 from int64); q_k = l_k*rw_k ; attribute = ((q0 a0 + q1 a1) + q2 a2) / ((q0 + q1) + q2)
 *   S7  texture gradients for mip selection: attribute differences to the pixel centres at +1 in x and in y on the same plane
 *   S8  blend per triangle with the target's RGBA8 storage: d = byte/255; src clamped to [0,1];
 *       rgb = src.rgb*src.a + d.rgb*(1 - src.a) ; a = src.a + d.a*(1 - src.a) ; stored with the UNORM8 conversion
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"
#include "oracle_shade.h"

typedef struct { int64_t X[3], Y[3]; float rw[3]; const uint8_t *vp[3]; int64_t area2; int clipped; float B[3][3]; } RasterTri;

#define RASTER_W_EPS 9.5367431640625e-07f       /* 2^-20 */
#define RASTER_GUARD 4.0f
#define RASTER_MAX_POLY 10
typedef struct { float c[7]; } ClipVertex;       /* x, y, z, w, b0, b1, b2 */

static float clip_distance(int plane, const ClipVertex *v) {
    const float x = v->c[0], y = v->c[1], z = v->c[2], w = v->c[3];
    switch (plane) {
    case 0: return w - RASTER_W_EPS;
    case 1: return z;
    case 2: return w - z;
    case 3: return x + RASTER_GUARD * w;
    case 4: return RASTER_GUARD * w - x;
    case 5: return y + RASTER_GUARD * w;
    default: return RASTER_GUARD * w - y;
    }
}

/* S0: clip the triangle; returns the polygon's vertex count (0 = nothing left), *clipped = 0 when the triangle was inside every plane */
static int clip_triangle(const float p[3][4], ClipVertex poly[RASTER_MAX_POLY], int *clipped) {
    int n = 3, all = 1;
    for (int k = 0; k < 3; k++) {
        for (int c = 0; c < 4; c++) poly[k].c[c] = p[k][c];
        for (int c = 0; c < 3; c++) poly[k].c[4 + c] = c == k ? 1.0f : 0.0f;
        for (int pl = 0; pl < 7; pl++) if (!(clip_distance(pl, &poly[k]) >= 0.0f)) all = 0;
    }
    *clipped = !all;
    if (all) return 3;
    for (int pl = 0; pl < 7 && n >= 3; pl++) {
        ClipVertex out[RASTER_MAX_POLY + 1]; int m = 0;
        for (int i = 0; i < n; i++) {
            const ClipVertex *A = &poly[i], *Bv = &poly[(i + 1) % n];
            const float dA = clip_distance(pl, A), dB = clip_distance(pl, Bv);
            const int inA = dA >= 0.0f, inB = dB >= 0.0f;
            if (inA && m < RASTER_MAX_POLY) out[m++] = *A;
            if (inA != inB && m < RASTER_MAX_POLY) {
                const ClipVertex *P = inA ? A : Bv, *Q = inA ? Bv : A;
                const float dP = inA ? dA : dB, dQ = inA ? dB : dA;
                const float t = dP / (dP - dQ);
                for (int c = 0; c < 7; c++) out[m].c[c] = P->c[c] + t * (Q->c[c] - P->c[c]);
                m++;
            }
        }
        n = m;
        for (int i = 0; i < n; i++) poly[i] = out[i];
    }
    return n >= 3 ? n : 0;
}

/* S1-S3 for one (sub-)triangle with clip-space corners v[3] (x, y, z, w, barycentrics) */
static int setup_triangle(const ClipVertex v[3], const uint8_t *const vp[3], int clipped, float vpX, float vpY, float vpW, float vpH, RasterTri *t) {
    t->clipped = clipped;
    for (int k = 0; k < 3; k++) {
        const float *p = v[k].c;
        if (!(p[3] > 0.0f)) return 0;
        for (int c = 0; c < 3; c++) t->B[k][c] = p[4 + c];
        float rw = 1.0f / p[3];
        float xs = ((p[0] * rw) * 0.5f + 0.5f) * vpW + vpX, ys = (0.5f - (p[1] * rw) * 0.5f) * vpH + vpY;   /* S1 */
        float xf = xs * 256.0f, yf = ys * 256.0f;
        if (!(fabsf(xf) <= 4194304.0f) || !(fabsf(yf) <= 4194304.0f)) return 0;              /* S2 */
        t->X[k] = (int64_t)lrintf(xf); t->Y[k] = (int64_t)lrintf(yf); t->rw[k] = rw; t->vp[k] = vp[k];
    }
    int64_t a = (t->X[1] - t->X[0]) * (t->Y[2] - t->Y[0]) - (t->Y[1] - t->Y[0]) * (t->X[2] - t->X[0]);      /* S3 */
    if (a == 0) return 0;
    if (a < 0) {
        int64_t tx = t->X[1]; t->X[1] = t->X[2]; t->X[2] = tx; int64_t ty = t->Y[1]; t->Y[1] = t->Y[2]; t->Y[2] = ty;
        float tr = t->rw[1]; t->rw[1] = t->rw[2]; t->rw[2] = tr;
        if (!clipped) { const uint8_t *tp = t->vp[1]; t->vp[1] = t->vp[2]; t->vp[2] = tp; }        /* clipped: the original vertices stay, the barycentric rows move */
        else for (int c = 0; c < 3; c++) { float tb = t->B[1][c]; t->B[1][c] = t->B[2][c]; t->B[2][c] = tb; }
        a = -a;
    }
    t->area2 = a;
    return 1;
}

static int64_t edge(const RasterTri *t, int a, int b, int64_t px, int64_t py) {
    return (t->X[b] - t->X[a]) * (py - t->Y[a]) - (t->Y[b] - t->Y[a]) * (px - t->X[a]);
}
static int edge_in(const RasterTri *t, int a, int b, int64_t e) {                            /* S4 */
    if (e > 0) return 1;
    if (e < 0) return 0;
    int64_t dx = t->X[b] - t->X[a], dy = t->Y[b] - t->Y[a];
    return dy < 0 || (dy == 0 && dx > 0);
}

/* perspective-correct weights at a fixed-point position (not necessarily covered) */
static void weights(const RasterTri *t, int64_t px, int64_t py, float q[3], float *qs) {     /* S6 */
    float area = (float)t->area2;
    float l0 = (float)edge(t, 1, 2, px, py) / area, l1 = (float)edge(t, 2, 0, px, py) / area, l2 = (float)edge(t, 0, 1, px, py) / area;
    q[0] = l0 * t->rw[0]; q[1] = l1 * t->rw[1]; q[2] = l2 * t->rw[2];
    *qs = (q[0] + q[1]) + q[2];
}
static float interp(const float q[3], float qs, float a0, float a1, float a2) { return ((q[0] * a0 + q[1] * a1) + q[2] * a2) / qs; }
/* attribute values at the corners of a (sub-)triangle from the original vertices' values a[3] (S0) */
static void corner_values(const RasterTri *t, const float a[3], float out[3]) {
    if (!t->clipped) { out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; return; }
    for (int k = 0; k < 3; k++) out[k] = (t->B[k][0] * a[0] + t->B[k][1] * a[1]) + t->B[k][2] * a[2];
}
static float interp_attr(const RasterTri *t, const float q[3], float qs, float a0, float a1, float a2) {
    const float a[3] = { a0, a1, a2 }; float c[3];
    corner_values(t, a, c);
    return interp(q, qs, c[0], c[1], c[2]);
}

/* Draw `count` instances (scene indices in `list`, draw order) into an RGBA8 target of w x h, rows [y0, y1). */
void oraster_draw(const OScene *s, const int *list, int count, uint8_t *target, int w, int h, int y0, int y1, int applyScissorsAndViewports) {
    for (int n = 0; n < count; n++) {
        const OInstanceDesc *d = &s->instances[list[n]];
        const OMesh *mesh = d->mesh;
        OCombiner cc; ocombiner_decode(d->shaderId, &cc);
        float vpX = 0.0f, vpY = 0.0f, vpW = (float)w, vpH = (float)h;
        int scL = 0, scT = 0, scR = w, scB = h;
        if (applyScissorsAndViewports) {                                                     /* ref:rt64_view.cpp:1114-1136 */
            if (d->scissorRect[2] > 0 && d->scissorRect[3] > 0) {
                scL = d->scissorRect[0]; scT = h - d->scissorRect[1] - d->scissorRect[3]; scR = d->scissorRect[0] + d->scissorRect[2]; scB = h - d->scissorRect[1];
            }
            if (d->viewportRect[2] > 0 && d->viewportRect[3] > 0) {
                vpX = (float)d->viewportRect[0]; vpY = (float)(h - d->viewportRect[1] - d->viewportRect[3]); vpW = (float)d->viewportRect[2]; vpH = (float)d->viewportRect[3];
            }
        }
        if (scL < 0) scL = 0;
        if (scT < y0) scT = y0;
        if (scR > w) scR = w;
        if (scB > y1) scB = y1;
        const int triCount = mesh->indexCount / 3;
        for (int tri = 0; tri < triCount; tri++) {
            float p[3][4]; const uint8_t *vp[3];
            for (int k = 0; k < 3; k++) { vp[k] = mesh->vertices + (size_t)mesh->indices[3 * tri + k] * (size_t)mesh->vertexStride; memcpy(p[k], vp[k], 16); }
            ClipVertex poly[RASTER_MAX_POLY]; int clipped = 0;
            const int nv = clip_triangle(p, poly, &clipped);                                  /* S0 */
            for (int sub = 1; sub + 1 < nv; sub++) {
            RasterTri t;
            const ClipVertex corners[3] = { poly[0], poly[sub], poly[sub + 1] };
            if (!setup_triangle(corners, vp, clipped, vpX, vpY, vpW, vpH, &t)) continue;
            int64_t minX = t.X[0], maxX = t.X[0], minY = t.Y[0], maxY = t.Y[0];
            for (int k = 1; k < 3; k++) { if (t.X[k] < minX) minX = t.X[k]; if (t.X[k] > maxX) maxX = t.X[k]; if (t.Y[k] < minY) minY = t.Y[k]; if (t.Y[k] > maxY) maxY = t.Y[k]; }
            int px0 = (int)(minX >> 8), px1 = (int)(maxX >> 8), py0 = (int)(minY >> 8), py1 = (int)(maxY >> 8);
            if (px0 < scL) px0 = scL;
            if (py0 < scT) py0 = scT;
            if (px1 > scR - 1) px1 = scR - 1;
            if (py1 > scB - 1) py1 = scB - 1;
            for (int py = py0; py <= py1; py++)
                for (int px = px0; px <= px1; px++) {
                    const int64_t cx = (int64_t)px * 256 + 128, cy = (int64_t)py * 256 + 128;
                    const int64_t e12 = edge(&t, 1, 2, cx, cy), e20 = edge(&t, 2, 0, cx, cy), e01 = edge(&t, 0, 1, cx, cy);
                    if (!edge_in(&t, 1, 2, e12) || !edge_in(&t, 2, 0, e20) || !edge_in(&t, 0, 1, e01)) continue;
                    float q[3], qs;
                    weights(&t, cx, cy, q, &qs);
                    of4 inputs[4]; memset(inputs, 0, sizeof(inputs));
                    for (int i = 0; i < cc.inputCount; i++) {
                        float a[3][4];
                        for (int k = 0; k < 3; k++) { a[k][3] = 1.0f; memcpy(a[k], t.vp[k] + cc.inputOffset[i], cc.opt_alpha ? 16 : 12); }   /* VS: float4(iInput, 1) */
                        inputs[i].x = interp_attr(&t, q, qs, a[0][0], a[1][0], a[2][0]); inputs[i].y = interp_attr(&t, q, qs, a[0][1], a[1][1], a[2][1]);
                        inputs[i].z = interp_attr(&t, q, qs, a[0][2], a[1][2], a[2][2]); inputs[i].w = interp_attr(&t, q, qs, a[0][3], a[1][3], a[2][3]);
                    }
                    of4 texVal0 = { 0.0f, 0.0f, 0.0f, 0.0f };
                    if (cc.useTextures[0] && d->diffuse) {
                        float uv[3][2], uraw[3][2];
                        for (int k = 0; k < 3; k++) memcpy(uraw[k], t.vp[k] + cc.uvOffset, 8);
                        for (int c = 0; c < 2; c++) { const float a[3] = { uraw[0][c], uraw[1][c], uraw[2][c] }; float o[3]; corner_values(&t, a, o); uv[0][c] = o[0]; uv[1][c] = o[1]; uv[2][c] = o[2]; }
                        float u = interp(q, qs, uv[0][0], uv[1][0], uv[2][0]), v = interp(q, qs, uv[0][1], uv[1][1], uv[2][1]);
                        float qx[3], qxs, qy[3], qys;                                       /* S7 */
                        weights(&t, cx + 256, cy, qx, &qxs); weights(&t, cx, cy + 256, qy, &qys);
                        of2 ddx = { interp(qx, qxs, uv[0][0], uv[1][0], uv[2][0]) - u, interp(qx, qxs, uv[0][1], uv[1][1], uv[2][1]) - v };
                        of2 ddy = { interp(qy, qys, uv[0][0], uv[1][0], uv[2][0]) - u, interp(qy, qys, uv[0][1], uv[1][1], uv[2][1]) - v };
                        float tex[4];
                        otex_sample_grad(d->diffuse, u, v, ddx, ddy, (int)d->filter, (int)d->hAddr, (int)d->vAddr, tex);
                        texVal0.x = tex[0]; texVal0.y = tex[1]; texVal0.z = tex[2]; texVal0.w = tex[3];
                    }
                    float src[4];
                    oshade_raster_pixel(&cc, inputs, texVal0, src);
                    for (int k = 0; k < 4; k++) src[k] = src[k] > 0.0f ? (src[k] < 1.0f ? src[k] : 1.0f) : 0.0f;      /* S8 */
                    uint8_t *dst = target + 4 * ((size_t)py * (size_t)w + (size_t)px);
                    float dr = from_unorm8(dst[0]), dg = from_unorm8(dst[1]), db = from_unorm8(dst[2]), da = from_unorm8(dst[3]);
                    const float ia = 1.0f - src[3];
                    dst[0] = to_unorm8(src[0] * src[3] + dr * ia); dst[1] = to_unorm8(src[1] * src[3] + dg * ia);
                    dst[2] = to_unorm8(src[2] * src[3] + db * ia); dst[3] = to_unorm8(src[3] + da * ia);
                }
            }
        }
    }
}
