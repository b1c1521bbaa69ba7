// raster_pixel.h -- the per-pixel part of the raster (HUD / background) pass: the triangle records of a draw list and the walk of
// one target pixel over them (coverage, generated pixel shader, blending in RGBA8).  Shared by raster_draw_kernel (raster.hip: one
// thread per target pixel) and lean_frame_kernel (passes.hip: the foreground list is blended over the pixel the kernel has just
// composed, before it is stored -- no second launch for the HUD).  Raster spec S0-S8: oracle/oracle_raster.c.
#pragma once
#include "shade.h"

struct RasterTri {                 // 128 bytes
    int32_t X[3], Y[3];            // 24.8 fixed point, orientation normalised (area2 > 0)
    float rw[3];
    uint32_t vtx[3];               // vertex numbers in the same (possibly swapped) order; a clipped record keeps the ORIGINAL triangle's order
    int32_t px0, py0, px1, py1;    // pixel bounding box clipped to the scissor; px0 > px1 = nothing to draw
    uint32_t inst;
    uint32_t clipped;              // S0: this record is a piece of a clipped triangle: corner k's attributes are B[k] . (original vertices' values)
    uint32_t extraFirst, extraCount;   // further pieces of the same source triangle (records [extraFirst, extraFirst + extraCount)), drawn right after this one
    float B[3][3];
    uint32_t pad[3];
};
static_assert(sizeof(RasterTri) == 128, "RasterTri");
#define RASTER_MAX_POLY 10         // a triangle against seven planes: at most 10 vertices = 8 pieces; pieces 1..7 of source triangle t live at triTotal + 7 t + (j - 1)
#define RASTER_EXTRA_PER_TRI 7


DEV int64_t edge_fn(const RasterTri &t, int a, int b, int64_t px, int64_t py) {
    return (int64_t)(t.X[b] - t.X[a]) * (py - (int64_t)t.Y[a]) - (int64_t)(t.Y[b] - t.Y[a]) * (px - (int64_t)t.X[a]);
}
DEV bool edge_in(const RasterTri &t, int a, int b, int64_t e) {               // S4: top-left rule, y down
    if (e > 0) return true;
    if (e < 0) return false;
    const int dx = t.X[b] - t.X[a], dy = t.Y[b] - t.Y[a];
    return dy < 0 || (dy == 0 && dx > 0);
}
struct Weights { float q0, q1, q2, qs; };
DEV Weights weights_at(const RasterTri &t, float area, int64_t px, int64_t py) {      // S6
    const float l0 = (float)edge_fn(t, 1, 2, px, py) / area, l1 = (float)edge_fn(t, 2, 0, px, py) / area, l2 = (float)edge_fn(t, 0, 1, px, py) / area;
    Weights w; w.q0 = l0 * t.rw[0]; w.q1 = l1 * t.rw[1]; w.q2 = l2 * t.rw[2]; w.qs = (w.q0 + w.q1) + w.q2;
    return w;
}
DEV float interp(const Weights &w, float a0, float a1, float a2) { return ((w.q0 * a0 + w.q1 * a1) + w.q2 * a2) / w.qs; }


// One pixel (x, y) against the whole list, in draw order.  wx0..wy1: the pixel rectangle of the calling wave (wave-uniform), a
// triangle whose bounding box misses it costs a few scalar instructions.  dstBits is the pixel's RGBA8 value: read from *dstPixel on
// the first covering triangle unless the caller already holds it (loaded = true); dirty tells the caller to write it back.
// One pixel (x, y) against one record.
DEV void raster_blend_record(const GpuRasterInstance *__restrict__ instances, const RasterTri &t, const GpuTexture *__restrict__ textures,
                             int x, int y, bool inside, int wx0, int wx1, int wy0, int wy1, const uint32_t *dstPixel, uint32_t &dstBits, bool &loaded, bool &dirty) {
    {
        if (t.px0 > wx1 || t.px1 < wx0 || t.py0 > wy1 || t.py1 < wy0) return;            // uniform: scalar compares
        if (!inside || x < t.px0 || x > t.px1 || y < t.py0 || y > t.py1) return;
        const int64_t cx = (int64_t)x * 256 + 128, cy = (int64_t)y * 256 + 128;
        const int64_t e12 = edge_fn(t, 1, 2, cx, cy), e20 = edge_fn(t, 2, 0, cx, cy), e01 = edge_fn(t, 0, 1, cx, cy);
        if (!edge_in(t, 1, 2, e12) || !edge_in(t, 2, 0, e20) || !edge_in(t, 0, 1, e01)) return;
        const GpuRasterInstance &in = instances[t.inst];
        const GpuCombiner cc = in.cc;
        const float area = (float)(e12 + e20 + e01);                                      // = area2 (the three edge functions sum to it)
        const Weights wq = weights_at(t, area, cx, cy);
        const uint8_t *v0 = in.vertices + (size_t)t.vtx[0] * in.vertexStride, *v1 = in.vertices + (size_t)t.vtx[1] * in.vertexStride, *v2 = in.vertices + (size_t)t.vtx[2] * in.vertexStride;
        const bool clipped = t.clipped != 0;
        // attribute at the record's corners from the original vertices' values (S0; identity for an unclipped triangle), then S6
        auto attr = [&](const Weights &w, float a0, float a1, float a2) -> float {
            if (clipped) {
                const float c0 = (t.B[0][0] * a0 + t.B[0][1] * a1) + t.B[0][2] * a2, c1 = (t.B[1][0] * a0 + t.B[1][1] * a1) + t.B[1][2] * a2, c2 = (t.B[2][0] * a0 + t.B[2][1] * a1) + t.B[2][2] * a2;
                return interp(w, c0, c1, c2);
            }
            return interp(w, a0, a1, a2);
        };
        VertexData vd;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            vd.input[k] = mk4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k < cc.inputCount) {
                const float *a0 = reinterpret_cast<const float *>(v0 + cc.inputOffset[k]), *a1 = reinterpret_cast<const float *>(v1 + cc.inputOffset[k]), *a2 = reinterpret_cast<const float *>(v2 + cc.inputOffset[k]);
                vd.input[k].x = attr(wq, a0[0], a1[0], a2[0]); vd.input[k].y = attr(wq, a0[1], a1[1], a2[1]); vd.input[k].z = attr(wq, a0[2], a1[2], a2[2]);
                vd.input[k].w = cc.optAlpha ? attr(wq, a0[3], a1[3], a2[3]) : attr(wq, 1.0f, 1.0f, 1.0f);        // VS: float4(iInput, 1)
            }
        }
        f4 texVal0 = mk4(0.0f, 0.0f, 0.0f, 0.0f);
        if (cc.useTex0 && in.texDiffuse >= 0) {
            const float *u0 = reinterpret_cast<const float *>(v0 + cc.uvOffset), *u1 = reinterpret_cast<const float *>(v1 + cc.uvOffset), *u2 = reinterpret_cast<const float *>(v2 + cc.uvOffset);
            const float u = attr(wq, u0[0], u1[0], u2[0]), v = attr(wq, u0[1], u1[1], u2[1]);
            const Weights wx = weights_at(t, area, cx + 256, cy), wy = weights_at(t, area, cx, cy + 256);         // S7
            f2 ddx, ddy;
            ddx.x = attr(wx, u0[0], u1[0], u2[0]) - u; ddx.y = attr(wx, u0[1], u1[1], u2[1]) - v;
            ddy.x = attr(wy, u0[0], u1[0], u2[0]) - u; ddy.y = attr(wy, u0[1], u1[1], u2[1]) - v;
            texVal0 = tex_sample_grad(tex_view(textures + in.texDiffuse), u, v, ddx, ddy, in.filter, in.hAddr, in.vAddr);
        }
        const f4 t1 = mk4(1.0f, 0.0f, 1.0f, 1.0f);                                       // rt64_shader.cpp:377 (TODO in the reference)
        f4 src;
        if (!cc.colorAlphaSame && cc.optAlpha) { src = color_formula(cc, false, true, vd, texVal0, t1); src.w = alpha_formula(cc, vd, texVal0, t1); }
        else src = color_formula(cc, cc.optAlpha, cc.optAlpha, vd, texVal0, t1);
        src.x = src.x > 0.0f ? fminf(src.x, 1.0f) : 0.0f; src.y = src.y > 0.0f ? fminf(src.y, 1.0f) : 0.0f;       // S8
        src.z = src.z > 0.0f ? fminf(src.z, 1.0f) : 0.0f; src.w = src.w > 0.0f ? fminf(src.w, 1.0f) : 0.0f;
        if (!loaded) { dstBits = *dstPixel; loaded = true; }
        const float dr = from_unorm8((uint8_t)(dstBits & 0xFF)), dg = from_unorm8((uint8_t)((dstBits >> 8) & 0xFF)), db = from_unorm8((uint8_t)((dstBits >> 16) & 0xFF)), da = from_unorm8((uint8_t)(dstBits >> 24));
        const float ia = 1.0f - src.w;
        dstBits = (uint32_t)to_unorm8(src.x * src.w + dr * ia) | ((uint32_t)to_unorm8(src.y * src.w + dg * ia) << 8)
                | ((uint32_t)to_unorm8(src.z * src.w + db * ia) << 16) | ((uint32_t)to_unorm8(src.w + da * ia) << 24);
        dirty = true;
    }
}

// One pixel (x, y) against the whole list, in draw order.  wx0..wy1: the pixel rectangle of the calling wave (wave-uniform), a
// triangle whose bounding box misses it costs a few scalar instructions.  dstBits is the pixel's RGBA8 value: read from *dstPixel on
// the first covering triangle unless the caller already holds it (loaded = true); dirty tells the caller to write it back.
// The further pieces of a clipped triangle (S0) follow their first piece (they are disjoint, so their own order is immaterial).
DEV void raster_blend_pixel(const GpuRasterInstance *__restrict__ instances, const RasterTri *__restrict__ tris, uint32_t triTotal, const GpuTexture *__restrict__ textures,
                            int x, int y, bool inside, int wx0, int wx1, int wy0, int wy1, const uint32_t *dstPixel, uint32_t &dstBits, bool &loaded, bool &dirty) {
    for (uint32_t n = 0; n < triTotal; n++) {
        const RasterTri &t = tris[n];
        raster_blend_record(instances, t, textures, x, y, inside, wx0, wx1, wy0, wy1, dstPixel, dstBits, loaded, dirty);
        const uint32_t extra = t.extraCount;                                             // uniform
        for (uint32_t e = 0; e < extra; e++)
            raster_blend_record(instances, tris[t.extraFirst + e], textures, x, y, inside, wx0, wx1, wy0, wy1, dstPixel, dstBits, loaded, dirty);
    }
}
