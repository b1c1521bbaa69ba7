// raster.hip -- raster (HUD / background) instance pass: software rasteriser + the generated raster pixel shader.
//
// Replaces the D3D12 graphics pipeline the reference draws its non-ray-traced instances with
// (/root/reference/src/rt64lib/private/rt64_shader.cpp:312-442 generateRasterGroup: pass-through vertex shader, pixel shader =
// diffuse texture sample + colour combiner, SRC_ALPHA / INV_SRC_ALPHA blending, no depth, no culling;
// private/rt64_view.cpp:1225-1254 drawInstances, :1292-1319 background pass + gBackground, :1657-1661 foreground pass).
// The rasteriser is fixed-function hardware there; here it follows the Direct3D 11 rules as the "Raster spec" S0-S8 written out
// in oracle/oracle_raster.c (24.8 fixed-point vertices, int64 edge functions, top-left rule, perspective-correct attributes,
// blending in the target's RGBA8 storage triangle by triangle).  Coverage is integer arithmetic: bit-exact with the oracle.
//
// MI355X shape: raster_setup_kernel turns every triangle of a draw list into a 96-byte record (one thread per triangle);
// raster_draw_kernel runs one thread per target pixel (32x2 pixels per wave), walks the records IN DRAW ORDER -- the record
// address is wave-uniform, so records arrive through scalar loads and a triangle whose bounding box misses the wave's 32x2
// pixels costs a few SALU instructions -- and keeps the destination pixel in registers until the list is exhausted
// (one read + one write of the target per pixel, however many layers blend).  HUD lists are hundreds of triangles; no binning.
#include <algorithm>
#include <cstring>
#include "kernels.h"
#include "shade.h"
#include "raster_pixel.h"

namespace {

// S0 (oracle/oracle_raster.c): plane distances of the homogeneous clipper, fp32, unfused.
#define RASTER_W_EPS 9.5367431640625e-07f       // 2^-20
#define RASTER_GUARD 4.0f
struct ClipVertex { float c[7]; };               // x, y, z, w, b0, b1, b2
DEV float clip_distance(int plane, const ClipVertex &v) {
    const float x = v.c[0], y = v.c[1], z = v.c[2], w = v.c[3];
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

// S1-S3 of one (sub-)triangle into a record; false = nothing to draw (the record keeps an empty pixel box).
DEV bool raster_setup_record(RasterTri &r, const ClipVertex v[3], const uint32_t vtx[3], bool clipped, uint32_t inst,
                             float vpX, float vpY, float vpW, float vpH, int scL, int scT, int scR, int scB) {
    memset(&r, 0, sizeof(r));
    r.inst = inst; r.px0 = 1; r.px1 = 0; r.py0 = 1; r.py1 = 0; r.clipped = clipped ? 1u : 0u;
    bool ok = true;
    int64_t X[3], Y[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float px = v[k].c[0], py = v[k].c[1], pw = v[k].c[3];
        if (!(pw > 0.0f)) ok = false;
        const float rw = 1.0f / pw;
        const float xs = ((px * rw) * 0.5f + 0.5f) * vpW + vpX, ys = (0.5f - (py * rw) * 0.5f) * vpH + vpY;     // S1
        const float xf = xs * 256.0f, yf = ys * 256.0f;
        if (!(fabsf(xf) <= 4194304.0f) || !(fabsf(yf) <= 4194304.0f)) ok = false;                          // S2
        X[k] = ok ? (int64_t)__float2int_rn(xf) : 0; Y[k] = ok ? (int64_t)__float2int_rn(yf) : 0;
        r.rw[k] = rw; r.vtx[k] = vtx[k];
#pragma unroll
        for (int c = 0; c < 3; c++) r.B[k][c] = v[k].c[4 + c];
    }
    int64_t area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);                          // S3
    if (area2 == 0) ok = false;
    if (area2 < 0) {
        int64_t tx = X[1]; X[1] = X[2]; X[2] = tx; int64_t ty = Y[1]; Y[1] = Y[2]; Y[2] = ty;
        float tr = r.rw[1]; r.rw[1] = r.rw[2]; r.rw[2] = tr;
        if (!clipped) { uint32_t tv = r.vtx[1]; r.vtx[1] = r.vtx[2]; r.vtx[2] = tv; }                       // clipped: the original vertices stay, the barycentric rows move
        else {
#pragma unroll
            for (int c = 0; c < 3; c++) { float tb = r.B[1][c]; r.B[1][c] = r.B[2][c]; r.B[2][c] = tb; }
        }
    }
    if (ok) {
#pragma unroll
        for (int k = 0; k < 3; k++) { r.X[k] = (int32_t)X[k]; r.Y[k] = (int32_t)Y[k]; }
        const int64_t minX = min(X[0], min(X[1], X[2])), maxX = max(X[0], max(X[1], X[2])), minY = min(Y[0], min(Y[1], Y[2])), maxY = max(Y[0], max(Y[1], Y[2]));
        r.px0 = max((int)(minX >> 8), scL); r.px1 = min((int)(maxX >> 8), scR - 1);
        r.py0 = max((int)(minY >> 8), scT); r.py1 = min((int)(maxY >> 8), scB - 1);
    }
    return ok;
}

// Short draw lists (a HUD is one or two instances) hand their instance table to the setup kernel BY VALUE, in the kernel arguments: the
// kernel reads it from there and writes the device copy the draw kernel (or the frame kernel, when the HUD is folded into it) reads
// afterwards -- a frame that re-stages its lists then has no table copy in front of the setup launch (5.6 us each on the stream).
#define RASTER_INLINE_MAX 16
struct RasterInlineTable { GpuRasterInstance inst[RASTER_INLINE_MAX]; };

template <class Table>
DEV void raster_setup_triangle(const Table &instances, uint32_t instanceCount, uint32_t triTotal, RasterTri *tris, int w, int h, int y0, int y1, int apply, uint32_t t) {
    if (t >= triTotal) return;
    uint32_t lo = 0, hi = instanceCount - 1;                                  // last instance with firstTri <= t
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (instances[mid].firstTri <= t) lo = mid; else hi = mid - 1; }
    const GpuRasterInstance &in = instances[lo];
    const uint32_t tri = t - in.firstTri;
    float vpX = 0.0f, vpY = 0.0f, vpW = (float)w, vpH = (float)h;
    int scL = 0, scT = 0, scR = w, scB = h;
    if (apply) {                                                               // rt64_view.cpp:1114-1136
        if (in.scissorRect[2] > 0 && in.scissorRect[3] > 0) { scL = in.scissorRect[0]; scT = h - in.scissorRect[1] - in.scissorRect[3]; scR = in.scissorRect[0] + in.scissorRect[2]; scB = h - in.scissorRect[1]; }
        if (in.viewportRect[2] > 0 && in.viewportRect[3] > 0) { vpX = (float)in.viewportRect[0]; vpY = (float)(h - in.viewportRect[1] - in.viewportRect[3]); vpW = (float)in.viewportRect[2]; vpH = (float)in.viewportRect[3]; }
    }
    scL = max(scL, 0); scT = max(scT, y0); scR = min(scR, w); scB = min(scB, y1);
    // S0: homogeneous clipping
    ClipVertex poly[RASTER_MAX_POLY + 1];
    uint32_t vtx[3];
    bool all = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        vtx[k] = in.indices[3 * tri + k];
        const float *p = reinterpret_cast<const float *>(in.vertices + (size_t)vtx[k] * in.vertexStride);
        poly[k].c[0] = p[0]; poly[k].c[1] = p[1]; poly[k].c[2] = p[2]; poly[k].c[3] = p[3];
        poly[k].c[4] = k == 0 ? 1.0f : 0.0f; poly[k].c[5] = k == 1 ? 1.0f : 0.0f; poly[k].c[6] = k == 2 ? 1.0f : 0.0f;
        for (int pl = 0; pl < 7; pl++) if (!(clip_distance(pl, poly[k]) >= 0.0f)) all = false;
    }
    const uint32_t extraBase = triTotal + RASTER_EXTRA_PER_TRI * t;
    if (all) {
        const ClipVertex c3[3] = { poly[0], poly[1], poly[2] };
        RasterTri r;
        raster_setup_record(r, c3, vtx, false, lo, vpX, vpY, vpW, vpH, scL, scT, scR, scB);
        r.extraFirst = extraBase; r.extraCount = 0;
        tris[t] = r;
        return;
    }
    int n = 3;
    for (int pl = 0; pl < 7 && n >= 3; pl++) {
        ClipVertex out[RASTER_MAX_POLY + 1]; int m = 0;
        for (int i = 0; i < n; i++) {
            const ClipVertex &A = poly[i], &Bv = poly[(i + 1) % n];
            const float dA = clip_distance(pl, A), dB = clip_distance(pl, Bv);
            const bool inA = dA >= 0.0f, inB = dB >= 0.0f;
            if (inA && m < RASTER_MAX_POLY) out[m++] = A;
            if (inA != inB && m < RASTER_MAX_POLY) {
                const ClipVertex &P = inA ? A : Bv, &Q = inA ? Bv : A;
                const float dP = inA ? dA : dB, dQ = inA ? dB : dA;
                const float tt = dP / (dP - dQ);
                for (int c = 0; c < 7; c++) out[m].c[c] = P.c[c] + tt * (Q.c[c] - P.c[c]);
                m++;
            }
        }
        n = m;
        for (int i = 0; i < n; i++) poly[i] = out[i];
    }
    if (n < 3) n = 0;
    uint32_t pieces = 0;
    RasterTri first; bool haveFirst = false;
    for (int sub = 1; sub + 1 < n; sub++) {
        const ClipVertex c3[3] = { poly[0], poly[sub], poly[sub + 1] };
        RasterTri r;
        raster_setup_record(r, c3, vtx, true, lo, vpX, vpY, vpW, vpH, scL, scT, scR, scB);       // a degenerate piece keeps an empty pixel box
        r.extraFirst = extraBase; r.extraCount = 0;
        if (!haveFirst) { first = r; haveFirst = true; }
        else tris[extraBase + pieces++] = r;
    }
    if (!haveFirst) { const ClipVertex z3[3] = { poly[0], poly[0], poly[0] }; raster_setup_record(first, z3, vtx, true, lo, vpX, vpY, vpW, vpH, scL, scT, scR, scB); first.px0 = 1; first.px1 = 0; first.py0 = 1; first.py1 = 0; first.extraFirst = extraBase; }
    first.extraCount = pieces;
    tris[t] = first;
}
__global__ __launch_bounds__(256) void raster_setup_kernel(const GpuRasterInstance *instances, uint32_t instanceCount, uint32_t triTotal,
                                                           RasterTri *tris, int w, int h, int y0, int y1, int apply) {
    raster_setup_triangle(instances, instanceCount, triTotal, tris, w, h, y0, y1, apply, blockIdx.x * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void raster_setup_inline_kernel(RasterInlineTable table, GpuRasterInstance *deviceTable, uint32_t instanceCount, uint32_t triTotal,
                                                                  RasterTri *tris, int w, int h, int y0, int y1, int apply) {
    if (blockIdx.x == 0) {
        const uint32_t words = instanceCount * (uint32_t)(sizeof(GpuRasterInstance) / 4);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&table);
        for (uint32_t k = threadIdx.x; k < words; k += 256) reinterpret_cast<uint32_t *>(deviceTable)[k] = src[k];
    }
    raster_setup_triangle(table.inst, instanceCount, triTotal, tris, w, h, y0, y1, apply, blockIdx.x * 256 + threadIdx.x);
}

// Everything a frame with changed tables queues in front of its first pass, as ONE launch: the frame-table upload (workgroups [0, copyBlocks): 16-byte words
// from the pinned upload ring, which the device reads across the host link) and the setup of up to RASTER_PROLOGUE_LISTS short draw lists, tables by value
// (workgroups from list.firstBlock on).  A host that re-sends its scene every frame -- the reference's own per-frame behaviour, `always_rebuild` -- queued a copy
// and two setup launches here, each 3-4 us long and 5 us behind the one before it.
__global__ __launch_bounds__(256) void frame_prologue_kernel(RasterPrologue a) {
    const uint32_t b = blockIdx.x;
    if (b < a.copyBlocks) {
        const uint32_t k = b * 256 + threadIdx.x;
        if (k < a.copyWords) static_cast<uint4 *>(a.copyDst)[k] = static_cast<const uint4 *>(a.copySrc)[k];
        return;
    }
#pragma unroll 1
    for (uint32_t l = 0; l < a.listCount; l++) {
        const RasterPrologueList &L = a.list[l];
        const uint32_t blocks = L.triTotal ? (L.triTotal + 255) / 256 : 1u;
        if (b < L.firstBlock || b >= L.firstBlock + blocks) continue;
        if (b == L.firstBlock) {
            const uint32_t words = L.instanceCount * (uint32_t)(sizeof(GpuRasterInstance) / 4);
            const uint32_t *src = reinterpret_cast<const uint32_t *>(L.inst);
            for (uint32_t k = threadIdx.x; k < words; k += 256) reinterpret_cast<uint32_t *>(L.deviceTable)[k] = src[k];
        }
        raster_setup_triangle(L.inst, L.instanceCount, L.triTotal, static_cast<RasterTri *>(L.tris), L.w, L.h, L.y0, L.y1, L.apply, (b - L.firstBlock) * 256 + threadIdx.x);
        return;
    }
}

__global__ __launch_bounds__(256) void raster_draw_kernel(const GpuRasterInstance *__restrict__ instances, const RasterTri *__restrict__ tris, uint32_t triTotal,
                                                          const GpuTexture *__restrict__ textures, uint8_t *target, int w, int y0, int y1, int gx0, int gy0, int stripRank, int stripCount, int clear) {
    const int bx = (int)blockIdx.x + gx0, by = (int)blockIdx.y + gy0;
    const int x = bx * 32 + (threadIdx.x & 31), y = y0 + by * 8 + (threadIdx.x >> 5);
    const bool inside = x < w && y < y1 && (((y - y0) / 16) % stripCount) == stripRank;
    // wave-uniform pixel rectangle of this wave: 32 x 2
    const int wx0 = bx * 32, wx1 = wx0 + 31, wy0 = __builtin_amdgcn_readfirstlane(y0 + by * 8 + (int)((threadIdx.x >> 6) * 2)), wy1 = wy0 + 1;
    // clear: the target starts as 0 (gBackground, rt64_view.cpp:1298-1319) -- the launch then covers the whole target and every pixel is stored, covered or not
    uint32_t dstBits = 0; bool loaded = clear != 0, dirty = clear != 0 && inside;
    const size_t i = (size_t)y * (size_t)w + (size_t)x;
    raster_blend_pixel(instances, tris, triTotal, textures, x, y, inside, wx0, wx1, wy0, wy1, reinterpret_cast<const uint32_t *>(target) + i, dstBits, loaded, dirty);
    if (dirty) reinterpret_cast<uint32_t *>(target)[i] = dstBits;
}

}  // namespace

size_t raster_tri_bytes(uint32_t triTotal) { return (size_t)triTotal * (1 + RASTER_EXTRA_PER_TRI) * sizeof(RasterTri); }       // first pieces + the clipper's further pieces

hipError_t launch_raster_setup(const GpuRasterInstance *instances, uint32_t instanceCount, uint32_t triTotal, void *tris, int w, int h, int y0, int y1, bool apply, hipStream_t s) {
    if (instanceCount == 0 || triTotal == 0) return hipSuccess;
    hipLaunchKernelGGL(raster_setup_kernel, dim3((triTotal + 255) / 256), dim3(256), 0, s, instances, instanceCount, triTotal, static_cast<RasterTri *>(tris), w, h, y0, y1, apply ? 1 : 0);
    return hipGetLastError();
}
bool raster_setup_takes_table_inline(uint32_t instanceCount) { return instanceCount >= 1 && instanceCount <= RASTER_INLINE_MAX; }
// hostTable: the list's instances in host memory (copied into the launch); deviceTable: where the kernel leaves them for the draw
hipError_t launch_raster_setup_inline(const GpuRasterInstance *hostTable, GpuRasterInstance *deviceTable, uint32_t instanceCount, uint32_t triTotal, void *tris, int w, int h, int y0, int y1, bool apply, hipStream_t s) {
    if (!raster_setup_takes_table_inline(instanceCount)) return hipErrorInvalidValue;
    RasterInlineTable table;
    memset(&table, 0, sizeof(table));
    memcpy(table.inst, hostTable, (size_t)instanceCount * sizeof(GpuRasterInstance));
    hipLaunchKernelGGL(raster_setup_inline_kernel, dim3(triTotal ? (triTotal + 255) / 256 : 1u), dim3(256), 0, s, table, deviceTable, instanceCount, triTotal, static_cast<RasterTri *>(tris), w, h, y0, y1, apply ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_raster_draw(const GpuRasterInstance *instances, const void *tris, uint32_t triTotal, const GpuTexture *textures, uint8_t *target,
                              int w, int y0, int y1, const int bounds[4], int stripRank, int stripCount, bool clear, hipStream_t s) {
    // grid over the list's bounding rectangle, aligned to the 32 x 8 block shape (rows relative to y0 keep the strip arithmetic); the whole target when it is cleared as well
    const int whole[4] = { 0, y0, w, y1 };
    if (clear) bounds = whole;
    const int gx0 = bounds[0] / 32, gx1 = (bounds[2] + 31) / 32, gy0 = (std::max(bounds[1], y0) - y0) / 8, gy1 = (std::min(bounds[3], y1) - y0 + 7) / 8;
    if ((triTotal == 0 && !clear) || gx1 <= gx0 || gy1 <= gy0) return hipSuccess;
    hipLaunchKernelGGL(raster_draw_kernel, dim3((unsigned)(gx1 - gx0), (unsigned)(gy1 - gy0)), dim3(256), 0, s, instances, static_cast<const RasterTri *>(tris), triTotal, textures,
                       target, w, y0, y1, gx0, gy0, stripRank, stripCount, clear ? 1 : 0);
    return hipGetLastError();
}

bool frame_prologue_takes(const RasterPrologue &a) {
    if (a.listCount > RASTER_PROLOGUE_LISTS || a.copyWords > RASTER_PROLOGUE_COPY_WORDS) return false;
    for (uint32_t l = 0; l < a.listCount; l++) if (a.list[l].instanceCount < 1 || a.list[l].instanceCount > RASTER_PROLOGUE_INSTANCES) return false;
    return a.listCount != 0 || a.copyWords != 0;
}
// `a`: copySrc / copyDst / copyWords and the lists filled in by the caller; the workgroup ranges are assigned here
hipError_t launch_frame_prologue(RasterPrologue &a, hipStream_t s) {
    if (!frame_prologue_takes(a)) return hipErrorInvalidValue;
    a.copyBlocks = (a.copyWords + 255) / 256;
    uint32_t blocks = a.copyBlocks;
    for (uint32_t l = 0; l < a.listCount; l++) { a.list[l].firstBlock = blocks; blocks += a.list[l].triTotal ? (a.list[l].triTotal + 255) / 256 : 1u; }
    hipLaunchKernelGGL(frame_prologue_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}
