// upscale.hip -- the temporal upscaler stage: jittered temporal-accumulation upsample of rtOutput to the display size.
//
// Stands where the reference calls a vendor SDK through `Upscaler::upscale` (private/rt64_upscaler.h:25-48; call site
// private/rt64_view.cpp:1584-1618: colour, flow, reactive mask, lock mask, depth, the frame's jitter in, rtOutputUpscaled out, which
// PostProcessPS then reads, :800-801).  DLSS / FSR2 / XeSS are SDKs that do not exist here; this kernel consumes the same inputs.
// Algorithm: "Upscale spec" U1-U8 in oracle/oracle_upscale.c (the scalar restatement the parity tests compare against) -- a 3 x 3
// Gaussian resample of the jittered render-size samples around each display pixel, history fetched along the dilated motion vector,
// clamped to the neighbourhood's colour box unless the lock mask holds it, blended by accumulated frame count / reactive mask.
//
// MI355X shape: one thread per display pixel, 32 x 8 pixels per workgroup; the nine taps of neighbouring lanes overlap (L1 / L2 hits),
// one 16-byte store per pixel; ~60 B of HBM traffic per display pixel -- bandwidth-trivial next to the ray passes.
#include "kernels.h"
#include "device_math.h"

namespace {

DEV float satf(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

__global__ __launch_bounds__(256) void taa_upsample_kernel(const float4 *__restrict__ color, const uint32_t *__restrict__ flow, const uint8_t *__restrict__ reactive,
                                                           const uint8_t *__restrict__ lock, const float *__restrict__ depth, int rw, int rh, float jx, float jy,
                                                           const float4 *__restrict__ prev, float4 *__restrict__ out, int dw, int dh, int haveHistory) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= dw || y >= dh) return;
    const float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;                 // U1
    const float rx = u * (float)rw, ry = v * (float)rh;
    int i0 = (int)floorf(rx - jx), k0 = (int)floorf(ry - jy);                                         // U2
    i0 = i0 < 0 ? 0 : (i0 > rw - 1 ? rw - 1 : i0); k0 = k0 < 0 ? 0 : (k0 > rh - 1 ? rh - 1 : k0);
    float sumW = 0.0f, sr = 0.0f, sg = 0.0f, sb = 0.0f;
    float mnr = INFINITY, mng = INFINITY, mnb = INFINITY, mxr = -INFINITY, mxg = -INFINITY, mxb = -INFINITY;
    float conf = 0.0f, bestDepth = INFINITY, fx = 0.0f, fy = 0.0f;
#pragma unroll
    for (int dk = -1; dk <= 1; dk++)
#pragma unroll
        for (int di = -1; di <= 1; di++) {                                                            // U3
            const int i = i0 + di, k = k0 + dk;
            const int ic = i < 0 ? 0 : (i > rw - 1 ? rw - 1 : i), kc = k < 0 ? 0 : (k > rh - 1 ? rh - 1 : k);
            const size_t q = (size_t)kc * (size_t)rw + (size_t)ic;
            const float sx = (float)i + 0.5f + jx, sy = (float)k + 0.5f + jy;
            const float dx = rx - sx, dy = ry - sy;
            const float w = exp2f(-((dx * dx + dy * dy) * 2.88539008f));
            const float4 c = color[q];
            sumW += w; sr += w * c.x; sg += w * c.y; sb += w * c.z;
            mnr = fminf(mnr, c.x); mng = fminf(mng, c.y); mnb = fminf(mnb, c.z); mxr = fmaxf(mxr, c.x); mxg = fmaxf(mxg, c.y); mxb = fmaxf(mxb, c.z);
            if (di == 0 && dk == 0) conf = w;
            const float z = depth[q];                                                                 // U4
            if (z < bestDepth) { bestDepth = z; const uint32_t fl = flow[q]; fx = f16_bits_to_f32((uint16_t)(fl & 0xFFFFu)); fy = f16_bits_to_f32((uint16_t)(fl >> 16)); }
        }
    const size_t q0 = (size_t)k0 * (size_t)rw + (size_t)i0;
    const float reac = satf(from_unorm8(reactive[q0])), lk = satf(from_unorm8(lock[q0]));
    const float cr = sr / sumW, cg = sg / sumW, cb = sb / sumW;
    float hr = 0.0f, hg = 0.0f, hb = 0.0f, N = 0.0f;
    const float pu = u + fx / (float)rw, pv = v + fy / (float)rh;                                      // U5
    if (haveHistory && pu >= 0.0f && pu <= 1.0f && pv >= 0.0f && pv <= 1.0f) {
        const float hx = pu * (float)dw - 0.5f, hy = pv * (float)dh - 0.5f;
        const float x0f = floorf(hx), y0f = floorf(hy), tx = hx - x0f, ty = hy - y0f;
        int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        x0 = x0 < 0 ? 0 : (x0 > dw - 1 ? dw - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > dw - 1 ? dw - 1 : x1);
        y0 = y0 < 0 ? 0 : (y0 > dh - 1 ? dh - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > dh - 1 ? dh - 1 : y1);
        const float4 c00 = prev[(size_t)y0 * dw + x0], c10 = prev[(size_t)y0 * dw + x1], c01 = prev[(size_t)y1 * dw + x0], c11 = prev[(size_t)y1 * dw + x1];
        { const float top = c00.x + tx * (c10.x - c00.x), bot = c01.x + tx * (c11.x - c01.x); hr = top + ty * (bot - top); }
        { const float top = c00.y + tx * (c10.y - c00.y), bot = c01.y + tx * (c11.y - c01.y); hg = top + ty * (bot - top); }
        { const float top = c00.z + tx * (c10.z - c00.z), bot = c01.z + tx * (c11.z - c01.z); hb = top + ty * (bot - top); }
        { const float top = c00.w + tx * (c10.w - c00.w), bot = c01.w + tx * (c11.w - c01.w); N = top + ty * (bot - top); }
    }
    float a = fmaxf(fmaxf(1.0f / (N + 1.0f), 0.1f * conf), reac);                                       // U7
    if (!(N > 0.0f)) a = 1.0f;
    float cr2 = fminf(fmaxf(hr, mnr), mxr), cg2 = fminf(fmaxf(hg, mng), mxg), cb2 = fminf(fmaxf(hb, mnb), mxb);   // U6
    cr2 = cr2 + lk * (hr - cr2); cg2 = cg2 + lk * (hg - cg2); cb2 = cb2 + lk * (hb - cb2);
    out[(size_t)y * (size_t)dw + (size_t)x] = make_float4(cr2 + a * (cr - cr2), cg2 + a * (cg - cg2), cb2 + a * (cb - cb2), fminf(N + 1.0f, 32.0f));
}

}  // namespace

hipError_t launch_taa_upsample(const ViewImages &I, int cur, int rw, int rh, float jx, float jy, const float *prev, float *out, int dw, int dh, bool haveHistory, hipStream_t s) {
    dim3 grid((unsigned)(dw + 31) / 32, (unsigned)(dh + 7) / 8);
    hipLaunchKernelGGL(taa_upsample_kernel, grid, dim3(256), 0, s, reinterpret_cast<const float4 *>(I.output), reinterpret_cast<const uint32_t *>(I.flow), I.reactiveMask, I.lockMask,
                       I.depth[cur], rw, rh, jx, jy, reinterpret_cast<const float4 *>(prev), reinterpret_cast<float4 *>(out), dw, dh, haveHistory ? 1 : 0);
    return hipGetLastError();
}
