// svgf.hip -- SVGF (spatiotemporal variance-guided filter) for the GI buffer.
//
// BASELINE.json's north_star puts an SVGF denoiser where the reference has vendor upscaler hooks (rt64_{dlss,fsr,xess}.cpp) and
// five 3x3 Gaussian passes over gIndirectLightAccum (rt64_view.cpp:1512-1530; GaussianFilterRGB3x3CS.hlsl).  Algorithm: Schied et
// al. 2017 (HPG), on the demodulated GI signal (albedo is multiplied back in ComposePS.hlsl:28-29):
//   temporal : colour history is the reference's own reprojection (IndirectRayGen.hlsl:43-56,126-127); luminance moments are
//              accumulated next to it in indirect_kernel (passes.hip) with the same reprojection and history length;
//   variance : max(0, mu2 - mu1^2), or a 7x7 bilateral spatial estimate x 4/history while history < 4;
//   a-trous  : 5 iterations (steps 1,2,4,8,16), 5x5 B3-spline, edge stops on depth gradient, normal (^128) and luminance
//              (sigma_l = 4 x sqrt of the 3x3-Gaussian-filtered variance); variance filtered with squared weights.
// Images between iterations are RGBA16F (rgb = colour, a = variance) in the reference's filter ping-pong buffers.
// Both kernels are plain HBM-bound image passes: 25 (49) taps x 8 B + guide reads per pixel, one thread per pixel, 32x8 tiles.
#include "kernels.h"
#include "device_math.h"

namespace {

DEV float lum3(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }

DEV float grad_z(const float *depth, int x, int y, int w, int h) {
    const float z = depth[(size_t)y * w + x];
    const float zx = depth[(size_t)y * w + (x + 1 < w ? x + 1 : x)], zy = depth[(size_t)(y + 1 < h ? y + 1 : y) * w + x];
    return fmaxf(fabsf(zx - z), fabsf(zy - z));
}

__global__ __launch_bounds__(256) void svgf_variance_kernel(const uint16_t *color, const float *moments, const int32_t *instanceId, const uint16_t *normal,
                                                            const float *depth, uint16_t *out, int w, int h) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const size_t i = (size_t)y * w + x;
    const f4 c = load_rgba16f(color, i);
    float var = 0.0f;
    if (instanceId[i] >= 0) {
        if (c.w >= 4.0f) { const float m1 = moments[2 * i], m2 = moments[2 * i + 1]; var = fmaxf(0.0f, m2 - m1 * m1); }
        else {
            const f3 np = xyz(load_rgba16f(normal, i));
            const float zp = depth[i], gz = grad_z(depth, x, y, w, h);
            float sw = 0.0f, s1 = 0.0f, s2 = 0.0f;
            for (int dy = -3; dy <= 3; dy++)
                for (int dx = -3; dx <= 3; dx++) {
                    const int qx = x + dx, qy = y + dy;
                    if (qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
                    const size_t j = (size_t)qy * w + qx;
                    if (instanceId[j] < 0) continue;
                    const f3 nq = xyz(load_rgba16f(normal, j));
                    const float dist = sqrtf((float)(dx * dx + dy * dy));
                    const float wz = expf(-fabsf(zp - depth[j]) / (1.0f * gz * dist + 1e-8f));
                    const float wn = powf(fmaxf(0.0f, dot3(np, nq)), 128.0f);
                    const float wt = wz * wn;
                    const f4 cq = load_rgba16f(color, j);
                    const float l = lum3(cq.x, cq.y, cq.z);
                    sw += wt; s1 += wt * l; s2 += wt * l * l;
                }
            if (sw > 0.0f) { const float m1 = s1 / sw, m2 = s2 / sw; var = fmaxf(0.0f, m2 - m1 * m1) * (4.0f / fmaxf(c.w, 1.0f)); }
        }
    }
    store_rgba16f(out, i, c.x, c.y, c.z, var);
}

__global__ __launch_bounds__(256) void svgf_atrous_kernel(const uint16_t *in, uint16_t *out, const int32_t *instanceId, const uint16_t *normal, const float *depth,
                                                          int w, int h, int step) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const size_t i = (size_t)y * w + x;
    const f4 c = load_rgba16f(in, i);
    if (instanceId[i] < 0) { store_rgba16f(out, i, c.x, c.y, c.z, c.w); return; }
    const float K[5] = { 1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f };
    const float G[3] = { 0.25f, 0.5f, 0.25f };
    float gv = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            int qx = x + dx, qy = y + dy;
            qx = qx < 0 ? 0 : (qx >= w ? w - 1 : qx); qy = qy < 0 ? 0 : (qy >= h ? h - 1 : qy);
            gv += G[dx + 1] * G[dy + 1] * load_rgba16f(in, (size_t)qy * w + qx).w;
        }
    const float phiL = 4.0f * sqrtf(fmaxf(0.0f, gv)) + 1e-6f;
    const f3 np = xyz(load_rgba16f(normal, i));
    const float zp = depth[i], gz = grad_z(depth, x, y, w, h), lp = lum3(c.x, c.y, c.z);
    float sw = 0.0f, sr = 0.0f, sg = 0.0f, sb = 0.0f, sv = 0.0f;
#pragma unroll
    for (int ky = -2; ky <= 2; ky++)
#pragma unroll
        for (int kx = -2; kx <= 2; kx++) {
            const int qx = x + kx * step, qy = y + ky * step;
            if (qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
            const size_t j = (size_t)qy * w + qx;
            if (instanceId[j] < 0) continue;
            const f4 cq = load_rgba16f(in, j);
            const float hk = K[kx + 2] * K[ky + 2];
            float wt = hk;
            if (kx != 0 || ky != 0) {
                const f3 nq = xyz(load_rgba16f(normal, j));
                const float dist = sqrtf((float)(kx * kx + ky * ky)) * (float)step;
                const float wz = expf(-fabsf(zp - depth[j]) / (1.0f * gz * dist + 1e-8f));
                const float wn = powf(fmaxf(0.0f, dot3(np, nq)), 128.0f);
                const float wl = expf(-fabsf(lp - lum3(cq.x, cq.y, cq.z)) / phiL);
                wt = hk * wz * wn * wl;
            }
            sw += wt; sr += wt * cq.x; sg += wt * cq.y; sb += wt * cq.z; sv += wt * wt * cq.w;
        }
    const float inv = 1.0f / sw;
    store_rgba16f(out, i, sr * inv, sg * inv, sb * inv, sv * inv * inv);
}

}  // namespace

hipError_t launch_svgf(const ViewImages &I, int cur, int width, int height, hipStream_t s) {
    dim3 grid((unsigned)(width + 31) / 32, (unsigned)(height + 7) / 8);
    hipLaunchKernelGGL(svgf_variance_kernel, grid, dim3(256), 0, s, I.indirectLight[cur], I.moments[cur], I.instanceId, I.normal[cur], I.depth[cur], I.filteredIndirect[0], width, height);
    for (int k = 0; k < 5; k++)
        hipLaunchKernelGGL(svgf_atrous_kernel, grid, dim3(256), 0, s, I.filteredIndirect[k % 2], I.filteredIndirect[(k % 2) ^ 1], I.instanceId, I.normal[cur], I.depth[cur], width, height, 1 << k);
    return hipGetLastError();
}
