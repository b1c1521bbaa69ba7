/* oracle_svgf.c -- SVGF denoiser of the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h).
 *
 * BASELINE.json's north_star replaces the reference's vendor upscaler hooks (ref:private/rt64_{dlss,fsr,xess}.cpp) and its
 * five 3x3 Gaussian passes over the GI buffer (ref:private/rt64_view.cpp:1512-1530) by an SVGF spatiotemporal filter.
 * There is no SVGF in the reference to restate; the published algorithm is
 *   Schied et al. 2017, "Spatiotemporal Variance-Guided Filtering" (HPG), sections 4.1-4.4,
 * applied to the demodulated GI signal (gIndirectLightAccum; albedo is multiplied back in ComposePS.hlsl:28-29):
 *   temporal  colour: the reference's own reprojection + history blend in IndirectRayGen.hlsl:43-56,126-127 (kept as is);
 *             luminance moments (mu1, mu2) accumulated with the same reprojection and the same history length
 *             (oracle_render.c: pass_indirect) -> variance = max(0, mu2 - mu1^2);
 *   variance  history < 4: 7x7 bilateral spatial estimate (section 4.2), scaled by 4 / history;
 *   a-trous   5 iterations, 5x5 B3-spline kernel, steps 1,2,4,8,16, edge-stopping weights (section 4.4):
 *               w_z = exp(-|z_p - z_q| / (sigma_z * gradz_p * |offset| + 1e-8)),  gradz = max forward difference of depth
 *               w_n = max(0, n_p . n_q)^128
 *               w_l = exp(-|l_p - l_q| / (sigma_l * sqrt(max(0, gauss3x3(var)_p)) + 1e-6)),  sigma_z = 1, sigma_l = 4
 *             colour' = sum(h w c_q) / sum(h w);  variance' = sum((h w)^2 var_q) / sum(h w)^2
 *   images    between iterations RGBA16F (rgb = colour, a = variance), like the reference's filter ping-pong buffers.
 * Shared contract with csrc/svgf.hip: same operation order; exp/pow differ in the last bits (tolerance in the tests).
 */
#include <math.h>
#include <string.h>
#include "oracle_internal.h"

static inline float lum(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }

static float grad_z(const OScene *s, int cur, int x, int y, int w, int h) {
    const float *d = s->depth[cur];
    float z = d[(size_t)y * w + x];
    float zx = d[(size_t)y * w + (x + 1 < w ? x + 1 : x)], zy = d[(size_t)(y + 1 < h ? y + 1 : y) * w + x];
    return fmaxf(fabsf(zx - z), fabsf(zy - z));
}

static void variance_pass(OScene *s, int cur, int w, int h, float *out) {
    const float *col = s->indirectLight[cur], *mom = s->moments[cur];
#pragma omp parallel for schedule(dynamic, 8)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            float r = col[4 * i], g = col[4 * i + 1], b = col[4 * i + 2], hist = col[4 * i + 3];
            float var = 0.0f;
            if (s->instanceId[i] >= 0) {
                if (hist >= 4.0f) var = fmaxf(0.0f, mom[2 * i + 1] - mom[2 * i] * mom[2 * i]);
                else {
                    of3 np = v3(s->normal[cur][4 * i], s->normal[cur][4 * i + 1], s->normal[cur][4 * i + 2]);
                    float zp = s->depth[cur][i], gz = grad_z(s, cur, x, y, w, h);
                    float sw = 0.0f, s1 = 0.0f, s2 = 0.0f;
                    for (int dy = -3; dy <= 3; dy++)
                        for (int dx = -3; dx <= 3; dx++) {
                            int qx = x + dx, qy = y + dy;
                            if (qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
                            size_t j = (size_t)qy * w + qx;
                            if (s->instanceId[j] < 0) continue;
                            of3 nq = v3(s->normal[cur][4 * j], s->normal[cur][4 * j + 1], s->normal[cur][4 * j + 2]);
                            float dist = sqrtf((float)(dx * dx + dy * dy));
                            float wz = expf(-fabsf(zp - s->depth[cur][j]) / (1.0f * gz * dist + 1e-8f));
                            float wn = powf(fmaxf(0.0f, v3dot(np, nq)), 128.0f);
                            float wt = wz * wn;
                            float l = lum(col[4 * j], col[4 * j + 1], col[4 * j + 2]);
                            sw += wt; s1 += wt * l; s2 += wt * l * l;
                        }
                    if (sw > 0.0f) { float m1 = s1 / sw, m2 = s2 / sw; var = fmaxf(0.0f, m2 - m1 * m1) * (4.0f / fmaxf(hist, 1.0f)); }
                }
            }
            out[4 * i] = q_f16(r); out[4 * i + 1] = q_f16(g); out[4 * i + 2] = q_f16(b); out[4 * i + 3] = q_f16(var);
        }
}

static void atrous_pass(const OScene *s, int cur, int w, int h, int step, const float *in, float *out) {
    static const float K[5] = { 1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f };
    static const float G[3] = { 0.25f, 0.5f, 0.25f };
#pragma omp parallel for schedule(dynamic, 8)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            float cr = in[4 * i], cg = in[4 * i + 1], cb = in[4 * i + 2], cv = in[4 * i + 3];
            if (s->instanceId[i] < 0) { out[4 * i] = cr; out[4 * i + 1] = cg; out[4 * i + 2] = cb; out[4 * i + 3] = cv; continue; }
            float gv = 0.0f;                                   /* 3x3 Gaussian of the variance (clamped addressing) */
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int qx = x + dx, qy = y + dy;
                    qx = qx < 0 ? 0 : (qx >= w ? w - 1 : qx); qy = qy < 0 ? 0 : (qy >= h ? h - 1 : qy);
                    gv += G[dx + 1] * G[dy + 1] * in[4 * ((size_t)qy * w + qx) + 3];
                }
            float phiL = 4.0f * sqrtf(fmaxf(0.0f, gv)) + 1e-6f;
            of3 np = v3(s->normal[cur][4 * i], s->normal[cur][4 * i + 1], s->normal[cur][4 * i + 2]);
            float zp = s->depth[cur][i], gz = grad_z(s, cur, x, y, w, h), lp = lum(cr, cg, cb);
            float sw = 0.0f, sr = 0.0f, sg = 0.0f, sb = 0.0f, sv = 0.0f;
            for (int ky = -2; ky <= 2; ky++)
                for (int kx = -2; kx <= 2; kx++) {
                    int qx = x + kx * step, qy = y + ky * step;
                    if (qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
                    size_t j = (size_t)qy * w + qx;
                    if (s->instanceId[j] < 0) continue;
                    float hk = K[kx + 2] * K[ky + 2], wt = hk;
                    if (kx != 0 || ky != 0) {
                        of3 nq = v3(s->normal[cur][4 * j], s->normal[cur][4 * j + 1], s->normal[cur][4 * j + 2]);
                        float dist = sqrtf((float)(kx * kx + ky * ky)) * (float)step;
                        float wz = expf(-fabsf(zp - s->depth[cur][j]) / (1.0f * gz * dist + 1e-8f));
                        float wn = powf(fmaxf(0.0f, v3dot(np, nq)), 128.0f);
                        float wl = expf(-fabsf(lp - lum(in[4 * j], in[4 * j + 1], in[4 * j + 2])) / phiL);
                        wt = hk * wz * wn * wl;
                    }
                    sw += wt; sr += wt * in[4 * j]; sg += wt * in[4 * j + 1]; sb += wt * in[4 * j + 2]; sv += wt * wt * in[4 * j + 3];
                }
            float inv = 1.0f / sw;                              /* sw >= K[2]*K[2] > 0: the centre tap always counts */
            out[4 * i] = q_f16(sr * inv); out[4 * i + 1] = q_f16(sg * inv); out[4 * i + 2] = q_f16(sb * inv); out[4 * i + 3] = q_f16(sv * inv * inv);
        }
}

void osvgf_filter(OScene *s, const OFrameParams *p, int cur) {
    int w = p->width, h = p->height;
    variance_pass(s, cur, w, h, s->filteredIndirect[0]);
    for (int k = 0; k < 5; k++)
        atrous_pass(s, cur, w, h, 1 << k, s->filteredIndirect[k % 2], s->filteredIndirect[(k % 2) ^ 1]);
}
