/* oracle_upscale.c -- the temporal upscaler stage of the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h).
 *
 * The reference hands rtOutput + flow + reactive mask + lock mask + depth + the frame's jitter to a vendor upscaler behind
 * `Upscaler::upscale` (ref:private/rt64_upscaler.h:25-48, call site ref:private/rt64_view.cpp:1584-1618) and PostProcessPS then reads
 * rtOutputUpscaled (ref:rt64_view.cpp:800-801).  DLSS / FSR2 / XeSS are closed or absent SDKs; the MI355X library ships its own
 * stage behind the same inputs -- a jittered temporal-accumulation upsample ("TAAU") -- and this file is its scalar restatement.
 * What is taken from the reference / the published FSR2 API it drives:
 *   render size per quality mode   ref:rt64_fsr.cpp:98-126 (Native 100 %, UltraQuality 77 %, FSR2 ratios 1.5 / 1.7 / 2.0 / 3.0),
 *                                  QualityMode::Auto by display size ref:rt64_upscaler.cpp:11-36
 *   jitter                         HaltonJitter(frameCount, phases) ref:rt64_common.h:347-361, only with an upscaler ref:rt64_view.cpp:1273-1281,
 *                                  phases = int(8 (display width / render width)^2)   (ffxFsr2GetJitterPhaseCount, ref:rt64_fsr.cpp:128-130)
 * Upscale spec (U1-U8; csrc/upscale.hip follows it operation by operation).  Display pixel (x, y), render size rw x rh, jitter j:
 *   U1  uv = ((x + .5) / dw, (y + .5) / dh);  r = (uv.x rw, uv.y rh)             position in render-pixel units
 *   U2  i0 = clamp(floor(r.x - j.x), 0, rw - 1), k0 likewise                      render pixel whose jittered sample is nearest
 *   U3  over the 3 x 3 pixels around (i0, k0) (indices clamped for the fetch, unclamped for the distance):
 *       s = (i + .5 + j.x, k + .5 + j.y);  w = exp2(-|r - s|^2 * 2.88539008);  cur = sum(w c) / sum(w);  cmin / cmax per channel;
 *       conf = w of the centre pixel
 *   U4  guides: flow of the 3 x 3 pixel with the smallest depth (first one on ties, rows then columns), reactive / lock of (i0, k0)
 *   U5  uvPrev = uv + (flow.x / rw, flow.y / rh);  history h = bilinear, clamped addressing, of the previous upscaled image at uvPrev
 *       when there is one and 0 <= uvPrev <= 1;  N = h.a (accumulated frames), else N = 0
 *   U6  hc = clamp(h.rgb, cmin, cmax);  hc = hc + sat(lock) (h.rgb - hc)                      locked pixels keep their history
 *   U7  a = max(1 / (N + 1), 0.1 conf, sat(reactive));  N = 0 -> a = 1;  out.rgb = hc + a (cur - hc);  out.a = min(N + 1, 32)
 *   U8  PostProcessPS samples the upscaled image (display size) instead of rtOutput
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

/* QualityMode numbering of RT64_UPSCALER_MODE_* (include/rt64.h): 0 auto, 1 ultra performance, 2 performance, 3 balanced, 4 quality,
 * 5 ultra quality, 6 native.  Returns 0 when `upscaler` selects no built-in stage (OFF, or a vendor SDK that does not exist here). */
int oracle_upscaler_info(int upscaler, int mode, int displayW, int displayH, int *renderW, int *renderH, int *phases) {
    if (!(upscaler == 1 || upscaler == 3)) return 0;                       /* RT64_UPSCALER_AUTO / RT64_UPSCALER_FSR -> the built-in stage */
    if (mode == 0) {                                                       /* getQualityAuto, ref:rt64_upscaler.cpp:11-36 */
        const uint64_t px = (uint64_t)displayW * (uint64_t)displayH;
        mode = px <= 1280ull * 720 ? 5 : (px <= 1920ull * 1080 ? 4 : (px <= 2560ull * 1440 ? 3 : (px <= 3840ull * 2160 ? 2 : 1)));
    }
    int w, h;
    if (mode == 6) { w = displayW; h = displayH; }
    else if (mode == 5) { w = (displayW * 77) / 100; h = (displayH * 77) / 100; }
    else {
        const float ratio = mode == 4 ? 1.5f : (mode == 3 ? 1.7f : (mode == 2 ? 2.0f : 3.0f));
        w = (int)((float)displayW / ratio); h = (int)((float)displayH / ratio);
    }
    if (w < 1) w = 1;
    if (h < 1) h = 1;
    *renderW = w; *renderH = h;
    const float q = (float)displayW / (float)w;
    int n = (int)(8.0f * (q * q));
    *phases = n < 1 ? 1 : n;
    return 1;
}

static float satf(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

void oupscale_frame(const float *color, const float *flow, const float *reactive, const float *lock, const float *depth, int rw, int rh,
                    float jx, float jy, const float *prev, float *out, int dw, int dh, int haveHistory) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;           /* U1 */
            const float rx = u * (float)rw, ry = v * (float)rh;
            int i0 = (int)floorf(rx - jx), k0 = (int)floorf(ry - jy);                                   /* U2 */
            i0 = i0 < 0 ? 0 : (i0 > rw - 1 ? rw - 1 : i0); k0 = k0 < 0 ? 0 : (k0 > rh - 1 ? rh - 1 : k0);
            float sumW = 0.0f, sum[3] = { 0.0f, 0.0f, 0.0f }, cmin[3] = { INFINITY, INFINITY, INFINITY }, cmax[3] = { -INFINITY, -INFINITY, -INFINITY };
            float conf = 0.0f, bestDepth = INFINITY, fx = 0.0f, fy = 0.0f;
            for (int dk = -1; dk <= 1; dk++)
                for (int di = -1; di <= 1; di++) {                                                      /* U3 */
                    const int i = i0 + di, k = k0 + dk;
                    const int ic = i < 0 ? 0 : (i > rw - 1 ? rw - 1 : i), kc = k < 0 ? 0 : (k > rh - 1 ? rh - 1 : k);
                    const size_t q = (size_t)kc * (size_t)rw + (size_t)ic;
                    const float sx = (float)i + 0.5f + jx, sy = (float)k + 0.5f + jy;
                    const float dx = rx - sx, dy = ry - sy;
                    const float w = exp2f(-((dx * dx + dy * dy) * 2.88539008f));
                    sumW += w;
                    for (int c = 0; c < 3; c++) { const float cv = color[4 * q + c]; sum[c] += w * cv; cmin[c] = fminf(cmin[c], cv); cmax[c] = fmaxf(cmax[c], cv); }
                    if (di == 0 && dk == 0) conf = w;
                    const float z = depth[q];                                                           /* U4 */
                    if (z < bestDepth) { bestDepth = z; fx = flow[2 * q]; fy = flow[2 * q + 1]; }
                }
            const size_t q0 = (size_t)k0 * (size_t)rw + (size_t)i0;
            const float reac = satf(reactive[q0]), lk = satf(lock[q0]);
            float cur[3]; for (int c = 0; c < 3; c++) cur[c] = sum[c] / sumW;
            float h[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, N = 0.0f;
            const float pu = u + fx / (float)rw, pv = v + fy / (float)rh;                                /* U5 */
            if (haveHistory && pu >= 0.0f && pu <= 1.0f && pv >= 0.0f && pv <= 1.0f) {
                const float hx = pu * (float)dw - 0.5f, hy = pv * (float)dh - 0.5f;
                const float x0f = floorf(hx), y0f = floorf(hy), tx = hx - x0f, ty = hy - y0f;
                int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
                x0 = x0 < 0 ? 0 : (x0 > dw - 1 ? dw - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > dw - 1 ? dw - 1 : x1);
                y0 = y0 < 0 ? 0 : (y0 > dh - 1 ? dh - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > dh - 1 ? dh - 1 : y1);
                for (int c = 0; c < 4; c++) {
                    const float c00 = prev[4 * ((size_t)y0 * dw + x0) + c], c10 = prev[4 * ((size_t)y0 * dw + x1) + c];
                    const float c01 = prev[4 * ((size_t)y1 * dw + x0) + c], c11 = prev[4 * ((size_t)y1 * dw + x1) + c];
                    const float top = c00 + tx * (c10 - c00), bot = c01 + tx * (c11 - c01);
                    h[c] = top + ty * (bot - top);
                }
                N = h[3];
            }
            float a = fmaxf(fmaxf(1.0f / (N + 1.0f), 0.1f * conf), reac);                                 /* U7 */
            if (!(N > 0.0f)) a = 1.0f;
            float *o = out + 4 * ((size_t)y * (size_t)dw + (size_t)x);
            for (int c = 0; c < 3; c++) {
                float hc = fminf(fmaxf(h[c], cmin[c]), cmax[c]);                                         /* U6 */
                hc = hc + lk * (h[c] - hc);
                o[c] = hc + a * (cur[c] - hc);
            }
            o[3] = fminf(N + 1.0f, 32.0f);
        }
}
