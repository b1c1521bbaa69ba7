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
//
// MI355X shape.  A guide kernel packs what the edge stops need into one 16-byte record per pixel (normal 3 x f16, valid flag,
// depth f32, depth gradient f32) so a tap costs two loads (8 B colour + 16 B guide) instead of four.  Per (tap, pixel) pair the three
// edge stops collapse into ONE v_exp_f32:
//     w = h * exp2(128 log2(max(0, n.n')) - log2e (|dz| / (gz dist + 1e-8) + |dl| / phi_l)).
// The a-trous kernel can register-block ATROUS_ROWS output rows spaced by the step per lane (8 x 5 fetched taps serve 4 outputs: 10
// taps per pixel instead of 25).  Measured at 1080p (five iterations + guide + variance): 4 rows 0.240 ms (133 VGPRs, 3 waves/SIMD),
// 2 rows 0.223 ms, 1 row 0.210 ms (68 VGPRs, 7 waves/SIMD): the pass is latency-bound and occupancy beats tap reuse, so 1 it is.
// Sky pixels (56 % of the sample frame) are written to both ping-pong images by the variance kernel and skipped by the iterations.
// Round 2 measured the other way to reuse taps as well: steps 1 and 2 with the workgroup's (64 + 4S) x (8 + 4S) input pixels staged in LDS
// (1.6 / 2.3 global loads per output instead of 50, same values into the same arithmetic).  Slower again -- C3 SVGF 0.240 against 0.210 ms,
// C5 0.800 against 0.751 (profiles/r02_experiments/atrous_lds_tiled*, the patch is there too): the staging barrier and 5 instead of 8
// waves per SIMD cost more than the loads they replace, which the L1 serves at its hit rate anyway.  Nor is it the bytes per tap: 12-byte guide
// records (`global_load_dwordx3`, the depth gradient moved to its own array; 20 instead of 24 B per tap) measured 0.212 / 0.753 ms.
#include "kernels.h"
#include "device_math.h"

#pragma clang fp contract(fast)      // image filter, tolerance-tested: let mul+add fuse (the geometry files keep -ffp-contract=off)
namespace {

DEV float lum3(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }

DEV float grad_z(const float *depth, int x, int y, int w, int h) {
    const float z = depth[(size_t)y * w + x];
    const float zx = depth[(size_t)y * w + (x + 1 < w ? x + 1 : x)], zy = depth[(size_t)(y + 1 < h ? y + 1 : y) * w + x];
    return fmaxf(fabsf(zx - z), fabsf(zy - z));
}

struct GuideRec { f3 n; float z, gz; bool valid; };
DEV GuideRec unpack_guide(uint4 g) {
    GuideRec r;
    r.n = mk3(f16_bits_to_f32((uint16_t)(g.x & 0xFFFFu)), f16_bits_to_f32((uint16_t)(g.x >> 16)), f16_bits_to_f32((uint16_t)(g.y & 0xFFFFu)));
    r.valid = (g.y >> 16) != 0u; r.z = __uint_as_float(g.z); r.gz = __uint_as_float(g.w);
    return r;
}
DEV f4 unpack_rgba16f(uint2 v) {
    return mk4(f16_bits_to_f32((uint16_t)(v.x & 0xFFFFu)), f16_bits_to_f32((uint16_t)(v.x >> 16)),
               f16_bits_to_f32((uint16_t)(v.y & 0xFFFFu)), f16_bits_to_f32((uint16_t)(v.y >> 16)));
}
DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
DEV float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
#define LOG2E 1.44269504088896f

// Variance of every pixel.  Pixels with four frames of history take it from the luminance moments (28 B of traffic per pixel).  Younger pixels --
// disocclusions, and every pixel of a mesh that deforms, whose history restarts each frame (C4) -- take the 7 x 7 bilateral estimate: a workgroup with
// such a pixel stages the (32 + 6) x (8 + 6) neighbourhood of its outputs in LDS once, as 16-byte records (normal 3 x f16 + valid flag, depth,
// luminance: the guide record with the luminance in the place of the depth gradient), and the 49 taps of a pixel are ds_read_b128s.  Per tap the
// two edge stops are one v_exp_f32 like in the a-trous kernel:  w = exp2(128 log2(max(0, n.n')) - log2e |dz| / (gz dist + 1e-8)),  the ten distinct
// values of log2e / (gz dist + 1e-8) computed once per pixel.  (Round 2: four scattered loads, expf, powf and sqrtf per tap: 196 us at 1440p on C4.)
#define VAR_TILE_W (32 + 6)
#define VAR_TILE_H (8 + 6)
// young != nullptr: bounce_resolve_kernel has already written the filter input of every pixel but those with fewer than four frames of history, and marked the
// 32-pixel row segments that hold such a pixel: a workgroup without a mark ends at once, the others estimate and store their young pixels only.
__global__ __launch_bounds__(256) void svgf_variance_kernel(const uint2 *__restrict__ color, const float2 *__restrict__ moments, const int32_t *__restrict__ instanceId,
                                                            const uint4 *__restrict__ guide, uint2 *__restrict__ out, uint2 *__restrict__ outSky, int w, int h, int y0, int y1, uint32_t *young) {
    __shared__ uint4 tile[VAR_TILE_W * VAR_TILE_H];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x = blockIdx.x * 32 + tx, y = y0 + blockIdx.y * 8 + ty;
    if (young) {
        bool marked = false;
        if (threadIdx.x < 8 && y0 + (int)blockIdx.y * 8 + (int)threadIdx.x < y1) {
            uint32_t *f = young + (size_t)(y0 + blockIdx.y * 8 + threadIdx.x) * (size_t)((w + 31) / 32) + blockIdx.x;
            marked = *f != 0u;
            if (marked) *f = 0u;                  // (cleared for the next frame)
        }
        if (!__syncthreads_or(marked ? 1 : 0)) return;
    }
    const bool inside = x < w && y < y1;
    const size_t i = inside ? (size_t)y * w + x : 0;
    uint2 cbits = make_uint2(0u, 0u);
    float history = 0.0f, var = 0.0f;
    bool spatial = false, valid = false;
    if (inside) {
        cbits = color[i];
        history = f16_bits_to_f32((uint16_t)(cbits.y >> 16));
        valid = instanceId[i] >= 0;
        if (valid) {
            if (history >= 4.0f) { if (!young) { const float2 m = moments[i]; var = svgf_moment_variance(m.x, m.y); } }
            else spatial = true;
        }
    }
    if (__syncthreads_or(spatial ? 1 : 0)) {         // workgroup-uniform
        const int bx = blockIdx.x * 32 - 3, by = y0 + blockIdx.y * 8 - 3;
        for (int t = threadIdx.x; t < VAR_TILE_W * VAR_TILE_H; t += 256) {
            const int qx = bx + t % VAR_TILE_W, qy = by + t / VAR_TILE_W;
            uint4 rec = make_uint4(0u, 0u, 0u, 0u);                       // outside the frame: not valid
            if (qx >= 0 && qx < w && qy >= 0 && qy < h) {
                const size_t j = (size_t)qy * w + qx;
                const uint4 g = guide[j];
                const f4 cq = unpack_rgba16f(color[j]);
                rec = make_uint4(g.x, g.y, g.z, __float_as_uint(lum3(cq.x, cq.y, cq.z)));
            }
            tile[t] = rec;
        }
        __syncthreads();
        if (spatial) {
            const uint4 *centre = tile + (ty + 3) * VAR_TILE_W + tx + 3;
            const GuideRec p = unpack_guide(*centre);                      // (.gz holds the luminance here; the gradient comes from the guide image)
            const float gz = __uint_as_float(guide[i].w);
            float sw = 0.0f, s1 = 0.0f, s2 = 0.0f;
            // rows rolled, the seven taps of a row unrolled: log2e / (gz |(ax, ay)| + 1e-8) for ax = 0..3 is made per row (fully unrolled the 49 taps
            // hoist their LDS reads and the ten constants into 118 VGPRs -- 4 waves per SIMD for a kernel that streams on most of the frame)
#pragma unroll 1
            for (int dy = -3; dy <= 3; dy++) {
                const float fy2 = (float)(dy * dy);
                float kz[4];
#pragma unroll
                for (int ax = 0; ax <= 3; ax++) kz[ax] = LOG2E * s_rcp(gz * s_sqrt((float)(ax * ax) + fy2) + 1e-8f);
                const uint4 *row = centre + dy * VAR_TILE_W;
#pragma unroll
                for (int dx = -3; dx <= 3; dx++) {
                    const GuideRec q = unpack_guide(row[dx]);
                    const float e = 128.0f * fast_log2(fmaxf(0.0f, dot3(p.n, q.n))) - fabsf(p.z - q.z) * kz[dx < 0 ? -dx : dx];
                    const float wt = q.valid ? fast_exp2(e) : 0.0f;
                    const float l = q.gz;
                    sw += wt; s1 += wt * l; s2 += wt * l * l;
                }
            }
            if (sw > 0.0f) { const float inv = s_rcp(sw), m1 = s1 * inv, m2 = s2 * inv; var = fmaxf(0.0f, m2 - m1 * m1) * (4.0f * s_rcp(fmaxf(history, 1.0f))); }
        }
    }
    if (inside && (!young || spatial)) {
        const uint2 v = make_uint2(cbits.x, (cbits.y & 0xFFFFu) | ((uint32_t)f32_to_f16_bits(var) << 16));
        out[i] = v;
        if (!valid) outSky[i] = v;        // pixels without a surface pass through the filter unchanged: written to the other ping-pong image here, skipped by every a-trous iteration
    }
}

// guide record of every pixel (runs before svgf_variance_kernel, which reads it too)
__global__ __launch_bounds__(256) void svgf_guide_kernel(const int32_t *instanceId, const uint16_t *normal, const float *depth, uint4 *guide, int w, int h, int y0, int y1) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= y1) return;
    const size_t i = (size_t)y * w + x;
    const uint2 n = reinterpret_cast<const uint2 *>(normal)[i];
    uint4 g;
    g.x = n.x; g.y = (n.y & 0xFFFFu) | (instanceId[i] >= 0 ? 0x10000u : 0u);
    g.z = __float_as_uint(depth[i]); g.w = __float_as_uint(grad_z(depth, x, y, w, h));
    guide[i] = g;
}

#define ATROUS_ROWS 1
// ComposePS of one pixel (ComposePS.hlsl:18-37 + the PostProcessPS passthrough; compose_post_kernel<false> of passes.hip, the same operations through the same
// helpers -- none of them contracted, whatever this file's pragma says) with the filtered GI value the last a-trous iteration has just rounded to RGBA16F:
// on frames with the SVGF denoiser the last iteration composes its pixel itself (one launch and one pass over the filtered image less per frame).
DEV void compose_pixel(const SvgfComposeFold &f, size_t i, uint2 filtered) {
#pragma clang fp contract(off)
    const f4 d = load_rgba8(f.diffuse, i);
    f3 result;
    if (d.w > RT_EPSILON) {
        const f3 diffuse = xyz(d);
        const f3 direct = xyz(load_rgba16f(f.filteredDirect, i)), indirect = xyz(unpack_rgba16f(filtered));
        result = diffuse * (direct + indirect);
        result = lerp3(diffuse, result, d.w);
        result = result + xyz(load_rgba16f(f.reflection, i));
        result = result + xyz(load_rgba16f(f.refraction, i));
        result = result + xyz(load_rgba16f(f.transparent, i));
    }
    else result = xyz(d);
    float4 v = make_float4(result.x, result.y, result.z, 1.0f);
    if (f.sppCount > 1) {       // spp_accumulate_kernel of passes.hip, folded in as well (same additions in the same order, one multiplication by 1.0f / count at the end)
        float4 *sum = reinterpret_cast<float4 *>(f.sppSum);
        if (f.sppSub > 0) { const float4 a = sum[i]; v.x = a.x + v.x; v.y = a.y + v.y; v.z = a.z + v.z; v.w = a.w + v.w; }
        if (f.sppSub + 1 < f.sppCount) {
            sum[i] = v;
            reinterpret_cast<float4 *>(f.output)[i] = make_float4(result.x, result.y, result.z, 1.0f);
            return;
        }
        const float inv = 1.0f / (float)f.sppCount;
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        reinterpret_cast<float4 *>(f.output)[i] = v;
        store_rgba8(f.final, i, v.x, v.y, v.z, 1.0f);
        return;
    }
    reinterpret_cast<float4 *>(f.output)[i] = v;
    if (f.writeFinal) store_rgba8(f.final, i, result.x, result.y, result.z, 1.0f);
}

template <bool COMPOSE>
__global__ __launch_bounds__(256, 6) void svgf_atrous_kernel(const uint2 *__restrict__ in, uint2 *__restrict__ out, const uint4 *__restrict__ guide, int w, int h, int step, int y0, int y1, SvgfComposeFold fold) {
    static_assert(!COMPOSE || ATROUS_ROWS == 1, "the folded Compose takes one output pixel per lane");
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int t = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int yb = y0 + (t / step) * (ATROUS_ROWS * step) + (t % step);     // this lane filters rows yb + j*step, j = 0..3, inside [y0, y1)
    if (x >= w || yb >= y1) return;

    f4 c[ATROUS_ROWS]; f3 np[ATROUS_ROWS]; float zp[ATROUS_ROWS], lp[ATROUS_ROWS], kl[ATROUS_ROWS], gzc[ATROUS_ROWS];
    bool live[ATROUS_ROWS];
    float sw[ATROUS_ROWS], sr[ATROUS_ROWS], sg[ATROUS_ROWS], sb[ATROUS_ROWS], sv[ATROUS_ROWS];
    bool any = false;
#pragma unroll
    for (int j = 0; j < ATROUS_ROWS; j++) {
        const int y = yb + j * step;
        live[j] = false; sw[j] = sr[j] = sg[j] = sb[j] = sv[j] = 0.0f;
        c[j] = mk4(0.0f, 0.0f, 0.0f, 0.0f); np[j] = mk3s(0.0f); zp[j] = lp[j] = kl[j] = gzc[j] = 0.0f;
        if (y >= y1) continue;
        const size_t i = (size_t)y * w + x;
        const GuideRec g = unpack_guide(guide[i]);
        if (!g.valid) continue;               // sky: svgf_variance_kernel wrote the pixel to both ping-pong images, nothing to filter or to copy
        c[j] = unpack_rgba16f(in[i]);
        live[j] = true; any = true;
        // 3x3 Gaussian of the variance drives the luminance edge stop
        const float G[3] = { 0.25f, 0.5f, 0.25f };
        float gv = 0.0f;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                int qx = x + dx, qy = y + dy;
                qx = qx < 0 ? 0 : (qx >= w ? w - 1 : qx); qy = qy < 0 ? 0 : (qy >= h ? h - 1 : qy);
                const uint16_t vbits = reinterpret_cast<const uint16_t *>(in)[((size_t)qy * w + qx) * 4 + 3];
                gv += G[dx + 1] * G[dy + 1] * f16_bits_to_f32(vbits);
            }
        const float phiL = 4.0f * s_sqrt(fmaxf(0.0f, gv)) + 1e-6f;
        kl[j] = LOG2E * s_rcp(phiL);
        np[j] = g.n; zp[j] = g.z; lp[j] = lum3(c[j].x, c[j].y, c[j].z);
        gzc[j] = g.gz;
    }
    if (!any) {
        // sky (or a row past the end): nothing to filter.  The folded Compose still owes the pixel: its GI value is what svgf_variance_kernel wrote to both images.
        if (COMPOSE && yb >= fold.oy0 && yb < fold.oy1) compose_pixel(fold, (size_t)yb * w + x, in[(size_t)yb * w + x]);
        return;
    }

    // Tap rows are a rolled loop (8 rows, 5 taps = 10 loads each).  Which of the lane's 4 pixels a row serves (|ky| <= 2) is wave-uniform, so that test is a scalar
    // branch.  Out-of-frame / sky taps are fetched from a clamped address and get weight 0.
    const float fstep = (float)step;
    struct RowTaps { uint4 g[5]; uint2 c[5]; bool rowIn; };
    auto load_row = [&](RowTaps &r, int ry) {
        const int qy = yb + (ry - 2) * step;
        r.rowIn = qy >= 0 && qy < h;
        const size_t rowBase = (size_t)(qy < 0 ? 0 : (qy >= h ? h - 1 : qy)) * w;
#pragma unroll
        for (int kx = -2; kx <= 2; kx++) {
            const int qx = x + kx * step;
            const size_t q = rowBase + (qx < 0 ? 0 : (qx >= w ? w - 1 : qx));
            r.g[kx + 2] = guide[q]; r.c[kx + 2] = in[q];
        }
    };
    auto filter_row = [&](const RowTaps &r, int ry) {
        if (!r.rowIn) return;
        GuideRec gq[5]; f4 cq[5]; float lq[5], hx[5];
#pragma unroll
        for (int kx = -2; kx <= 2; kx++) {
            const int qx = x + kx * step;
            gq[kx + 2] = unpack_guide(r.g[kx + 2]);
            cq[kx + 2] = unpack_rgba16f(r.c[kx + 2]);
            const bool tapOk = qx >= 0 && qx < w && gq[kx + 2].valid;
            const float K[5] = { 1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f };
            hx[kx + 2] = tapOk ? K[kx + 2] : 0.0f;
            lq[kx + 2] = lum3(cq[kx + 2].x, cq[kx + 2].y, cq[kx + 2].z);
        }
#pragma unroll
        for (int j = 0; j < ATROUS_ROWS; j++) {
            const int ky = ry - 2 - j;                               // wave-uniform
            if (ky < -2 || ky > 2) continue;
            const int ay = ky < 0 ? -ky : ky;
            const float hy = ay == 0 ? 3.0f / 8.0f : (ay == 1 ? 1.0f / 4.0f : 1.0f / 16.0f);
            float kzd[3];                                            // log2e / (gz |(ax, ay)| step + 1e-8) for ax = 0, 1, 2
#pragma unroll
            for (int ax = 0; ax <= 2; ax++) kzd[ax] = LOG2E * s_rcp(1.0f * gzc[j] * (s_sqrt((float)(ax * ax + ay * ay)) * fstep) + 1e-8f);
#pragma unroll
            for (int kx = -2; kx <= 2; kx++) {
                const int ax = kx < 0 ? -kx : kx;
                float e = 128.0f * fast_log2(fmaxf(0.0f, dot3(np[j], gq[kx + 2].n))) - (fabsf(zp[j] - gq[kx + 2].z) * kzd[ax] + fabsf(lp[j] - lq[kx + 2]) * kl[j]);
                if (kx == 0) e = ky == 0 ? 0.0f : e;                                     // the centre tap carries the plain kernel weight
                const float wt = (hx[kx + 2] * hy) * fast_exp2(fminf(e, 0.0f));
                const f4 t = cq[kx + 2];
                sw[j] += wt; sr[j] += wt * t.x; sg[j] += wt * t.y; sb[j] += wt * t.z; sv[j] += wt * wt * t.w;
            }
        }
    };
    // (double-buffering the rows costs 50 VGPRs and spills at 3 waves/SIMD: measured slower, so rows are fetched one at a time)
#pragma unroll 1
    for (int ry = 0; ry < ATROUS_ROWS + 4; ry++) {
        RowTaps A;
        load_row(A, ry);
        filter_row(A, ry);
    }
#pragma unroll
    for (int j = 0; j < ATROUS_ROWS; j++) {
        if (!live[j]) continue;
        const float inv = s_rcp(sw[j]);
        uint2 v;
        v.x = (uint32_t)f32_to_f16_bits(sr[j] * inv) | ((uint32_t)f32_to_f16_bits(sg[j] * inv) << 16);
        v.y = (uint32_t)f32_to_f16_bits(sb[j] * inv) | ((uint32_t)f32_to_f16_bits(sv[j] * inv * inv) << 16);
        out[(size_t)(yb + j * step) * w + x] = v;
        if (COMPOSE && yb >= fold.oy0 && yb < fold.oy1) compose_pixel(fold, (size_t)yb * w + x, v);
    }
}

}  // namespace

// The denoiser in two halves.  launch_svgf_inputs: guide records of rows [gy0, gy1), variance (= the a-trous input image, both ping-pong images for sky
// pixels) of rows [vy0, vy1); the 7 x 7 variance estimate reads guide records and colour 3 rows around its rows, the guide's depth gradient one row below.
// launch_svgf_atrous: the five iterations over rows [y0, y1).  An image-tile partition passes its rows + the halo: the iterations reach
// 2 * (1 + 2 + 4 + 8 + 16) = SVGF_ATROUS_HALO_ROWS rows.  Taps may fall outside [y0, y1) but inside the frame: they only reach output rows that are
// themselves farther than the halo from the rows the device owns.  With halo RECOMPUTE (SVGF_HALO_ROWS = 62 + 3 + 1) both halves run on rows + halo;
// with halo EXCHANGE (rt64_host.cpp) the inputs are made for the device's own rows only and the halo rows of both images arrive from the neighbours.
hipError_t launch_svgf_inputs(const ViewImages &I, int cur, int width, int height, int gy0, int gy1, int vy0, int vy1, bool inputByResolve, hipStream_t s) {
    if (gy1 > gy0) {
        dim3 grid((unsigned)(width + 31) / 32, (unsigned)(gy1 - gy0 + 7) / 8);
        hipLaunchKernelGGL(svgf_guide_kernel, grid, dim3(256), 0, s, I.instanceId, I.normal[cur], I.depth[cur], I.svgfGuide, width, height, gy0, gy1);
    }
    if (vy1 > vy0) {
        dim3 grid((unsigned)(width + 31) / 32, (unsigned)(vy1 - vy0 + 7) / 8);
        hipLaunchKernelGGL(svgf_variance_kernel, grid, dim3(256), 0, s, reinterpret_cast<const uint2 *>(I.indirectLight[cur]), reinterpret_cast<const float2 *>(I.moments[cur]), I.instanceId, I.svgfGuide,
                           reinterpret_cast<uint2 *>(I.filteredIndirect[0]), reinterpret_cast<uint2 *>(I.filteredIndirect[1]), width, height, vy0, vy1, inputByResolve ? I.svgfYoung : nullptr);
    }
    return hipGetLastError();
}
// Iterations [first, last) of the five over rows [y0, y1); `fold` (may be nullptr) = Compose of the rows [oy0, oy1) inside iteration 4.
// [oy0, oy1) = the rows whose RESULT is wanted (the device's own rows of an image-tile partition; the whole range otherwise): iteration k only has to cover
// them + what the iterations after it still reach, 2 * (2^(k+1) + ... + 16) = 64 - 2^(k+2) rows on either side -- 60, 56, 48, 32, 0 -- instead of the full
// halo five times (392 halo-row passes per band instead of 620).
hipError_t launch_svgf_atrous(const ViewImages &I, int width, int height, int y0, int y1, int oy0, int oy1, int first, int last, const SvgfComposeFold *fold, hipStream_t s) {
    if (y1 <= y0) return hipSuccess;
    const SvgfComposeFold none = {};
    for (int k = first; k < last; k++) {
        const int step = 1 << k, reach = 64 - (4 << k);
        const int ky0 = y0 > oy0 - reach ? y0 : oy0 - reach, ky1 = y1 < oy1 + reach ? y1 : oy1 + reach, rows = ky1 - ky0;
        if (rows <= 0) continue;
        const unsigned laneRows = (unsigned)((rows + ATROUS_ROWS * step - 1) / (ATROUS_ROWS * step)) * (unsigned)step;     // lanes per column
        dim3 agrid((unsigned)(width + 63) / 64, (laneRows + 3) / 4);
        const uint2 *in = reinterpret_cast<const uint2 *>(I.filteredIndirect[k % 2]); uint2 *out = reinterpret_cast<uint2 *>(I.filteredIndirect[(k % 2) ^ 1]);
        if (k == 4 && fold) hipLaunchKernelGGL(svgf_atrous_kernel<true>, agrid, dim3(256), 0, s, in, out, I.svgfGuide, width, height, step, ky0, ky1, *fold);
        else hipLaunchKernelGGL(svgf_atrous_kernel<false>, agrid, dim3(256), 0, s, in, out, I.svgfGuide, width, height, step, ky0, ky1, none);
    }
    return hipGetLastError();
}
