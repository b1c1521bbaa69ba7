// passes.hip -- the per-frame ray passes of the render path as HIP kernels (wave64, persistent workgroups).
//
// MI355X-side replacement of the DispatchRays / Dispatch / Draw sequence of View::render
// (/root/reference/src/rt64lib/private/rt64_view.cpp:1321-1650) and of the HLSL entry points it runs:
//   primary_trace + primary_shade  <- PrimaryRayGen (shaders/PrimaryRayGen.hlsl:31-198) + surface any-hit
//   direct                         <- DirectRayGen (shaders/DirectRayGen.hlsl:14-65) + shadow any-hit
//   indirect                       <- IndirectRayGen (shaders/IndirectRayGen.hlsl:31-137)
//   refraction / reflection        <- RefractionRayGen.hlsl:19-117 / ReflectionRayGen.hlsl:25-143
//   gaussian                       <- GaussianFilterRGB3x3CS.hlsl:21-82
//   compose_post                   <- ComposePS.hlsl:18-37 + PostProcessPS.hlsl:13-36 (fused: one read of the G-buffer)
// Differences in structure (not in results): primary visibility is a pure traversal kernel that writes a 16-byte hit
// record per pixel, and shading runs as a second, traversal-free kernel over coherent hit records; images keep the
// reference's storage formats so every quantisation point is preserved.
//
// Launch shape: RT_GRID_BLOCKS persistent workgroups of 256 threads walk 16x16-pixel tiles round-robin; a wave owns an
// 8x8 pixel block (coherent rays, coalesced 8-row image stores).  Consecutive workgroup ids land on different XCDs, so
// neighbouring tiles spread over all eight L2s while each tile's BVH nodes stay hot in its own.
#include "kernels.h"
#include "shade.h"
#include "raster_pixel.h"

// This file is compiled twice.  passes_simple.hip defines RT_ASSUME_SIMPLE and includes it: the same ray kernels for frames whose
// textures all have power-of-two sizes and whose instances are all shadow-opaque (rule O2) -- the non-power-of-two addressing and the
// shadow any-hit program are not compiled in, which takes a third of the instructions and most of the register spills out of the
// shading kernels (the sample scene is such a frame).  FrameParams::simpleKernels routes a launch to the `_simple` twin.
#ifdef RT_ASSUME_SIMPLE
#define RT_LAUNCHER(name) name##_simple
#define RT_ROUTE_SIMPLE(call)
#else
#define RT_LAUNCHER(name) name
#define RT_ROUTE_SIMPLE(call) if (P.simpleKernels) return call
#endif
#ifndef RT_ABLATE
#define RT_ABLATE 0              // diagnostic builds (never shipped, tools/exp/r03_bounce_ablate.sh): parts of bounce_trace_plain_kernel stubbed out to price them
#endif

namespace {

struct Pixel { uint32_t x, y; bool valid; };

// Pixel tiles.  A workgroup (4 waves) owns one tile per loop trip; two shapes:
//   SQUARE 16x16, wave = 8x8 pixels   : traversal kernels (coherent rays share BVH nodes)
//   ROWS   32x8,  wave = 32x2 pixels  : shading kernels (a wave's image store covers whole 128/256-byte lines)
// Tile rows owned by this device: strips stripRank, stripRank + stripCount, ... of the 16-row strips in [tileY0, tileY1).
enum TileShape { TILE_SQUARE = 0, TILE_ROWS = 1 };
#ifndef RT_WAVE_BLOCK_W
#define RT_WAVE_BLOCK_W 8        // pixels per row of a wave's block inside a SQUARE tile: 8 (8 x 8, shipped) or 16 (16 x 4: experiment of round 4, tools/exp/r04_wave_block.sh)
#endif
#define RT_WAVE_BLOCK_H (64 / RT_WAVE_BLOCK_W)
template <int SHAPE> struct TileDim { static constexpr int W = SHAPE == TILE_SQUARE ? 16 : 32, H = SHAPE == TILE_SQUARE ? 16 : 8; };

DEV uint32_t owned_strips(PRef P) {
    const uint32_t all = (uint32_t)(P.tileY1 - P.tileY0 + 15) / 16;
    return all > (uint32_t)P.stripRank ? (all - (uint32_t)P.stripRank + (uint32_t)P.stripCount - 1) / (uint32_t)P.stripCount : 0u;
}
template <int SHAPE = TILE_SQUARE> DEV uint32_t tile_count(PRef P) {
    return (uint32_t)((P.width + TileDim<SHAPE>::W - 1) / TileDim<SHAPE>::W) * owned_strips(P) * (16u / TileDim<SHAPE>::H);
}
template <int SHAPE = TILE_SQUARE> DEV Pixel tile_pixel_at(PRef P, uint32_t tile, uint32_t wave, uint32_t lane) {
    constexpr uint32_t TW = TileDim<SHAPE>::W, TH = TileDim<SHAPE>::H, perStrip = 16u / TH;
    const uint32_t tilesX = ((uint32_t)P.width + TW - 1) / TW;
    const uint32_t tx = tile % tilesX, lt = tile / tilesX;
    const uint32_t strip = (lt / perStrip) * (uint32_t)P.stripCount + (uint32_t)P.stripRank;
    Pixel p;
    if (SHAPE == TILE_SQUARE) {
#if RT_WAVE_BLOCK_W == 16
        p.x = tx * TW + (lane & 15); p.y = wave * 4 + (lane >> 4);           // a wave owns 16 x 4 pixels: its stores are whole 64-byte pieces of image rows
#else
        p.x = tx * TW + (wave & 1) * 8 + (lane & 7); p.y = (wave >> 1) * 8 + (lane >> 3);
#endif
    }
    else { p.x = tx * TW + (lane & 31); p.y = wave * 2 + (lane >> 5); }
    p.y += (uint32_t)P.tileY0 + strip * 16 + (lt % perStrip) * TH;
    p.valid = p.x < (uint32_t)P.width && p.y < (uint32_t)P.tileY1;
    return p;
}
template <int SHAPE = TILE_SQUARE> DEV Pixel tile_pixel(PRef P, uint32_t tile) {
    return tile_pixel_at<SHAPE>(P, tile, threadIdx.x >> 6, threadIdx.x & 63);
}

DEV bool row_owned(PRef P, int y) { return (((y - P.tileY0) / 16) % P.stripCount) == P.stripRank; }

// wordsPerLane: uint32 words of the stack array per lane (RT_STACK_LDS, or RT_STACK_LDS_CACHED / 2 for the int16 stacks of the cached kernels)
DEV TraceStack make_stack(PRef P, uint32_t *ldsStack, uint32_t wordsPerLane = RT_STACK_LDS) {
    TraceStack s;
    uint32_t *block = ldsStack + (threadIdx.x >> 6) * wordsPerLane * RT_LANES;       // this wave's [entry][lane] block
    s.lds = (LdsU32Ptr)(block + (threadIdx.x & 63u));
    s.lds16 = (LdsI16Ptr)block + (threadIdx.x & 63u);
    s.spill = (GlobalU32Ptr)(P.traversalStack + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (RT_STACK_SPILL_HEADER + RT_STACK_SPILL) + RT_STACK_SPILL_HEADER);
    s.cache = nullptr; s.ldsEntries = RT_STACK_LDS;
    return s;
}
// this lane's columns of the light-candidate arrays, [wave][slots][lane] (floats, then bytes)
DEV void light_columns(ShadeEnv &env, float *intensities, uint8_t *indices, uint32_t slots) {
    const uint32_t at = (threadIdx.x >> 6) * slots * RT_LANES + (threadIdx.x & 63u);
    env.lightIntensity = intensities + at; env.lightIndex = indices + at;
}

// LDS scene cache, filled once per workgroup (all threads call it; ends with a barrier).  Layout in 16-byte words:
//   [0, 4m)            one 64-byte record per TLAS leaf slot: (M[c], M[4+c], M[8+c], M[12+c]) for c = 0..2 of worldToObject, then
//                      (instance | flags << 8 | word offset of its BLAS nodes << 16, depth bias, tris pointer lo, hi)
//   [4m, 4m + 4 nT)    TLAS nodes, nT = max(m - 1, 1)
//   [cacheNodeOffset)  the BLAS nodes of every instance (offsets assigned by View::update)
// The host enables it (FrameParams::cacheWords != 0) when all of that is at most RT_CACHE_MAX_WORDS: small scenes, like the sample.
// The image is assembled once per table change in HBM (scene_cache_image_kernel: the pointer chase tlasIndex -> instance -> node array
// happens there, three dependent round trips); a workgroup's fill is then one flat copy whose loads are all in flight together.
#ifndef RT_ASSUME_SIMPLE
DEV uint32_t cache_child_ref(uint32_t c) { return (c & RT64_LEAF_BIT) ? (c == RT64_NO_CHILD ? c : (0xFFFF8000u | (c & RT_CACHE_INDEX_MASK))) : c; }
DEV void copy_cache_nodes(const GpuNode *nodes, uint32_t count, u32x4 *dst) {      // word 3 of a node = (left, right, parent, pad): the child references become 16-bit
    typedef const u32x4 __attribute__((address_space(1))) *G4;
    G4 src = reinterpret_cast<G4>(reinterpret_cast<uintptr_t>(nodes));
    for (uint32_t t = threadIdx.x; t < 4u * count; t += blockDim.x) {
        u32x4 w = src[t];
        if ((t & 3u) == 3u) { w.x = cache_child_ref(w.x); w.y = cache_child_ref(w.y); }
        dst[t] = w;
    }
}
// blasOnly: the head (instance records + TLAS nodes) arrived with the table upload, written by the host (View::update: hostCache); only the BLAS node arrays are copied
__global__ __launch_bounds__(RT_BLOCK) void scene_cache_image_kernel(const GpuInstance *instances, const uint32_t *tlasIndex, const GpuNode *tlasNodes, uint32_t m, u32x4 *cache, int blasOnly) {
    typedef u32x4 W4;
    const uint32_t T = blockDim.x, tid = threadIdx.x;

    for (uint32_t k = tid; k < m && !blasOnly; k += T) {
        const uint32_t inst = tlasIndex[k];
        const GpuInstance &in = instances[inst];
        const float *M = in.worldToObject;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            W4 w; w.x = __float_as_uint(M[c]); w.y = __float_as_uint(M[4 + c]); w.z = __float_as_uint(M[8 + c]); w.w = __float_as_uint(M[12 + c]);
            cache[4 * k + c] = w;
        }
        const uint64_t tp = reinterpret_cast<uint64_t>(in.tris);
        W4 info; info.x = inst | ((in.flags & 0xFFu) << 8) | (in.cacheNodeOffset << 16); info.y = __float_as_uint(in.material.depthBias); info.z = (uint32_t)tp; info.w = (uint32_t)(tp >> 32);
        cache[4 * k + 3] = info;
    }
    if (!blasOnly) copy_cache_nodes(tlasNodes, m > 1 ? m - 1 : 1u, cache + 4 * m);
    for (uint32_t k = 0; k < m; k++) {                       // uniform: every thread walks the same instance list
        const GpuInstance &in = instances[tlasIndex[k]];
        copy_cache_nodes(in.nodes, in.triCount > 1 ? in.triCount - 1 : 1u, cache + in.cacheNodeOffset);
    }
}
#endif
DEV void fill_scene_cache(PRef P, u32x4_lds *cache) {
    typedef const u32x4 __attribute__((address_space(1))) *G4;
    G4 src = reinterpret_cast<G4>(reinterpret_cast<uintptr_t>(P.cacheImage));
    const uint32_t words = P.cacheWords;
    for (uint32_t t = threadIdx.x; t < words; t += RT_BLOCK) cache[t] = src[t];
    __syncthreads();
}

DEV void flush_env(PRef P, const ShadeEnv &env, int pass, int rayCounter, uint32_t rays) {
    flush_counts(P, env.cnt, pass);
    if (!P.countTraversal) return;
    unsigned long long a = rays, b = env.shadowRays;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_down(a, d, 64); b += __shfl_down(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { unsigned long long *ctr = P.counters + (size_t)(blockIdx.x % RT_COUNTER_STRIPES) * CTR_COUNT; if (a) atomicAdd(&ctr[rayCounter], a); if (b) atomicAdd(&ctr[CTR_SHADOW], b); }
}

// ---- surface rays ---------------------------------------------------------------------------------------------------------
// Sort key = t - depthBias (WithDistanceBias, Instances.hlsli:17-19); ties keep the first-come hit like the reference's
// strict '<' insertion (rt64_shader.cpp:557).  Hits on instances flagged opaque (rule O1) shorten tmax (R5):
// lim = (t - depthBias) + maxDepthBias.
//   KLIST = false: every instance of the frame is provably opaque -> the list degenerates to the closest hit (registers).
//   KLIST = true : sorted insertion into the per-pixel list in HBM, 16 entries + 1 scratch slot, exactly the any-hit of
//                  rt64_shader.cpp:547-581 (a hit that lands in slot 15 commits tmax; nhits keeps counting).
struct SurfaceHit { float key, t, u, v; uint32_t instance, prim; bool hit; };

template <bool KLIST, bool CACHED = false>
DEV uint32_t trace_surface(PRef P, ShadeEnv &env, IRef I, size_t pixel, f3 o, f3 d, const RayDiff &rayDiff,
                           uint32_t px, uint32_t py, SurfaceHit &best) {
    float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    best.hit = false; best.key = INFINITY;
    uint32_t nhits = 0;
    const size_t stride = (size_t)P.width * (size_t)P.height;
    trace_ray<CACHED>(P, oo, dd, RT_RAY_MIN_DISTANCE, RT_RAY_MAX_DISTANCE, true, env.stk,
              [&](float t, float u, float v, uint32_t instance, uint32_t prim, float &tmax, uint32_t instFlags, float instDepthBias) -> bool {
                  const GpuInstance &in = P.instances[instance];
                  const float key = t - instDepthBias;
                  if (!KLIST) {
                      if (key < best.key) { best.key = key; best.t = t; best.u = u; best.v = v; best.instance = instance; best.prim = prim; best.hit = true; }
                  }
                  else {
                      if (in.cc.optTextureEdge) {            // IgnoreHit() before the hit is stored (rt64_shader.cpp:502-511)
                          HitRecord tmp;
                          if (!surface_anyhit(P, instance, prim, t, u, v, d, rayDiff, px, py, tmp)) return false;
                      }
                      uint32_t hi = nhits < RT64_MAX_HIT_QUERIES ? nhits : RT64_MAX_HIT_QUERIES;
                      while (hi > 0) {
                          const uint4 prev = I.klistA[(size_t)(hi - 1) * stride + pixel];
                          if (!(key < __uint_as_float(prev.x))) break;
                          I.klistA[(size_t)hi * stride + pixel] = prev;
                          I.klistB[(size_t)hi * stride + pixel] = I.klistB[(size_t)(hi - 1) * stride + pixel];
                          hi--;
                      }
                      if (hi < RT64_MAX_HIT_QUERIES) {
                          I.klistA[(size_t)hi * stride + pixel] = make_uint4(__float_as_uint(key), __float_as_uint(u), __float_as_uint(v), prim);
                          I.klistB[(size_t)hi * stride + pixel] = make_uint2(__float_as_uint(t), instance);
                          ++nhits;
                          if (hi == RT64_MAX_HIT_QUERIES - 1 && t < tmax) tmax = t;      // not IgnoreHit(): the hit is committed
                      }
                      if (!(instFlags & GPU_INST_OPAQUE)) return false;
                  }
                  const float lim = key + P.maxDepthBias;
                  if (lim < tmax) tmax = lim;
                  return false;
              }, env.cnt);
    return KLIST ? nhits : (best.hit ? 1u : 0u);
}

// Entry `hit` of the list as a shaded record (the any-hit program runs here, on coherent data, instead of during traversal).
// Reads beyond the 17 allocated slots return an empty record like an out-of-bounds typed UAV load.
// Entry `hit` of the pixel's sorted list as (instance, primitive, t, u, v), without running the any-hit program.
template <bool KLIST>
DEV bool surface_entry(PRef P, IRef I, size_t pixel, uint32_t hit, const SurfaceHit &best, SurfaceHit &e) {
    if (!KLIST) { e = best; return true; }
    if (hit > RT64_MAX_HIT_QUERIES) return false;
    const size_t stride = (size_t)P.width * (size_t)P.height;
    const uint4 a = I.klistA[(size_t)hit * stride + pixel];
    const uint2 b = I.klistB[(size_t)hit * stride + pixel];
    e.instance = b.y; e.prim = a.w; e.t = __uint_as_float(b.x); e.u = __uint_as_float(a.y); e.v = __uint_as_float(a.z); e.hit = true; e.key = __uint_as_float(a.x);
    return true;
}
template <bool KLIST>
DEV bool surface_record(PRef P, IRef I, size_t pixel, uint32_t hit, const SurfaceHit &best, f3 dir,
                        const RayDiff &rayDiff, uint32_t px, uint32_t py, HitRecord &r) {
    if (!KLIST) return surface_anyhit(P, best.instance, best.prim, best.t, best.u, best.v, dir, rayDiff, px, py, r);
    if (hit > RT64_MAX_HIT_QUERIES) return false;
    const size_t stride = (size_t)P.width * (size_t)P.height;
    const uint4 a = I.klistA[(size_t)hit * stride + pixel];
    const uint2 b = I.klistB[(size_t)hit * stride + pixel];
    return surface_anyhit(P, b.y, a.w, __uint_as_float(b.x), __uint_as_float(a.y), __uint_as_float(a.z), dir, rayDiff, px, py, r);
}

#ifndef TRACE_WAVES
#define TRACE_WAVES 4          // waves/SIMD the register allocator must fit for pure-traversal kernels (5 spills, measured no faster)
#endif
// ---- primary visibility --------------------------------------------------------------------------------------------------

template <bool KLIST, bool CACHED = false>
__global__ __launch_bounds__(RT_BLOCK, KLIST ? 2 : TRACE_WAVES) void primary_trace_kernel(FrameParams Pv, ViewImages Iv, int32_t *hitInstance) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = CACHED ? RT_STACK_LDS_CACHED / 2 : RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    extern __shared__ u32x4_lds dynLds[];
    if (CACHED) fill_scene_cache(P, dynLds);
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    env.lightIntensity = nullptr; env.lightIndex = nullptr;              // pure visibility: no light is picked in this kernel
    if (CACHED) env.stk.use_cache(dynLds);
    uint32_t rays = 0;
    const uint32_t tiles = tile_count(P);
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this trip's view of the frame constants and the image table (read where used, never carried across trips)
        Pixel p = tile_pixel(P, tile);
        if (!p.valid) continue;
        f3 o, d; f2 ndc;
        primary_ray(P, p.x, p.y, o, d, ndc);
        const size_t i = (size_t)p.y * (size_t)P.width + p.x;
        RayDiff rayDiff;
        if (KLIST) {    // the texture-edge any-hit samples with the primary ray differentials (PrimaryRayGen.hlsl:55-59)
            f3 cU = mk3(P.cameraU[0], P.cameraU[1], P.cameraU[2]), cV = mk3(P.cameraV[0], P.cameraV[1], P.cameraV[2]), cW = mk3(P.cameraW[0], P.cameraW[1], P.cameraW[2]);
            rayDiff.dOdx = mk3s(0.0f); rayDiff.dOdy = mk3s(0.0f);
            compute_ray_diffs((cU * ndc.x + cV * ndc.y) + cW, cU, cV, P.resolution[2], P.resolution[3], rayDiff.dDdx, rayDiff.dDdy);
        }
        SurfaceHit h;
        const uint32_t nhits = trace_surface<KLIST, CACHED>(P, env, I, i, o, d, rayDiff, p.x, p.y, h);
        rays++;
        uint4 rec;
        if (KLIST) {
            I.klistCount[i] = nhits;
            if (nhits) {
                const uint4 a = I.klistA[i]; const uint2 b = I.klistB[i];
                rec = make_uint4(b.x, a.y, a.z, a.w); hitInstance[i] = (int32_t)b.y;
            }
            else { rec.x = rec.y = rec.z = rec.w = 0xFFFFFFFFu; hitInstance[i] = -1; }
        }
        else if (h.hit) { rec.x = __float_as_uint(h.t); rec.y = __float_as_uint(h.u); rec.z = __float_as_uint(h.v); rec.w = h.prim; hitInstance[i] = (int32_t)h.instance; }
        else { rec.x = rec.y = rec.z = rec.w = 0xFFFFFFFFu; hitInstance[i] = -1; }
        reinterpret_cast<uint4 *>(I.primaryHit)[i] = rec;
    }
    flush_env(P, env, PASS_PRIMARY_TRACE, CTR_PRIMARY, rays);
}

// ---- PrimaryRayGen resolve + G-buffer -------------------------------------------------------------------------------------

// TRANSPARENT_LIGHT: some instance is not provably opaque, so the 'transparent geometry that needs lighting' path of
// PrimaryRayGen.hlsl:136-148 (one random light + shadow ray from inside the resolve loop) can be reached.
#ifndef SHADE_TILE
#define SHADE_TILE TILE_ROWS
#endif
#ifndef DIRECT_TILE
#define DIRECT_TILE TILE_SQUARE
#endif
#ifndef SHADE_WAVES
#define SHADE_WAVES 3
#endif
#ifndef DIRECT_WAVES
#define DIRECT_WAVES 3
#endif
// Without SLP pairs (csrc/Makefile) these two sit at 150 / 138 VGPRs: compiled for 4 waves per SIMD (128 VGPRs, 68 / 16 bytes of scratch per lane) they measure C5 2.89 -> 2.81 ms,
// C3 / C4 unchanged (tools/exp/r04_waves_noslp.sh; the frame kernel at 4 waves: C2 0.154 -> 0.164, the two-phase bounce walk at 6: no change).
#ifndef HIT_WAVES
#define HIT_WAVES 4                   // bounce_hit_kernel walking from the LDS scene cache (the uncached form keeps DIRECT_WAVES: its 46 KB of LDS bound it to 3 anyway, and asked for 4 the compiler settles on 2)
#endif
#ifndef REFLECT_WAVES
#define REFLECT_WAVES 4               // reflection_kernel walking from the LDS scene cache
#endif
// FULL = false ("lean" frame): no pass of this frame consumes the view direction, reflection / refraction / transparent
// accumulators, motion vectors or upscaler masks, so they are not written (42 of 94 bytes per pixel); RT64_ReadbackDevice
// re-runs the FULL variant on demand (View::materialise in rt64_host.cpp).
// What PrimaryRayGen resolves for one pixel (the values its image stores take).
struct PrimaryResolve {
    f3 position, normal, specular, transparent;
    f4 color;
    float flowX, flowY, reactiveMask, lockMask, depth, reflA, refrA;
    int instanceId;
};

// PrimaryRayGen.hlsl:47-196 for one pixel whose visibility is already known (`best` / the k-buffer): the resolve loop over the
// sorted hits + the background term.  Shared by primary_shade_kernel (hit records from HBM) and lean_frame_kernel (hit in registers).
template <bool TRANSPARENT_LIGHT, bool KLIST, bool FULL>
DEV void resolve_primary(PRef P, IRef I, ShadeEnv &env, uint32_t px, uint32_t py, size_t i, f3 rayOrigin, f3 rayDirection, f2 d,
                         const SurfaceHit &best, uint32_t nhits, PrimaryResolve &R) {
    f3 cU = mk3(P.cameraU[0], P.cameraU[1], P.cameraU[2]), cV = mk3(P.cameraV[0], P.cameraV[1], P.cameraV[2]), cW = mk3(P.cameraW[0], P.cameraW[1], P.cameraW[2]);
    f3 nonNormRayDir = (cU * d.x + cV * d.y) + cW;
    float reflA = 0.0f, refrA = 0.0f;

    f2 screenUV; screenUV.x = ((float)px + P.pixelJitter[0]) / (float)P.width; screenUV.y = ((float)py + P.pixelJitter[1]) / (float)P.height;
    // The background / sky colour only enters through `bgColor * resColor.a` after the resolve loop: it is fetched there, and
    // only by pixels that are not fully covered (PrimaryRayGen.hlsl:47-53 samples up front; same values, fewer fetches).
    f2 prevBgPos, curBgPos; prevBgPos.x = prevBgPos.y = curBgPos.x = curBgPos.y = 0.0f;
    if (FULL) {
        f3 bgPosition = rayOrigin + rayDirection * RT_RAY_MAX_DISTANCE;
        prevBgPos = world_to_screen(cmat(P.prevViewProj).m, bgPosition); curBgPos = world_to_screen(cmat(P.viewProj).m, bgPosition);
    }

    RayDiff rayDiff;
    rayDiff.dOdx = mk3s(0.0f); rayDiff.dOdy = mk3s(0.0f); rayDiff.dDdx = mk3s(0.0f); rayDiff.dDdy = mk3s(0.0f);

    f3 resPosition = mk3s(0.0f), resNormal = -rayDirection, resSpecular = mk3s(0.0f), resTransparent = mk3s(0.0f), resTransparentLight = mk3s(0.0f);
    bool resTransparentLightComputed = false;
    f4 resColor = mk4(0, 0, 0, 1);
    float resFlowX = (curBgPos.x - prevBgPos.x) * (float)P.width, resFlowY = (curBgPos.y - prevBgPos.y) * (float)P.height;
    float resReactiveMask = 0.0f, resLockMask = 0.0f, resDepth = 1.0f;
    int resInstanceId = -1;
    const f3 ambient = mk3(P.ambientBaseColor[0], P.ambientBaseColor[1], P.ambientBaseColor[2]) + mk3(P.ambientNoGIColor[0], P.ambientNoGIColor[1], P.ambientNoGIColor[2]);

    if (nhits) compute_ray_diffs(nonNormRayDir, cU, cV, P.resolution[2], P.resolution[3], rayDiff.dDdx, rayDiff.dDdy);
    for (uint32_t hit = 0; hit < nhits; hit++) {
        SurfaceHit e;
        if (!surface_entry<KLIST>(P, I, i, hit, best, e)) continue;
        bool covered = false;
        // one pass per distinct instance among the wave's lanes: the any-hit program and the material run on scalar registers
        waterfall(e.instance, [&](uint32_t instanceId) {
        const InstView iv = inst_view(P, instanceId);
        HitRecord r;
        if (!surface_anyhit_view(P, iv, instanceId, e.prim, e.t, e.u, e.v, rayDirection, rayDiff, px, py, r)) return;
        f4 hitColor = r.color;
        float alphaContrib = resColor.w * hitColor.w;
        if (alphaContrib >= RT_EPSILON) {
            const RT64_MATERIAL &m = iv.material;
            resLockMask += m.lockMask * alphaContrib;
            bool usesLighting = m.lightGroupMaskBits > 0;
            bool applyLighting = usesLighting && (hitColor.w > RT_APPLY_LIGHTS_MINIMUM_ALPHA);
            f3 vertexPosition = rayOrigin + rayDirection * (r.dist + m.depthBias);
            f3 vertexNormal = r.normal;
            f3 specular = ld_v3(m.specularColor) * r.specular;
            bool storeHit = false;
            if (m.fogEnabled) {
                f4 fog = fog_from_camera(P, m, vertexPosition);
                resTransparent = resTransparent + xyz(fog) * (fog.w * alphaContrib);
                alphaContrib *= (1.0f - fog.w);
            }
            if (m.reflectionFactor > RT_EPSILON) {
                float fresnelAmount = fresnel_reflect_amount(vertexNormal, rayDirection, m.reflectionFactor, m.reflectionFresnelFactor);
                float reflectAmount = fresnelAmount * alphaContrib;
                reflA = reflectAmount;
                alphaContrib *= (1.0f - fresnelAmount);
                storeHit = true;
                resLockMask += reflectAmount;
            }
            f3 resColorAdd = xyz(hitColor) * alphaContrib;
            if (applyLighting) {
                storeHit = true;
                resColor.x += resColorAdd.x; resColor.y += resColorAdd.y; resColor.z += resColorAdd.z;
            }
            else if (TRANSPARENT_LIGHT && usesLighting) {
                if (!resTransparentLightComputed) {
                    resTransparentLight = compute_lights_random(P, env, px, py, rayDirection, instanceId, vertexPosition, vertexNormal, specular, 1, true);
                    resTransparentLightComputed = true;
                }
                resTransparent = resTransparent + resColorAdd * ((ambient + ld_v3(m.selfLight)) + resTransparentLight);
            }
            else resTransparent = resTransparent + resColorAdd * (ambient + ld_v3(m.selfLight));
            resColor.w *= (1.0f - hitColor.w);
            if (m.refractionFactor > RT_EPSILON) { storeHit = true; refrA = resColor.w; resColor.w = 0.0f; }
            if (storeHit && resInstanceId < 0) {
                f2 prevPos, curPos; prevPos.x = prevPos.y = curPos.x = curPos.y = 0.0f;
                if (FULL) { prevPos = world_to_screen(cmat(P.prevViewProj).m, vertexPosition - r.flow); curPos = world_to_screen(cmat(P.viewProj).m, vertexPosition); }
                resPosition = vertexPosition; resNormal = vertexNormal; resSpecular = specular; resInstanceId = (int)instanceId;
                resFlowX = (curPos.x - prevPos.x) * (float)P.width; resFlowY = (curPos.y - prevPos.y) * (float)P.height;
                if (FULL) {
                    f4 projPos = mul4(cmat(P.viewProj).m, mk4(vertexPosition.x, vertexPosition.y, vertexPosition.z, 1.0f));
                    resDepth = s_div(projPos.z, projPos.w);
                }
            }
        }
        if (resColor.w <= RT_EPSILON) covered = true;
        });
        if (covered) break;
    }
    resReactiveMask += fmaxf(resTransparent.x, fmaxf(resTransparent.y, resTransparent.z));
    if (resColor.w != 0.0f) {
        const f3 bgColor = sky_over_background_2d(P, screenUV);
        resColor.x += bgColor.x * resColor.w; resColor.y += bgColor.y * resColor.w; resColor.z += bgColor.z * resColor.w;
    }
    resColor.w = 1.0f - resColor.w;
    R.position = resPosition; R.normal = resNormal; R.specular = resSpecular; R.transparent = resTransparent; R.color = resColor;
    R.flowX = resFlowX; R.flowY = resFlowY; R.reactiveMask = resReactiveMask; R.lockMask = resLockMask; R.depth = resDepth; R.reflA = reflA; R.refrA = refrA;
    R.instanceId = resInstanceId;
}

// The image stores of PrimaryRayGen for one pixel (FULL: the reference's whole G-buffer; lean: what DirectRayGen + Compose read).
template <bool FULL>
DEV void store_primary(PRef P, IRef I, size_t i, int cur, f3 rayDirection, const PrimaryResolve &R) {
    if (FULL) {
        store_rgba16f(I.viewDirection, i, rayDirection.x, rayDirection.y, rayDirection.z, 0.0f);
        store_rgba16f(I.reflection, i, 0.0f, 0.0f, 0.0f, R.reflA);
        store_rgba16f(I.refraction, i, 0.0f, 0.0f, 0.0f, R.refrA);
    }
    // Lean frame: DirectRayGen reads position / normal / specular only where a surface was hit, so miss pixels skip those stores.
    if (FULL || R.instanceId >= 0) {
        reinterpret_cast<float4 *>(I.shadingPosition)[i] = make_float4(R.position.x, R.position.y, R.position.z, 0.0f);
        store_rgba16f(I.shadingNormal, i, R.normal.x, R.normal.y, R.normal.z, 0.0f);
        store_rgba16f(I.shadingSpecular, i, R.specular.x, R.specular.y, R.specular.z, 0.0f);
    }
    store_rgba8(I.diffuse, i, R.color.x, R.color.y, R.color.z, R.color.w);
    I.instanceId[i] = R.instanceId;
    if (FULL) {
        I.firstInstanceId[i] = R.instanceId;                    // CopyResource(rtFirstInstanceId, rtInstanceId), rt64_view.cpp:1383
        store_rgba16f(I.transparent, i, R.transparent.x, R.transparent.y, R.transparent.z, 1.0f);
        reinterpret_cast<uint32_t *>(I.flow)[i] = (uint32_t)f32_to_f16_bits(-R.flowX) | ((uint32_t)f32_to_f16_bits(R.flowY) << 16);
        I.reactiveMask[i] = to_unorm8(fminf(R.reactiveMask, 0.9f));
        I.lockMask[i] = to_unorm8(P.binaryLockMask ? (R.lockMask >= 0.5f ? 1.0f : 0.0f) : fminf(R.lockMask, 1.0f));
        // history guides of the temporal / SVGF passes: no consumer on a lean frame
        store_rgba16f(I.normal[cur], i, R.normal.x, R.normal.y, R.normal.z, 0.0f);
        I.depth[cur][i] = R.depth;
    }
}

template <bool TRANSPARENT_LIGHT, bool KLIST, bool FULL>
__global__ __launch_bounds__(RT_BLOCK, SHADE_WAVES) void primary_shade_kernel(FrameParams Pv, ViewImages Iv, const int32_t *hitInstance, int cur) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    __shared__ float ldsLightIntensity[(RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    __shared__ uint8_t ldsLightIndex[(RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    light_columns(env, ldsLightIntensity, ldsLightIndex, RT64_MAX_LIGHTS + 1);
    const uint32_t tiles = tile_count<SHADE_TILE>(P);
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this trip's view of the frame constants and the image table (read where used, never carried across trips)
        Pixel p = tile_pixel<SHADE_TILE>(P, tile);
        if (!p.valid) continue;
        const uint32_t px = p.x, py = p.y;
        const size_t i = (size_t)py * (size_t)P.width + px;
        f3 rayOrigin, rayDirection; f2 d;
        primary_ray(P, px, py, rayOrigin, rayDirection, d);
        const uint4 hrec = reinterpret_cast<const uint4 *>(I.primaryHit)[i];
        const int hInst = hitInstance[i];
        SurfaceHit best;
        best.hit = hInst >= 0; best.instance = (uint32_t)hInst; best.prim = hrec.w;
        best.t = __uint_as_float(hrec.x); best.u = __uint_as_float(hrec.y); best.v = __uint_as_float(hrec.z);
        const uint32_t nhits = KLIST ? I.klistCount[i] : (hInst >= 0 ? 1u : 0u);
        PrimaryResolve R;
        resolve_primary<TRANSPARENT_LIGHT, KLIST, FULL>(P, I, env, px, py, i, rayOrigin, rayDirection, d, best, nhits, R);

        store_primary<FULL>(P, I, i, cur, rayDirection, R);
    }
    flush_env(P, env, PASS_PRIMARY_SHADE, CTR_PRIMARY, 0);
}

// ---- DirectRayGen ----------------------------------------------------------------------------------------------------------

DEV float history_weight(PRef P, IRef I, size_t i, uint32_t px, uint32_t py, f3 normal, int cur, long &prevIndex) {
    // DirectRayGen.hlsl:31-45 / IndirectRayGen.hlsl:43-56
    uint32_t fl = reinterpret_cast<const uint32_t *>(I.flow)[i];
    float fx = f16_bits_to_f32((uint16_t)(fl & 0xFFFFu)), fy = f16_bits_to_f32((uint16_t)(fl >> 16));
    int ix = (int)((float)px + 0.5f + fx), iy = (int)((float)py + 0.5f + fy);
    const int prev = cur ^ 1;
    float prevDepth = 0.0f; f3 prevNormal = mk3s(0.0f);
    prevIndex = -1;
    if (ix >= 0 && iy >= 0 && ix < P.width && iy < P.height) {       // out-of-bounds image loads return 0
        size_t j = (size_t)iy * (size_t)P.width + (size_t)ix;
        prevDepth = I.depth[prev][j]; prevNormal = xyz(load_rgba16f(I.normal[prev], j));
        prevIndex = (long)j;
    }
    float weightDepth = fabsf(I.depth[cur][i] - prevDepth) / 0.01f;
    float weightNormal = s_pow(fmaxf(0.0f, dot3(prevNormal, normal)), 128.0f);
    return expf(-weightDepth) * weightNormal;
}

// ComposePS + PostProcessPS of a LEAN frame for one pixel (same arithmetic as compose_post_kernel<true>): everything it reads is
// the pixel's own diffuse colour and direct light, so DirectRayGen's kernel finishes the pixel instead of a separate launch
// (one kernel boundary less per frame: ~11 us of kernel + the inter-kernel gap and cache refill).
DEV f3 compose_lean_value(PRef P, f4 d, f3 directStored) {
    f3 result;
    if (d.w > RT_EPSILON) {
        f3 diffuse = xyz(d);
        f3 indirect = mk3(q_f16(P.ambientBaseColor[0] + P.ambientNoGIColor[0]), q_f16(P.ambientBaseColor[1] + P.ambientNoGIColor[1]), q_f16(P.ambientBaseColor[2] + P.ambientNoGIColor[2]));
        result = diffuse * (directStored + indirect);
        result = lerp3(diffuse, result, d.w);
    }
    else result = xyz(d);
    return result;
}
DEV void compose_lean_pixel(PRef P, IRef I, size_t i, f3 directStored) {
    f4 d = load_rgba8(I.diffuse, i);
    f3 result;
    if (d.w > RT_EPSILON) {
        f3 diffuse = xyz(d);
        f3 indirect = mk3(q_f16(P.ambientBaseColor[0] + P.ambientNoGIColor[0]), q_f16(P.ambientBaseColor[1] + P.ambientNoGIColor[1]), q_f16(P.ambientBaseColor[2] + P.ambientNoGIColor[2]));
        result = diffuse * (directStored + indirect);
        result = lerp3(diffuse, result, d.w);
    }
    else result = xyz(d);
    reinterpret_cast<float4 *>(I.output)[i] = make_float4(result.x, result.y, result.z, 1.0f);
    if (!P.separatePost) store_rgba8(I.final, i, result.x, result.y, result.z, 1.0f);
}

// CACHED variants of the ray kernels keep the scene cache and the light-selection columns in dynamic LDS:
//   [scene cache: P.cacheWords x 16 B][light intensities: slots x RT_BLOCK floats][light indices: slots x RT_BLOCK bytes], slots = min(lights, 16) + 1
DEV uint32_t light_slots(PRef P) { return (P.lightCount < RT64_MAX_LIGHTS ? P.lightCount : (uint32_t)RT64_MAX_LIGHTS) + 1u; }
DEV void cached_env(PRef P, ShadeEnv &env, u32x4_lds *dynLds) {
    fill_scene_cache(P, dynLds);
    env.stk.use_cache(dynLds);
    float *li = reinterpret_cast<float *>(dynLds + P.cacheWords);
    light_columns(env, li, reinterpret_cast<uint8_t *>(li + light_slots(P) * blockDim.x), light_slots(P));
}

// DirectRayGen.hlsl:47-58 for one lit pixel: sampled lights + self light + eye light (before the temporal accumulation).
template <bool CACHED>
DEV f3 direct_light_pixel(PRef P, ShadeEnv &env, uint32_t px, uint32_t py, f3 rayDirection, int instanceId, f3 position, f3 normal, f3 specular) {
    f3 selfLight; float specularExponent;
    waterfall((uint32_t)instanceId, [&](uint32_t k) { const RT64_MATERIAL mk = load_const(&P.instances[k].material); selfLight = ld_v3(mk.selfLight); specularExponent = mk.specularExponent; });
    f3 resDirect = compute_lights_random<CACHED>(P, env, px, py, rayDirection, (uint32_t)instanceId, position, normal, specular, P.maxLights, true);
    resDirect = resDirect + selfLight;
    float eyeLambert = fmaxf(dot3(normal, -rayDirection), 0.0f);
    f3 eyeReflected = reflect3(rayDirection, normal);
    float eyeSpec = s_pow(fmaxf(saturatef(dot3(eyeReflected, -rayDirection)), 0.0f), specularExponent);
    f3 eyeD = mk3(P.eyeLightDiffuseColor[0], P.eyeLightDiffuseColor[1], P.eyeLightDiffuseColor[2]), eyeS = mk3(P.eyeLightSpecularColor[0], P.eyeLightSpecularColor[1], P.eyeLightSpecularColor[2]);
    return resDirect + (eyeD * eyeLambert + eyeS * (specular * eyeSpec));
}

template <bool FULL, bool CACHED = false>
__global__ __launch_bounds__(RT_BLOCK, DIRECT_WAVES) void direct_kernel(FrameParams Pv, ViewImages Iv, int cur) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = CACHED ? RT_STACK_LDS_CACHED / 2 : RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    __shared__ float ldsLightIntensity[CACHED ? 1 : (RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    __shared__ uint8_t ldsLightIndex[CACHED ? 1 : (RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    extern __shared__ u32x4_lds dynLds[];
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    light_columns(env, ldsLightIntensity, ldsLightIndex, RT64_MAX_LIGHTS + 1);
    if (CACHED) cached_env(P, env, dynLds);
    const uint32_t tiles = tile_count<DIRECT_TILE>(P);
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this trip's view of the frame constants and the image table (read where used, never carried across trips)
        Pixel p = tile_pixel<DIRECT_TILE>(P, tile);
        if (!p.valid) continue;
        const uint32_t px = p.x, py = p.y;
        const size_t i = (size_t)py * (size_t)P.width + px;
        const int instanceId = I.instanceId[i];
        if (instanceId < 0) {
            store_rgba16f(I.directLight[cur], i, 1.0f, 1.0f, 1.0f, 0.0f);
            if (FULL) store_rgba16f(I.filteredDirect[1], i, 1.0f, 1.0f, 1.0f, 0.0f);
            else compose_lean_pixel(P, I, i, mk3s(1.0f));
            continue;
        }
        f3 o, rayDirection; f2 ndc;
        primary_ray(P, px, py, o, rayDirection, ndc);
        const float4 pos4 = reinterpret_cast<const float4 *>(I.shadingPosition)[i];
        f3 position = mk3(pos4.x, pos4.y, pos4.z), normal = xyz(load_rgba16f(I.shadingNormal, i)), specular = xyz(load_rgba16f(I.shadingSpecular, i));
        f3 newDirect = mk3s(0.0f); float historyLength = 0.0f;
        if (P.diReproject) {
            long j; float w = history_weight(P, I, i, px, py, normal, cur, j);
            f4 prevAccum = j >= 0 ? load_rgba16f(I.directLight[cur ^ 1], (size_t)j) : mk4(0, 0, 0, 0);
            newDirect = xyz(prevAccum); historyLength = prevAccum.w * w;
        }
        const f3 resDirect = direct_light_pixel<CACHED>(P, env, px, py, rayDirection, instanceId, position, normal, specular);
        historyLength = fminf(historyLength + 1.0f, 64.0f);
        newDirect = lerp3(newDirect, resDirect, s_rcp(historyLength));
        store_rgba16f(I.directLight[cur], i, newDirect.x, newDirect.y, newDirect.z, historyLength);
        if (FULL) store_rgba16f(I.filteredDirect[1], i, newDirect.x, newDirect.y, newDirect.z, historyLength);
        else compose_lean_pixel(P, I, i, mk3(q_f16(newDirect.x), q_f16(newDirect.y), q_f16(newDirect.z)));      // Compose reads the RGBA16F value
    }
    flush_env(P, env, PASS_DIRECT, CTR_PRIMARY, 0);
}

// ---- lean frame in one kernel ----------------------------------------------------------------------------------------------
// A lean frame (every instance provably opaque, no fog / reflection / refraction / GI / motion blur) is pixel-local from the primary
// ray to the back buffer, so one kernel carries each pixel through PrimaryRayGen (visibility + resolve), DirectRayGen and Compose /
// PostProcess with the intermediate G-buffer values in registers.  Every value that the separate kernels pass through an image is
// rounded here exactly as that image's format rounds it (RGBA16F normal / specular / direct light, RGBA8 diffuse), so the back buffer
// is bit-identical to the three-kernel path.  It still stores what View::materialise needs to rebuild the full G-buffer on demand:
// the hit record (16 + 4 B) and the direct-light accumulation (8 B); rtOutput is written only when PostProcess runs separately.
// FULL = true is the same kernel for a frame that is NOT lean but whose instances are all provably opaque (GI, denoiser, reflection,
// refraction, fog, motion blur downstream): it writes the reference's whole G-buffer like primary_shade_kernel<.., FULL> and
// DirectRayGen's two images like direct_kernel<true>, and leaves Compose to its own pass.  ownedY0 / ownedY1: the rows DirectRayGen
// covers (the frame parameters may include a denoiser halo above and below them, which only the G-buffer part renders).
#ifndef LEAN_WAVES
#define LEAN_WAVES 3          // waves per SIMD of the one-kernel frame: 3 (168 VGPRs) measured 12 % faster than 2 (193 VGPRs, no spills); 4 (128 VGPRs, 62 spilled dwords in the simple build) measured 30 % slower
#endif
// BLOCK = 256: one 16 x 16 tile per workgroup trip, a wave owns an 8 x 8 quadrant (scenes with the LDS scene cache: its fill is shared by four waves).
// BLOCK = 64 : the per-wave form for scenes that walk from HBM / L2 -- one 8 x 8 wave-tile per workgroup trip.  Nothing is shared between the waves of
//              such a frame, and with one wave per workgroup every wave's registers AND its LDS are free the moment it ends: a tile whose four quadrants
//              cost very differently (the silhouette of a dense mesh) no longer holds three finished waves' resources until its slowest wave is done,
//              and the dispatcher balances 4 x as many, 4 x smaller jobs.  Wave-tile -> workgroup map (wave_tile_of): the four quadrants of a tile go to
//              workgroups 8 apart -- the same XCD under round-robin placement, so a tile's rays still share one L2.
DEV bool wave_tile_of(uint32_t seq, uint32_t tiles, uint32_t &tileSeq, uint32_t &quadrant) {
    const uint32_t xcd = seq & 7u, j = seq >> 3;
    quadrant = j & 3u; tileSeq = (j >> 2) * 8u + xcd;
    return tileSeq < tiles;
}
template <bool CACHED, bool FULL, int WAVES, int BLOCK = RT_BLOCK>
__global__ __launch_bounds__(BLOCK, WAVES) void lean_frame_kernel(FrameParams Pv, ViewImages Iv, int32_t *hitInstance, int cur, int ownedY0, int ownedY1) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = CACHED ? RT_STACK_LDS_CACHED / 2 : RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * BLOCK];
    // dynamic LDS: [scene cache (CACHED)][light candidates: intensities, indices -- sized by the frame's light count]
    extern __shared__ u32x4_lds dynLds[];
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    if (CACHED) cached_env(P, env, dynLds);
    else { float *li = reinterpret_cast<float *>(dynLds); light_columns(env, li, reinterpret_cast<uint8_t *>(li + light_slots(P) * BLOCK), light_slots(P)); }
    uint32_t rays = 0;
    TraceCounts primaryCnt = TraceCounts();
    const uint32_t tiles = tile_count(P);
    const uint32_t trips = BLOCK == RT_BLOCK ? tiles : ((tiles + 7u) / 8u) * 32u;
    if (P.tileTiming && (threadIdx.x & 63u) == 0u) {      // profiling aid (device option tile_timing): two records per wave, written where they are taken so that nothing stays live across the frame
        uint32_t hwId; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwId));
        P.tileTiming[2 * ((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6))] = make_uint4((uint32_t)__builtin_amdgcn_s_memrealtime(), (uint32_t)__builtin_amdgcn_s_memtime(), hwId, 1u);
    }
    // (one-wave workgroups, device option tile_order) what a trip cost -- the most node + triangle visits any of its lanes made -- goes to its tile's entry for the next
    // frame's order: lanes leave a trip at different places, so each notes its visits with an LDS max where it leaves (note_cost) and the wave, converged again at the
    // top of the next trip and behind the loop, hands the trip's figure to the tile (flush_cost)
    __shared__ uint32_t ldsTripCost;
    uint32_t tileOfTrip = 0xFFFFFFFFu, visitsBefore = 0;
    const bool costed = BLOCK != RT_BLOCK && P.tileCost != nullptr;
    if (costed && threadIdx.x == 0) ldsTripCost = 0;
    auto note_cost = [&]() { if (costed) atomicMax(&ldsTripCost, env.cnt.nodes + env.cnt.tris - visitsBefore); };
    auto flush_cost = [&](PRef P) {
        if (!costed || tileOfTrip == 0xFFFFFFFFu) return;
        __syncthreads();             // (one wave: orders the lanes' LDS notes before lane 0 reads them)
        if (threadIdx.x == 0) { atomicMax(&P.tileCost[tileOfTrip], ldsTripCost); ldsTripCost = 0; }
        tileOfTrip = 0xFFFFFFFFu;
    };
    for (uint32_t trip = blockIdx.x; trip < trips; trip += gridDim.x) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this tile's view of the frame constants and the image table: read where used, never carried across tiles
        // tiles are walked from the bottom of the frame up: the expensive ones (geometry) start first and the cheap ones (sky, at the
        // top of a typical frame) fill the tail of the launch
        Pixel p;
        if (BLOCK == RT_BLOCK) p = tile_pixel(P, tiles - 1u - trip);
        else {
            uint32_t tileSlot, quadrant;
            flush_cost(P);
            if (!wave_tile_of(trip, tiles, tileSlot, quadrant)) continue;
            // longest-first (device option tile_order): slot -> tile through last frame's cost order; the four quadrants of a tile keep their XCD
            tileOfTrip = P.tileOrder ? P.tileOrder[tileSlot] : tileSlot;
            p = tile_pixel_at(P, tiles - 1u - tileOfTrip, quadrant, threadIdx.x & 63u);
            visitsBefore = env.cnt.nodes + env.cnt.tris;
        }
        if (!p.valid) continue;
        const uint32_t px = p.x, py = p.y;
        const size_t i = (size_t)py * (size_t)P.width + px;
        f3 o, rayDirection; f2 ndc;
        primary_ray(P, px, py, o, rayDirection, ndc);
        RayDiff noDiff; noDiff.dOdx = noDiff.dOdy = noDiff.dDdx = noDiff.dDdy = mk3s(0.0f);
        SurfaceHit h;
        const TraceCounts before = env.cnt;
        const uint32_t nhits = trace_surface<false, CACHED>(P, env, I, i, o, rayDirection, noDiff, px, py, h);
        primaryCnt.nodes += env.cnt.nodes - before.nodes; primaryCnt.tris += env.cnt.tris - before.tris;
        rays++;
        // A lean frame keeps nothing but the back buffer: hitInstance == nullptr says so, and a reader of any other image gets it from
        // View::materialise, which runs the FULL variant of this kernel on the kept frame parameters (same rays, same arithmetic).
        const bool keepRecords = FULL || hitInstance != nullptr;
        if (keepRecords) {
            uint4 rec;
            if (h.hit) { rec.x = __float_as_uint(h.t); rec.y = __float_as_uint(h.u); rec.z = __float_as_uint(h.v); rec.w = h.prim; hitInstance[i] = (int32_t)h.instance; }
            else { rec.x = rec.y = rec.z = rec.w = 0xFFFFFFFFu; hitInstance[i] = -1; }
            reinterpret_cast<uint4 *>(I.primaryHit)[i] = rec;
        }

        PrimaryResolve R;
        resolve_primary<false, false, FULL>(P, I, env, px, py, i, o, rayDirection, ndc, h, nhits, R);
        if (FULL) {
            store_primary<true>(P, I, i, cur, rayDirection, R);
            if ((int)py < ownedY0 || (int)py >= ownedY1) { note_cost(); continue; }          // halo row: G-buffer only
        }
        const f4 diffuse = mk4(q_unorm8(R.color.x), q_unorm8(R.color.y), q_unorm8(R.color.z), q_unorm8(R.color.w));     // rtDiffuse is RGBA8
        f3 direct = mk3s(1.0f); float historyLength = 0.0f;                                                            // DirectRayGen.hlsl:19
        if (R.instanceId >= 0) {
            const f3 normal = mk3(q_f16(R.normal.x), q_f16(R.normal.y), q_f16(R.normal.z));                            // rtShadingNormal / rtShadingSpecular are RGBA16F
            const f3 specular = mk3(q_f16(R.specular.x), q_f16(R.specular.y), q_f16(R.specular.z));
            const f3 resDirect = direct_light_pixel<CACHED>(P, env, px, py, rayDirection, R.instanceId, R.position, normal, specular);
            historyLength = 1.0f;
            direct = lerp3(mk3s(0.0f), resDirect, s_rcp(historyLength));
        }
        if (keepRecords) store_rgba16f(I.directLight[cur], i, direct.x, direct.y, direct.z, historyLength);
        if (FULL) { store_rgba16f(I.filteredDirect[1], i, direct.x, direct.y, direct.z, historyLength); note_cost(); continue; }
        const f3 result = compose_lean_value(P, diffuse, mk3(q_f16(direct.x), q_f16(direct.y), q_f16(direct.z)));
        if (P.separatePost) reinterpret_cast<float4 *>(I.output)[i] = make_float4(result.x, result.y, result.z, 1.0f);
        else {
            // the back-buffer pixel, and over it the foreground (HUD) draw list when the host folded it into this kernel: blended here
            // in RGBA8 exactly as raster_draw_kernel would blend it over the stored pixel (rt64_view.cpp:1657-1661), without the launch
            uint32_t bits = (uint32_t)to_unorm8(result.x) | ((uint32_t)to_unorm8(result.y) << 8) | ((uint32_t)to_unorm8(result.z) << 16) | ((uint32_t)to_unorm8(1.0f) << 24);
            if (P.rasterFgCount) {
                const int wx0 = __builtin_amdgcn_readfirstlane((int)(px & ~(uint32_t)(RT_WAVE_BLOCK_W - 1))), wy0 = __builtin_amdgcn_readfirstlane((int)(py & ~(uint32_t)(RT_WAVE_BLOCK_H - 1)));      // the wave's pixel block (8 x 8)
                bool loaded = true, dirty = false;
                raster_blend_pixel(P.rasterFg, static_cast<const RasterTri *>(P.rasterFgTris), P.rasterFgCount, P.textures, (int)px, (int)py, true, wx0, wx0 + RT_WAVE_BLOCK_W - 1, wy0, wy0 + RT_WAVE_BLOCK_H - 1, nullptr, bits, loaded, dirty);
            }
            reinterpret_cast<uint32_t *>(I.final)[i] = bits;
            if (P.finalPacked) {        // the gather's send buffer: row r of the owned rows, strips back to back (same order as RT64_CopyDeviceImage)
                const uint32_t rel = py - (uint32_t)P.tileY0, prow = ((rel >> 4) / (uint32_t)P.stripCount) * 16u + (rel & 15u);
                P.finalPacked[(size_t)prow * (size_t)P.width + px] = bits;
            }
        }
        note_cost();
    }
    flush_cost(P);
    if (P.tileTiming) {       // end record: clock, then the most node + triangle visits any lane of the wave made and the wave's total (<< 8 | 1: the valid mark)
        uint32_t worst = env.cnt.nodes + env.cnt.tris, total = worst;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { worst = max(worst, (uint32_t)__shfl_xor((int)worst, d, 64)); total += (uint32_t)__shfl_xor((int)total, d, 64); }
        if ((threadIdx.x & 63u) == 0u)
            P.tileTiming[2 * ((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) + 1] = make_uint4((uint32_t)__builtin_amdgcn_s_memrealtime(), (uint32_t)__builtin_amdgcn_s_memtime(), worst, (total << 8) | 1u);
#ifdef RT_PROFILE_TRIPS
        uint32_t tn = env.cnt.tripsNode, tl = env.cnt.tripsLeaf, sm = env.cnt.spills, ss = env.cnt.spills;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { tn = max(tn, (uint32_t)__shfl_xor((int)tn, d, 64)); tl = max(tl, (uint32_t)__shfl_xor((int)tl, d, 64)); sm = max(sm, (uint32_t)__shfl_xor((int)sm, d, 64)); ss += (uint32_t)__shfl_xor((int)ss, d, 64); }
        if ((threadIdx.x & 63u) == 0u)
            P.tileTiming[(size_t)RT_TIMING_WAVES * 2 + (size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)] = make_uint4(tn, tl, sm, ss);
#endif
    }
    // counters: the primary rays' visits under PASS_PRIMARY_TRACE, the shadow rays' under PASS_DIRECT (same split as the separate kernels)
    TraceCounts directCnt = TraceCounts(); directCnt.nodes = env.cnt.nodes - primaryCnt.nodes; directCnt.tris = env.cnt.tris - primaryCnt.tris;
    flush_counts(P, primaryCnt, PASS_PRIMARY_TRACE);
    env.cnt = directCnt;
    flush_env(P, env, PASS_DIRECT, CTR_PRIMARY, rays);
}

// ---- bounce-ray resolve shared by Indirect / Refraction / Reflection ------------------------------------------------------

// ---- second bounce of a GI ray (extension gi_bounces = 2; rules B1-B3 at gi_ray_radiance, oracle/oracle_render.c) -----------------------------------------
// The radiance ONE further ray brings back from the surface a GI ray resolved to: direction cosine-weighted about the resolved normal from blue-noise slice
// `noiseFrame` at the pixel's coordinates, the same resolve / light sample / sky term as the first bounce (IndirectRayGen.hlsl:58-131) with the constant
// ambient term as the light its surface receives besides its direct light.  Walked and shaded by the calling lane; its visits go to the caller's counters.
template <bool KLIST, bool CACHED>
DEV f3 second_bounce_radiance(PRef P, ShadeEnv &env, IRef I, size_t i, uint32_t px, uint32_t py, f3 rayOrigin, f3 normal, uint32_t noiseFrame, f3 ambientBase, f3 ambient) {
    const f3 rayDirection = cos_hemisphere_blue_noise(P, px, py, noiseFrame, normal);
    RayDiff rd; rd.dOdx = rd.dOdy = rd.dDdx = rd.dDdy = mk3s(0.0f);
    SurfaceHit best;
    const uint32_t nhits = trace_surface<KLIST, CACHED>(P, env, I, i, rayOrigin, rayDirection, rd, px, py, best);
    f3 resPosition = mk3s(0.0f), resNormal = mk3s(0.0f), resSpecular = mk3s(0.0f); f4 resColor = mk4(0, 0, 0, 1); int resInstanceId = -1;
    for (uint32_t hit = 0; hit < nhits; hit++) {
        HitRecord r;
        if (!surface_record<KLIST>(P, I, i, hit, best, rayDirection, rd, px, py, r)) continue;
        f4 hitColor = r.color;
        float alphaContrib = resColor.w * hitColor.w;
        if (alphaContrib >= RT_EPSILON) {
            const RT64_MATERIAL &m = P.instances[r.instanceId].material;
            resPosition = rayOrigin + rayDirection * (r.dist + m.depthBias);
            resNormal = r.normal; resSpecular = ld_v3(m.specularColor) * r.specular;
            resColor.x += hitColor.x * alphaContrib; resColor.y += hitColor.y * alphaContrib; resColor.z += hitColor.z * alphaContrib;
            resColor.w *= (1.0f - hitColor.w);
            resInstanceId = (int)r.instanceId;
        }
        if (resColor.w <= RT_EPSILON) break;
    }
    f3 resIndirect = ambientBase;
    if (resInstanceId >= 0) {
        f3 directLight = compute_lights_random<CACHED>(P, env, px, py, rayDirection, (uint32_t)resInstanceId, resPosition, resNormal, resSpecular, 1, true) + ld_v3(P.instances[resInstanceId].material.selfLight);
        f3 indirectLight = ((xyz(resColor) * (1.0f - resColor.w)) * (ambient + directLight)) * P.giDiffuseStrength;
        resIndirect = resIndirect + indirectLight;
    }
    if (resColor.w != 0.0f) resIndirect = resIndirect + sky_over_background_envmap(P, rayDirection) * (P.giSkyStrength * resColor.w);
    else resIndirect = resIndirect + mk3s(0.0f) * (P.giSkyStrength * resColor.w);
    return resIndirect;
}

template <bool KLIST, bool SECOND = false>      // SECOND: extension gi_bounces = 2 (its own instantiation: the reference's one-bounce kernels stay as they were)
__global__ __launch_bounds__(RT_BLOCK, 3) void indirect_kernel(FrameParams Pv, ViewImages Iv, int cur, int writeFiltered) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    __shared__ float ldsLightIntensity[(RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    __shared__ uint8_t ldsLightIndex[(RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    light_columns(env, ldsLightIntensity, ldsLightIndex, RT64_MAX_LIGHTS + 1);
    uint32_t rays = 0;
    const f3 ambientBase = mk3(P.ambientBaseColor[0], P.ambientBaseColor[1], P.ambientBaseColor[2]);
    const f3 ambient = ambientBase + mk3(P.ambientNoGIColor[0], P.ambientNoGIColor[1], P.ambientNoGIColor[2]);
    const uint32_t tiles = tile_count(P);
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this trip's view of the frame constants and the image table (read where used, never carried across trips)
        Pixel p = tile_pixel(P, tile);
        if (!p.valid) continue;
        const uint32_t px = p.x, py = p.y;
        const size_t i = (size_t)py * (size_t)P.width + px;
        const int instanceId = I.instanceId[i];
        if (!(instanceId >= 0 && P.giSamples > 0)) {
            store_rgba16f(I.indirectLight[cur], i, ambient.x, ambient.y, ambient.z, 0.0f);
            reinterpret_cast<float2 *>(I.moments[cur])[i] = make_float2(0.0f, 0.0f);
            if (writeFiltered) store_rgba16f(I.filteredIndirect[1], i, ambient.x, ambient.y, ambient.z, 0.0f);
            continue;
        }
        const float4 pos4 = reinterpret_cast<const float4 *>(I.shadingPosition)[i];
        f3 rayOrigin = mk3(pos4.x, pos4.y, pos4.z), shadingNormal = xyz(load_rgba16f(I.shadingNormal, i));
        f3 newIndirect = mk3s(0.0f); float historyLength = 0.0f;
        float2 prevM = make_float2(0.0f, 0.0f); float sumL = 0.0f, sumL2 = 0.0f;      // SVGF luminance moments (svgf.hip)
        if (P.giReproject) {
            long j; float w = history_weight(P, I, i, px, py, shadingNormal, cur, j);
            f4 prevAccum = j >= 0 ? load_rgba16f(I.indirectLight[cur ^ 1], (size_t)j) : mk4(0, 0, 0, 0);
            if (j >= 0) prevM = reinterpret_cast<const float2 *>(I.moments[cur ^ 1])[j];
            newIndirect = xyz(prevAccum); historyLength = prevAccum.w * w;
        }
        uint32_t maxSamples = P.giSamples; const uint32_t blueNoiseMult = 64u / P.giSamples;
        while (maxSamples > 0) {
            f3 rayDirection = cos_hemisphere_blue_noise(P, px, py, P.frameCount + maxSamples * blueNoiseMult, shadingNormal);
            RayDiff rd; rd.dOdx = rd.dOdy = rd.dDdx = rd.dDdy = mk3s(0.0f);
            SurfaceHit best;
            const uint32_t nhits = trace_surface<KLIST>(P, env, I, i, rayOrigin, rayDirection, rd, px, py, best);
            rays++;
            const f3 bgColor = sky_over_background_envmap(P, rayDirection);
            f3 resPosition = mk3s(0.0f), resNormal = mk3s(0.0f), resSpecular = mk3s(0.0f); f4 resColor = mk4(0, 0, 0, 1); int resInstanceId = -1;
            for (uint32_t hit = 0; hit < nhits; hit++) {
                HitRecord r;
                if (!surface_record<KLIST>(P, I, i, hit, best, rayDirection, rd, px, py, r)) continue;
                f4 hitColor = r.color;
                float alphaContrib = resColor.w * hitColor.w;
                if (alphaContrib >= RT_EPSILON) {
                    const RT64_MATERIAL &m = P.instances[r.instanceId].material;
                    resPosition = rayOrigin + rayDirection * (r.dist + m.depthBias);
                    resNormal = r.normal; resSpecular = ld_v3(m.specularColor) * r.specular;
                    resColor.x += hitColor.x * alphaContrib; resColor.y += hitColor.y * alphaContrib; resColor.z += hitColor.z * alphaContrib;
                    resColor.w *= (1.0f - hitColor.w);
                    resInstanceId = (int)r.instanceId;
                }
                if (resColor.w <= RT_EPSILON) break;
            }
            f3 resIndirect = ambientBase;
            if (resInstanceId >= 0) {
                f3 directLight = compute_lights_random(P, env, px, py, rayDirection, (uint32_t)resInstanceId, resPosition, resNormal, resSpecular, 1, true) + ld_v3(P.instances[resInstanceId].material.selfLight);
                f3 incoming = ambient;
                if (SECOND) {                   // extension, rules B1-B3 (oracle/oracle_render.c: gi_ray_radiance)
                    incoming = second_bounce_radiance<KLIST, false>(P, env, I, i, px, py, resPosition, resNormal, P.frameCount + maxSamples * blueNoiseMult + (blueNoiseMult > 1u ? blueNoiseMult / 2u : 1u), ambientBase, ambient);
                    rays++;
                }
                f3 indirectLight = ((xyz(resColor) * (1.0f - resColor.w)) * (incoming + directLight)) * P.giDiffuseStrength;
                resIndirect = resIndirect + indirectLight;
            }
            resIndirect = resIndirect + bgColor * (P.giSkyStrength * resColor.w);
            historyLength = fminf(historyLength + 1.0f, 64.0f);
            newIndirect = lerp3(newIndirect, resIndirect, s_rcp(historyLength));
            { const float l = 0.2126f * resIndirect.x + 0.7152f * resIndirect.y + 0.0722f * resIndirect.z; sumL += l; sumL2 += l * l; }
            maxSamples--;
        }
        store_rgba16f(I.indirectLight[cur], i, newIndirect.x, newIndirect.y, newIndirect.z, historyLength);
        {
            const float nS = (float)P.giSamples, alphaM = fminf(nS / historyLength, 1.0f);
            reinterpret_cast<float2 *>(I.moments[cur])[i] = make_float2(lerpf(prevM.x, sumL / nS, alphaM), lerpf(prevM.y, sumL2 / nS, alphaM));
        }
        if (writeFiltered) store_rgba16f(I.filteredIndirect[1], i, newIndirect.x, newIndirect.y, newIndirect.z, historyLength);
    }
    flush_env(P, env, PASS_INDIRECT, CTR_INDIRECT, rays);
}

// ---- IndirectRayGen as a wavefront pair (opaque frames) --------------------------------------------------------------------
// The one-kernel form above (bounce traversal + any-hit + texturing + light pick + shadow traversal) is 71 KB of code at 237
// VGPRs: it overflows the 64 KB instruction cache two CUs share and runs at 2 waves/SIMD.  When every instance is provably opaque
// (rule O1: the hit list degenerates to the closest hit) the pass splits like the primary pass does:
//   bounce_trace_{plain,refill}_kernel : bounce direction + closest-hit traversal only -> 32-byte record per (sample, pixel)
//                          { t, u, v, primitive | dir.xyz, instance }, ray ids compacted into hit / miss lists (wave ballot)
//   bounce_hit_kernel / bounce_miss_kernel : sample radiance of the listed rays (surface any-hit + light pick + shadow ray / sky)
//   bounce_resolve_kernel : per pixel: temporal accumulation over the samples, moments, stores.
// Same arithmetic in the same order as indirect_kernel<false>; records and results only carry values across launch boundaries.
// Wave-ballot compaction of the bounce rays by outcome: every traced ray appends its id (sample * pixels + pixel) to the hit list or
// to the miss list of ITS WORKGROUP (a private segment of the list arrays): one LDS atomic per wave and list, rank inside the ballot
// gives the slot, no global atomics (two chip-wide counters saturate at ~90 appends/us).  The shading kernels below run with the same
// grid, workgroup b on the segments of workgroup b: lanes that shade a surface and trace a shadow ray are no longer interleaved with
// lanes that only look up the sky (the one-kernel form ran at 30 % VALU lane utilisation).
struct BounceSegments { uint32_t perBlock; };        // list entries reserved per workgroup and list
// Tiles of the bounce walk's workgroups.  One workgroup per tile (frames of up to RT_MAX_BOUNCE_GROUPS tiles; bigger ones give workgroup b the tiles b,
// b + grid, ...), walked from the bottom of the frame up like the one-kernel frame: the dispatcher hands the next tile to whichever CU has a free slot,
// geometry (the bottom of a typical frame) first, sky last, and a workgroup whose tiles show no surface ends before it fills its scene cache
// (bounce_any_surface).  (Round 3; until then a persistent grid walked tiles b, b + grid, ...: whenever the grid shares a large factor with the tiles of a
// row, a workgroup's tiles line up in one column of the picture and the launch waits for the workgroups that drew the sphere -- C5 bounce kernels between
// 1.50 and 1.95 ms for grids of 1920 ... 5120; consecutive blocks of tiles per workgroup are worse still: profiles/r03_experiments/r03_grid_sweep*.txt.)
DEV uint32_t bounce_tiles_per_group(uint32_t tiles) { return (tiles + gridDim.x - 1) / gridDim.x; }
DEV bool bounce_tile_of(uint32_t tiles, uint32_t per, uint32_t k, uint32_t &tile) {         // k-th tile of this workgroup
    const uint32_t at = blockIdx.x + k * gridDim.x;
    tile = tiles - 1u - at;
    return k < per && at < tiles;
}
// Does any pixel of this workgroup's tiles show a surface (i.e. has a bounce ray)?  Workgroup-uniform; every thread calls it.
DEV bool bounce_any_surface(PRef P, IRef I, uint32_t tiles, uint32_t per) {
    bool any = false;
    for (uint32_t k = 0, tile; bounce_tile_of(tiles, per, k, tile); k++) {
        const Pixel p = tile_pixel(P, tile);
        if (p.valid && I.instanceId[(size_t)p.y * (size_t)P.width + p.x] >= 0) any = true;
    }
    return __syncthreads_or(any ? 1 : 0) != 0;
}
DEV uint32_t bounce_segment_size(PRef P) {
    const uint32_t tiles = tile_count(P);
    return ((tiles + gridDim.x - 1) / gridDim.x) * RT_BLOCK * P.giSamples;
}
// called by the lanes that hold a finished ray (any subset of the wave); ldsCount = the workgroup's two running counts
DEV void bounce_append(IRef I, uint32_t *ldsCount, uint32_t segment, size_t missBase, bool hit, uint32_t id) {
    const unsigned long long hm = __ballot(hit), mm = __ballot(!hit);
    const uint32_t lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)(hm | mm)) - 1;
    uint32_t hbase = 0, mbase = 0;
    if ((int)lane == leader) {
        if (hm) hbase = atomicAdd(&ldsCount[0], (uint32_t)__popcll(hm));
        if (mm) mbase = atomicAdd(&ldsCount[1], (uint32_t)__popcll(mm));
    }
    hbase = __shfl(hbase, leader, 64); mbase = __shfl(mbase, leader, 64);
    const unsigned long long below = (1ull << lane) - 1ull;
    const size_t seg = (size_t)blockIdx.x * segment;
    if (hit) I.bounceLists[seg + hbase + (uint32_t)__popcll(hm & below)] = id;
    else I.bounceLists[missBase + seg + mbase + (uint32_t)__popcll(mm & below)] = id;
}
DEV size_t bounce_miss_base(PRef P, uint32_t segment) { return (size_t)gridDim.x * segment; }

DEV f3 bounce_sky_term(PRef P, f3 rayDirection);

template <bool CACHED>
__global__ __launch_bounds__(RT_BLOCK, TRACE_WAVES) void bounce_trace_plain_kernel(FrameParams Pv, ViewImages Iv) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = CACHED ? RT_STACK_LDS_CACHED / 2 : RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    __shared__ uint32_t ldsCount[2];
    extern __shared__ u32x4_lds dynLds[];
    const uint32_t tiles = tile_count(P), per = bounce_tiles_per_group(tiles);
    if (!bounce_any_surface(P, I, tiles, per)) { if (threadIdx.x < 2) I.bounceCounts[2 * blockIdx.x + threadIdx.x] = 0; return; }      // nothing but sky: no ray, no cache fill
    if (threadIdx.x < 2) ldsCount[threadIdx.x] = 0;
    if (CACHED) fill_scene_cache(P, dynLds);
    __syncthreads();
    const uint32_t segment = bounce_segment_size(P); const size_t missBase = bounce_miss_base(P, segment);
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    env.lightIntensity = nullptr; env.lightIndex = nullptr;
    if (CACHED) env.stk.use_cache(dynLds);
    uint32_t rays = 0;
    const size_t stride = (size_t)P.width * (size_t)P.height;
    for (uint32_t k = 0, tile; bounce_tile_of(tiles, per, k, tile); k++) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this trip's view of the frame constants and the image table (read where used, never carried across trips)
        Pixel p = tile_pixel(P, tile);
        if (!p.valid) continue;
        const uint32_t px = p.x, py = p.y;
        const size_t i = (size_t)py * (size_t)P.width + px;
        if (I.instanceId[i] < 0) continue;
        const float4 pos4 = reinterpret_cast<const float4 *>(I.shadingPosition)[i];
        const f3 rayOrigin = mk3(pos4.x, pos4.y, pos4.z), shadingNormal = xyz(load_rgba16f(I.shadingNormal, i));
        const f3 ambientBase = mk3(P.ambientBaseColor[0], P.ambientBaseColor[1], P.ambientBaseColor[2]);
        const uint32_t blueNoiseMult = 64u / P.giSamples;
        for (uint32_t smp = P.giSamples; smp > 0; smp--) {
            const f3 rayDirection = cos_hemisphere_blue_noise(P, px, py, P.frameCount + smp * blueNoiseMult, shadingNormal);
            RayDiff rd; rd.dOdx = rd.dOdy = rd.dDdx = rd.dDdy = mk3s(0.0f);
            SurfaceHit best;
#if RT_ABLATE >= 2
            best.hit = false;                  // 2, 4: no walk
#else
            trace_surface<false, CACHED>(P, env, I, i, rayOrigin, rayDirection, rd, px, py, best);
#endif
            rays++;
            const size_t id = (size_t)(smp - 1) * stride + i;
            if (!best.hit) {
                // a ray that leaves the scene is finished here: its radiance is the sky term (what bounce_miss_kernel computes from a record and a
                // list entry -- one 16-byte store instead of a 32-byte record, a list append, and a kernel that reads both back).  (Round 3 measured the
                // third place for it: the direction and a mark in the result slot, the sky lookup in bounce_resolve_kernel.  C5: this kernel 1.334 -> 1.254 ms,
                // the resolve 0.098 -> 0.266 ms -- worse; the lookup costs 0.08 ms here.)
#if RT_ABLATE == 1 || RT_ABLATE == 2
                const f3 resIndirect = ambientBase + rayDirection * (P.giSkyStrength * 1.0f);       // 1, 2: no sky lookup
#else
                const f3 resIndirect = ambientBase + bounce_sky_term(P, rayDirection) * (P.giSkyStrength * 1.0f);
#endif
                I.bounceResults[id] = BounceRadiance{ resIndirect.x, resIndirect.y, resIndirect.z };
                continue;
            }
            uint4 a, b;
            a.x = __float_as_uint(best.t); a.y = __float_as_uint(best.u); a.z = __float_as_uint(best.v); a.w = best.prim;
            b.x = __float_as_uint(rayDirection.x); b.y = __float_as_uint(rayDirection.y); b.z = __float_as_uint(rayDirection.z);
            b.w = best.instance;
            uint4 *rec = I.bounceRecords + id * 2;
            rec[0] = a; rec[1] = b;
            bounce_append(I, ldsCount, segment, missBase, true, (uint32_t)id);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) I.bounceCounts[2 * blockIdx.x + threadIdx.x] = ldsCount[threadIdx.x];
    flush_env(P, env, PASS_INDIRECT, CTR_INDIRECT, rays);
}

// ---- the bounce walk in two phases (small scenes, several samples per pixel) ---------------------------------------------------------------------
// Bounce rays differ in LENGTH: on a scene of a few instances many of them miss every instance box and end after one or two TLAS nodes, the ones that
// enter an instance go on for ten to twenty steps, and the pixels of a tile that show the sky have no ray at all -- a wave of the plain walk keeps
// those lanes idle until its longest ray is done (26 % VALU lane utilisation, C5).  Here a workgroup takes up to four (tile, sample) items -- 256
// rays each -- through
//   phase 1: ray generation + the TLAS part of the walk only, every lane busy.  A ray that reaches no TLAS leaf has finished exactly as trace_ray
//            would finish it (same nodes, same tests, same count): it takes its sky term and is done.  A ray that reaches a leaf is a SURVIVOR: its
//            direction goes to its record, its (item, pixel slot) to a list in LDS (wave ballot + one LDS atomic per wave), nothing is counted;
//   phase 2: the lanes of the workgroup take the survivors densely from the list and walk each one in full, from the root (trace_surface: the walk
//            the plain kernel makes, counted here), then record the hit or take the sky term.
// Per ray the operations and their order are those of the plain kernel, so hits, records and visit counters are bit-identical
// (tests/test_gpu_features.py::test_bounce_walk_in_two_phases_matches_the_plain_walk); the survivors' TLAS steps run twice.
// Measured (MI355X, ms of the three bounce kernels, plain walk on a persistent grid of 2048 -> two phases, one workgroup per tile, 5 waves per SIMD): C5 (4 samples)
// 1.81 -> 1.49, C4 (2 samples) 0.529 -> 0.487, C3 (1 sample) 0.233 -> 0.235: the host takes this form for frames with two or more samples per pixel.
#define SPLIT_ITEMS 4u
#define SPLIT_WAVES 5          // waves per SIMD (96 VGPRs, 13 dwords spilled outside the walk): 3 % faster than 4 on C4 / C5
template <bool CACHED>
DEV bool tlas_reaches_a_leaf(PRef P, const f3 &o, const f3 &d, const TraceStack &stk, uint32_t &nodesVisited) {
    nodesVisited = 0;
    if (P.instanceCount == 0) return false;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    RaySpace W;
    make_ray_space(oo, dd, W);
    const uint32_t tlasOff = 4u * P.cacheInstances;
    int sp = 0;
    uint32_t cur = 0;
    for (;;) {
        while (!(cur & RT64_LEAF_BIT)) {
            const GpuNode nd = CACHED ? load_node_lds(stk.cache + tlasOff + 4u * cur) : load_node(P.tlasNodes + cur);
            nodesVisited++;
            float tl, tr;
            const bool hl = box_hit(W, nd.lmin, nd.lmax, RT_RAY_MIN_DISTANCE, RT_RAY_MAX_DISTANCE, tl);
            const bool hr = box_hit(W, nd.rmin, nd.rmax, RT_RAY_MIN_DISTANCE, RT_RAY_MAX_DISTANCE, tr);
            const bool both = hl && hr, rightFirst = tr < tl;
            const uint32_t nearChild = both ? (rightFirst ? nd.right : nd.left) : (hl ? nd.left : nd.right);
            if (both) stk.template push<CACHED>(sp, rightFirst ? nd.left : nd.right);
            if (hl || hr) cur = nearChild;
            else { if (sp == 0) return false; cur = stk.template pop<CACHED>(sp); }
        }
        if (cur != RT64_NO_CHILD) return true;
        if (sp == 0) return false;
        cur = stk.template pop<CACHED>(sp);
    }
}

template <bool CACHED>
__global__ __launch_bounds__(RT_BLOCK, CACHED ? SPLIT_WAVES : TRACE_WAVES) void bounce_trace_split_kernel(FrameParams Pv, ViewImages Iv) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = CACHED ? RT_STACK_LDS_CACHED / 2 : RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    __shared__ uint32_t ldsCount[2];
    __shared__ uint32_t ldsSurvivors;
    __shared__ uint16_t ldsList[SPLIT_ITEMS * RT_BLOCK];
    extern __shared__ u32x4_lds dynLds[];
    const uint32_t tiles = tile_count(P), S = P.giSamples, blueNoiseMult = 64u / S, per = bounce_tiles_per_group(tiles);
    if (!bounce_any_surface(P, I, tiles, per)) { if (threadIdx.x < 2) I.bounceCounts[2 * blockIdx.x + threadIdx.x] = 0; return; }      // nothing but sky: no ray, no cache fill
    if (threadIdx.x < 2) ldsCount[threadIdx.x] = 0;
    if (threadIdx.x == 2) ldsSurvivors = 0;
    if (CACHED) fill_scene_cache(P, dynLds);
    __syncthreads();
    const uint32_t segment = bounce_segment_size(P); const size_t missBase = bounce_miss_base(P, segment);
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    env.lightIntensity = nullptr; env.lightIndex = nullptr;
    if (CACHED) env.stk.use_cache(dynLds);
    uint32_t rays = 0;
    const size_t stride = (size_t)P.width * (size_t)P.height;
    const uint32_t myTiles = blockIdx.x < tiles ? (tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u, items = myTiles * S;
    // item w of this workgroup: its (w / S)-th tile (bounce_tile_of), sample S - (w % S) (the plain kernel's order)
    for (uint32_t w0 = 0; w0 < items; w0 += SPLIT_ITEMS) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this round's view of the frame constants and the image table
        const f3 ambientBase = mk3(P.ambientBaseColor[0], P.ambientBaseColor[1], P.ambientBaseColor[2]);
        // ---- phase 1 ----
        for (uint32_t k = 0; k < SPLIT_ITEMS && w0 + k < items; k++) {
            const uint32_t w = w0 + k, tile = tiles - 1u - (blockIdx.x + (w / S) * gridDim.x), smp = S - (w % S);
            const Pixel p = tile_pixel(P, tile);
            bool survivor = false;
            if (p.valid) {
                const size_t i = (size_t)p.y * (size_t)P.width + p.x;
                if (I.instanceId[i] >= 0) {
                    const float4 pos4 = reinterpret_cast<const float4 *>(I.shadingPosition)[i];
                    const f3 rayOrigin = mk3(pos4.x, pos4.y, pos4.z), shadingNormal = xyz(load_rgba16f(I.shadingNormal, i));
                    const f3 rayDirection = cos_hemisphere_blue_noise(P, p.x, p.y, P.frameCount + smp * blueNoiseMult, shadingNormal);
                    uint32_t visited;
                    survivor = tlas_reaches_a_leaf<CACHED>(P, rayOrigin, rayDirection, env.stk, visited);
                    const size_t id = (size_t)(smp - 1) * stride + i;
                    if (survivor) {
                        uint4 b; b.x = __float_as_uint(rayDirection.x); b.y = __float_as_uint(rayDirection.y); b.z = __float_as_uint(rayDirection.z); b.w = 0xFFFFFFFFu;
                        I.bounceRecords[id * 2 + 1] = b;
                    }
                    else {
                        env.cnt.nodes += visited; rays++;
                        const f3 resIndirect = ambientBase + bounce_sky_term(P, rayDirection) * (P.giSkyStrength * 1.0f);
                        I.bounceResults[id] = BounceRadiance{ resIndirect.x, resIndirect.y, resIndirect.z };
                    }
                }
            }
            const unsigned long long sm = __ballot(survivor);
            if (sm) {
                const uint32_t lane = threadIdx.x & 63u;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ldsSurvivors, (uint32_t)__popcll(sm));
                base = (uint32_t)__shfl((int)base, 0, 64);
                if (survivor) ldsList[base + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull))] = (uint16_t)((k << 8) | threadIdx.x);
            }
        }
        __syncthreads();
        // ---- phase 2 ----
        const uint32_t n = ldsSurvivors;
        for (uint32_t e = threadIdx.x; e < n; e += RT_BLOCK) {
            const uint32_t entry = ldsList[e], w = w0 + (entry >> 8), slot = entry & 255u;
            const uint32_t tile = tiles - 1u - (blockIdx.x + (w / S) * gridDim.x), smp = S - (w % S);
            const Pixel p = tile_pixel_at(P, tile, slot >> 6, slot & 63u);
            const size_t i = (size_t)p.y * (size_t)P.width + p.x, id = (size_t)(smp - 1) * stride + i;
            const float4 pos4 = reinterpret_cast<const float4 *>(I.shadingPosition)[i];
            const uint4 b = I.bounceRecords[id * 2 + 1];
            const f3 rayOrigin = mk3(pos4.x, pos4.y, pos4.z), rayDirection = mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
            RayDiff rd; rd.dOdx = rd.dOdy = rd.dDdx = rd.dDdy = mk3s(0.0f);
            SurfaceHit best;
            trace_surface<false, CACHED>(P, env, I, i, rayOrigin, rayDirection, rd, p.x, p.y, best);
            rays++;
            if (!best.hit) {
                const f3 resIndirect = ambientBase + bounce_sky_term(P, rayDirection) * (P.giSkyStrength * 1.0f);
                I.bounceResults[id] = BounceRadiance{ resIndirect.x, resIndirect.y, resIndirect.z };
                continue;
            }
            uint4 a;
            a.x = __float_as_uint(best.t); a.y = __float_as_uint(best.u); a.z = __float_as_uint(best.v); a.w = best.prim;
            uint4 *rec = I.bounceRecords + id * 2;
            rec[0] = a; rec[1] = make_uint4(b.x, b.y, b.z, best.instance);
            bounce_append(I, ldsCount, segment, missBase, true, (uint32_t)id);
        }
        __syncthreads();
        if (threadIdx.x == 0) ldsSurvivors = 0;
        __syncthreads();
    }
    if (threadIdx.x < 2) I.bounceCounts[2 * blockIdx.x + threadIdx.x] = ldsCount[threadIdx.x];
    flush_env(P, env, PASS_INDIRECT, CTR_INDIRECT, rays);
}

// (Handing the rays of a tile to the lanes in direction order instead -- 32 bins of (d.y level, sign d.x, sign d.z), counted and
// ranked in LDS, bit-identical results -- was measured and removed: C5 indirect 2.52 against 2.34 ms, commit 10c0030 and
// profiles/r02_experiments/bounce_binned_*.  The lanes of a wave diverge by WHICH of the three code paths -- node test, instance
// entry, triangle test -- their next step needs, and rays that start together and point the same way still take those steps at
// different trips.)
// Bounce rays are incoherent: in a plain one-ray-per-lane walk the wave waits for its longest ray (measured 27 % VALU lane
// utilisation).  Each wave therefore streams through its work list -- (tile, pixel slot, sample) in tile order -- and lanes whose
// ray has finished are REFILLED: when fewer than BOUNCE_MIN_LIVE lanes are still walking, the walk pauses (RayWalk::run), the
// finished lanes claim the next stream positions by rank in the idle ballot, generate their rays and join the walk.
#define BOUNCE_MIN_LIVE 40
__global__ __launch_bounds__(RT_BLOCK, TRACE_WAVES) void bounce_trace_refill_kernel(FrameParams Pv, ViewImages Iv) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    __shared__ uint32_t ldsCount[2];
    if (threadIdx.x < 2) ldsCount[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t segment = bounce_segment_size(P); const size_t missBase = bounce_miss_base(P, segment);
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    env.lightIntensity = nullptr; env.lightIndex = nullptr;
    uint32_t rays = 0;
    const size_t stride = (size_t)P.width * (size_t)P.height;
    const uint32_t tiles = tile_count(P), wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t S = P.giSamples, blueNoiseMult = 64u / S;
    const uint32_t myTiles = blockIdx.x < tiles ? (tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
    const uint32_t end = myTiles * 64u * S;        // stream positions of this wave: ((tileSeq * 64 + slot) * S + sample)
    uint32_t next = 0;                              // wave-uniform
    RayWalk walk; walk.alive = false;
    SurfaceHit best; best.hit = false; best.key = INFINITY;
    f3 rayDirection = mk3s(0.0f); size_t recIndex = 0; bool holding = false;     // holding: this lane owns a ray whose record is not written yet
    for (;;) {
        // ---- retire finished rays, refill idle lanes ----
        if (holding && !walk.alive) {
            uint4 a, b;
            a.x = __float_as_uint(best.t); a.y = __float_as_uint(best.u); a.z = __float_as_uint(best.v); a.w = best.prim;
            b.x = __float_as_uint(rayDirection.x); b.y = __float_as_uint(rayDirection.y); b.z = __float_as_uint(rayDirection.z);
            b.w = best.hit ? best.instance : 0xFFFFFFFFu;
            I.bounceRecords[recIndex] = a; I.bounceRecords[recIndex + 1] = b;
            bounce_append(I, ldsCount, segment, missBase, best.hit, (uint32_t)(recIndex / 2));
            holding = false;
        }
        while (next < end) {
            const unsigned long long idle = __ballot(!holding);
            if (idle == 0ull) break;
            if (!holding) {
                const uint32_t pos = next + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (pos < end) {
                    const uint32_t smp = pos % S, slot = (pos / S) & 63u, seq = pos / (S * 64u);
                    const Pixel p = tile_pixel_at(P, blockIdx.x + seq * gridDim.x, wave, slot);
                    const size_t i = (size_t)p.y * (size_t)P.width + p.x;
                    if (p.valid && I.instanceId[i] >= 0) {
                        const float4 pos4 = reinterpret_cast<const float4 *>(I.shadingPosition)[i];
                        const f3 rayOrigin = mk3(pos4.x, pos4.y, pos4.z), shadingNormal = xyz(load_rgba16f(I.shadingNormal, i));
                        rayDirection = cos_hemisphere_blue_noise(P, p.x, p.y, P.frameCount + (S - smp) * blueNoiseMult, shadingNormal);
                        const float oo[3] = { rayOrigin.x, rayOrigin.y, rayOrigin.z }, dd[3] = { rayDirection.x, rayDirection.y, rayDirection.z };
                        walk.begin(P, oo, dd, RT_RAY_MIN_DISTANCE, RT_RAY_MAX_DISTANCE);
                        best.hit = false; best.key = INFINITY;
                        recIndex = ((size_t)(S - smp - 1) * stride + i) * 2;
                        holding = true; rays++;
                    }
                }
            }
            next += (uint32_t)__popcll(idle);
        }
        if (__ballot(holding) == 0ull) break;
        // ---- walk until the ray ends or too few lanes are left walking (then refill) ----
        if (holding)
            walk.run<BOUNCE_MIN_LIVE>(P, true, env.stk,
                     [&](float t, float u, float v, uint32_t instance, uint32_t prim, float &tmax) -> bool {      // trace_surface<false>
                         const float key = t - P.instances[instance].material.depthBias;
                         if (key < best.key) { best.key = key; best.t = t; best.u = u; best.v = v; best.instance = instance; best.prim = prim; best.hit = true; }
                         const float lim = key + P.maxDepthBias;
                         if (lim < tmax) tmax = lim;
                         return false;
                     }, env.cnt, next < end);
    }
    __syncthreads();
    if (threadIdx.x < 2) I.bounceCounts[2 * blockIdx.x + threadIdx.x] = ldsCount[threadIdx.x];
    flush_env(P, env, PASS_INDIRECT, CTR_INDIRECT, rays);
}

// ---- IndirectRayGen shading on the compacted lists ---------------------------------------------------------------------------------
// bounce_hit_kernel   : one list entry per lane: surface any-hit on the recorded hit, light pick + shadow ray -> sample radiance
// bounce_miss_kernel  : one list entry per lane: sky / background environment lookup                          -> sample radiance
// bounce_resolve_kernel: per pixel, samples in the reference's order: temporal accumulation, luminance moments, stores.
// Per-sample arithmetic and its order are those of indirect_kernel<false>.
DEV f3 bounce_sky_term(PRef P, f3 rayDirection) {
    return sky_over_background_envmap(P, rayDirection);
}

template <bool CACHED, bool SECOND = false>    // SECOND: extension gi_bounces = 2 (its own instantiation: the reference's one-bounce kernel stays as it was)
__global__ __launch_bounds__(RT_BLOCK, CACHED ? HIT_WAVES : DIRECT_WAVES) void bounce_hit_kernel(FrameParams Pv, ViewImages Iv) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = CACHED ? RT_STACK_LDS_CACHED / 2 : RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    __shared__ float ldsLightIntensity[CACHED ? 1 : (RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    __shared__ uint8_t ldsLightIndex[CACHED ? 1 : (RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    extern __shared__ u32x4_lds dynLds[];
    const uint32_t n = I.bounceCounts[2 * blockIdx.x];      // the segment bounce_trace's workgroup blockIdx.x filled
    if (n == 0) return;                                     // (workgroup-uniform) no hit listed: nothing to shade, no cache fill
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    light_columns(env, ldsLightIntensity, ldsLightIndex, RT64_MAX_LIGHTS + 1);
    if (CACHED) cached_env(P, env, dynLds);
    const f3 ambientBase = mk3(P.ambientBaseColor[0], P.ambientBaseColor[1], P.ambientBaseColor[2]);
    const f3 ambient = ambientBase + mk3(P.ambientNoGIColor[0], P.ambientNoGIColor[1], P.ambientNoGIColor[2]);
    const uint32_t stride = (uint32_t)P.width * (uint32_t)P.height;
    const uint32_t segment = bounce_segment_size(P);
    uint32_t rays = 0;                                      // second-bounce rays (gi_bounces = 2)
    for (uint32_t e = threadIdx.x; e < n; e += RT_BLOCK) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this trip's view of the frame constants and the image table (read where used, never carried across trips)
        const uint32_t id = I.bounceLists[(size_t)blockIdx.x * segment + e], i = id % stride;
        const uint32_t px = i % (uint32_t)P.width, py = i / (uint32_t)P.width;
        const uint4 a = I.bounceRecords[(size_t)id * 2], b = I.bounceRecords[(size_t)id * 2 + 1];
        const f3 rayDirection = mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
        const float4 pos4 = reinterpret_cast<const float4 *>(I.shadingPosition)[i];
        const f3 rayOrigin = mk3(pos4.x, pos4.y, pos4.z);
        RayDiff rd; rd.dOdx = rd.dOdy = rd.dDdx = rd.dDdy = mk3s(0.0f);
        f3 resPosition = mk3s(0.0f), resNormal = mk3s(0.0f), resSpecular = mk3s(0.0f); f4 resColor = mk4(0, 0, 0, 1); int resInstanceId = -1;
        HitRecord r;
        if (surface_anyhit(P, b.w, a.w, __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), rayDirection, rd, px, py, r)) {
            f4 hitColor = r.color;
            float alphaContrib = resColor.w * hitColor.w;
            if (alphaContrib >= RT_EPSILON) {
                const RT64_MATERIAL &m = P.instances[r.instanceId].material;
                resPosition = rayOrigin + rayDirection * (r.dist + m.depthBias);
                resNormal = r.normal; resSpecular = ld_v3(m.specularColor) * r.specular;
                resColor.x += hitColor.x * alphaContrib; resColor.y += hitColor.y * alphaContrib; resColor.z += hitColor.z * alphaContrib;
                resColor.w *= (1.0f - hitColor.w);
                resInstanceId = (int)r.instanceId;
            }
        }
        f3 resIndirect = ambientBase;
        if (resInstanceId >= 0) {
            f3 directLight = compute_lights_random<CACHED>(P, env, px, py, rayDirection, (uint32_t)resInstanceId, resPosition, resNormal, resSpecular, 1, true) + ld_v3(P.instances[resInstanceId].material.selfLight);
            f3 incoming = ambient;
            if (SECOND) {                       // extension, rules B1-B3 (oracle/oracle_render.c: gi_ray_radiance): the second bounce of this GI ray, walked and shaded by the lane that shades the first hit
                const uint32_t blueNoiseMult = 64u / P.giSamples, noiseStep = blueNoiseMult > 1u ? blueNoiseMult / 2u : 1u;
                incoming = second_bounce_radiance<false, CACHED>(P, env, I, i, px, py, resPosition, resNormal, P.frameCount + (id / stride + 1u) * blueNoiseMult + noiseStep, ambientBase, ambient);
                rays++;
            }
            f3 indirectLight = ((xyz(resColor) * (1.0f - resColor.w)) * (incoming + directLight)) * P.giDiffuseStrength;
            resIndirect = resIndirect + indirectLight;
        }
        if (resColor.w != 0.0f) resIndirect = resIndirect + bounce_sky_term(P, rayDirection) * (P.giSkyStrength * resColor.w);
        else resIndirect = resIndirect + mk3s(0.0f) * (P.giSkyStrength * resColor.w);
        I.bounceResults[id] = BounceRadiance{ resIndirect.x, resIndirect.y, resIndirect.z };
    }
    flush_env(P, env, PASS_INDIRECT, CTR_INDIRECT, rays);
}

__global__ __launch_bounds__(RT_BLOCK) void bounce_miss_kernel(FrameParams Pv, ViewImages Iv) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    const f3 ambientBase = mk3(P.ambientBaseColor[0], P.ambientBaseColor[1], P.ambientBaseColor[2]);
    const uint32_t segment = bounce_segment_size(P), n = I.bounceCounts[2 * blockIdx.x + 1];
    const size_t base = bounce_miss_base(P, segment) + (size_t)blockIdx.x * segment;
    for (uint32_t e = threadIdx.x; e < n; e += RT_BLOCK) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this trip's view of the frame constants and the image table (read where used, never carried across trips)
        const uint32_t id = I.bounceLists[base + e];
        const uint4 b = I.bounceRecords[(size_t)id * 2 + 1];
        const f3 rayDirection = mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
        const f3 resIndirect = ambientBase + bounce_sky_term(P, rayDirection) * (P.giSkyStrength * 1.0f);
        I.bounceResults[id] = BounceRadiance{ resIndirect.x, resIndirect.y, resIndirect.z };
    }
}

__global__ __launch_bounds__(256) void bounce_resolve_kernel(FrameParams Pv, ViewImages Iv, int cur, int writeFiltered, int writeGuide) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = P.tileY0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= P.width || y >= P.tileY1 || !row_owned(P, y)) return;
    const uint32_t px = (uint32_t)x, py = (uint32_t)y;
    const size_t i = (size_t)py * (size_t)P.width + px, stride = (size_t)P.width * (size_t)P.height;
    if (writeGuide) {
        // the SVGF guide record of the pixel (svgf.hip: svgf_guide_kernel, the same bytes): this kernel already has the pixel's id, normal and depth in flight, so the
        // denoiser's first launch -- a pass over the same three images -- is folded into it on frames whose GI runs as the wavefront chain (C5: 44 us + a launch gap)
        const uint2 n = reinterpret_cast<const uint2 *>(I.normal[cur])[i];
        const float *depth = I.depth[cur];
        const float z = depth[i], zx = depth[(size_t)y * P.width + (x + 1 < P.width ? x + 1 : x)], zy = depth[(size_t)(y + 1 < P.height ? y + 1 : y) * P.width + x];
        uint4 g;
        g.x = n.x; g.y = (n.y & 0xFFFFu) | (I.instanceId[i] >= 0 ? 0x10000u : 0u);
        g.z = __float_as_uint(z); g.w = __float_as_uint(fmaxf(fabsf(zx - z), fabsf(zy - z)));
        I.svgfGuide[i] = g;
    }
    // writeGuide & 2: the filter's INPUT as well -- colour + variance in filteredIndirect[0], what svgf_variance_kernel writes for every pixel with four frames of
    // history (variance from the luminance moments this kernel has just made) and for every pixel without a surface (both ping-pong images); pixels with a shorter
    // history are marked in I.svgfYoung and svgf_variance_kernel then runs its 7 x 7 estimate only where a mark is set (C5, static scene: 40 -> 6 us)
    const bool writeInput = (writeGuide & 2) != 0;
    if (I.instanceId[i] < 0) {
        const float ax = P.ambientBaseColor[0] + P.ambientNoGIColor[0], ay = P.ambientBaseColor[1] + P.ambientNoGIColor[1], az = P.ambientBaseColor[2] + P.ambientNoGIColor[2];
        store_rgba16f(I.indirectLight[cur], i, ax, ay, az, 0.0f);
        reinterpret_cast<float2 *>(I.moments[cur])[i] = make_float2(0.0f, 0.0f);
        if (writeFiltered) store_rgba16f(I.filteredIndirect[1], i, ax, ay, az, 0.0f);
        if (writeInput) { store_rgba16f(I.filteredIndirect[0], i, ax, ay, az, 0.0f); store_rgba16f(I.filteredIndirect[1], i, ax, ay, az, 0.0f); }
        return;
    }
    const f3 shadingNormal = xyz(load_rgba16f(I.shadingNormal, i));
    f3 newIndirect = mk3s(0.0f); float historyLength = 0.0f;
    float2 prevM = make_float2(0.0f, 0.0f); float sumL = 0.0f, sumL2 = 0.0f;
    if (P.giReproject) {
        long j; float w = history_weight(P, I, i, px, py, shadingNormal, cur, j);
        f4 prevAccum = j >= 0 ? load_rgba16f(I.indirectLight[cur ^ 1], (size_t)j) : mk4(0, 0, 0, 0);
        if (j >= 0) prevM = reinterpret_cast<const float2 *>(I.moments[cur ^ 1])[j];
        newIndirect = xyz(prevAccum); historyLength = prevAccum.w * w;
    }
    for (uint32_t smp = P.giSamples; smp > 0; smp--) {
        const BounceRadiance v = I.bounceResults[(size_t)(smp - 1) * stride + i];
        const f3 resIndirect = mk3(v.r, v.g, v.b);
        historyLength = fminf(historyLength + 1.0f, 64.0f);
        newIndirect = lerp3(newIndirect, resIndirect, s_rcp(historyLength));
        { const float l = 0.2126f * resIndirect.x + 0.7152f * resIndirect.y + 0.0722f * resIndirect.z; sumL += l; sumL2 += l * l; }
    }
    store_rgba16f(I.indirectLight[cur], i, newIndirect.x, newIndirect.y, newIndirect.z, historyLength);
    {
        const float nS = (float)P.giSamples, alphaM = fminf(nS / historyLength, 1.0f);
        const float2 m = make_float2(lerpf(prevM.x, sumL / nS, alphaM), lerpf(prevM.y, sumL2 / nS, alphaM));
        reinterpret_cast<float2 *>(I.moments[cur])[i] = m;
        if (writeInput) {
            const bool young = q_f16(historyLength) < 4.0f;               // (the filter reads the history length back from the RGBA16F image)
            store_rgba16f(I.filteredIndirect[0], i, newIndirect.x, newIndirect.y, newIndirect.z, young ? 0.0f : svgf_moment_variance(m.x, m.y));
            if (young) I.svgfYoung[(size_t)y * (size_t)((P.width + 31) / 32) + (size_t)(x >> 5)] = 1u;
        }
    }
    if (writeFiltered) store_rgba16f(I.filteredIndirect[1], i, newIndirect.x, newIndirect.y, newIndirect.z, historyLength);
}

DEV f3 hlsl_refract(f3 i, f3 n, float eta) {
    float cosi = dot3(n, i);
    float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
    if (k < 0.0f) return mk3s(0.0f);
    return i * eta - n * (eta * cosi + sqrtf(k));
}

template <bool KLIST>
__global__ __launch_bounds__(RT_BLOCK) void refraction_kernel(FrameParams Pv, ViewImages Iv) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    constexpr uint32_t STACK_WORDS = RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    __shared__ float ldsLightIntensity[(RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    __shared__ uint8_t ldsLightIndex[(RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    light_columns(env, ldsLightIntensity, ldsLightIndex, RT64_MAX_LIGHTS + 1);
    uint32_t rays = 0;
    const f3 ambient = mk3(P.ambientBaseColor[0], P.ambientBaseColor[1], P.ambientBaseColor[2]) + mk3(P.ambientNoGIColor[0], P.ambientNoGIColor[1], P.ambientNoGIColor[2]);
    const uint32_t tiles = tile_count(P);
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this trip's view of the frame constants and the image table (read where used, never carried across trips)
        Pixel p = tile_pixel(P, tile);
        if (!p.valid) continue;
        const uint32_t px = p.x, py = p.y;
        const size_t i = (size_t)py * (size_t)P.width + px;
        const int instanceId = I.instanceId[i];
        f4 refr = load_rgba16f(I.refraction, i);
        const float refractionAlpha = refr.w;
        if (instanceId < 0 || refractionAlpha <= RT_EPSILON) continue;
        const float4 pos4 = reinterpret_cast<const float4 *>(I.shadingPosition)[i];
        f3 rayOrigin = mk3(pos4.x, pos4.y, pos4.z), viewDirection = xyz(load_rgba16f(I.viewDirection, i)), shadingNormal = xyz(load_rgba16f(I.shadingNormal, i));
        f3 rayDirection = hlsl_refract(viewDirection, shadingNormal, P.instances[instanceId].material.refractionFactor);
        f2 screenUV; screenUV.x = ((float)px + P.pixelJitter[0]) / (float)P.width; screenUV.y = ((float)py + P.pixelJitter[1]) / (float)P.height;
        const f3 bgColor = sky_over_background_2d(P, screenUV);
        RayDiff rd; rd.dOdx = rd.dOdy = rd.dDdx = rd.dDdy = mk3s(0.0f);
        SurfaceHit best;
        const uint32_t nhits = trace_surface<KLIST>(P, env, I, i, rayOrigin, rayDirection, rd, px, py, best);
        rays++;
        f3 resPosition = mk3s(0.0f), resNormal = mk3s(0.0f), resSpecular = mk3s(0.0f), resTransparent = mk3s(0.0f); f4 resColor = mk4(0, 0, 0, 1); int resInstanceId = -1;
        for (uint32_t hit = 0; hit < nhits; hit++) {
            HitRecord r;
            if (!surface_record<KLIST>(P, I, i, hit, best, rayDirection, rd, px, py, r)) continue;
            f4 hitColor = r.color;
            float alphaContrib = resColor.w * hitColor.w;
            if (alphaContrib >= RT_EPSILON) {
                const RT64_MATERIAL &m = P.instances[r.instanceId].material;
                bool usesLighting = m.lightGroupMaskBits > 0;
                f3 vertexPosition = rayOrigin + rayDirection * (r.dist + m.depthBias);
                if (m.fogEnabled) {
                    f4 fog = fog_from_camera(P, m, vertexPosition);
                    resTransparent = resTransparent + xyz(fog) * (fog.w * alphaContrib);
                    alphaContrib *= (1.0f - fog.w);
                }
                if (usesLighting) {
                    resColor.x += hitColor.x * alphaContrib; resColor.y += hitColor.y * alphaContrib; resColor.z += hitColor.z * alphaContrib;
                    resPosition = vertexPosition; resNormal = r.normal; resSpecular = ld_v3(m.specularColor) * r.specular; resInstanceId = (int)r.instanceId;
                }
                else resTransparent = resTransparent + (xyz(hitColor) * alphaContrib) * (ambient + ld_v3(m.selfLight));
                resColor.w *= (1.0f - hitColor.w);
            }
            if (resColor.w <= RT_EPSILON) break;
        }
        f3 rgb = xyz(resColor);
        if (resInstanceId >= 0) {
            f3 directLight = compute_lights_random(P, env, px, py, rayDirection, (uint32_t)resInstanceId, resPosition, resNormal, resSpecular, 1, true) + ld_v3(P.instances[resInstanceId].material.selfLight);
            rgb = rgb * (ambient + directLight);
        }
        rgb = rgb + (bgColor * resColor.w + resTransparent);
        store_rgba16f(I.refraction, i, refr.x + rgb.x * refractionAlpha, refr.y + rgb.y * refractionAlpha, refr.z + rgb.z * refractionAlpha, refr.w);
    }
    flush_env(P, env, PASS_REFRACTION, CTR_REFRACTION, rays);
}

// pass = which of the frame's reflection passes this is, last = whether it is the last one, parity = the frame's ping-pong index.  A pass only has work where the
// pass before it left a reflection alpha above EPSILON (a mirror seen in a mirror); that pass says so in I.reflectFlags[parity][pass], and a pass whose flag is
// clear ends before it reads a pixel (the second pass of the sample scene's reflective floor: 32 us of a C5 frame).  The frame's last pass clears the other
// parity's flags for the next frame.  A stale set flag only costs the scan it used to cost.
// CACHED (round 4; opaque frames of scenes with the LDS scene cache): the walk of the mirrored ray and of its shadow ray from LDS like the frame kernel's -- scene cache +
// int16 stacks + light columns sized by the frame's light count: 32 KB of LDS per workgroup on the sample scene where the uncached form holds 46.6 KB (24 KB of
// stacks, 21.7 KB of columns for 17 lights whatever the frame has), i.e. 5 instead of 3 workgroups per CU, and every node visit a ds_read_b128 instead of an L2
// round trip.  The cache is filled when the workgroup starts, like everywhere else (a fill that waits for the first tile with a mirror measured slower in round 3:
// the vote and the fill were serial in workgroups of one tile).
template <bool KLIST, bool CACHED = false>
__global__ __launch_bounds__(RT_BLOCK, CACHED ? REFLECT_WAVES : 1) void reflection_kernel(FrameParams Pv, ViewImages Iv, int pass, int last, int parity) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    if (last && blockIdx.x == 0 && threadIdx.x < 4) I.reflectFlags[(parity ^ 1) * 4 + threadIdx.x] = 0;
    if (pass > 0 && pass < 4 && I.reflectFlags[parity * 4 + pass] == 0u) return;          // (workgroup-uniform: written by the launch before this one; passes beyond the fourth always scan)
    const uint32_t frameTag = P.frameCount + 1u;       // marks the continuation state this frame wrote (never 0: the images start zeroed)
    constexpr uint32_t STACK_WORDS = CACHED ? RT_STACK_LDS_CACHED / 2 : RT_STACK_LDS;
    __shared__ uint32_t ldsStack[STACK_WORDS * RT_BLOCK];
    __shared__ float ldsLightIntensity[CACHED ? 1 : (RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    __shared__ uint8_t ldsLightIndex[CACHED ? 1 : (RT64_MAX_LIGHTS + 1) * RT_BLOCK];
    extern __shared__ u32x4_lds dynLds[];
    ShadeEnv env; env.stk = make_stack(P, ldsStack, STACK_WORDS); env.cnt = TraceCounts(); env.shadowRays = 0;
    light_columns(env, ldsLightIntensity, ldsLightIndex, RT64_MAX_LIGHTS + 1);
    if (CACHED) cached_env(P, env, dynLds);
    uint32_t rays = 0;
    const f3 ambient = mk3(P.ambientBaseColor[0], P.ambientBaseColor[1], P.ambientBaseColor[2]) + mk3(P.ambientNoGIColor[0], P.ambientNoGIColor[1], P.ambientNoGIColor[2]);
    const uint32_t tiles = tile_count(P);
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        PRef P = *kernel_params_here(); IRef I = *kernel_images_here();      // this trip's view of the frame constants and the image table (read where used, never carried across trips)
        Pixel p = tile_pixel(P, tile);
        if (!p.valid) continue;
        const uint32_t px = p.x, py = p.y;
        const size_t i = (size_t)py * (size_t)P.width + px;
        f4 refl = load_rgba16f(I.reflection, i);
        const float reflectionAlpha = refl.w;
        // the surface this pass mirrors: the G-buffer's (pass 0), or what the pass before left in the continuation state (ViewImages::reflState0 / 1)
        int instanceId; f3 shadingPosition, viewDirection, shadingNormal;
        if (pass == 0) {
            instanceId = I.instanceId[i];
            if (instanceId < 0 || reflectionAlpha <= RT_EPSILON) continue;
            const float4 pos4 = reinterpret_cast<const float4 *>(I.shadingPosition)[i];
            shadingPosition = mk3(pos4.x, pos4.y, pos4.z); viewDirection = xyz(load_rgba16f(I.viewDirection, i)); shadingNormal = xyz(load_rgba16f(I.shadingNormal, i));
        }
        else {
            if (reflectionAlpha <= RT_EPSILON || I.reflTag[i] != frameTag) continue;        // (an alpha above EPSILON was left by a pass that also wrote the state)
            const uint4 s0 = I.reflState0[i], s1 = I.reflState1[i];
            instanceId = (int)s0.w;
            shadingPosition = mk3(__uint_as_float(s0.x), __uint_as_float(s0.y), __uint_as_float(s0.z));
            viewDirection = xyz(unpack_rgba16f_bits(s1.x, s1.y)); shadingNormal = xyz(unpack_rgba16f_bits(s1.z, s1.w));
        }
        f3 rayDirection = reflect3(viewDirection, shadingNormal);
        float newReflectionAlpha = 0.0f;
        const f3 bgColor = sky_over_background_envmap(P, rayDirection);
        RayDiff rd; rd.dOdx = rd.dOdy = rd.dDdx = rd.dDdy = mk3s(0.0f);
        SurfaceHit best;
        const uint32_t nhits = trace_surface<KLIST, CACHED>(P, env, I, i, shadingPosition, rayDirection, rd, px, py, best);
        rays++;
        const RT64_MATERIAL &pm = P.instances[instanceId].material;
        f3 resPosition = mk3s(0.0f), resNormal = mk3s(0.0f), resSpecular = mk3s(0.0f), resTransparent = mk3s(0.0f); f4 resColor = mk4(0, 0, 0, 1); int resInstanceId = -1;
        for (uint32_t hit = 0; hit < nhits; hit++) {
            HitRecord r;
            if (!surface_record<KLIST>(P, I, i, hit, best, rayDirection, rd, px, py, r)) continue;
            f4 hitColor = r.color;
            float alphaContrib = resColor.w * hitColor.w;
            if (alphaContrib >= RT_EPSILON) {
                const RT64_MATERIAL &m = P.instances[r.instanceId].material;
                bool usesLighting = m.lightGroupMaskBits > 0;
                f3 vertexPosition = shadingPosition + rayDirection * (r.dist + m.depthBias);
                if (m.fogEnabled) {
                    f4 fog = fog_from_origin(m, vertexPosition, shadingPosition);
                    resTransparent = resTransparent + xyz(fog) * (fog.w * alphaContrib);
                    alphaContrib *= (1.0f - fog.w);
                }
                f3 vertexNormal = r.normal;
                f3 specular = ld_v3(m.specularColor) * r.specular;
                if (m.reflectionFactor > RT_EPSILON) {
                    float fresnelAmount = fresnel_reflect_amount(vertexNormal, rayDirection, m.reflectionFactor, pm.reflectionFresnelFactor);   // sic: [instanceId], ReflectionRayGen.hlsl:98
                    newReflectionAlpha += fresnelAmount * alphaContrib * reflectionAlpha;
                }
                if (usesLighting) { resColor.x += hitColor.x * alphaContrib; resColor.y += hitColor.y * alphaContrib; resColor.z += hitColor.z * alphaContrib; }
                else resTransparent = resTransparent + (xyz(hitColor) * alphaContrib) * (ambient + ld_v3(m.selfLight));
                resPosition = vertexPosition; resNormal = vertexNormal; resSpecular = specular; resInstanceId = (int)r.instanceId;
                resColor.w *= (1.0f - hitColor.w);
            }
            if (resColor.w <= RT_EPSILON) break;
        }
        f3 rgb = xyz(resColor);
        if (resInstanceId >= 0) {
            f3 directLight = compute_lights_random<CACHED>(P, env, px, py, rayDirection, (uint32_t)resInstanceId, resPosition, resNormal, resSpecular, 1, false) + ld_v3(P.instances[resInstanceId].material.selfLight);
            rgb = rgb * (ambient + directLight);
            // ReflectionRayGen.hlsl:117-124: the next pass continues from here (values rounded as the images the reference rewrites round them: RGBA32F / RGBA16F)
            I.reflState0[i] = make_uint4(__float_as_uint(resPosition.x), __float_as_uint(resPosition.y), __float_as_uint(resPosition.z), (uint32_t)resInstanceId);
            I.reflState1[i] = make_uint4(pack_rgba16f_lo(rayDirection.x, rayDirection.y), pack_rgba16f_lo(rayDirection.z, 0.0f), pack_rgba16f_lo(resNormal.x, resNormal.y), pack_rgba16f_lo(resNormal.z, 0.0f));
            I.reflTag[i] = frameTag;
        }
        rgb = rgb + (bgColor * resColor.w + resTransparent);
        const f3 HighlightColor = mk3(1.0f, 1.05f, 1.2f), ShadowColor = mk3(0.1f, 0.05f, 0.0f);
        float shine = pm.reflectionShineFactor;
        rgb = lerp3(rgb, HighlightColor, s_pow(fmaxf(rayDirection.y, 0.0f) * shine, 3.0f));
        rgb = lerp3(rgb, ShadowColor, s_pow(fmaxf(-rayDirection.y, 0.0f) * shine, 3.0f));
        float k = reflectionAlpha * saturatef(1.0f - newReflectionAlpha);
        store_rgba16f(I.reflection, i, refl.x + rgb.x * k, refl.y + rgb.y * k, refl.z + rgb.z * k, saturatef(newReflectionAlpha));
        if (pass < 3 && q_f16(saturatef(newReflectionAlpha)) > RT_EPSILON) I.reflectFlags[parity * 4 + pass + 1] = 1u;      // this pixel goes on to the next pass
    }
    flush_env(P, env, PASS_REFLECTION, CTR_REFLECTION, rays);
}

// ---- GaussianFilterRGB3x3CS ------------------------------------------------------------------------------------------------

DEV f3 bilinear_clamp_rgb(const uint16_t *img, int w, int h, float u, float v) {   // LINEAR + CLAMP static sampler, rt64_device.cpp:737-742
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y), fx = x - x0f, fy = y - y0f;
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 >= w ? w - 1 : x0); x1 = x1 < 0 ? 0 : (x1 >= w ? w - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 >= h ? h - 1 : y0); y1 = y1 < 0 ? 0 : (y1 >= h ? h - 1 : y1);
    f3 c00 = xyz(load_rgba16f(img, (size_t)y0 * w + x0)), c10 = xyz(load_rgba16f(img, (size_t)y0 * w + x1));
    f3 c01 = xyz(load_rgba16f(img, (size_t)y1 * w + x0)), c11 = xyz(load_rgba16f(img, (size_t)y1 * w + x1));
    f3 top = lerp3(c00, c10, fx), bot = lerp3(c01, c11, fx);
    return lerp3(top, bot, fy);
}

__global__ __launch_bounds__(256) void gaussian_kernel(const uint16_t *in, uint16_t *out, int w, int h, int y0, int y1) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= y1) return;
    const float texelX = 1.0f / (float)w, texelY = 1.0f / (float)h;
    const float k00 = 0.077847f, k01 = 0.123317f, k11 = 0.195346f;
    float wt[4];
    const bool xl = x == 0, xr = x == w - 1, yt = y == 0, yb = y == h - 1;
    if (x > 0 && y > 0 && x < w - 1 && y < h - 1) { wt[0] = k00 + k01 + k01 + k11; wt[1] = k00 + k01; wt[2] = k00 + k01; wt[3] = k00; }
    else if (xl && yt) { wt[0] = k11 / 0.519827f; wt[1] = k01 / 0.519827f; wt[2] = k01 / 0.519827f; wt[3] = k00 / 0.519827f; }
    else if (xr && yt) { wt[0] = (k01 + k11) / 0.519827f; wt[1] = 0.0f; wt[2] = 0.201164f / 0.519827f; wt[3] = 0.0f; }
    else if (xl && yb) { wt[0] = (k01 + k11) / 0.519827f; wt[1] = (k00 + k01) / 0.519827f; wt[2] = 0.0f; wt[3] = 0.0f; }
    else if (xr && yb) { wt[0] = (k00 + k01 + k01 + k11) / 0.519827f; wt[1] = wt[2] = wt[3] = 0.0f; }
    else if (xl) { wt[0] = (k01 + k11) / 0.720991f; wt[1] = (k00 + k01) / 0.720991f; wt[2] = k01 / 0.720991f; wt[3] = k00 / 0.720991f; }
    else if (xr) { wt[0] = (k00 + k01 + k01 + k11) / 0.720991f; wt[1] = 0.0f; wt[2] = (k00 + k01) / 0.720991f; wt[3] = 0.0f; }
    else if (yt) { wt[0] = (k01 + k11) / 0.720991f; wt[1] = k01 / 0.720991f; wt[2] = (k00 + k01) / 0.720991f; wt[3] = k00 / 0.720991f; }
    else { wt[0] = (k00 + k01 + k01 + k11) / 0.720991f; wt[1] = (k00 + k01) / 0.720991f; wt[2] = 0.0f; wt[3] = 0.0f; }
    const float off[3][2] = { { 0.5f + -k01 / (k01 + k11), 0.5f + -k01 / (k01 + k11) }, { 0.5f + 1.0f, 0.5f + -k00 / (k00 + k01) }, { 0.5f + -k00 / (k00 + k01), 0.5f + 1.0f } };
    f3 smp[4];
#pragma unroll
    for (int k = 0; k < 3; k++) smp[k] = bilinear_clamp_rgb(in, w, h, ((float)x + off[k][0]) * texelX, ((float)y + off[k][1]) * texelY);
    smp[3] = (x + 1 < w && y + 1 < h) ? xyz(load_rgba16f(in, (size_t)(y + 1) * w + (x + 1))) : mk3s(0.0f);
    const size_t i = (size_t)y * w + x;
    f4 old = load_rgba16f(out, i);
    store_rgba16f(out, i, smp[0].x * wt[0] + smp[1].x * wt[1] + smp[2].x * wt[2] + smp[3].x * wt[3],
                  smp[0].y * wt[0] + smp[1].y * wt[1] + smp[2].y * wt[2] + smp[3].y * wt[3],
                  smp[0].z * wt[0] + smp[1].z * wt[1] + smp[2].z * wt[2] + smp[3].z * wt[3], old.w);
}

// ---- ComposePS + PostProcessPS (fused) -------------------------------------------------------------------------------------

// LEAN: direct light straight from the raw accumulation, constant ambient for the indirect term (giSamples == 0), and no
// reflection / refraction / transparent reads -- all of them are exact zeros on a lean frame.
template <bool LEAN>
__global__ __launch_bounds__(256) void compose_post_kernel(FrameParams Pv, ViewImages Iv, int cur, int writeFinal) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = P.tileY0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= P.width || y >= P.tileY1 || !row_owned(P, y)) return;
    const size_t i = (size_t)y * (size_t)P.width + x;
    f4 d = load_rgba8(I.diffuse, i);
    f3 result;
    if (d.w > RT_EPSILON) {
        f3 diffuse = xyz(d);
        f3 direct, indirect;
        if (LEAN) {
            direct = xyz(load_rgba16f(I.directLight[cur], i));
            indirect = mk3(q_f16(P.ambientBaseColor[0] + P.ambientNoGIColor[0]), q_f16(P.ambientBaseColor[1] + P.ambientNoGIColor[1]), q_f16(P.ambientBaseColor[2] + P.ambientNoGIColor[2]));
        }
        else { direct = xyz(load_rgba16f(I.filteredDirect[1], i)); indirect = xyz(load_rgba16f(I.filteredIndirect[1], i)); }
        result = diffuse * (direct + indirect);
        result = lerp3(diffuse, result, d.w);
        if (!LEAN) {
            result = result + xyz(load_rgba16f(I.reflection, i));
            result = result + xyz(load_rgba16f(I.refraction, i));
            result = result + xyz(load_rgba16f(I.transparent, i));
        }
    }
    else result = xyz(d);
    reinterpret_cast<float4 *>(I.output)[i] = make_float4(result.x, result.y, result.z, 1.0f);
    if (!P.separatePost && writeFinal) store_rgba8(I.final, i, result.x, result.y, result.z, 1.0f);   // PostProcessPS passthrough (motionBlurStrength == 0, render size == screen size)
}

// Longest-first order of the one-kernel frame's tiles (device option tile_order; scenes that walk from HBM): tiles sorted by the cost the frame just recorded, most
// expensive first -- a counting sort over min(cost, 1023) in one workgroup (a 1080p frame has 8 160 tiles) -- and the costs cleared for the next frame.  Ties land in
// whatever order the atomics resolve: any permutation renders the same picture.
__global__ __launch_bounds__(1024) void tile_order_kernel(uint32_t *cost, uint32_t *order, uint32_t n) {
    __shared__ uint32_t bucket[1024];
    bucket[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) atomicAdd(&bucket[1023u - min(cost[i], 1023u)], 1u);
    __syncthreads();
    // exclusive scan of the 1024 counts: a wave scans its 64, then the 16 wave totals are added up by every thread
    const uint32_t mine = bucket[threadIdx.x];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64); if ((threadIdx.x & 63u) >= (uint32_t)d) incl += up; }
    __shared__ uint32_t waveTotal[16];
    if ((threadIdx.x & 63u) == 63u) waveTotal[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += waveTotal[w];
    bucket[threadIdx.x] = base + incl - mine;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint32_t b = 1023u - min(cost[i], 1023u);
        order[atomicAdd(&bucket[b], 1u)] = i;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) cost[i] = 0;
}

// Extension primary_spp (rule P3, oracle/oracle_render.c): rtOutput of sub-frame `sub` added to the running sum of the frame's sub-frames, in order; the last
// sub-frame turns the sum into the mean (one multiplication by 1.0f / count), stores it as rtOutput and its PostProcessPS passthrough as the back buffer.
__global__ __launch_bounds__(256) void spp_accumulate_kernel(FrameParams Pv, ViewImages Iv, float4 *sum, int sub, int count) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = P.tileY0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= P.width || y >= P.tileY1 || !row_owned(P, y)) return;
    const size_t i = (size_t)y * (size_t)P.width + x;
    float4 v = reinterpret_cast<const float4 *>(I.output)[i];
    if (sub > 0) { const float4 a = sum[i]; v.x = a.x + v.x; v.y = a.y + v.y; v.z = a.z + v.z; v.w = a.w + v.w; }
    if (sub + 1 < count) { sum[i] = v; return; }
    const float inv = 1.0f / (float)count;
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    reinterpret_cast<float4 *>(I.output)[i] = v;
    store_rgba8(I.final, i, v.x, v.y, v.z, 1.0f);
}

// PostProcessPS.hlsl:13-36 as its own pass: the screen-size back buffer resampled from the render-size output with the static
// sampler of rt64_device.cpp:958-973 (MIN_MAG_MIP_LINEAR, WRAP), plus the motion-blur gather along gFlow.  Only launched when
// the render size differs from the screen size (RT64_VIEW_DESC.resolutionScale) or motionBlurStrength > 0.
DEV int wrapi(int i, int n) { int j = i % n; return j < 0 ? j + n : j; }
DEV f4 sample_output_linear_wrap(const float *img, int w, int h, float u, float v) {
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y), fx = x - x0f, fy = y - y0f;
    const int x0 = wrapi((int)x0f, w), x1 = wrapi((int)x0f + 1, w), y0 = wrapi((int)y0f, h), y1 = wrapi((int)y0f + 1, h);
    const float4 c00 = reinterpret_cast<const float4 *>(img)[(size_t)y0 * w + x0], c10 = reinterpret_cast<const float4 *>(img)[(size_t)y0 * w + x1];
    const float4 c01 = reinterpret_cast<const float4 *>(img)[(size_t)y1 * w + x0], c11 = reinterpret_cast<const float4 *>(img)[(size_t)y1 * w + x1];
    f4 r;
    { const float top = c00.x + fx * (c10.x - c00.x), bot = c01.x + fx * (c11.x - c01.x); r.x = top + fy * (bot - top); }
    { const float top = c00.y + fx * (c10.y - c00.y), bot = c01.y + fx * (c11.y - c01.y); r.y = top + fy * (bot - top); }
    { const float top = c00.z + fx * (c10.z - c00.z), bot = c01.z + fx * (c11.z - c01.z); r.z = top + fy * (bot - top); }
    r.w = 1.0f;
    return r;
}
DEV f2 sample_flow_linear_wrap(const uint16_t *img, int w, int h, float u, float v) {
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y), fx = x - x0f, fy = y - y0f;
    const int x0 = wrapi((int)x0f, w), x1 = wrapi((int)x0f + 1, w), y0 = wrapi((int)y0f, h), y1 = wrapi((int)y0f + 1, h);
    auto ld = [&](int xx, int yy) { const uint32_t p = reinterpret_cast<const uint32_t *>(img)[(size_t)yy * w + xx]; f2 r; r.x = f16_bits_to_f32((uint16_t)(p & 0xFFFFu)); r.y = f16_bits_to_f32((uint16_t)(p >> 16)); return r; };
    const f2 c00 = ld(x0, y0), c10 = ld(x1, y0), c01 = ld(x0, y1), c11 = ld(x1, y1);
    f2 r;
    { const float top = c00.x + fx * (c10.x - c00.x), bot = c01.x + fx * (c11.x - c01.x); r.x = top + fy * (bot - top); }
    { const float top = c00.y + fx * (c10.y - c00.y), bot = c01.y + fx * (c11.y - c01.y); r.y = top + fy * (bot - top); }
    return r;
}
__global__ __launch_bounds__(256) void post_process_kernel(FrameParams Pv, ViewImages Iv) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    const int sw = (int)P.resolution[2], sh = (int)P.resolution[3];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= sw || y >= sh) return;
    // viewport + scissor of the full-screen triangle: the screen, or the rectangles of the first ray-traced instance (rt64_view.cpp:1258-1271,1624-1626)
    const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
    if (x < P.rtScissor[0] || x >= P.rtScissor[2] || y < P.rtScissor[1] || y >= P.rtScissor[3]) return;
    if (!(cx >= P.rtViewport[0]) || !(cx < P.rtViewport[0] + P.rtViewport[2]) || !(cy >= P.rtViewport[1]) || !(cy < P.rtViewport[1] + P.rtViewport[3])) return;
    const float u = (cx - P.rtViewport[0]) / P.rtViewport[2], v = (cy - P.rtViewport[1]) / P.rtViewport[3];       // FullScreenVS interpolant at the pixel centre
    f4 color; bool blurred = false;
    if (P.motionBlurStrength > 0.0f && P.motionBlurSamples > 0) {
        const f2 fl = sample_flow_linear_wrap(I.flow, P.width, P.height, u, v);
        const float flx = fl.x / P.resolution[0], fly = fl.y / P.resolution[1];
        const float flowLength = sqrtf(flx * flx + fly * fly);
        if (flowLength > 1e-6f) {
            const float sampleStep = P.motionBlurStrength / (float)P.motionBlurSamples;
            float sr = 0.0f, sg = 0.0f, sb = 0.0f, sumWeight = 0.0f;
            const float su = u - (flx * P.motionBlurStrength / 2.0f), sv = v - (fly * P.motionBlurStrength / 2.0f);
            for (uint32_t k = 0; k < P.motionBlurSamples; k++) {
                float uu = su + flx * (float)k * sampleStep, vv = sv + fly * (float)k * sampleStep;
                uu = fminf(fmaxf(uu, 0.0f), 1.0f); vv = fminf(fmaxf(vv, 0.0f), 1.0f);
                const f4 c = sample_output_linear_wrap(P.postSource, P.postSourceW, P.postSourceH, uu, vv);
                sr += c.x * 1.0f; sg += c.y * 1.0f; sb += c.z * 1.0f; sumWeight += 1.0f;
            }
            color = mk4(sr / sumWeight, sg / sumWeight, sb / sumWeight, 1.0f);
            blurred = true;
        }
    }
    if (!blurred) color = sample_output_linear_wrap(P.postSource, P.postSourceW, P.postSourceH, u, v);
    store_rgba8(I.final, (size_t)y * (size_t)sw + x, color.x, color.y, color.z, 1.0f);
}

// IndirectRayGen with giSamples == 0 (IndirectRayGen.hlsl:135): every pixel gets ambientBase + ambientNoGI, history 0.
__global__ __launch_bounds__(256) void indirect_constant_kernel(FrameParams Pv, ViewImages Iv, int cur) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = P.tileY0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= P.width || y >= P.tileY1 || !row_owned(P, y)) return;
    const size_t i = (size_t)y * (size_t)P.width + x;
    const float r = P.ambientBaseColor[0] + P.ambientNoGIColor[0], g = P.ambientBaseColor[1] + P.ambientNoGIColor[1], b = P.ambientBaseColor[2] + P.ambientNoGIColor[2];
    store_rgba16f(I.indirectLight[cur], i, r, g, b, 0.0f);
    store_rgba16f(I.filteredIndirect[1], i, r, g, b, 0.0f);
}

// The reflection passes keep their continuation state beside the G-buffer (ViewImages::reflState0 / 1); the reference rewrites the G-buffer itself
// (ReflectionRayGen.hlsl:117-124).  A reader of gShadingPosition / gViewDirection / gShadingNormal / gInstanceId gets the reference's bytes through this
// kernel: every pixel the frame's passes tagged takes its last state (View::applyReflectionState, on readback only).
__global__ __launch_bounds__(256) void apply_reflection_state_kernel(ViewImages I, int width, int y0, int y1, uint32_t frameTag) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= width || y >= y1) return;
    const size_t i = (size_t)y * (size_t)width + x;
    if (I.reflTag[i] != frameTag) return;
    const uint4 s0 = I.reflState0[i], s1 = I.reflState1[i];
    reinterpret_cast<float4 *>(I.shadingPosition)[i] = make_float4(__uint_as_float(s0.x), __uint_as_float(s0.y), __uint_as_float(s0.z), 0.0f);
    reinterpret_cast<uint2 *>(I.viewDirection)[i] = make_uint2(s1.x, s1.y);
    reinterpret_cast<uint2 *>(I.shadingNormal)[i] = make_uint2(s1.z, s1.w);
    I.instanceId[i] = (int32_t)s0.w;
}

__global__ __launch_bounds__(256) void clear_final_kernel(FrameParams Pv, ViewImages Iv) {
    PRef P = *kernel_params(); IRef I = *kernel_images(); (void)Pv; (void)Iv; (void)I;
    if (P.separatePost) {         // the back buffer has the screen size, the frame is not partitioned
        const int sw = (int)P.resolution[2], sh = (int)P.resolution[3];
        const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
        if (x < sw && y < sh) store_rgba8(I.final, (size_t)y * (size_t)sw + x, 0.0f, 0.0f, 0.0f, 1.0f);
        return;
    }
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = P.tileY0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= P.width || y >= P.tileY1 || !row_owned(P, y)) return;
    store_rgba8(I.final, (size_t)y * (size_t)P.width + x, 0.0f, 0.0f, 0.0f, 1.0f);   // cleared back buffer, rt64_device.cpp:996-997
}

}  // namespace

#ifndef RT_ASSUME_SIMPLE
// One spill slab per lane of every workgroup of the largest grid a frame of `width` x `rows` can launch: the persistent kernels use
// at most RT_GRID_BLOCKS workgroups, the one-kernel frame one per 16 x 16 tile.
size_t rt_stack_spill_bytes(int width, int rows) {
    const size_t tiles = (size_t)((width + 15) / 16) * (size_t)((rows + 15) / 16);
    size_t blocks = tiles > (size_t)RT_GRID_BLOCKS ? tiles : (size_t)RT_GRID_BLOCKS;
    if (blocks > RT_MAX_FRAME_GROUPS) blocks = RT_MAX_FRAME_GROUPS;
    blocks += 8;          // the per-wave frame rounds its grid up to whole groups of 8 tiles (32 one-wave workgroups)       // no launch has more workgroups than that (launch_lean_frame, sparse_grid)
    return blocks * RT_BLOCK * (RT_STACK_SPILL_HEADER + RT_STACK_SPILL) * sizeof(uint32_t);      // (a lane's entries + the header in front of them: trace.h)
}
#endif


// Grid of a ray kernel: one persistent workgroup per CU slot (RT_GRID_BLOCKS), or one per tile when the device's share of the
// frame has fewer tiles than that (small frames, a 1/8 strip share): workgroups without a tile only cost launch time.
static unsigned rt_grid(const FrameParams &P) {
    const unsigned all = (unsigned)(P.tileY1 - P.tileY0 + 15) / 16;
    const unsigned strips = all > (unsigned)P.stripRank ? (all - (unsigned)P.stripRank + (unsigned)P.stripCount - 1) / (unsigned)P.stripCount : 0u;
    const unsigned tilesSquare = (unsigned)((P.width + 15) / 16) * strips, tilesRows = (unsigned)((P.width + 31) / 32) * strips * 2u;
    const unsigned tiles = tilesSquare > tilesRows ? tilesSquare : tilesRows;
    return tiles < 1u ? 1u : (tiles < (unsigned)RT_GRID_BLOCKS ? tiles : (unsigned)RT_GRID_BLOCKS);
}
#define LAUNCH_RAY(kernel, ...) do { hipLaunchKernelGGL(kernel, dim3(rt_grid(P)), dim3(RT_BLOCK), 0, s, __VA_ARGS__); return hipGetLastError(); } while (0)
// dynamic LDS of a CACHED kernel: scene cache, plus the light-selection columns when the kernel picks lights
static size_t cached_lds_bytes(const FrameParams &P, bool lights) {
    const size_t slots = (P.lightCount < RT64_MAX_LIGHTS ? P.lightCount : (uint32_t)RT64_MAX_LIGHTS) + 1u;
    return (size_t)P.cacheWords * 16 + (lights ? (slots * RT_BLOCK * 5 + 15) / 16 * 16 : 0);
}
#define LAUNCH_RAY_LDS(kernel, bytes, ...) do { hipLaunchKernelGGL(kernel, dim3(rt_grid(P)), dim3(RT_BLOCK), bytes, s, __VA_ARGS__); return hipGetLastError(); } while (0)

#ifndef RT_ASSUME_SIMPLE
hipError_t launch_scene_cache_image(const GpuInstance *instances, const uint32_t *tlasIndex, const GpuNode *tlasNodes, uint32_t cacheInstances, void *image, bool blasOnly, hipStream_t s) {
    hipLaunchKernelGGL(scene_cache_image_kernel, dim3(1), dim3(RT_BLOCK), 0, s, instances, tlasIndex, tlasNodes, cacheInstances, static_cast<u32x4 *>(image), blasOnly ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_primary_trace(const FrameParams &P, const ViewImages &I, int32_t *hitInstance, bool klist, hipStream_t s) {
    if (klist) LAUNCH_RAY(primary_trace_kernel<true>, P, I, hitInstance);
    if (P.cacheWords) LAUNCH_RAY_LDS((primary_trace_kernel<false, true>), cached_lds_bytes(P, false), P, I, hitInstance);
    LAUNCH_RAY(primary_trace_kernel<false>, P, I, hitInstance);
}
#endif

hipError_t RT_LAUNCHER(launch_primary_shade)(const FrameParams &P, const ViewImages &I, const int32_t *hitInstance, int cur, bool transparentLighting, bool lean, hipStream_t s) {
    RT_ROUTE_SIMPLE(launch_primary_shade_simple(P, I, hitInstance, cur, transparentLighting, lean, s));
    if (transparentLighting) LAUNCH_RAY((primary_shade_kernel<true, true, true>), P, I, hitInstance, cur);
    if (lean) LAUNCH_RAY((primary_shade_kernel<false, false, false>), P, I, hitInstance, cur);
    LAUNCH_RAY((primary_shade_kernel<false, false, true>), P, I, hitInstance, cur);
}
hipError_t RT_LAUNCHER(launch_direct)(const FrameParams &P, const ViewImages &I, int cur, bool lean, hipStream_t s) {
    RT_ROUTE_SIMPLE(launch_direct_simple(P, I, cur, lean, s));
    if (P.cacheWords) {
        if (lean) LAUNCH_RAY_LDS((direct_kernel<false, true>), cached_lds_bytes(P, true), P, I, cur);
        LAUNCH_RAY_LDS((direct_kernel<true, true>), cached_lds_bytes(P, true), P, I, cur);
    }
    if (lean) LAUNCH_RAY(direct_kernel<false>, P, I, cur);
    LAUNCH_RAY(direct_kernel<true>, P, I, cur);
}
// One workgroup per tile: the hardware dispatcher hands the next tile to whichever CU has a free slot (LEAN_WAVES workgroups per
// CU), which together with the bottom-up tile order (geometry first) is a longest-job-first schedule; a resident round of
// persistent workgroups with a static round-robin walk measured 8 % slower on the full frame (181 against 165 us) and keeps every
// register file full until the launch ends, so nothing on another stream (the RCCL gather) can run beside it.
#ifndef PERWAVE_WAVES
#define PERWAVE_WAVES 3        // waves per SIMD of the per-wave (64-thread workgroup) form
#endif
hipError_t RT_LAUNCHER(launch_lean_frame)(const FrameParams &P, const ViewImages &I, int32_t *hitInstance, int cur, bool full, int ownedY0, int ownedY1, unsigned maxGroups, bool perWave, hipStream_t s) {
    RT_ROUTE_SIMPLE(launch_lean_frame_simple(P, I, hitInstance, cur, full, ownedY0, ownedY1, maxGroups, perWave, s));
    const unsigned strips = (unsigned)(P.tileY1 - P.tileY0 + 15) / 16, owned = strips > (unsigned)P.stripRank ? (strips - (unsigned)P.stripRank + (unsigned)P.stripCount - 1) / (unsigned)P.stripCount : 0u;
    const unsigned tiles = (unsigned)((P.width + 15) / 16) * owned;
    // ... up to 8192 workgroups; bigger frames give every workgroup ceil(tiles / 8192) tiles (round-robin, same bottom-up order), which
    // amortises the scene-cache fill again (1440p: 2 tiles per workgroup, 4K: 4)
    if (maxGroups < 1u || maxGroups > RT_MAX_FRAME_GROUPS) maxGroups = RT_MAX_FRAME_GROUPS;
    const size_t lds = cached_lds_bytes(P, true);       // the scene cache (when the frame has one) + the light-candidate columns of this frame's light count
    if (P.cacheWords) {
        const unsigned perGroup = (tiles + maxGroups - 1u) / maxGroups, grid = tiles < 1u ? 1u : (tiles + perGroup - 1u) / perGroup;
        if (full) hipLaunchKernelGGL((lean_frame_kernel<true, true, LEAN_WAVES>), dim3(grid), dim3(RT_BLOCK), lds, s, P, I, hitInstance, cur, ownedY0, ownedY1);
        else hipLaunchKernelGGL((lean_frame_kernel<true, false, LEAN_WAVES>), dim3(grid), dim3(RT_BLOCK), lds, s, P, I, hitInstance, cur, ownedY0, ownedY1);
    }
    else if (perWave) {
        // one wave per workgroup: trips = 32 per 8 tiles (wave_tile_of), up to 4 x maxGroups workgroups, a multiple of 32 so that every trip of a workgroup stays on its XCD's tiles
        const unsigned trips = ((tiles + 7u) / 8u) * 32u, cap = maxGroups * 4u, perGroup = (trips + cap - 1u) / cap;
        unsigned grid = trips < 1u ? 32u : (trips + perGroup - 1u) / perGroup;
        grid = (grid + 31u) / 32u * 32u;
        const size_t ldsWave = cached_lds_bytes(P, true) / (RT_BLOCK / 64) + 16;
        if (full) hipLaunchKernelGGL((lean_frame_kernel<false, true, PERWAVE_WAVES, 64>), dim3(grid), dim3(64), ldsWave, s, P, I, hitInstance, cur, ownedY0, ownedY1);
        else hipLaunchKernelGGL((lean_frame_kernel<false, false, PERWAVE_WAVES, 64>), dim3(grid), dim3(64), ldsWave, s, P, I, hitInstance, cur, ownedY0, ownedY1);
    }
    else {
        const unsigned perGroup = (tiles + maxGroups - 1u) / maxGroups, grid = tiles < 1u ? 1u : (tiles + perGroup - 1u) / perGroup;
        if (full) hipLaunchKernelGGL((lean_frame_kernel<false, true, LEAN_WAVES>), dim3(grid), dim3(RT_BLOCK), lds, s, P, I, hitInstance, cur, ownedY0, ownedY1);
        else hipLaunchKernelGGL((lean_frame_kernel<false, false, LEAN_WAVES>), dim3(grid), dim3(RT_BLOCK), lds, s, P, I, hitInstance, cur, ownedY0, ownedY1);
    }
    return hipGetLastError();
}
hipError_t RT_LAUNCHER(launch_indirect)(const FrameParams &P, const ViewImages &I, int cur, bool writeFiltered, bool klist, int walk, unsigned groups, int writeGuide, hipStream_t s) {
    RT_ROUTE_SIMPLE(launch_indirect_simple(P, I, cur, writeFiltered, klist, walk, groups, writeGuide, s));
    const bool second = P.giBounces >= 2u;
    if (klist && second) LAUNCH_RAY((indirect_kernel<true, true>), P, I, cur, writeFiltered ? 1 : 0);
    if (klist) LAUNCH_RAY(indirect_kernel<true>, P, I, cur, writeFiltered ? 1 : 0);
    if ((P.giSamples == 0 || !I.bounceRecords) && second) LAUNCH_RAY((indirect_kernel<false, true>), P, I, cur, writeFiltered ? 1 : 0);
    if (P.giSamples == 0 || !I.bounceRecords) LAUNCH_RAY(indirect_kernel<false>, P, I, cur, writeFiltered ? 1 : 0);
    // grid of the bounce kernels: one workgroup per tile up to `groups` workgroups (0 = RT_MAX_BOUNCE_GROUPS), then tiles b, b + grid, ... (bounce_tile_of)
    unsigned grid = rt_grid(P);
    {
        const unsigned all = (unsigned)(P.tileY1 - P.tileY0 + 15) / 16, strips = all > (unsigned)P.stripRank ? (all - (unsigned)P.stripRank + (unsigned)P.stripCount - 1) / (unsigned)P.stripCount : 0u;
        const unsigned tiles = (unsigned)((P.width + 15) / 16) * strips, most = P.cacheWords ? RT_MAX_BOUNCE_GROUPS : RT_MAX_FRAME_GROUPS /* the HBM spill slab of the traversal stacks is sized for that many workgroups */, cap = groups && groups < most ? groups : most;
        const unsigned per = tiles > cap ? (tiles + cap - 1) / cap : 1u;
        if (walk != BOUNCE_WALK_REFILL) grid = tiles < 1u ? 1u : (tiles + per - 1) / per;
    }
    if (walk == BOUNCE_WALK_REFILL) hipLaunchKernelGGL(bounce_trace_refill_kernel, dim3(grid), dim3(RT_BLOCK), 0, s, P, I);
    else if (walk == BOUNCE_WALK_SPLIT && P.cacheWords) hipLaunchKernelGGL(bounce_trace_split_kernel<true>, dim3(grid), dim3(RT_BLOCK), cached_lds_bytes(P, false), s, P, I);
    else if (walk == BOUNCE_WALK_SPLIT) hipLaunchKernelGGL(bounce_trace_split_kernel<false>, dim3(grid), dim3(RT_BLOCK), 0, s, P, I);
    else if (P.cacheWords) hipLaunchKernelGGL(bounce_trace_plain_kernel<true>, dim3(grid), dim3(RT_BLOCK), cached_lds_bytes(P, false), s, P, I);
    else hipLaunchKernelGGL(bounce_trace_plain_kernel<false>, dim3(grid), dim3(RT_BLOCK), 0, s, P, I);
    // same grid for the three kernels: workgroup b shades the segments workgroup b of bounce_trace filled (lengths stay on the device)
    if (P.cacheWords && second) hipLaunchKernelGGL((bounce_hit_kernel<true, true>), dim3(grid), dim3(RT_BLOCK), cached_lds_bytes(P, true), s, P, I);
    else if (second) hipLaunchKernelGGL((bounce_hit_kernel<false, true>), dim3(grid), dim3(RT_BLOCK), 0, s, P, I);
    else if (P.cacheWords) hipLaunchKernelGGL(bounce_hit_kernel<true>, dim3(grid), dim3(RT_BLOCK), cached_lds_bytes(P, true), s, P, I);
    else hipLaunchKernelGGL(bounce_hit_kernel<false>, dim3(grid), dim3(RT_BLOCK), 0, s, P, I);
    if (walk == BOUNCE_WALK_REFILL) hipLaunchKernelGGL(bounce_miss_kernel, dim3(grid), dim3(RT_BLOCK), 0, s, P, I);      // the other walks finish their misses themselves
    dim3 rgrid((unsigned)(P.width + 31) / 32, (unsigned)(P.tileY1 - P.tileY0 + 7) / 8);
    hipLaunchKernelGGL(bounce_resolve_kernel, rgrid, dim3(256), 0, s, P, I, cur, writeFiltered ? 1 : 0, writeGuide);
    return hipGetLastError();
}
#ifndef RT_ASSUME_SIMPLE
hipError_t launch_indirect_constant(const FrameParams &P, const ViewImages &I, int cur, hipStream_t s) {
    dim3 grid((unsigned)(P.width + 31) / 32, (unsigned)(P.tileY1 - P.tileY0 + 7) / 8);
    hipLaunchKernelGGL(indirect_constant_kernel, grid, dim3(256), 0, s, P, I, cur);
    return hipGetLastError();
}
#endif

// Refraction / reflection only have work where the primary hit has the factor: most tiles return at once, so these two take one
// workgroup per tile (up to 8192) and let the dispatcher balance them (C5 reflection: 0.315 -> 0.26 ms against the persistent grid).
static unsigned sparse_grid(const FrameParams &P) {
    const unsigned all = (unsigned)(P.tileY1 - P.tileY0 + 15) / 16;
    const unsigned strips = all > (unsigned)P.stripRank ? (all - (unsigned)P.stripRank + (unsigned)P.stripCount - 1) / (unsigned)P.stripCount : 0u;
    const unsigned tiles = (unsigned)((P.width + 15) / 16) * strips;
    return tiles < 1u ? 1u : (tiles < RT_MAX_FRAME_GROUPS ? tiles : RT_MAX_FRAME_GROUPS);
}
hipError_t RT_LAUNCHER(launch_refraction)(const FrameParams &P, const ViewImages &I, bool klist, hipStream_t s) {
    RT_ROUTE_SIMPLE(launch_refraction_simple(P, I, klist, s));
    if (klist) hipLaunchKernelGGL(refraction_kernel<true>, dim3(sparse_grid(P)), dim3(RT_BLOCK), 0, s, P, I);
    else hipLaunchKernelGGL(refraction_kernel<false>, dim3(sparse_grid(P)), dim3(RT_BLOCK), 0, s, P, I);
    return hipGetLastError();
}
hipError_t RT_LAUNCHER(launch_reflection)(const FrameParams &P, const ViewImages &I, bool klist, int pass, bool last, int parity, hipStream_t s) {
    RT_ROUTE_SIMPLE(launch_reflection_simple(P, I, klist, pass, last, parity, s));
    if (klist) hipLaunchKernelGGL(reflection_kernel<true>, dim3(sparse_grid(P)), dim3(RT_BLOCK), 0, s, P, I, pass, last ? 1 : 0, parity);
    else if (P.cacheWords) hipLaunchKernelGGL((reflection_kernel<false, true>), dim3(sparse_grid(P)), dim3(RT_BLOCK), cached_lds_bytes(P, true), s, P, I, pass, last ? 1 : 0, parity);
    else hipLaunchKernelGGL(reflection_kernel<false>, dim3(sparse_grid(P)), dim3(RT_BLOCK), 0, s, P, I, pass, last ? 1 : 0, parity);
    return hipGetLastError();
}

#ifndef RT_ASSUME_SIMPLE
hipError_t launch_gaussian(const uint16_t *in, uint16_t *out, int width, int height, int y0, int y1, hipStream_t s) {
    dim3 grid((unsigned)(width + 31) / 32, (unsigned)(y1 - y0 + 7) / 8);
    hipLaunchKernelGGL(gaussian_kernel, grid, dim3(256), 0, s, in, out, width, height, y0, y1);
    return hipGetLastError();
}
hipError_t launch_compose_post(const FrameParams &P, const ViewImages &I, int cur, bool lean, bool writeFinal, hipStream_t s) {
    dim3 grid((unsigned)(P.width + 31) / 32, (unsigned)(P.tileY1 - P.tileY0 + 7) / 8);
    if (lean) hipLaunchKernelGGL(compose_post_kernel<true>, grid, dim3(256), 0, s, P, I, cur, writeFinal ? 1 : 0);
    else hipLaunchKernelGGL(compose_post_kernel<false>, grid, dim3(256), 0, s, P, I, cur, writeFinal ? 1 : 0);
    return hipGetLastError();
}
unsigned lean_frame_tiles(const FrameParams &P) {
    const unsigned strips = (unsigned)(P.tileY1 - P.tileY0 + 15) / 16, owned = strips > (unsigned)P.stripRank ? (strips - (unsigned)P.stripRank + (unsigned)P.stripCount - 1) / (unsigned)P.stripCount : 0u;
    return (unsigned)((P.width + 15) / 16) * owned;
}
hipError_t launch_tile_order(uint32_t *cost, uint32_t *order, uint32_t tiles, hipStream_t s) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, cost, order, tiles);
    return hipGetLastError();
}
hipError_t launch_spp_accumulate(const FrameParams &P, const ViewImages &I, float *sum, int sub, int count, hipStream_t s) {
    dim3 grid((unsigned)(P.width + 31) / 32, (unsigned)(P.tileY1 - P.tileY0 + 7) / 8);
    hipLaunchKernelGGL(spp_accumulate_kernel, grid, dim3(256), 0, s, P, I, reinterpret_cast<float4 *>(sum), sub, count);
    return hipGetLastError();
}
hipError_t launch_post_process(const FrameParams &P, const ViewImages &I, hipStream_t s) {
    dim3 grid((unsigned)((int)P.resolution[2] + 31) / 32, (unsigned)((int)P.resolution[3] + 7) / 8);
    hipLaunchKernelGGL(post_process_kernel, grid, dim3(256), 0, s, P, I);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void stack_slab_init_kernel(uint32_t *slab, size_t lanes, const uint32_t *flag) {
    const size_t lane = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (lane >= lanes) return;
    const uint64_t p = reinterpret_cast<uint64_t>(flag);
    uint32_t *h = slab + lane * (RT_STACK_SPILL_HEADER + RT_STACK_SPILL);
    h[0] = (uint32_t)p; h[1] = (uint32_t)(p >> 32);
}
hipError_t launch_stack_slab_init(uint32_t *slab, size_t lanes, const uint32_t *flagDevicePointer, hipStream_t s) {
    if (!lanes) return hipSuccess;
    hipLaunchKernelGGL(stack_slab_init_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, slab, lanes, flagDevicePointer);
    return hipGetLastError();
}
hipError_t launch_apply_reflection_state(const ViewImages &I, int width, int y0, int y1, uint32_t frameTag, hipStream_t s) {
    if (y1 <= y0) return hipSuccess;
    dim3 grid((unsigned)(width + 31) / 32, (unsigned)(y1 - y0 + 7) / 8);
    hipLaunchKernelGGL(apply_reflection_state_kernel, grid, dim3(256), 0, s, I, width, y0, y1, frameTag);
    return hipGetLastError();
}
hipError_t launch_clear_final(const FrameParams &P, const ViewImages &I, hipStream_t s) {
    dim3 grid((unsigned)(P.width + 31) / 32, (unsigned)(P.tileY1 - P.tileY0 + 7) / 8);
    if (P.separatePost) grid = dim3((unsigned)((int)P.resolution[2] + 31) / 32, (unsigned)((int)P.resolution[3] + 7) / 8);
    hipLaunchKernelGGL(clear_final_kernel, grid, dim3(256), 0, s, P, I);
    return hipGetLastError();
}
#endif

