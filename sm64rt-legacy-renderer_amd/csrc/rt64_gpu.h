// rt64_gpu.h -- plain structures shared by the host side and the HIP kernels of librt64.so.
//
// HBM layout of the render path (see DESIGN.md "Data layout"):
//   * per mesh      : raw interleaved vertex buffer + u32 index buffer (as handed to RT64_SetMesh),
//                     BLAS = GpuNode[max(n-1,1)] (64 B each) + GpuTri[n] (48 B each, Morton order) + BlasHeader
//   * per frame     : GpuInstance[] (transforms, material, combiner, pointers), TLAS GpuNode[], GpuTexture[] table,
//                     RT64_LIGHT[]
//   * per view      : G-buffer images in the reference's storage formats (rt64_view.cpp:152-241)
#pragma once
#include <stdint.h>
#include "../../include/rt64.h"

#define RT64_LEAF_BIT 0x80000000u
#define RT64_NO_CHILD 0xFFFFFFFFu
#define RT64_MAX_MIPS 16
#define RT64_MAX_LIGHTS 16          // Lights.hlsli:25
#define RT64_MAX_HIT_QUERIES 16     // GlobalHitBuffers.hlsli:8

// LBVH inner node, 64 B = four 16-byte loads.  Both children's boxes live in the parent so one fetch decides both.
// Child ids: bit 31 = leaf (low bits = slot in Morton order), RT64_NO_CHILD = absent (single-leaf tree).
struct alignas(16) GpuNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t left, right, parent, pad;   // pad: the other end j of the node's leaf range [min(i, j), max(i, j)] (Karras' split search; 0 in a single-leaf tree)
};
static_assert(sizeof(GpuNode) == 64, "GpuNode");

// BLAS leaf: one triangle, positions only (Moller-Trumbore operands), 48 B = three 16-byte loads.
struct alignas(16) GpuTri {
    float v0[3]; uint32_t prim;      // prim = PrimitiveIndex() (triangle number in the index buffer)
    float v1[3]; uint32_t pad1;
    float v2[3]; uint32_t pad2;
};
static_assert(sizeof(GpuTri) == 48, "GpuTri");
// The one-step-per-trip walk (trace_ray_stepwise, trace.h) -- the walk of every ray kernel that does not hold the LDS scene cache -- fetches a leaf's 48-byte
// record with the same four 16-byte loads as a 64-byte node: 16 bytes past the record.  EVERY array reached through GpuInstance::tris therefore ends with
// this much readable slack behind its last record (gpu_tri_array_bytes is the one place that says how much to allocate).
#define GPU_TRI_FETCH_SLACK_BYTES (sizeof(GpuNode) - sizeof(GpuTri))
static_assert(GPU_TRI_FETCH_SLACK_BYTES == 16 && GPU_TRI_FETCH_SLACK_BYTES <= sizeof(GpuTri), "a triangle is fetched as one node-sized record");
inline size_t gpu_tri_array_bytes(size_t triangles) { return triangles * sizeof(GpuTri) + GPU_TRI_FETCH_SLACK_BYTES; }

struct BlasHeader {
    float bmin[3]; uint32_t count;
    float bmax[3]; uint32_t depth;   // inner nodes on the longest root-to-leaf path (single-workgroup builder; the multi-kernel path stores 255 = "deep")
};

struct GpuTexture {
    const uint8_t *texels;           // RGBA8, mips packed back to back
    uint32_t width, height, mips, pow2;  // pow2: width and height are powers of two (every mip then is): wrap / mirror by mask
    uint32_t mipOffset[RT64_MAX_MIPS];   // in texels
};

// Decoded colour combiner + vertex layout (reference: rt64_shader.cpp:32-96).
struct GpuCombiner {
    int8_t c[2][4];
    int8_t inputCount, useTex0, useTex1, vertexUV;
    int8_t doSingle[2], doMultiply[2], doMix[2];
    int8_t colorAlphaSame, optAlpha, optTextureEdge, optNoise;
    int16_t vertexSize, normalOffset, uvOffset, inputOffset[4];
};

enum : uint32_t {
    GPU_INST_CULL_DISABLE = 1u << 0,     // RT64_INSTANCE_DISABLE_BACKFACE_CULLING
    GPU_INST_OPAQUE       = 1u << 1,     // static opacity rule O1: every hit stores alpha 255
    GPU_INST_NORMAL_MAP   = 1u << 2,     // RT64_SHADER_NORMAL_MAP_ENABLED
    GPU_INST_SPECULAR_MAP = 1u << 3,     // RT64_SHADER_SPECULAR_MAP_ENABLED
    GPU_INST_SHADOW_OPAQUE = 1u << 4,    // rule O2: the first shadow-ray hit saturates (no opt_alpha, or alpha * shadowAlphaMultiplier provably >= 0.999)
};

struct alignas(16) GpuInstance {
    float objectToWorld[16];             // row-major, row-vector convention (p' = p * M)
    float objectToWorldNormal[16];
    float objectToWorldPrevious[16];
    float worldToObject[16];
    const GpuNode *nodes;
    const GpuTri *tris;
    const uint8_t *vertices;
    const uint32_t *indices;
    const BlasHeader *header;
    RT64_MATERIAL material;
    GpuCombiner cc;
    int32_t texDiffuse, texNormal, texSpecular;
    uint32_t filter, hAddr, vAddr;
    uint32_t flags;
    uint32_t triCount;
    uint32_t meshVersion;                // bumps on every RT64_SetMesh: part of the frame-table cache key
    uint32_t cacheNodeOffset;            // LDS scene cache: where this instance's BLAS nodes start, in 16-byte words (FrameParams::cacheWords != 0)
};

// One raster (non-ray-traced) instance of the frame, draw order: background list, then foreground list (rt64_view.cpp:1138-1147).
struct alignas(16) GpuRasterInstance {
    const uint8_t *vertices;             // vertex buffer as passed to RT64_SetMesh (position float4 at offset 0 = clip space)
    const uint32_t *indices;
    uint32_t vertexStride, triCount;
    uint32_t firstTri;                   // prefix sum of triCount inside its list
    GpuCombiner cc;
    int32_t texDiffuse;
    uint32_t filter, hAddr, vAddr;
    int32_t scissorRect[4], viewportRect[4];     // RT64_RECT x, y, w, h (origin bottom-left); w or h <= 0 = unset
    uint32_t meshVersion;
    uint32_t texSerial;                  // identity of the diffuse texture object (texDiffuse is only its slot in this frame's table)
};

// Constant block of one frame (reference: GlobalParams.hlsli:8-43 / rt64_view.cpp:961-1028), passed by value.
struct FrameParams {
    float view[16], viewI[16], prevViewI[16], projection[16], projectionI[16], viewProj[16], prevViewProj[16];
    float cameraU[4], cameraV[4], cameraW[4];
    float viewport[4], resolution[4];
    float ambientBaseColor[4], ambientNoGIColor[4], eyeLightDiffuseColor[4], eyeLightSpecularColor[4];
    float skyDiffuseMultiplier[4], skyHSLModifier[4];
    float pixelJitter[2];
    float skyYawOffset, giDiffuseStrength, giSkyStrength, motionBlurStrength;
    int32_t skyPlaneTexIndex;
    uint32_t randomSeed, diSamples, giSamples, diReproject, giReproject, binaryLockMask, maxLights, motionBlurSamples;
    uint32_t visualizationMode, frameCount;
    // --- additions of this implementation ---
    int32_t width, height;               // render size
    int32_t tileY0, tileY1;              // row range owned by this device
    int32_t stripRank, stripCount;       // interleaved 16-row strips inside the range (count 1 = all)
    float maxDepthBias;
    GpuTexture background;               // gBackground: raster background target (screen size RGBA8); texels == nullptr: no background instances
    float rtViewport[4]; int32_t rtScissor[4];   // rectangle the ray-traced picture is drawn into (x, y, w, h / left, top, right, bottom; screen pixels)
    // LDS scene cache (trace.h): when the TLAS + every BLAS node array + one 64-byte record per instance fit in LDS next to the
    // traversal stacks, the ray kernels copy them in once per workgroup and walk from there.  cacheWords = size in 16-byte words, 0 = off.
    uint32_t cacheWords, cacheInstances;
    const void *cacheImage;              // the cache's contents as one flat array in HBM (cacheWords x 16 B, built by scene_cache_image_kernel when the tables change): a workgroup fills its LDS copy with independent loads
    uint32_t simpleKernels;              // 1: every texture of the frame has power-of-two sizes and every instance is shadow-opaque: the launchers take the kernels of passes_simple.hip
    uint32_t separatePost;               // 1: render size != screen size or motion blur on -> PostProcessPS runs as post_process_kernel
    const float *postSource; int32_t postSourceW, postSourceH;      // image PostProcessPS samples: rtOutput (render size), or rtOutputUpscaled (screen size) behind an upscaler (rt64_view.cpp:800-801)
    // Foreground (HUD) raster list folded into the one-kernel lean frame: table + triangle records of raster.hip, 0 triangles = not folded.
    const GpuRasterInstance *rasterFg; const void *rasterFgTris; uint32_t rasterFgCount, rasterFgPad;
    uint32_t *finalPacked;               // RT64_SetDeviceGatherTarget: the owned rows of the back buffer, packed strip after strip (nullptr = off)
    float skyBase[4];                    // ComputeSkyPlaneUV: base u, base v, 0.25 * ratioDivision, 0.25
    // Level 0 of the sky plane once more, in 4 x 4-texel tiles (64 B each): the environment lookups of bounce / reflection rays are random
    // accesses, and a bilinear footprint then touches 1.56 cache lines on average instead of 2.1 (nullptr: not a power-of-two texture)
    const uint32_t *skyTiled; uint32_t skyTiledLog2W, skyTiledLog2H;
    uint32_t lightCount, instanceCount, countTraversal;
    // Longest-first order of the one-kernel frame's tiles on scenes that walk from HBM (one-wave workgroups): tileOrder[slot] = tile (bottom-up number) by last frame's
    // cost, most expensive first (nullptr: slot = tile); tileCost[tile] = the most node + triangle visits any lane of the tile made this frame (atomicMax; nullptr: not recorded)
    uint32_t *tileCost; const uint32_t *tileOrder;
    uint32_t giBounces;                  // extension (device option gi_bounces): 2 = a GI ray that resolves to a surface sends a second ray from there (rules B1-B3, oracle/oracle_render.c); otherwise the reference's one bounce
    const GpuInstance *instances;
    const GpuNode *tlasNodes;
    const uint32_t *tlasIndex;           // TLAS leaf slot -> instance
    const GpuTexture *textures;
    const RT64_LIGHT *lights;
    const uint8_t *blueNoise;            // 512x512 RGBA8
    uint32_t *traversalStack;            // overflow stack, per resident lane
    uint4 *tileTiming;                   // profiling aid (device option tile_timing), nullptr = off: two records per wave of the one-kernel frame: at its start { 100 MHz chip-wide clock, shader clock, HW_ID, 1 } and at its end { clock, shader clock, 0, 1 }
    unsigned long long *counters;        // [0] nodes [1] triangles [2] primary rays [3] shadow rays [4] indirect [5] reflection [6] refraction
};

struct BounceRadiance { float r, g, b; };

// G-buffer images of one view (device pointers), in the reference's formats.
struct ViewImages {
    uint16_t *viewDirection;             // RGBA16F
    float *shadingPosition;              // RGBA32F
    uint16_t *shadingNormal, *shadingSpecular;   // RGBA16F
    uint8_t *diffuse;                    // RGBA8
    int32_t *instanceId, *firstInstanceId;
    uint16_t *directLight[2], *indirectLight[2], *filteredDirect[2], *filteredIndirect[2];   // RGBA16F
    uint16_t *reflection, *refraction, *transparent;   // RGBA16F
    uint16_t *flow;                      // RG16F
    uint8_t *reactiveMask, *lockMask;    // R8
    uint16_t *normal[2];                 // RGBA16F
    float *depth[2];                     // R32F
    float *output;                       // RGBA32F (rtOutput)
    uint8_t *final;                      // RGBA8 back buffer
    uint32_t *primaryHit;                // RGBA32UI: t, u, v bits, instance << 24 | primitive
    float *moments[2];                   // SVGF: RG32F luminance moments
    uint4 *bounceRecords;                // IndirectRayGen wavefront: 2 x uint4 per (GI sample, pixel), sized for bounceSamples
    uint32_t *bounceLists, *bounceCounts; // ids of the traced rays compacted by outcome: [0, cap) hits, [cap, 2 cap) misses; counts[2]
    BounceRadiance *bounceResults;       // radiance of every (GI sample, pixel), 12 bytes (round 3: was a float4 with an unused lane -- 4 x 16 B read by the resolve and written by the walk / hit kernels per pixel of C5)
    uint4 *svgfGuide;                    // SVGF: 16-B guide record per pixel (normal 3 x f16, valid, depth, depth gradient)
    uint32_t *svgfYoung;                 // SVGF: [row][32-pixel segment] != 0: the segment has a pixel with fewer than 4 frames of history (set by bounce_resolve_kernel when it also writes the filter input, read and cleared by svgf_variance_kernel, which then only runs where one is set)
    // Continuation state of the reflection passes (ReflectionRayGen.hlsl:117-124 rewrites gShadingPosition / gViewDirection / gShadingNormal / gInstanceId in place so that
    // its next pass continues from the mirrored hit).  Kept beside the G-buffer here -- nobody but the next reflection pass and a debug readback reads the rewritten
    // values -- so that the passes touch no image another pass of the frame reads and can run on a second stream from the moment the G-buffer exists:
    // [0] = position.xyz + instance id bits, [1] = view direction RGBA16F | shading normal RGBA16F; reflTag[i] = frame tag of the last write (readback folds the
    // tagged pixels back into the G-buffer: View::applyReflectionState).  nullptr until a frame has a reflective material.
    uint4 *reflState0, *reflState1; uint32_t *reflTag;
    uint32_t *reflectFlags;              // [2][4]: does any pixel go on to reflection pass p of the frame of this parity?  (set by pass p - 1, the other parity cleared by the frame's last pass; a later pass with nothing to do ends at once)
    // Per-pixel sorted hit list (k-buffer), [RT64_MAX_HIT_QUERIES + 1][pixels]; allocated only while some instance is not
    // provably opaque.  The reference keeps 17 x 34 B per pixel for every frame (rt64_view.cpp:237-241).
    uint4 *klistA;                       // sort key (t - depthBias) bits, u bits, v bits, primitive
    uint2 *klistB;                       // t bits, instance
    uint32_t *klistCount;                // payload.nhits of the primary ray (per pixel)
};

// ---- image-tile partition of a frame over `count` devices -------------------------------------------------------------
// bands == 0: interleaved 16-row strips -- rank r owns strips r, r + count, ... (RT64_SetDeviceInterleave); bands == 1: contiguous bands
// of ceil(height / count) rows (RT64_SetDeviceTile; frames with a spatial filter).  A rank's buffer holds its rows packed in ascending order.
#if defined(__HIPCC__)
#define RT64_HD __host__ __device__ inline
#else
#define RT64_HD inline
#endif
#define RT64_GATHER_MAX_RANKS 64
struct GatherLayout { int height, count, mode; int starts[RT64_GATHER_MAX_RANKS + 1]; };   // mode 2 (cost-balanced bands): rank r owns rows [starts[r], starts[r + 1])
RT64_HD int gather_band_rows(int height, int count) { return (height + count - 1) / count; }
// the rank that owns frame row y, and (*packed) the row's position in that rank's packed buffer
RT64_HD int gather_row_owner(const GatherLayout &L, int y, int *packed) {
    if (L.count <= 1) { *packed = y; return 0; }
    if (L.mode == 2) { int r = 0; while (r + 1 < L.count && y >= L.starts[r + 1]) r++; *packed = y - L.starts[r]; return r; }
    if (L.mode == 1) { const int b = gather_band_rows(L.height, L.count); *packed = y % b; return y / b; }
    const int strip = y / 16;
    *packed = (strip / L.count) * 16 + (y & 15);
    return strip % L.count;
}
// rows rank `rank` owns
RT64_HD int gather_owned_rows(const GatherLayout &L, int rank) {
    if (L.count <= 1) return L.height;
    if (L.mode == 2) return L.starts[rank + 1] - L.starts[rank];
    if (L.mode == 1) { const int b = gather_band_rows(L.height, L.count), y0 = rank * b, y1 = (rank + 1) * b; return (y1 < L.height ? y1 : L.height) - (y0 < L.height ? y0 : L.height); }
    int rows = 0;
    for (int y = rank * 16; y < L.height; y += L.count * 16) rows += (y + 16 <= L.height ? 16 : L.height - y);
    return rows;
}
// rows of the largest share (what every rank's slot of the bucket on rank 0 is sized for)
RT64_HD int gather_max_owned_rows(const GatherLayout &L) {
    if (L.mode == 2) { int m = 0; for (int r = 0; r < L.count; r++) { const int n = L.starts[r + 1] - L.starts[r]; m = n > m ? n : m; } return m; }
    return L.mode == 1 ? gather_band_rows(L.height, L.count) : (((L.height + 15) / 16 + L.count - 1) / L.count) * 16;
}
RT64_HD GatherLayout gather_layout(int height, int count, int mode) { GatherLayout L; L.height = height; L.count = count < 1 ? 1 : count; L.mode = mode; for (int i = 0; i <= RT64_GATHER_MAX_RANKS; i++) L.starts[i] = 0; return L; }

#define RT_COUNTER_STRIPES 64          // copies of the counter block, chosen by workgroup number: the per-wave atomics of a counting frame spread over 64 lines
enum { CTR_NODES = 0, CTR_TRIS, CTR_PRIMARY, CTR_SHADOW, CTR_INDIRECT, CTR_REFLECTION, CTR_REFRACTION,
       CTR_PASS_BASE,                    // then {nodes, triangles} per pass:
       CTR_COUNT = CTR_PASS_BASE + 2 * 6 };
enum { PASS_PRIMARY_TRACE = 0, PASS_PRIMARY_SHADE, PASS_DIRECT, PASS_INDIRECT, PASS_REFRACTION, PASS_REFLECTION };
