// kernels.h -- host-callable launchers of the HIP kernels in librt64.so.
#pragma once
#include <hip/hip_runtime.h>
#include "rt64_gpu.h"

// ---- lbvh.hip ---------------------------------------------------------------------------------------------------
#define LBVH_SMALL_MAX 4096u          // leaves handled by the single-workgroup LDS builder
enum { LBVH_MODE_TRIANGLES = 0, LBVH_MODE_INSTANCES = 1 };

struct LbvhArgs {
    int mode, refit;
    uint32_t n;
    // LBVH_MODE_TRIANGLES
    const uint8_t *vertices; uint32_t vertexStride; const uint32_t *indices;
    // LBVH_MODE_INSTANCES
    const GpuInstance *instances;
    // outputs
    GpuNode *nodes;                   // max(n-1, 1)
    GpuTri *tris;                     // n (triangles mode)
    BlasHeader *header;
    uint32_t *sortedIndex, *morton, *leafParent;   // n each
    // scratch for the large path (allocated by the caller when n > LBVH_SMALL_MAX)
    void *scratch; size_t scratchBytes;
};

size_t lbvh_small_lds_bytes(uint32_t n);
size_t lbvh_large_scratch_bytes(uint32_t n);
hipError_t lbvh_launch(const LbvhArgs &args, hipStream_t stream);
hipError_t lbvh_launch_large(const LbvhArgs &args, hipStream_t stream);
hipError_t lbvh_launch_batch(const LbvhArgs *deviceArgs, uint32_t count, uint32_t maxN, hipStream_t stream);   // one workgroup per tree, n <= LBVH_SMALL_MAX each

// ---- bc7.hip ------------------------------------------------------------------------------------------------------
// Decode `blocksX * blocksY` BC7 blocks into an RGBA8 image of width x height texels.
// RGBA8 image (row-major) -> the same texels in 4 x 4 tiles of 64 bytes (width, height multiples of 4).
hipError_t tile_texture_launch(const uint8_t *rgba, uint32_t *tiled, uint32_t width, uint32_t height, hipStream_t stream);
hipError_t bc7_decode_launch(const uint8_t *blocks, uint8_t *rgba, uint32_t width, uint32_t height, hipStream_t stream);

// ---- passes.hip ---------------------------------------------------------------------------------------------------
#define RT_CACHE_MAX_WORDS 1536       // LDS scene cache, nodes + instance records: at most 24 KB next to the 8 KB stack (int16 entries) and the light columns of a workgroup, four workgroups per CU
#define RT_STACK_LDS 24               // traversal stack entries per lane in LDS of the kernels without the scene cache; deeper levels go to the HBM spill slab
#define RT_STACK_LDS_CACHED 16        // traversal stack entries (LDS only, no spill path) of the kernels that hold the LDS scene cache: a power of two
#define RT_STACK_SPILL 84             // entries per lane in the HBM slab behind the LDS entries
#define RT_STACK_SPILL_HEADER 2       // uint32 words in front of EVERY lane's entries: the address of the overflow word (host-pinned memory); a lane's slab is RT_STACK_SPILL_HEADER + RT_STACK_SPILL words
hipError_t launch_stack_slab_init(uint32_t *slab, size_t lanes, const uint32_t *flagDevicePointer, hipStream_t s);      // writes every lane's header
#define RT_GRID_BLOCKS 2048           // persistent grid of every ray kernel (8 workgroups of 256 per CU)
#define RT_TIMING_WAVES ((8192u + 8u) * 4u)   // waves the tile-timing buffer has records for (profiling aid)
#define RT_MAX_BOUNCE_GROUPS 65536u    // largest grid of the bounce kernels (one workgroup per 16 x 16 tile up to 4K and beyond; hit-list segments and counts are sized for it)
#define RT_MAX_FRAME_GROUPS 8192u     // largest grid of the one-workgroup-per-tile kernels (bigger frames give every workgroup a few tiles)
size_t rt_stack_spill_bytes(int width, int rows);        // bytes FrameParams::traversalStack needs for a frame of that size

hipError_t launch_scene_cache_image(const GpuInstance *instances, const uint32_t *tlasIndex, const GpuNode *tlasNodes, uint32_t cacheInstances, void *image, bool blasOnly, hipStream_t s);   // LDS scene cache contents, flat (FrameParams::cacheImage)
hipError_t launch_primary_trace(const FrameParams &P, const ViewImages &I, int32_t *hitInstance, bool klist, hipStream_t s);
hipError_t launch_primary_shade(const FrameParams &P, const ViewImages &I, const int32_t *hitInstance, int cur, bool transparentLighting, bool lean, hipStream_t s);
hipError_t launch_direct(const FrameParams &P, const ViewImages &I, int cur, bool lean, hipStream_t s);
enum { BOUNCE_WALK_PLAIN = 0, BOUNCE_WALK_REFILL = 1, BOUNCE_WALK_SPLIT = 2 };      // how bounce_trace hands rays to lanes (passes.hip)
// walk: how bounce_trace hands rays to lanes; groups: grid of the bounce kernels (0 = the persistent grid)
hipError_t launch_indirect(const FrameParams &P, const ViewImages &I, int cur, bool writeFiltered, bool klist, int walk, unsigned groups, int writeGuide, hipStream_t s);      // writeGuide: bounce_resolve_kernel also writes the SVGF guide records of its rows (wavefront chain only: !klist, bounce records allocated)
hipError_t launch_indirect_constant(const FrameParams &P, const ViewImages &I, int cur, hipStream_t s);
hipError_t launch_refraction(const FrameParams &P, const ViewImages &I, bool klist, hipStream_t s);
hipError_t launch_reflection(const FrameParams &P, const ViewImages &I, bool klist, int pass, bool last, int parity, hipStream_t s);      // pass / last / parity: see reflection_kernel (ViewImages::reflectFlags)
hipError_t launch_gaussian(const uint16_t *in, uint16_t *out, int width, int height, int y0, int y1, hipStream_t s);
hipError_t launch_compose_post(const FrameParams &P, const ViewImages &I, int cur, bool lean, bool writeFinal, hipStream_t s);
// A lean frame in one launch: primary visibility + resolve + direct light + compose (passes.hip, lean_frame_kernel).
// maxGroups: cap of the grid (RT_MAX_FRAME_GROUPS; device option max_frame_groups lowers it so that small frames exercise the several-tiles-per-workgroup walk)
// perWave: frames without the LDS scene cache run one wave (an 8 x 8 wave-tile) per workgroup instead of one 16 x 16 tile
hipError_t launch_lean_frame(const FrameParams &P, const ViewImages &I, int32_t *hitInstance, int cur, bool full, int ownedY0, int ownedY1, unsigned maxGroups, bool perWave, hipStream_t s);
hipError_t launch_post_process(const FrameParams &P, const ViewImages &I, hipStream_t s);      // PostProcessPS as its own pass (resolution scale / motion blur)
// passes_simple.hip: the same launchers over kernels compiled without non-power-of-two texture addressing and without the shadow any-hit
// program; the launchers above route to them when FrameParams::simpleKernels is set
hipError_t launch_primary_shade_simple(const FrameParams &P, const ViewImages &I, const int32_t *hitInstance, int cur, bool transparentLighting, bool lean, hipStream_t s);
hipError_t launch_direct_simple(const FrameParams &P, const ViewImages &I, int cur, bool lean, hipStream_t s);
hipError_t launch_indirect_simple(const FrameParams &P, const ViewImages &I, int cur, bool writeFiltered, bool klist, int walk, unsigned groups, int writeGuide, hipStream_t s);
hipError_t launch_refraction_simple(const FrameParams &P, const ViewImages &I, bool klist, hipStream_t s);
hipError_t launch_reflection_simple(const FrameParams &P, const ViewImages &I, bool klist, int pass, bool last, int parity, hipStream_t s);
hipError_t launch_lean_frame_simple(const FrameParams &P, const ViewImages &I, int32_t *hitInstance, int cur, bool full, int ownedY0, int ownedY1, unsigned maxGroups, bool perWave, hipStream_t s);
hipError_t launch_clear_final(const FrameParams &P, const ViewImages &I, hipStream_t s);
hipError_t launch_apply_reflection_state(const ViewImages &I, int width, int y0, int y1, uint32_t frameTag, hipStream_t s);      // rows [y0, y1): the continuation state the reflection passes tagged `frameTag` folded back into the G-buffer (readback only)
unsigned lean_frame_tiles(const FrameParams &P);        // tiles of the one-kernel frame's launch (the device's rows, 16 x 16)
hipError_t launch_tile_order(uint32_t *cost, uint32_t *order, uint32_t tiles, hipStream_t s);
hipError_t launch_spp_accumulate(const FrameParams &P, const ViewImages &I, float *sum, int sub, int count, hipStream_t s);

// ---- raster.hip ----------------------------------------------------------------------------------------------------
size_t raster_tri_bytes(uint32_t triTotal);       // setup records of a draw list
// Triangle setup of a draw list for a w x h target, rows [y0, y1); `apply` = drawInstances' applyScissorsAndViewports (rt64_view.cpp:1225).
bool raster_setup_takes_table_inline(uint32_t instanceCount);      // short lists: the table travels in the kernel arguments (launch_raster_setup_inline), no copy
hipError_t launch_raster_setup_inline(const GpuRasterInstance *hostTable, GpuRasterInstance *deviceTable, uint32_t instanceCount, uint32_t triTotal, void *tris, int w, int h, int y0, int y1, bool apply, hipStream_t s);
hipError_t launch_raster_setup(const GpuRasterInstance *instances, uint32_t instanceCount, uint32_t triTotal, void *tris, int w, int h, int y0, int y1, bool apply, hipStream_t s);
// Shade + blend the list, in order, into the RGBA8 target; `bounds` = pixel rectangle [x0, y0, x1, y1) the list can touch.
hipError_t launch_raster_draw(const GpuRasterInstance *instances, const void *tris, uint32_t triTotal, const GpuTexture *textures, uint8_t *target,
                              int w, int y0, int y1, const int bounds[4], int stripRank, int stripCount, bool clear, hipStream_t s);       // clear: the whole target starts as 0 (no memset in front)
// One launch for what a frame with changed tables queues before its first pass: the frame-table upload (read from the pinned upload ring) + the setup of short draw lists.
#define RASTER_PROLOGUE_LISTS 3
#define RASTER_PROLOGUE_INSTANCES 8
#define RASTER_PROLOGUE_COPY_WORDS (4096u)          // 64 KB of frame tables: beyond that the copy engine's launch is the smaller part
struct RasterPrologueList { GpuRasterInstance inst[RASTER_PROLOGUE_INSTANCES]; GpuRasterInstance *deviceTable; void *tris; uint32_t instanceCount, triTotal; int32_t w, h, y0, y1, apply; uint32_t firstBlock; };
struct RasterPrologue { RasterPrologueList list[RASTER_PROLOGUE_LISTS]; const void *copySrc; void *copyDst; uint32_t copyWords, copyBlocks, listCount, pad; };
static_assert(sizeof(RasterPrologue) <= 3584, "frame_prologue_kernel's arguments");
bool frame_prologue_takes(const RasterPrologue &a);
hipError_t launch_frame_prologue(RasterPrologue &a, hipStream_t s);

// ---- svgf.hip ------------------------------------------------------------------------------------------------------
// variance estimate + 5 a-trous iterations over the GI buffer; result in filteredIndirect[1]
#define SVGF_HALO_ROWS 66            // rows of neighbourhood the SVGF result of a row depends on (62 a-trous + 3 variance + 1 gradient)
#define GAUSSIAN_HALO_ROWS 5         // five 3x3 passes
#define SVGF_ATROUS_HALO_ROWS 62     // rows of a-trous INPUT (variance image + guide records) the result of a row depends on: what a halo exchange ships
#define SVGF_INPUT_HALO_ROWS 4       // rows of G-buffer + GI around the rows whose a-trous input is made (3 variance taps + 1 depth gradient)
hipError_t launch_svgf_inputs(const ViewImages &I, int cur, int width, int height, int gy0, int gy1, int vy0, int vy1, bool inputByResolve, hipStream_t s);      // inputByResolve: bounce_resolve_kernel wrote the filter input and marked the young pixels (ViewImages::svgfYoung)
// Compose folded into the last a-trous iteration (svgf.hip): what compose_post_kernel<false> reads and writes, for the rows [oy0, oy1) of the frame
struct SvgfComposeFold { const uint8_t *diffuse; const uint16_t *filteredDirect, *reflection, *refraction, *transparent; float *output; uint8_t *final; int oy0, oy1, writeFinal;
                         float *sppSum; int sppSub, sppCount; };      // extension primary_spp (rule P3): the composed value goes into the running sum of the frame's sub-frames; the last one stores the mean and the back buffer (sppCount <= 1: off)
hipError_t launch_svgf_atrous(const ViewImages &I, int width, int height, int y0, int y1, int oy0, int oy1, int first, int last, const SvgfComposeFold *fold, hipStream_t s);

// ---- upscale.hip ---------------------------------------------------------------------------------------------------
// Temporal upscaler stage (Upscaler::upscale, rt64_view.cpp:1584-1618): rtOutput + flow + masks + depth (render size rw x rh, jitter jx / jy)
// and the previous upscaled image -> `out` (display size dw x dh, RGBA32F: colour + accumulated frame count).
hipError_t launch_taa_upsample(const ViewImages &I, int cur, int rw, int rh, float jx, float jy, const float *prev, float *out, int dw, int dh, bool haveHistory, hipStream_t s);

// ---- gather.hip -----------------------------------------------------------------------------------------------------
// Rank 0 of a multi-GPU gather: frame row y <- row gather_row_owner(y) of the owner's packed buffer (`own` for rank 0, bucket + r * slotBytes for rank r).
hipError_t launch_gather_assemble(const uint8_t *own, const uint8_t *bucket, size_t slotBytes, uint8_t *frame, int width, const GatherLayout &L, hipStream_t s);
hipError_t launch_row_hit_count(const int32_t *hitInstance, uint32_t *counts, int width, int height, hipStream_t s);      // counts[y] = pixels of row y whose primary ray hit geometry
