// rt64_host.cpp -- host-side object model, frame orchestration and the extern "C" ABI of librt64.so.
//
// Mirrors the reference's L1/L2 layers (Device > Scene > {View, Instance}; Mesh / Texture / Shader hang off Device):
//   /root/reference/src/rt64lib/private/rt64_{device,scene,view,instance,mesh,texture,shader}.cpp
// with the D3D12 plumbing replaced by one HIP stream per device: uploads are hipMemcpyAsync on that stream, BLAS builds are
// kernels queued at RT64_SetMesh, and RT64_DrawDevice runs View::update + View::render as a fixed sequence of kernels and
// returns after the stream has drained (the reference presents and waits for GPU idle, rt64_device.cpp:1006-1025).
// Errors: C++ exceptions are caught at the ABI boundary, stored for RT64_GetLastError() and turned into NULL / no-op
// (rt64_common.h:379-383).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"
#include "device_math.h"

#define RT64_EXPORT extern "C" __attribute__((visibility("default")))

namespace rt64 {

static std::string GlobalLastError;

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    char msg_[512]; snprintf(msg_, sizeof(msg_), "HIP call " #call " failed: %s", hipGetErrorString(e_)); throw std::runtime_error(msg_); } } while (0)

template <class T> struct DevArray {
    T *ptr = nullptr; size_t count = 0;
    void reserve(size_t n) { if (n <= count) return; release(); HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T))); count = n; }
    void release() { if (ptr) hipFree(ptr); ptr = nullptr; count = 0; }
    size_t bytes() const { return count * sizeof(T); }
    ~DevArray() { release(); }
    DevArray() = default; DevArray(const DevArray &) = delete; DevArray &operator=(const DevArray &) = delete;
};

struct Scene; struct View; struct Instance; struct Mesh; struct Texture; struct Shader;

// ---- matrices (row-major, row-vector convention) ----------------------------------------------------------------------

struct Mat4 { float m[16]; };

static Mat4 mat_from(const RT64_MATRIX4 &s) { Mat4 r; memcpy(r.m, s.m, 64); return r; }
static Mat4 mat_identity() { Mat4 r = {}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
static Mat4 mat_mul(const Mat4 &A, const Mat4 &B) {
    Mat4 R;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
        R.m[i * 4 + j] = A.m[i * 4 + 0] * B.m[0 * 4 + j] + A.m[i * 4 + 1] * B.m[1 * 4 + j] + A.m[i * 4 + 2] * B.m[2 * 4 + j] + A.m[i * 4 + 3] * B.m[3 * 4 + j];
    return R;
}
static Mat4 mat_transpose(const Mat4 &A) { Mat4 R; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) R.m[i * 4 + j] = A.m[j * 4 + i]; return R; }

// General inverse by cofactors in double precision, rounded once to float.  (The reference uses DirectXMath's float
// XMMatrixInverse: rt64_view.cpp:368,981-984.)
static Mat4 mat_inverse(const Mat4 &M) {
    double m[16], inv[16];
    for (int i = 0; i < 16; i++) m[i] = (double)M.m[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    Mat4 R = {};
    if (det == 0.0) return R;
    double r = 1.0 / det;
    for (int i = 0; i < 16; i++) R.m[i] = (float)(inv[i] * r);
    return R;
}

// XMMatrixPerspectiveFovRH (rt64_view.cpp:1766).
static Mat4 mat_perspective_fov_rh(float fov, float aspect, float zn, float zf) {
    float s = sinf(0.5f * fov), c = cosf(0.5f * fov);
    float h = c / s, w = h / aspect, range = zf / (zn - zf);
    Mat4 P = {};
    P.m[0] = w; P.m[5] = h; P.m[10] = range; P.m[11] = -1.0f; P.m[14] = range * zn;
    return P;
}

struct V3 { float x, y, z; };
static V3 v3(float x, float y, float z) { V3 r = { x, y, z }; return r; }
static V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static float vlen(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static V3 vnorm(V3 a) { float l = vlen(a); return l > 0.0f ? v3(a.x / l, a.y / l, a.z / l) : a; }            // rt64_common.h:318-321
static V3 vcross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static V3 mat_point(const Mat4 &M, V3 p) { return v3(p.x * M.m[0] + p.y * M.m[4] + p.z * M.m[8] + 1.0f * M.m[12], p.x * M.m[1] + p.y * M.m[5] + p.z * M.m[9] + 1.0f * M.m[13], p.x * M.m[2] + p.y * M.m[6] + p.z * M.m[10] + 1.0f * M.m[14]); }
static V3 mat_vector(const Mat4 &M, V3 p) { return v3(p.x * M.m[0] + p.y * M.m[4] + p.z * M.m[8] + 0.0f * M.m[12], p.x * M.m[1] + p.y * M.m[5] + p.z * M.m[9] + 0.0f * M.m[13], p.x * M.m[2] + p.y * M.m[6] + p.z * M.m[10] + 0.0f * M.m[14]); }

// ---- colour combiner decode (rt64_shader.cpp:32-96) ---------------------------------------------------------------------

static GpuCombiner decode_combiner(uint32_t shaderId) {
    GpuCombiner cc = {};
    for (int i = 0; i < 4; i++) { cc.c[0][i] = (int8_t)((shaderId >> (i * 3)) & 7); cc.c[1][i] = (int8_t)((shaderId >> (12 + i * 3)) & 7); }
    for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) {
        int v = cc.c[i][j];
        if (v >= 1 && v <= 4 && v > cc.inputCount) cc.inputCount = (int8_t)v;
        if (v == 5 || v == 6) cc.useTex0 = 1;
        if (v == 7) cc.useTex1 = 1;
    }
    for (int i = 0; i < 2; i++) {
        cc.doSingle[i] = cc.c[i][2] == 0;
        cc.doMultiply[i] = cc.c[i][1] == 0 && cc.c[i][3] == 0;
        cc.doMix[i] = cc.c[i][1] == cc.c[i][3];
    }
    cc.colorAlphaSame = (shaderId & 0xfff) == ((shaderId >> 12) & 0xfff);
    cc.optAlpha = (shaderId & (1u << 24)) != 0;
    cc.optTextureEdge = (shaderId & (1u << 26)) != 0;
    cc.optNoise = (shaderId & (1u << 27)) != 0;
    cc.vertexUV = cc.useTex0 || cc.useTex1;
    int sz = 16;                                         // position float4
    cc.normalOffset = (int16_t)sz; sz += 12;
    cc.uvOffset = (int16_t)sz; if (cc.vertexUV) sz += 8;
    for (int i = 0; i < cc.inputCount; i++) { cc.inputOffset[i] = (int16_t)sz; sz += cc.optAlpha ? 16 : 12; }
    cc.vertexSize = (int16_t)sz;
    return cc;
}

// ---- objects ---------------------------------------------------------------------------------------------------------------

#define RT64_RENDER_STREAMS_MAX 4
#define RT64_LEAN_HOLDOFF_FRAMES 4     // frames that store their whole G-buffer after a scene change had to materialise a lean frame (Device::beforeSceneMutation)

struct Options {
    bool countTraversal = false, profilePasses = true, syncPresent = true, alwaysRebuild = false, leanFrames = true, fusedLean = true, foldForeground = true;
    int profileEvery = 1;         // with profile_passes: record the pass events on every n-th frame only (sampled timings; accumFrames counts the sampled frames)
    bool leanRecords = false;     // 1: the one-kernel lean frame also stores the hit records and the direct-light image (otherwise View::materialise re-traces them on demand)
    bool framePrologue = true;    // a frame with changed tables queues its table upload and the setup of its short raster lists as ONE launch (frame_prologue_kernel), and clears gBackground inside the draw that fills it
    bool hostTlas = true;         // the TLAS of up to RT64_HOST_TLAS_MAX instances is built on the host and travels in the table upload (0: always the GPU builder)
    bool simpleKernels = true;    // frames whose textures are all power-of-two sized and whose instances are all shadow-opaque run the kernels of passes_simple.hip
    bool ldsCache = true;         // small scenes: BVH nodes + instance records cached in LDS by the ray kernels (0: always walk from HBM/L2)
    bool spinPresent = true;      // RT64_DrawDevice waits for the frame by polling the stream (0: blocking hipStreamSynchronize)
    int perWaveFrame = -1;        // one-kernel frame of scenes without the LDS scene cache as one-wave workgroups (8 x 8 wave-tiles): 1 on, 0 off (16 x 16 tiles, four waves), -1 auto (on)
    bool tileTiming = false;      // profiling aid: the one-kernel frame records when each of its waves started and ended (RT64_ReadbackTileTiming)
    int bounceGroups = -1;        // cap of the bounce kernels' grid: 0 one workgroup per tile (up to RT_MAX_BOUNCE_GROUPS), n at most n workgroups (tiles b, b + n, ...), -1 auto (per tile for the two-phase walk, 1024 for the plain walk)
    int bounceSplit = -1;         // two-phase bounce walk (TLAS part first, survivors compacted through LDS): 1 on, 0 off, -1 auto (scenes with the LDS scene cache, two or more GI samples per pixel)
    int bounceRefill = -1;        // bounce-ray traversal with wave-ballot refill: 1 on, 0 off, -1 auto (on when the scene has >= 64 Ki triangles)
    int denoiserMode = 1;          // 0 = reference 5x Gaussian, 1 = SVGF
    bool tileOrder = true;         // one-kernel frame of scenes without the LDS scene cache (one-wave workgroups): tiles start in the order of their cost in the frame before, most expensive first
    bool foldCompose = true;       // frames with the SVGF denoiser: ComposePS inside the last a-trous iteration (0: compose_post_kernel, its own launch)
    bool foldVariance = true;      // ... and the filter's input (variance from the moments) for every pixel with four frames of history; svgf_variance_kernel then only runs where a younger pixel is marked (0: it makes every pixel's input)
    bool foldGuide = true;         // frames with the wavefront GI chain + SVGF: bounce_resolve_kernel writes the filter's guide records (0: svgf_guide_kernel, its own launch)
    bool overlapFrames = true;     // enqueued (sync_present = 0) pixel-local frames alternate over the render streams (three by default): frame k+1 starts while the last waves of frame k are still walking (Device::draw)
    bool reflectionEarly = true;   // ... and they fork as soon as the G-buffer exists (beside the GI chain too), not only beside the a-trous iterations (0: round 3's placement)
    bool overlapReflection = true; // frames with reflection passes AND the SVGF denoiser: the reflection launches run on a second stream beside the a-trous iterations (they share no image)
    bool haloExchange = false;     // band partitions of GI + SVGF frames: ship the filter input of the halo rows between the devices of the gather (RCCL) instead of re-rendering them
    bool haloDryRun = false;       // timing aid: an exchanging band runs its frame but moves no halo rows (what one rank's GPU work costs, measured on one device; results outside the band's interior are then wrong)
    int haloMargin = SVGF_INPUT_HALO_ROWS;   // with a halo exchange: rows of G-buffer + GI kept around the band (temporal history under camera motion); at least SVGF_INPUT_HALO_ROWS
    int maxReflections = 2;        // rt64_view.cpp:60 (inspector-only knob in the reference)
    // Path-tracing extensions beyond the reference (one bounce per GI ray, IndirectRayGen.hlsl:58-131; one primary sample per pixel, rt64.h:172-182); DESIGN.md 4:
    int giBounces = 1;             // 2: a GI ray that resolves to a surface sends a second cosine-weighted ray from there; what it finds stands where the constant ambient term stands at the first hit
    int primarySpp = 1;            // N: RT64_DrawDevice renders N jittered sub-frames (every pass up to Compose each) and presents the mean of their composed outputs
    unsigned maxFrameGroups = RT_MAX_FRAME_GROUPS;   // grid cap of the one-kernel frame (tests lower it: several tiles per workgroup on a small frame)
};

struct Device {
    int hipDevice = 0;
    hipStream_t stream = nullptr;         // the render stream of the frame in hand: streams[cur]
    // Render streams (option overlap_frames; RT64_RENDER_STREAMS = 1 .. 4, default 3).  A frame that is pixel-local from its primary rays to the back buffer, uploads nothing and has no temporal
    // consumer ("pure": the one-kernel lean frame on unchanged tables) reads only what earlier frames left untouched and writes only per-slot storage -- back
    // buffer, gather send buffer, traversal spill slab, tile-cost order, all indexed by `cur` -- so consecutive pure frames may run side by side: with
    // sync_present = 0 they alternate between the two streams, and frame k+1's first waves fill the wave slots the tail of frame k has left empty.  Everything
    // else -- an upload, a build, a frame with history, a readback, any API call that touches device memory -- first makes the current stream wait for the other
    // one (joinStreams), after which streams[cur] is ordered behind everything enqueued so far, exactly as with one stream.
    hipStream_t streams[RT64_RENDER_STREAMS_MAX] = {}; int cur = 0, streamCount = 3;      // (RT64_RENDER_STREAMS = 1 .. 4 at device creation; default 3: measured)
    hipEvent_t streamJoin[RT64_RENDER_STREAMS_MAX] = {};
    bool streamBusy[RT64_RENDER_STREAMS_MAX] = {};      // streams[k] may hold work streams[cur] has not waited for
    bool framePure = false, lastFramePure = false;
    void joinStreams() {
        for (int k = 0; k < streamCount; k++) {
            if (k == cur || !streamBusy[k]) continue;
            HIP_CHECK(hipEventRecord(streamJoin[k], streams[k]));
            HIP_CHECK(hipStreamWaitEvent(stream, streamJoin[k], 0));
            streamBusy[k] = false;
        }
    }
    // ... and the other direction: work that is not slot-local (always enqueued on the stream that is current at the time, behind a join) has to be seen by every
    // later frame on ANY stream.  `orderedEpoch` counts such work; a stream that is switched to without having seen the latest epoch first waits for the stream
    // that is being left (which has, by induction) -- one event at the first switch onto each stream after an upload, nothing in the steady state.
    unsigned long long orderedEpoch = 1, seenEpoch[RT64_RENDER_STREAMS_MAX] = {};
    hipEvent_t streamCatchUp = nullptr;
    void noteOrderedWork() { seenEpoch[cur] = ++orderedEpoch; }
    void impure() { framePure = false; joinStreams(); noteOrderedWork(); }      // this frame enqueues something that is not slot-local: it runs behind every earlier frame
    void enter() { use(); joinStreams(); noteOrderedWork(); lastFramePure = false; }   // API entry that touches device memory outside RT64_DrawDevice / RT64_SubmitGather
    void switchStream() {
        const int prev = cur;
        streamBusy[prev] = true; cur = (cur + 1) % streamCount; stream = streams[cur];
        if (seenEpoch[cur] != orderedEpoch) {
            HIP_CHECK(hipEventRecord(streamCatchUp, streams[prev]));
            HIP_CHECK(hipStreamWaitEvent(stream, streamCatchUp, 0));
            seenEpoch[cur] = orderedEpoch;
        }
    }
    int width = 0, height = 0, pendingWidth = 0, pendingHeight = 0;
    int tileY0 = 0, tileY1 = 0; bool tileSet = false;
    int stripRank = 0, stripCount = 1;
    // The partition the LAST frame was rendered with: what a readback describes (a host may set another partition and read the frame it already has before it draws the next).
    struct DrawnPartition { int tileY0 = 0, tileY1 = 0, stripRank = 0, stripCount = 1; bool valid = false; } drawn;
    std::vector<Scene *> scenes;
    Options opt;
    RT64_FRAME_STATS stats = {}, accum = {}; bool statsPending = false, statsHaveView = false;
    bool profNow = false, statsProfiled = false; unsigned profCounter = 0;       // this frame records its pass events (option profile_every)
    double hostUpdateUs = 0.0, hostRenderUs = 0.0, hostStageUs[16] = {}, hostEventUs = 0.0; unsigned long long hostFrames = 0;      // host-side cost of View::update / View::render (RT64_HOST_TIMING=1 prints them)
    void finishStats();
    uint32_t *traversalOverflow = nullptr;      // pinned host word the traversal stacks report a dropped entry to (TraceStack::report_overflow); read after every frame the host waits for
    void checkTraversalOverflow();
    DevArray<uint32_t> spillStack[RT64_RENDER_STREAMS_MAX];                                     // HBM half of the traversal stacks, one slab per render stream (indexed by the launch's lanes)
    void *gatherTarget = nullptr; size_t gatherTargetBytes = 0;          // RT64_SetDeviceGatherTarget
    hipEvent_t frameWait = nullptr;       // set by the gather: the next frame's stream waits for this event before its first launch (the slot that frame writes is free then); applied by Device::draw once it knows the stream
    uint8_t *finalOverride = nullptr;     // direct gather (RT64_SetGatherDirect): the back buffer of the frame in hand IS the gather's frame slot -- on rank 0 its own memory, on the others rank 0's through an IPC mapping (peer stores over xGMI)
    hipStream_t auxStream = nullptr; hipEvent_t forkEvent = nullptr, joinEvent = nullptr;      // second stream of a frame whose reflection passes run beside its denoiser (created on first use)
    // Halo exchange of the SVGF filter input between the bands of a partition (RT64_SetDeviceHaloExchange / option halo_exchange): transport and layout
    struct HaloLink {
        RT64_HALO_EXCHANGE fn = nullptr; void *user = nullptr; int rank = 0, count = 0; std::vector<int> starts;       // the host's transport
        struct Gather *gather = nullptr;                                                                                 // RCCL: the device's gather (bands), set by RT64_CreateGather
        hipEvent_t ready = nullptr, done = nullptr;
        uint8_t *pinned = nullptr; size_t pinnedBytes = 0;
    } halo;
    bool haloActive() const;
    DevArray<unsigned long long> counters;
    DevArray<uint4> tileTiming; unsigned tileTimingWaves = 0;        // option tile_timing: records of the last one-kernel frame
    DevArray<uint8_t> blueNoise;
    uint8_t *pinned[2] = { nullptr, nullptr }; size_t pinnedBytes[2] = { 0, 0 };
    enum { EV_BEGIN, EV_BUILD, EV_PRIMARY_TRACE, EV_PRIMARY, EV_DIRECT, EV_INDIRECT, EV_REFL, EV_DENOISE, EV_END, EV_COUNT };
    // One set of pass events per sub-frame of a primary_spp frame (the pass timings of such a frame are the sums over its sub-frames; beyond EV_SETS sub-frames the
    // last EV_SETS are measured and scaled up), plus a pair around the reflection passes when they run on the second stream beside the denoiser.
    enum { EV_SETS = 8 };
    hipEvent_t eventSets[EV_SETS][EV_COUNT] = {}, auxEvents[EV_SETS][2] = {};
    // A mark with no GPU work since the previous mark reuses that mark's event: every hipEventRecord is a barrier packet (~4 us of GPU idle).
    int eventAliases[EV_SETS][EV_COUNT] = {}; bool auxTimed[EV_SETS] = {}; int evSet = 0, evSetsUsed = 1, evSubFrames = 1; int lastMark = EV_BEGIN; bool workSinceMark = false;
    hipEvent_t *events = eventSets[0]; int *eventAlias = eventAliases[0];
    void beginEventSet(int sub) {           // first thing of sub-frame `sub` (0: the frame's start, before the builds)
        evSet = sub % EV_SETS; events = eventSets[evSet]; eventAlias = eventAliases[evSet]; auxTimed[evSet] = false;
        HIP_CHECK(hipEventRecord(events[EV_BEGIN], stream)); eventAlias[EV_BEGIN] = EV_BEGIN; lastMark = EV_BEGIN; workSinceMark = false;
    }
    void endEventSet() {                    // the end mark shares the last mark's event when nothing was launched after it (one barrier packet less per frame)
        if (workSinceMark) { HIP_CHECK(hipEventRecord(events[EV_END], stream)); eventAlias[EV_END] = EV_END; }
        else eventAlias[EV_END] = eventAlias[lastMark];
    }

    Device(int w, int h, int dev);
    ~Device();
    void use() const { HIP_CHECK(hipSetDevice(hipDevice)); }
    // Pinned upload buffer of the synchronous texture uploads (the caller waits for the copy before the pool is written again).
    // Everything that is queued without a wait -- mesh arrays, frame tables, raster lists -- is staged in the upload ring below.
    void *staging(size_t bytes, int pool = 0) {
        if (bytes > pinnedBytes[pool]) {
            if (pinned[pool]) { HIP_CHECK(hipStreamSynchronize(stream)); hipHostFree(pinned[pool]); }
            pinnedBytes[pool] = std::max(bytes, (size_t)1 << 20);
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&pinned[pool]), pinnedBytes[pool], hipHostMallocDefault));
        }
        return pinned[pool];
    }
    // Upload ring (pinned): RT64_SetMesh stages its arrays here and queues the copies without waiting; a wrap-around drains the
    // stream first, so a region is never rewritten while a copy may still read it.
    uint8_t *ring = nullptr; size_t ringBytes = 0, ringHead = 0;
    void *ringAlloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        auto drain = [&]() { for (hipStream_t st : streams) if (st) HIP_CHECK(hipStreamSynchronize(st)); };       // (table uploads are queued on whichever render stream their frame runs on)
        if (bytes > ringBytes) {
            drain();
            if (ring) hipHostFree(ring);
            ringBytes = std::max(bytes * 2, (size_t)32 << 20); ringHead = 0;
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ring), ringBytes, hipHostMallocDefault));
        }
        if (ringHead + bytes > ringBytes) { drain(); ringHead = 0; }
        void *p = ring + ringHead; ringHead += bytes;
        return p;
    }
    // Meshes whose BLAS has to be (re)built or refitted: recorded by RT64_SetMesh, executed together by flushMeshBuilds() at the
    // next frame (or before anything else reads a BLAS) -- small trees in ONE launch, one workgroup each.
    std::vector<Mesh *> dirtyMeshes; DevArray<LbvhArgs> buildArgs;
    void flushMeshBuilds();
    // A lean frame keeps its G-buffer implicit: View::materialise re-reads the frame's meshes, textures and tables.  Whoever is about to change
    // or free one of them calls this first, so the images are produced from the bytes the frame was rendered with; the next few frames then
    // store their whole G-buffer themselves (a host that changes its scene every frame pays for one materialise, not one per frame).
    void beforeSceneMutation();
    unsigned leanHoldoff = 0;
    void draw(int vsyncInterval, float deltaTimeMs);
    float aspect() const { return (float)width / (float)height; }
    // Rows of [tileY0, tileY1) in strips stripRank, stripRank + stripCount, ... (16 rows each).
    template <class F> void forEachOwnedStrip(F &&f) const {
        for (int y = tileY0 + stripRank * 16; y < tileY1; y += stripCount * 16) f(y, std::min(y + 16, tileY1));
    }
    int ownedRows() const { int r = 0; forEachOwnedStrip([&](int a, int b) { r += b - a; }); return r; }
};

struct Texture {
    Device *device; int width = 0, height = 0, mips = 0;
    DevArray<uint8_t> texels; uint32_t mipOffset[RT64_MAX_MIPS] = {};
    uint8_t minAlpha = 255, maxAlpha = 255;
    int currentIndex = -1;
    uint32_t serial;                                   // unique per texture object: part of the raster-list cache key (a slot index alone does not identify a texture)
    explicit Texture(Device *d) : device(d) { static uint32_t globalTextureSerial = 0; serial = ++globalTextureSerial; }
    void setRGBA8(const void *bytes, int byteCount, int w, int h, int rowPitch);
    void setDDS(const void *bytes, int byteCount);
};

struct Shader {
    Device *device; uint32_t shaderId, filter, hAddr, vAddr; int flags; GpuCombiner cc;
};

struct Mesh {
    Device *device; int flags;
    int vertexCount = 0, vertexStride = 0, indexCount = 0;
    std::vector<uint8_t> hostVertices;                 // kept for the opacity rule (alpha bounds of the vertex inputs)
    DevArray<uint8_t> vertices; DevArray<uint32_t> indices;
    DevArray<GpuNode> nodes; DevArray<GpuTri> tris; DevArray<BlasHeader> header;
    DevArray<uint32_t> sortedIndex, morton, leafParent; DevArray<uint8_t> buildScratch;
    uint32_t blasCount = 0;                            // leaves of the current BLAS (0 = none)
    float hostBmin[3] = { 0, 0, 0 }, hostBmax[3] = { 0, 0, 0 };     // bounds of the indexed positions (== BlasHeader bounds: a min / max over the same floats), for the host-side TLAS build
    uint32_t topologyVersion = 0, depth = 255;      // tree depth (== BlasHeader::depth), computed on the host when the topology is decided (Mesh::set): refits keep it
    uint32_t treeDepth() const { return blasCount == 0 ? 0u : depth; }
    bool buildPending = false, pendingRefit = false;   // RT64_SetMesh recorded a build / refit that Device::flushMeshBuilds has not run yet
    ~Mesh();
    uint32_t version = 0;
    std::map<int, std::pair<float, float>> alphaBounds; uint32_t alphaBoundsVersion = 0;
    Mesh(Device *d, int f) : device(d), flags(f) {}
    void set(const void *vertexArray, int vcount, int vstride, const unsigned int *indexArray, int icount);
    std::pair<float, float> inputAlphaBounds(int offset);
};

struct Instance {
    Scene *scene;
    Mesh *mesh = nullptr; Texture *diffuse = nullptr, *normal = nullptr, *specular = nullptr; Shader *shader = nullptr;
    Mat4 transform = mat_identity(), previousTransform = mat_identity();
    RT64_MATERIAL material = {}; RT64_RECT scissorRect = {}, viewportRect = {}; unsigned int flags = 0;
    explicit Instance(Scene *s);
    ~Instance();
};

struct Scene {
    Device *device; RT64_SCENE_DESC desc = {};
    std::vector<RT64_LIGHT> lights;
    std::vector<Instance *> instances; std::vector<View *> views;
    explicit Scene(Device *d);
    ~Scene();
};

struct RenderInstance { Instance *instance; };

struct View {
    Scene *scene;
    // RT64_VIEW_DESC state + inspector-only knobs, defaults of rt64_view.cpp:47-66
    float resolutionScale = 1.0f, motionBlurStrength = 0.0f; uint32_t diSamples = 0, giSamples = 0, maxLights = 12, motionBlurSamples = 32;
    bool denoiserEnabled = false;
    // RT64_VIEW_DESC.upscaler / upscalerMode: AUTO and FSR select the built-in temporal upscaler (upscale.hip); the vendor SDKs do not exist here
    int upscaler = RT64_UPSCALER_OFF, upscalerMode = RT64_UPSCALER_MODE_AUTO; float upscalerSharpness = 0.0f;
    bool upscaleActive = false; int jitterPhases = 1; float pixelJitter[2] = { 0.0f, 0.0f };
    float *upscaled[2] = { nullptr, nullptr }; int upW = 0, upH = 0, upSwap = 0; bool upValid = false;
    Texture *skyPlane = nullptr;
    DevArray<uint32_t> skyTiled; uint32_t skyTiledSerial = 0; uint32_t skyTiledLog2[2] = { 0, 0 };     // level 0 of the sky plane in 4 x 4 tiles (FrameParams::skyTiled)
    Mat4 view = mat_identity(), projection = mat_identity(), viewI = mat_identity(), projectionI = mat_identity(), viewProj = mat_identity(), prevViewI = mat_identity(), prevViewProj = mat_identity();
    float fov = 0.0f, nearDist = 0.0f, farDist = 0.0f; bool canReproject = true, matricesValid = false, perspectiveSet = false;
    uint32_t frameCount = 0; bool rtSwap = false, skipReprojection = true;
    int imgW = 0, imgH = 0;                   // render size: lround(screen * resolutionScale), rt64_view.cpp:138-139
    int finalW = 0, finalH = 0;               // back buffer = screen size
    // rectangle of the ray-traced picture: the first ray-traced instance's scissor / viewport when it has any (rt64_view.cpp:1258-1271)
    float rtViewport[4] = { 0, 0, 0, 0 }; int rtScissor[4] = { 0, 0, 0, 0 }; bool rtRect = false;
    bool needSpillSlab = false;                // some walk of this frame can outgrow its LDS stack entries (see View::update)
    uint32_t cacheWords = 0;                   // LDS scene cache size in 16-byte words (0: the scene does not fit / option lds_cache = 0)
    bool separatePost() const { return upscaleActive || rtRect || imgW != finalW || imgH != finalH || (motionBlurStrength > 0.0f && motionBlurSamples > 0); }
    // device images
    ViewImages img = {};
    std::vector<void *> allocations; uint32_t bounceSamples = 0;
    DevArray<int32_t> hitInstance;
    // per-frame tables
    std::vector<RenderInstance> rtInstances, rasterBg, rasterFg;
    // raster pass (raster.hip): device tables of the two lists, triangle setup records, gBackground target
    struct RasterList {
        DevArray<GpuRasterInstance> table; DevArray<uint8_t> tris;      // device instance table + triangle setup records
        std::vector<uint8_t> uploaded;                                   // bytes of the table the records were built from (cache key)
        uint32_t triTotal = 0; int w = 0, h = 0, y0 = 0, y1 = 0; bool apply = false, ready = false, changed = false;
        bool contentChanged = false;                                     // `changed` and the bytes really differ (always_rebuild re-stages identical lists)
        int bounds[4] = { 0, 0, 0, 0 };                                  // conservative pixel rectangle [x0, y0, x1, y1) the list can touch
    };
    RasterList rasterBgEnv, rasterBgScreen, rasterFgScreen;             // bg -> gBackground (no scissors), bg -> back buffer, fg -> back buffer
    DevArray<uint8_t> background; int backgroundW = 0, backgroundH = 0;
    void prepareRasterList(const std::vector<RenderInstance> &list, RasterList &rl, int w, int h, int y0, int y1, bool apply);
    void drawRasterList(RasterList &rl, uint8_t *target, bool clear = false);
    // What View::update has decided to put on the stream but not queued yet: the table upload and the setup of short lists leave as one launch (flushPrologue);
    // whatever else needs the uploaded tables first (a TLAS built by kernels, the scene-cache image, a materialise) calls flushTableCopy and gets the plain copy.
    RasterPrologue prologue = {}; size_t prologueCopyBytes = 0;
    void flushTableCopy();
    void flushPrologue();
    std::vector<Texture *> usedTextures;
    // The frame tables live in ONE device allocation in their staging order (instances | textures | lights) so that a changed
    // frame costs one host-to-device copy (each copy packet is ~10 us of stream time, whatever its size).
    template <class T> struct TablePtr { T *ptr = nullptr; };
    // ... and there are several such allocations (TABLE_SLOTS, one per render stream): tables that CHANGED are uploaded into the next slot, never over the ones the frame
    // before was rendered with.  The frame before -- still running on another stream, or kept implicit as a lean frame whose G-buffer a reader may ask for later
    // (View::materialise) -- keeps its instances, textures, lights, TLAS and scene-cache image, so a host that moves an instance every frame (a game) pays neither a
    // materialise nor a join for it: its frames stay lean and stay side by side.  Unchanged tables stay in their slot.
    enum { TABLE_SLOTS = 2 * RT64_RENDER_STREAMS_MAX };
    struct TableSlot {
        DevArray<uint8_t> dTables; TablePtr<GpuInstance> dInstances; TablePtr<GpuTexture> dTextures; TablePtr<RT64_LIGHT> dLights;
        DevArray<GpuNode> tlasNodes; DevArray<uint32_t> tlasIndex, tlasMorton, tlasLeafParent; DevArray<BlasHeader> tlasHeader; DevArray<uint8_t> tlasScratch;
        // where the slot's TLAS lives: the arrays above (GPU build), or the tail of dTables (a few instances: built on the host, uploaded with the tables)
        const GpuNode *tlasNodesAt = nullptr; const uint32_t *tlasIndexAt = nullptr, *tlasMortonAt = nullptr; const BlasHeader *tlasHeaderAt = nullptr;
        std::vector<uint8_t> uploadedTables;      // bytes of the slot's instance/texture/light tables (cache key)
        DevArray<uint8_t> cacheImage; bool cacheImageValid = false;      // FrameParams::cacheImage of frames without a host-built TLAS: rebuilt after every table upload that leaves the cache enabled
        // ... or, with a host-built TLAS, inside the table allocation: its head (instance records + TLAS nodes) is written by the host into the same upload as the tables and
        // only the BLAS node arrays behind it are copied by a kernel -- when a mesh of the frame was built, refitted or replaced, not when an instance moved
        const uint8_t *cacheImageAt = nullptr; std::vector<uint64_t> cacheBlasKey;
        unsigned readers = 0;                     // render streams (bit mask) whose frames have read the slot since its last upload
    } tab[TABLE_SLOTS];
    int tabCur = 0;
    std::vector<uint8_t> tableScratch;
    float maxDepthBias = 0.0f;
    bool anyNonOpaque = false, anyReflection = false, anyRefraction = false, anyFog = false;
    bool simpleFrame = false;                 // every texture of the frame is a power of two in both sizes and every instance is shadow-opaque (passes_simple.hip)
    bool leanFrame = false;                   // last frame skipped the images no pass consumed (see materialise)
    bool fusedFullFrame = false;              // full frame whose primary + direct passes ran as lean_frame_kernel<.., FULL>
    bool packedFinal = false;                 // the frame also wrote its owned back-buffer rows to the device's gather target
    bool fusedStoreless = false;              // ... and that kernel stored the back buffer only (no hit records, no direct-light image)
    bool fusedFrame = false;                  // ... and ran as lean_frame_kernel: rtOutput was not written either (unless PostProcess ran separately)
    FrameParams lastParams; int lastCur = 0;
    // the last frame's reflection passes left their continuation state beside the G-buffer: a readback of the four images the reference rewrites folds it back first
    bool reflStatePending = false; uint32_t reflStateTag = 0; int reflStateY0 = 0, reflStateY1 = 0;
    void applyReflectionState();
    // extension primary_spp (rules P1-P4, oracle/oracle_render.c): the frame as `subFrames` complete sub-frames; Device::draw drives them
    int subFrame = 0, subFrames = 1; DevArray<float> sppSum;
    // longest-first tile order of the one-kernel frame on scenes that walk from HBM (device option tile_order): last frame's cost per tile and the order made from it
    DevArray<uint32_t> tileCost[RT64_RENDER_STREAMS_MAX], tileOrder[RT64_RENDER_STREAMS_MAX]; uint32_t tileOrderTiles = 0; bool tileOrderValid[RT64_RENDER_STREAMS_MAX] = {};      // (one set per render stream: Device::streams)
    uint8_t *finalBuf[RT64_RENDER_STREAMS_MAX] = {};      // the back buffer, one per render stream; img.final is the one of the frame in hand / last drawn

    explicit View(Scene *s);
    ~View();
    void releaseImages();
    void createImages(int w, int h, int screenW, int screenH);
    void update();
    void render();
    void fillParams(FrameParams &P);
    void materialise();
};

// ---- Device -----------------------------------------------------------------------------------------------------------------

static std::string assets_dir() {
    if (const char *e = getenv("RT64_ASSETS_DIR")) return e;
    // librt64.so lives in <repo>/sm64rt-legacy-renderer_amd/; the blue-noise table in <repo>/assets/
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&assets_dir), &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.find_last_of('/');
        if (s != std::string::npos) return p.substr(0, s) + "/../assets";
    }
    return "assets";
}

Device::Device(int w, int h, int dev) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) throw std::runtime_error("No HIP device available: librt64.so needs an AMD GPU (there is no CPU fallback).");
    if (dev < 0) { HIP_CHECK(hipGetDevice(&dev)); }
    if (dev >= count) throw std::runtime_error("Requested HIP device index is out of range.");
    hipDevice = dev;
    use();
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (w <= 0 || h <= 0) throw std::runtime_error("Invalid device size.");
    width = pendingWidth = w; height = pendingHeight = h; tileY0 = 0; tileY1 = h;
    if (const char *e = getenv("RT64_RENDER_STREAMS")) streamCount = std::min(std::max(atoi(e), 1), RT64_RENDER_STREAMS_MAX);
    for (int k = 0; k < streamCount; k++) {
        HIP_CHECK(hipStreamCreateWithFlags(&streams[k], hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&streamJoin[k], hipEventDisableTiming));
    }
    HIP_CHECK(hipEventCreateWithFlags(&streamCatchUp, hipEventDisableTiming));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&traversalOverflow), 64, hipHostMallocMapped)); *traversalOverflow = 0;
    stream = streams[0]; cur = 0; seenEpoch[0] = orderedEpoch;
    for (auto &set : eventSets) for (auto &ev : set) HIP_CHECK(hipEventCreate(&ev));
    for (auto &set : auxEvents) for (auto &ev : set) HIP_CHECK(hipEventCreate(&ev));
    counters.reserve((size_t)CTR_COUNT * RT_COUNTER_STRIPES);
    HIP_CHECK(hipMemsetAsync(counters.ptr, 0, counters.bytes(), stream));
    // Blue-noise table (Device::loadBlueNoise, rt64_device.cpp:794-797): 512x512 RGBA8.
    blueNoise.reserve(512 * 512 * 4);
    std::string path = assets_dir() + "/bluenoise_512x512_rgba8.bin";
    std::vector<uint8_t> bn(512 * 512 * 4, 128);
    if (FILE *f = fopen(path.c_str(), "rb")) {
        size_t got = fread(bn.data(), 1, bn.size(), f);
        fclose(f);
        if (got != bn.size()) throw std::runtime_error("Blue-noise table " + path + " is truncated.");
    }
    else throw std::runtime_error("Blue-noise table " + path + " not found (the library looks in <its directory>/../assets; RT64_ASSETS_DIR overrides): soft shadows and GI sampling need it.");
    HIP_CHECK(hipMemcpy(blueNoise.ptr, bn.data(), bn.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipStreamSynchronize(stream));
}

Device::~Device() {
    hipSetDevice(hipDevice);
    for (hipStream_t st : streams) if (st) hipStreamSynchronize(st);
    if (getenv("RT64_HOST_TIMING") && hostFrames) fprintf(stderr, "RT64 host timing: %llu frames, View::update %.1f us, View::render %.1f us per frame\n", hostFrames, hostUpdateUs / (double)hostFrames, hostRenderUs / (double)hostFrames);
    if (getenv("RT64_HOST_TIMING") && hostFrames) { fprintf(stderr, "  launch host us by stage (up to each event mark):"); for (int i = 0; i < 16; i++) if (hostStageUs[i] > 0.0) fprintf(stderr, " [%d] %.1f", i, hostStageUs[i] / (double)hostFrames); fprintf(stderr, "  event records %.1f\n", hostEventUs / (double)hostFrames); }
    auto scenesCopy = scenes;
    for (Scene *s : scenesCopy) delete s;                 // rt64_device.cpp:97-100
    for (auto &set : eventSets) for (auto &ev : set) if (ev) hipEventDestroy(ev);
    for (auto &set : auxEvents) for (auto &ev : set) if (ev) hipEventDestroy(ev);
    for (auto *p : pinned) if (p) hipHostFree(p);
    if (auxStream) { hipStreamSynchronize(auxStream); hipStreamDestroy(auxStream); }
    if (forkEvent) hipEventDestroy(forkEvent);
    if (joinEvent) hipEventDestroy(joinEvent);
    if (halo.pinned) hipHostFree(halo.pinned);
    if (halo.ready) hipEventDestroy(halo.ready);
    if (halo.done) hipEventDestroy(halo.done);
    if (ring) hipHostFree(ring);
    if (traversalOverflow) hipHostFree(traversalOverflow);
    for (hipEvent_t ev : streamJoin) if (ev) hipEventDestroy(ev);
    if (streamCatchUp) hipEventDestroy(streamCatchUp);
    for (hipStream_t st : streams) if (st) hipStreamDestroy(st);
}

// ---- Texture ----------------------------------------------------------------------------------------------------------------

void Texture::setRGBA8(const void *bytes, int byteCount, int w, int h, int rowPitch) {
    if (!bytes || w <= 0 || h <= 0 || rowPitch < w * 4 || (long long)rowPitch * h > (long long)byteCount + (rowPitch - w * 4)) throw std::runtime_error("RT64_CreateTexture: invalid RGBA8 description.");
    device->enter();
    width = w; height = h; mips = 1; mipOffset[0] = 0;     // mip generation is compiled out in the reference (rt64_device.cpp:758-762)
    texels.reserve((size_t)w * h * 4);
    uint8_t *stage = static_cast<uint8_t *>(device->staging((size_t)w * h * 4));
    uint8_t mn = 255, mx = 0;
    for (int y = 0; y < h; y++) {
        const uint8_t *src = static_cast<const uint8_t *>(bytes) + (size_t)y * rowPitch;
        memcpy(stage + (size_t)y * w * 4, src, (size_t)w * 4);
        for (int x = 0; x < w; x++) { uint8_t a = src[4 * x + 3]; mn = std::min(mn, a); mx = std::max(mx, a); }
    }
    minAlpha = mn; maxAlpha = mx;
    HIP_CHECK(hipMemcpyAsync(texels.ptr, stage, (size_t)w * h * 4, hipMemcpyHostToDevice, device->stream));
    HIP_CHECK(hipStreamSynchronize(device->stream));        // the reference submits + waits per texture (rt64_texture.cpp:130-137)
}

static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

void Texture::setDDS(const void *data, int byteCount) {
    const uint8_t *bytes = static_cast<const uint8_t *>(data);
    if (!bytes || byteCount < 128 || memcmp(bytes, "DDS ", 4) != 0) throw std::runtime_error("RT64_CreateTexture: not a DDS file.");
    uint32_t h = rd32(bytes + 12), w = rd32(bytes + 16), mipCount = rd32(bytes + 28);
    uint32_t pfFlags = rd32(bytes + 80), fourCC = rd32(bytes + 84), rgbBits = rd32(bytes + 88);
    size_t off = 128; bool bc7 = false, rgba = false;
    if ((pfFlags & 0x4) && fourCC == 0x30315844u) {
        if (byteCount < 148) throw std::runtime_error("RT64_CreateTexture: truncated DX10 DDS header.");
        uint32_t fmt = rd32(bytes + 128); off = 148;
        if (fmt == 98 || fmt == 99) bc7 = true; else if (fmt == 28 || fmt == 29) rgba = true;
        else throw std::runtime_error("RT64_CreateTexture: unsupported DXGI format in DDS (supported: BC7_UNORM, R8G8B8A8_UNORM).");
    }
    else if ((pfFlags & 0x40) && rgbBits == 32 && rd32(bytes + 92) == 0x000000FFu) rgba = true;
    else throw std::runtime_error("RT64_CreateTexture: unsupported DDS pixel format.");
    if (mipCount == 0) mipCount = 1;
    if (mipCount > RT64_MAX_MIPS || w == 0 || h == 0) throw std::runtime_error("RT64_CreateTexture: invalid DDS dimensions.");
    device->enter();
    width = (int)w; height = (int)h; mips = (int)mipCount;
    size_t totalTexels = 0, totalSrc = 0;
    { uint32_t mw = w, mh = h; for (uint32_t m = 0; m < mipCount; m++) { mipOffset[m] = (uint32_t)totalTexels; totalTexels += (size_t)mw * mh; totalSrc += bc7 ? (size_t)((mw + 3) / 4) * ((mh + 3) / 4) * 16 : (size_t)mw * mh * 4; mw = mw > 1 ? mw / 2 : 1; mh = mh > 1 ? mh / 2 : 1; } }
    if (off + totalSrc > (size_t)byteCount) throw std::runtime_error("RT64_CreateTexture: DDS data is truncated.");
    texels.reserve(totalTexels * 4);
    uint8_t *stage = static_cast<uint8_t *>(device->staging(std::max(totalSrc, totalTexels * 4)));
    memcpy(stage, bytes + off, totalSrc);
    if (bc7) {
        DevArray<uint8_t> blocks; blocks.reserve(totalSrc);
        HIP_CHECK(hipMemcpyAsync(blocks.ptr, stage, totalSrc, hipMemcpyHostToDevice, device->stream));
        uint32_t mw = w, mh = h; size_t so = 0;
        for (uint32_t m = 0; m < mipCount; m++) {
            HIP_CHECK(bc7_decode_launch(blocks.ptr + so, texels.ptr + (size_t)mipOffset[m] * 4, mw, mh, device->stream));
            so += (size_t)((mw + 3) / 4) * ((mh + 3) / 4) * 16; mw = mw > 1 ? mw / 2 : 1; mh = mh > 1 ? mh / 2 : 1;
        }
        HIP_CHECK(hipStreamSynchronize(device->stream));
    }
    else {
        HIP_CHECK(hipMemcpyAsync(texels.ptr, stage, totalTexels * 4, hipMemcpyHostToDevice, device->stream));
        HIP_CHECK(hipStreamSynchronize(device->stream));
    }
    // Alpha bounds for the opacity rule (read back once; texture creation is synchronous like the reference's).
    HIP_CHECK(hipMemcpy(stage, texels.ptr, totalTexels * 4, hipMemcpyDeviceToHost));
    uint8_t mn = 255, mx = 0;
    for (size_t i = 0; i < totalTexels; i++) { uint8_t a = stage[4 * i + 3]; mn = std::min(mn, a); mx = std::max(mx, a); }
    minAlpha = mn; maxAlpha = mx;
}

// ---- Mesh -------------------------------------------------------------------------------------------------------------------

static uint32_t host_morton30(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t v[3] = { x & 1023u, y & 1023u, z & 1023u }, code = 0;
    for (int k = 0; k < 3; k++) {
        uint32_t t = v[k];
        t = (t | (t << 16)) & 0x030000FFu; t = (t | (t << 8)) & 0x0300F00Fu; t = (t | (t << 4)) & 0x030C30C3u; t = (t | (t << 2)) & 0x09249249u;
        code |= t << k;
    }
    return code;
}
// Depth of the LBVH the device builder will make over these triangles (inner nodes on the longest root-to-leaf path == BlasHeader::depth),
// computed on the host so that View::update never waits for a build to learn it: the same leaf boxes, scene box, Morton keys and key order as
// lbvh.hip (Geometry spec G1-G3: IEEE min / max, (c - min) * scale truncated, 62-bit keys code << 32 | triangle), then the depth of the radix
// tree over the sorted keys -- Karras' tree splits a key range where the highest differing bit of its end keys flips, so the depth follows from
// the keys alone.  A few microseconds for the few hundred triangles of a game mesh; trees of the multi-kernel builder report 255 ("deep").
static uint32_t host_blas_depth(const uint8_t *vertices, size_t stride, const unsigned int *indices, uint32_t n) {
    if (n <= 1) return 1;
    if (n > LBVH_SMALL_MAX) return 255;
    std::vector<float> lo(3 * (size_t)n), hi(3 * (size_t)n);
    float smin[3] = { INFINITY, INFINITY, INFINITY }, smax[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t t = 0; t < n; t++) {
        float v[3][3];
        for (int k = 0; k < 3; k++) memcpy(v[k], vertices + (size_t)indices[3 * t + k] * stride, 12);
        for (int k = 0; k < 3; k++) {
            const float mn = fminf(fminf(v[0][k], v[1][k]), v[2][k]), mx = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]);
            lo[3 * (size_t)t + k] = mn; hi[3 * (size_t)t + k] = mx; smin[k] = fminf(smin[k], mn); smax[k] = fmaxf(smax[k], mx);
        }
    }
    std::vector<unsigned long long> keys(n);
    float scale[3];
    for (int k = 0; k < 3; k++) { const float ext = smax[k] - smin[k]; scale[k] = ext > 0.0f ? 1024.0f / ext : 0.0f; }
    for (uint32_t t = 0; t < n; t++) {
        uint32_t q[3];
        for (int k = 0; k < 3; k++) {
            const float c = (lo[3 * (size_t)t + k] + hi[3 * (size_t)t + k]) * 0.5f, f = (c - smin[k]) * scale[k];
            q[k] = !(f >= 1.0f) ? 0u : (f >= 1023.0f ? 1023u : (uint32_t)(int)f);       // (int)f clamped to [0, 1023]; NaN -> 0 like v_cvt_i32_f32
        }
        keys[t] = ((unsigned long long)host_morton30(q[0], q[1], q[2]) << 32) | t;
    }
    std::sort(keys.begin(), keys.end());
    struct Range { uint32_t l, r, d; };
    std::vector<Range> todo; todo.push_back({ 0u, n - 1, 1u });
    uint32_t deepest = 1;
    while (!todo.empty()) {
        const Range g = todo.back(); todo.pop_back();
        deepest = std::max(deepest, g.d);
        const unsigned long long bit = 1ull << (63 - __builtin_clzll(keys[g.l] ^ keys[g.r]));
        uint32_t a = g.l, b = g.r;                      // first key of the range with `bit` set (keys[l] has it clear, keys[r] set)
        while (a + 1 < b) { const uint32_t m = a + (b - a) / 2; if (keys[m] & bit) b = m; else a = m; }
        if (a > g.l) todo.push_back({ g.l, a, g.d + 1 });
        if (b < g.r) todo.push_back({ b, g.r, g.d + 1 });
    }
    return deepest;
}

void Mesh::set(const void *vertexArray, int vcount, int vstride, const unsigned int *indexArray, int icount) {
    if (!vertexArray || !indexArray || vcount <= 0 || icount <= 0 || vstride < 12) throw std::runtime_error("RT64_SetMesh: invalid arguments.");
    // An index past the vertex array would make the BLAS builder and the any-hit vertex fetches read outside the buffer (a GPU memory fault).
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    const int usedIndices = (flags & RT64_MESH_RAYTRACE_ENABLED) ? (icount / 3) * 3 : icount;
    for (int i = 0; i < icount; i++) {
        if (indexArray[i] >= (unsigned int)vcount) throw std::runtime_error("RT64_SetMesh: index " + std::to_string(i) + " (" + std::to_string(indexArray[i]) + ") is out of range for " + std::to_string(vcount) + " vertices.");
        if (i < usedIndices) {
            float p[3]; memcpy(p, static_cast<const uint8_t *>(vertexArray) + (size_t)indexArray[i] * vstride, 12);
            for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], p[k]); mx[k] = fmaxf(mx[k], p[k]); }
        }
    }
    memcpy(hostBmin, mn, 12); memcpy(hostBmax, mx, 12);
    device->enter();
    if (vertices.ptr) device->beforeSceneMutation();       // a kept lean frame may still read this mesh's arrays (a first upload changes nothing a frame has seen)
    // rt64_mesh.cpp:30-39,76-82: a change of counts/stride discards the BLAS even if updatable.
    const bool sameShape = vertices.ptr && vertexCount == vcount && vertexStride == vstride && indexCount == icount;
    // A build that is still only recorded has to run on the arrays it was recorded for when this call is going to REFIT its tree (the
    // reference's command list holds the build, then the update, in that order): launch it now, before the new arrays are queued behind it.
    // (Two refits, or anything followed by a full build, merge into the last one: only the latest positions matter there.)
    if (buildPending && !pendingRefit && (flags & RT64_MESH_RAYTRACE_ENABLED) && (flags & RT64_MESH_RAYTRACE_UPDATABLE) && sameShape && blasCount == (uint32_t)icount / 3)
        device->flushMeshBuilds();
    const size_t vbytes = (size_t)vcount * vstride, ibytes = (size_t)icount * 4;
    hostVertices.assign(static_cast<const uint8_t *>(vertexArray), static_cast<const uint8_t *>(vertexArray) + vbytes);
    vertices.reserve(vbytes); indices.reserve((size_t)icount);
    // rt64_mesh.cpp:53,96: the arrays are copied during the call; the device copies are queued behind them on the stream.
    uint8_t *stage = static_cast<uint8_t *>(device->ringAlloc(vbytes + ibytes));
    memcpy(stage, vertexArray, vbytes); memcpy(stage + vbytes, indexArray, ibytes);
    HIP_CHECK(hipMemcpyAsync(vertices.ptr, stage, vbytes, hipMemcpyHostToDevice, device->stream));
    HIP_CHECK(hipMemcpyAsync(indices.ptr, stage + vbytes, ibytes, hipMemcpyHostToDevice, device->stream));
    static uint32_t globalMeshVersion = 0;      // unique across meshes: a recycled allocation never aliases a cached frame table
    vertexCount = vcount; vertexStride = vstride; indexCount = icount; version = ++globalMeshVersion;
    if (flags & RT64_MESH_RAYTRACE_ENABLED) {                // rt64_mesh.cpp:114-126
        const uint32_t n = (uint32_t)icount / 3;
        if (n == 0) throw std::runtime_error("RT64_SetMesh: a ray-traced mesh needs at least one triangle.");
        const bool refit = (flags & RT64_MESH_RAYTRACE_UPDATABLE) && sameShape && blasCount == n;   // rt64_mesh.cpp:129,149-157
        // A refit keeps the topology of the tree that exists (or is about to exist, if its build is still pending).
        pendingRefit = buildPending ? (pendingRefit && refit) : refit;
        if (!pendingRefit) depth = host_blas_depth(static_cast<const uint8_t *>(vertexArray), (size_t)vstride, indexArray, n);
        nodes.reserve(std::max<size_t>(n - 1, 1)); tris.reserve((gpu_tri_array_bytes(n) + sizeof(GpuTri) - 1) / sizeof(GpuTri)); header.reserve(1);      // records + the fetch slack behind the last one (rt64_gpu.h: GPU_TRI_FETCH_SLACK_BYTES)
        if (tris.bytes() < gpu_tri_array_bytes(n)) throw std::runtime_error("RT64_SetMesh: triangle array without its fetch slack.");
        sortedIndex.reserve(n); morton.reserve(n); leafParent.reserve(n);
        if (n > LBVH_SMALL_MAX) buildScratch.reserve(lbvh_large_scratch_bytes(n));
        blasCount = n;
        if (!buildPending) { buildPending = true; device->dirtyMeshes.push_back(this); }
    }
}

Mesh::~Mesh() {
    if (buildPending) { auto &d = device->dirtyMeshes; d.erase(std::remove(d.begin(), d.end(), this), d.end()); }
}

// The recorded BLAS work of every mesh set since the last frame (the reference executes its upload command list the same way at the
// next preRender, rt64_device.cpp:979-983).
void Device::flushMeshBuilds() {
    if (dirtyMeshes.empty()) return;
    use();
    impure();
    std::vector<LbvhArgs> small; uint32_t maxN = 0;
    for (Mesh *m : dirtyMeshes) {
        LbvhArgs a = {};
        a.mode = LBVH_MODE_TRIANGLES; a.refit = m->pendingRefit ? 1 : 0; a.n = m->blasCount;
        a.vertices = m->vertices.ptr; a.vertexStride = (uint32_t)m->vertexStride; a.indices = m->indices.ptr;
        a.nodes = m->nodes.ptr; a.tris = m->tris.ptr; a.header = m->header.ptr; a.sortedIndex = m->sortedIndex.ptr; a.morton = m->morton.ptr; a.leafParent = m->leafParent.ptr;
        if (a.n > LBVH_SMALL_MAX) { a.scratch = m->buildScratch.ptr; a.scratchBytes = m->buildScratch.bytes(); HIP_CHECK(lbvh_launch_large(a, stream)); }
        else { small.push_back(a); maxN = std::max(maxN, a.n); }
        if (!m->pendingRefit) m->topologyVersion++;
        m->buildPending = false; m->pendingRefit = false;
    }
    dirtyMeshes.clear();
    if (small.size() == 1) HIP_CHECK(lbvh_launch(small[0], stream));
    else if (!small.empty()) {
        const size_t bytes = small.size() * sizeof(LbvhArgs);
        buildArgs.reserve(small.size());
        void *stage = ringAlloc(bytes);
        memcpy(stage, small.data(), bytes);
        HIP_CHECK(hipMemcpyAsync(buildArgs.ptr, stage, bytes, hipMemcpyHostToDevice, stream));
        HIP_CHECK(lbvh_launch_batch(buildArgs.ptr, (uint32_t)small.size(), maxN, stream));
    }
    workSinceMark = true;
}

void Device::beforeSceneMutation() {
    for (Scene *sc : scenes) for (View *v : sc->views) if (v->leanFrame) { v->materialise(); leanHoldoff = RT64_LEAN_HOLDOFF_FRAMES; }
}

std::pair<float, float> Mesh::inputAlphaBounds(int offset) {
    if (alphaBoundsVersion != version) { alphaBounds.clear(); alphaBoundsVersion = version; }
    auto it = alphaBounds.find(offset);
    if (it != alphaBounds.end()) return it->second;
    float mn = INFINITY, mx = -INFINITY;
    for (int v = 0; v < vertexCount; v++) {
        if ((size_t)v * vertexStride + offset + 16 > hostVertices.size()) { mn = NAN; mx = NAN; break; }
        float a; memcpy(&a, hostVertices.data() + (size_t)v * vertexStride + offset + 12, 4);
        if (!(a >= mn)) mn = a;
        if (!(a <= mx)) mx = a;
    }
    return alphaBounds[offset] = std::make_pair(mn, mx);
}

// ---- Scene / Instance ---------------------------------------------------------------------------------------------------------

Scene::Scene(Device *d) : device(d) {
    // rt64_scene.cpp:19-27
    desc.eyeLightDiffuseColor = { 0.08f, 0.08f, 0.08f }; desc.eyeLightSpecularColor = { 0.04f, 0.04f, 0.04f };
    desc.skyDiffuseMultiplier = { 1.0f, 1.0f, 1.0f }; desc.skyHSLModifier = { 0.0f, 0.0f, 0.0f };
    desc.skyYawOffset = 0.0f; desc.giDiffuseStrength = 0.7f; desc.giSkyStrength = 0.35f;
    d->scenes.push_back(this);
}
Scene::~Scene() {
    device->scenes.erase(std::remove(device->scenes.begin(), device->scenes.end(), this), device->scenes.end());
    auto viewsCopy = views; for (View *v : viewsCopy) delete v;                 // rt64_scene.cpp:43-51
    auto instancesCopy = instances; for (Instance *i : instancesCopy) delete i;
}
Instance::Instance(Scene *s) : scene(s) { s->instances.push_back(this); }
Instance::~Instance() { auto &v = scene->instances; v.erase(std::remove(v.begin(), v.end(), this), v.end()); }

// ---- View ---------------------------------------------------------------------------------------------------------------------

// Render size and jitter phase count of the built-in temporal upscaler for a display size; false = (upscaler, mode) selects none.
// Quality table: rt64_fsr.cpp:98-126 (Native 100 %, UltraQuality 77 %, FSR2's published 1.5 / 1.7 / 2.0 / 3.0 ratios), Auto by display
// size rt64_upscaler.cpp:11-36; phases = int(8 (display / render)^2) is what ffxFsr2GetJitterPhaseCount (rt64_fsr.cpp:128-130) returns.
static bool upscaler_info(int upscaler, int mode, int displayW, int displayH, int &renderW, int &renderH, int &phases) {
    if (!(upscaler == RT64_UPSCALER_AUTO || upscaler == RT64_UPSCALER_FSR)) return false;
    if (mode == RT64_UPSCALER_MODE_AUTO) {
        const uint64_t px = (uint64_t)displayW * (uint64_t)displayH;
        mode = px <= 1280ull * 720 ? RT64_UPSCALER_MODE_ULTRA_QUALITY : (px <= 1920ull * 1080 ? RT64_UPSCALER_MODE_QUALITY : (px <= 2560ull * 1440 ? RT64_UPSCALER_MODE_BALANCED :
               (px <= 3840ull * 2160 ? RT64_UPSCALER_MODE_PERFORMANCE : RT64_UPSCALER_MODE_ULTRA_PERFORMANCE)));
    }
    int w, h;
    if (mode == RT64_UPSCALER_MODE_NATIVE) { w = displayW; h = displayH; }
    else if (mode == RT64_UPSCALER_MODE_ULTRA_QUALITY) { w = (displayW * 77) / 100; h = (displayH * 77) / 100; }
    else {
        const float ratio = mode == RT64_UPSCALER_MODE_QUALITY ? 1.5f : (mode == RT64_UPSCALER_MODE_BALANCED ? 1.7f : (mode == RT64_UPSCALER_MODE_PERFORMANCE ? 2.0f : 3.0f));
        w = (int)((float)displayW / ratio); h = (int)((float)displayH / ratio);
    }
    renderW = std::max(w, 1); renderH = std::max(h, 1);
    const float q = (float)displayW / (float)renderW;
    phases = std::max((int)(8.0f * (q * q)), 1);
    return true;
}
static float halton_sequence(int i, int b) {          // rt64_common.h:347-357
    float f = 1.0f, r = 0.0f;
    while (i > 0) { f = f / (float)b; r = r + f * (float)(i % b); i = i / b; }
    return r;
}

View::View(Scene *s) : scene(s) {
    s->views.push_back(this);
    createImages(s->device->width, s->device->height, s->device->width, s->device->height);
}
View::~View() {
    auto &v = scene->views; v.erase(std::remove(v.begin(), v.end(), this), v.end());
    releaseImages();
}
void View::releaseImages() {
    for (void *p : allocations) hipFree(p);
    allocations.clear(); img = ViewImages(); bounceSamples = 0; leanFrame = false; fusedFrame = false; for (auto &fb : finalBuf) fb = nullptr;
    for (auto &u : upscaled) { if (u) hipFree(u); u = nullptr; }
    upW = upH = 0; upValid = false;
}

void View::createImages(int w, int h, int screenW, int screenH) {       // View::createOutputBuffers, rt64_view.cpp:105-298 (same formats)
    scene->device->use();
    scene->device->impure();
    releaseImages();
    const size_t n = (size_t)w * h;
    auto alloc = [&](size_t bytes) { void *p = nullptr; HIP_CHECK(hipMalloc(&p, bytes)); HIP_CHECK(hipMemsetAsync(p, 0, bytes, scene->device->stream)); allocations.push_back(p); return p; };
    img.viewDirection = static_cast<uint16_t *>(alloc(n * 8)); img.shadingPosition = static_cast<float *>(alloc(n * 16));
    img.shadingNormal = static_cast<uint16_t *>(alloc(n * 8)); img.shadingSpecular = static_cast<uint16_t *>(alloc(n * 8));
    img.diffuse = static_cast<uint8_t *>(alloc(n * 4));
    img.instanceId = static_cast<int32_t *>(alloc(n * 4)); img.firstInstanceId = static_cast<int32_t *>(alloc(n * 4));
    for (int i = 0; i < 2; i++) {
        img.directLight[i] = static_cast<uint16_t *>(alloc(n * 8)); img.indirectLight[i] = static_cast<uint16_t *>(alloc(n * 8));
        img.filteredDirect[i] = static_cast<uint16_t *>(alloc(n * 8)); img.filteredIndirect[i] = static_cast<uint16_t *>(alloc(n * 8));
        img.normal[i] = static_cast<uint16_t *>(alloc(n * 8)); img.depth[i] = static_cast<float *>(alloc(n * 4));
        img.moments[i] = static_cast<float *>(alloc(n * 16));
        if (i == 0) img.svgfGuide = static_cast<uint4 *>(alloc(n * 16));
        if (i == 0) { const size_t young = (size_t)((w + 31) / 32) * (size_t)h * 4; img.svgfYoung = static_cast<uint32_t *>(alloc(young)); }
        if (i == 0) { img.reflectFlags = static_cast<uint32_t *>(alloc(64)); }
    }
    img.reflection = static_cast<uint16_t *>(alloc(n * 8)); img.refraction = static_cast<uint16_t *>(alloc(n * 8)); img.transparent = static_cast<uint16_t *>(alloc(n * 8));
    img.flow = static_cast<uint16_t *>(alloc(n * 4));
    img.reactiveMask = static_cast<uint8_t *>(alloc(n)); img.lockMask = static_cast<uint8_t *>(alloc(n));
    img.output = static_cast<float *>(alloc(n * 16));
    for (int k = 0; k < scene->device->streamCount; k++) finalBuf[k] = static_cast<uint8_t *>(alloc((size_t)screenW * screenH * 4));
    img.final = finalBuf[scene->device->cur];
    img.primaryHit = static_cast<uint32_t *>(alloc(n * 16));
    hitInstance.reserve(n);
    HIP_CHECK(hipMemsetAsync(hitInstance.ptr, 0xFF, n * 4, scene->device->stream));
    imgW = w; imgH = h; finalW = screenW; finalH = screenH;
    skipReprojection = true;                  // rt64_view.cpp:143
    fprintf(stdout, "Render buffer: %dX%d\n", w, h);      // rt64_view.cpp:150
}

// Static opacity rule O1: an instance is opaque when every hit it can produce stores alpha 255 in the RGBA8
// hit colour, so nothing behind it can contribute (PrimaryRayGen.hlsl:150,174).  Decided from bounds on the combiner's
// alpha sources: vertex input alphas over the mesh, texel alphas over the texture.
static float instance_alpha_lower_bound(const Instance *inst, const GpuCombiner &cc) {
    if (cc.optNoise || cc.optTextureEdge) return -1.0f;
    float lo;
    if (!cc.optAlpha) lo = 1.0f;
    else {
        float l[4], h[4];
        for (int k = 0; k < 4; k++) {
            int item = cc.c[1][k];
            if (item == 0) { l[k] = h[k] = 0.0f; }
            else if (item <= 4) { auto b = inst->mesh->inputAlphaBounds(cc.inputOffset[item - 1]); l[k] = b.first; h[k] = b.second; }
            else if (item <= 6) { l[k] = (float)inst->diffuse->minAlpha / 255.0f; h[k] = (float)inst->diffuse->maxAlpha / 255.0f; }
            else { l[k] = h[k] = 1.0f; }
        }
        if (cc.doSingle[1]) lo = l[3];
        else if (cc.doMultiply[1]) lo = std::min(std::min(l[0] * l[2], l[0] * h[2]), std::min(h[0] * l[2], h[0] * h[2]));
        else return -1.0f;
    }
    return lo;
}
static bool instance_is_opaque(const Instance *inst, const GpuCombiner &cc) {
    const float lo = instance_alpha_lower_bound(inst, cc);
    return lo >= 0.0f && inst->material.solidAlphaMultiplier * lo >= 0.999f;
}
// Rule O2: the shadow any-hit (rt64_shader.cpp:611-659) saturates payload.shadowHit on the first hit when
// alpha * shadowAlphaMultiplier is provably >= 0.999 for the whole instance; a combiner without opt_alpha always does (:661).
static bool instance_is_shadow_opaque(const Instance *inst, const GpuCombiner &cc) {
    if (!cc.optAlpha) return true;
    const float lo = instance_alpha_lower_bound(inst, cc);
    return lo >= 0.0f && inst->material.shadowAlphaMultiplier * lo >= 0.999f;
}

// Build (or reuse) the device table and the triangle setup records of one raster draw list.  The records only depend on the
// table bytes, the target size, the row range and the scissor mode: an unchanged HUD costs nothing here after its first frame.
void View::prepareRasterList(const std::vector<RenderInstance> &list, RasterList &rl, int w, int h, int y0, int y1, bool apply) {
    rl.changed = false; rl.contentChanged = false;
    if (list.empty()) { rl.contentChanged = rl.ready; rl.triTotal = 0; rl.ready = false; rl.uploaded.clear(); return; }
    Device *dev = scene->device;
    std::vector<GpuRasterInstance> hst(list.size());
    uint32_t triTotal = 0;
    float bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
    for (size_t i = 0; i < list.size(); i++) {
        Instance *inst = list[i].instance;
        GpuRasterInstance &g = hst[i];
        memset(&g, 0, sizeof(g));
        g.vertices = inst->mesh->vertices.ptr; g.indices = inst->mesh->indices.ptr;
        g.vertexStride = (uint32_t)inst->mesh->vertexStride; g.triCount = (uint32_t)inst->mesh->indexCount / 3; g.firstTri = triTotal;
        g.cc = inst->shader->cc;
        g.texDiffuse = inst->diffuse->currentIndex; g.filter = inst->shader->filter; g.hAddr = inst->shader->hAddr; g.vAddr = inst->shader->vAddr;
        g.texSerial = inst->diffuse->serial;
        g.scissorRect[0] = inst->scissorRect.x; g.scissorRect[1] = inst->scissorRect.y; g.scissorRect[2] = inst->scissorRect.w; g.scissorRect[3] = inst->scissorRect.h;
        g.viewportRect[0] = inst->viewportRect.x; g.viewportRect[1] = inst->viewportRect.y; g.viewportRect[2] = inst->viewportRect.w; g.viewportRect[3] = inst->viewportRect.h;
        g.meshVersion = inst->mesh->version;
        // the raster input layout reads position float4 + the attributes the combiner uses (rt64_shader.cpp:389-398)
        if (g.cc.vertexSize > inst->mesh->vertexStride || inst->mesh->vertexStride < 16) throw std::runtime_error("Raster instance mesh vertex stride is smaller than the layout its shader reads.");
        triTotal += g.triCount;
        // conservative screen bounds of the instance (same transform as raster spec S1, +-1 pixel)
        float vpX = 0.0f, vpY = 0.0f, vpW = (float)w, vpH = (float)h;
        if (apply && inst->viewportRect.w > 0 && inst->viewportRect.h > 0) { vpX = (float)inst->viewportRect.x; vpY = (float)(h - inst->viewportRect.y - inst->viewportRect.h); vpW = (float)inst->viewportRect.w; vpH = (float)inst->viewportRect.h; }
        for (int v = 0; v < inst->mesh->vertexCount; v++) {
            float p[4]; memcpy(p, inst->mesh->hostVertices.data() + (size_t)v * inst->mesh->vertexStride, 16);
            // a vertex outside the clipper's planes (w, near / far, guard band): the clipped pieces can land anywhere on the target
            if (!(p[3] > 1e-3f) || !(fabsf(p[0]) <= 4.0f * p[3]) || !(fabsf(p[1]) <= 4.0f * p[3])) { bx0 = by0 = -1e9f; bx1 = by1 = 1e9f; continue; }
            const float rw = 1.0f / p[3], xs = ((p[0] * rw) * 0.5f + 0.5f) * vpW + vpX, ys = (0.5f - (p[1] * rw) * 0.5f) * vpH + vpY;
            bx0 = std::min(bx0, xs); bx1 = std::max(bx1, xs); by0 = std::min(by0, ys); by1 = std::max(by1, ys);
        }
    }
    const size_t bytes = hst.size() * sizeof(GpuRasterInstance);
    const bool sameContent = rl.ready && rl.uploaded.size() == bytes && memcmp(rl.uploaded.data(), hst.data(), bytes) == 0 &&
                             rl.w == w && rl.h == h && rl.y0 == y0 && rl.y1 == y1 && rl.apply == apply;
    if (sameContent && !dev->opt.alwaysRebuild) return;
    dev->impure();
    rl.contentChanged = !sameContent;
    rl.table.reserve(hst.size()); rl.tris.reserve(std::max<size_t>(raster_tri_bytes(triTotal), 16));
    const bool inlineTable = raster_setup_takes_table_inline((uint32_t)hst.size());      // a short list rides in the setup kernel's arguments: no copy on the stream
    if (!inlineTable) {
        uint8_t *stage = static_cast<uint8_t *>(dev->ringAlloc(bytes));       // pinned upload ring: queued without a wait
        memcpy(stage, hst.data(), bytes);
        HIP_CHECK(hipMemcpyAsync(rl.table.ptr, stage, bytes, hipMemcpyHostToDevice, dev->stream));
    }
    rl.uploaded.assign(reinterpret_cast<uint8_t *>(hst.data()), reinterpret_cast<uint8_t *>(hst.data()) + bytes);
    rl.triTotal = triTotal; rl.w = w; rl.h = h; rl.y0 = y0; rl.y1 = y1; rl.apply = apply; rl.ready = true; rl.changed = true;
    rl.bounds[0] = std::max(0, (int)std::floor(std::min(bx0, 1e9f)) - 1); rl.bounds[1] = std::max(y0, (int)std::floor(std::min(by0, 1e9f)) - 1);
    rl.bounds[2] = std::min(w, (int)std::ceil(std::max(bx1, -1e9f)) + 2); rl.bounds[3] = std::min(y1, (int)std::ceil(std::max(by1, -1e9f)) + 2);
    if (inlineTable && dev->opt.framePrologue && hst.size() <= RASTER_PROLOGUE_INSTANCES && prologue.listCount < RASTER_PROLOGUE_LISTS) {       // leaves with the frame's other uploads (flushPrologue)
        RasterPrologueList &L = prologue.list[prologue.listCount++];
        memset(&L, 0, sizeof(L));
        memcpy(L.inst, hst.data(), bytes);
        L.deviceTable = rl.table.ptr; L.tris = rl.tris.ptr; L.instanceCount = (uint32_t)hst.size(); L.triTotal = triTotal; L.w = w; L.h = h; L.y0 = y0; L.y1 = y1; L.apply = apply ? 1 : 0;
    }
    else if (inlineTable) HIP_CHECK(launch_raster_setup_inline(hst.data(), rl.table.ptr, (uint32_t)hst.size(), triTotal, rl.tris.ptr, w, h, y0, y1, apply, dev->stream));
    else HIP_CHECK(launch_raster_setup(rl.table.ptr, (uint32_t)hst.size(), triTotal, rl.tris.ptr, w, h, y0, y1, apply, dev->stream));
}

void View::flushTableCopy() {
    if (!prologueCopyBytes) return;
    HIP_CHECK(hipMemcpyAsync(prologue.copyDst, prologue.copySrc, prologueCopyBytes, hipMemcpyHostToDevice, scene->device->stream));
    prologueCopyBytes = 0; prologue.copySrc = nullptr; prologue.copyDst = nullptr;
}
void View::flushPrologue() {
    Device *dev = scene->device;
    if (prologue.listCount == 0) { flushTableCopy(); return; }        // tables alone: the runtime's own copy (measured equal: moving_instance 0.1685 ms either way)
    prologue.copyWords = (uint32_t)((prologueCopyBytes + 15) / 16);
    if (prologueCopyBytes > (size_t)RASTER_PROLOGUE_COPY_WORDS * 16) { flushTableCopy(); prologue.copyWords = 0; }
    HIP_CHECK(launch_frame_prologue(prologue, dev->stream));
    prologueCopyBytes = 0;
    memset(&prologue, 0, sizeof(prologue));
    dev->workSinceMark = true;
}

void View::drawRasterList(RasterList &rl, uint8_t *target, bool clear) {
    if (!rl.ready || rl.triTotal == 0) return;
    Device *dev = scene->device;
    HIP_CHECK(launch_raster_draw(rl.table.ptr, rl.tris.ptr, rl.triTotal, tab[tabCur].dTextures.ptr, target, rl.w, rl.y0, rl.y1, rl.bounds, dev->stripRank, dev->stripCount, clear, dev->stream));
    dev->workSinceMark = true;
}

// ---- TLAS of a few instances, built on the host ----------------------------------------------------------------------------------------
// The same tree the GPU builder makes (Geometry spec G1-G6 of oracle/oracle_bvh.c / lbvh.hip: instance boxes from the eight transformed
// corners, 30-bit Morton codes of the box centres, stable sort by (code, instance), Karras' radix tree, children's boxes in the parent),
// bit for bit -- the operations are IEEE min / max / fma in the same order.  For the handful of instances of a small frame it costs a few
// microseconds of host time and travels in the frame-table upload, where the single-workgroup kernel costs ~17 us of serial GPU time
// (dependent cold loads) plus a launch on every frame whose tables changed.
#define RT64_HOST_TLAS_MAX 64
// v_min_f32 / v_max_f32 order -0 below +0; the host's fminf / fmaxf may return either.  (No NaN reaches here that the GPU builder would survive.)
static inline float gmin(float a, float b) { return (a < b || (a == b && std::signbit(a))) ? a : b; }
static inline float gmax(float a, float b) { return (a > b || (a == b && !std::signbit(a))) ? a : b; }
static void host_build_tlas(const GpuInstance *inst, const float (*meshMin)[3], const float (*meshMax)[3], uint32_t n,
                            GpuNode *nodes, uint32_t *sortedIndex, uint32_t *morton, BlasHeader *header) {
    std::vector<float> lo(3 * (size_t)n), hi(3 * (size_t)n);
    float smin[3] = { INFINITY, INFINITY, INFINITY }, smax[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = 0; i < n; i++) {                                    // G1
        float bmn[3] = { INFINITY, INFINITY, INFINITY }, bmx[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int c = 0; c < 8; c++) {
            const float p[3] = { (c & 1) ? meshMax[i][0] : meshMin[i][0], (c & 2) ? meshMax[i][1] : meshMin[i][1], (c & 4) ? meshMax[i][2] : meshMin[i][2] };
            float w[3]; g_xform_point(inst[i].objectToWorld, p, w);
            for (int k = 0; k < 3; k++) { bmn[k] = gmin(bmn[k], w[k]); bmx[k] = gmax(bmx[k], w[k]); }
        }
        for (int k = 0; k < 3; k++) { lo[3 * i + k] = bmn[k]; hi[3 * i + k] = bmx[k]; smin[k] = gmin(smin[k], bmn[k]); smax[k] = gmax(smax[k], bmx[k]); }
    }
    std::vector<unsigned long long> keys(n);                             // G2, G3
    for (uint32_t i = 0; i < n; i++) {
        uint32_t q[3];
        for (int k = 0; k < 3; k++) {
            const float ext = smax[k] - smin[k], scale = ext > 0.0f ? 1024.0f / ext : 0.0f;
            const float c = (lo[3 * i + k] + hi[3 * i + k]) * 0.5f, f = (c - smin[k]) * scale;
            const int qi = (int)f;
            q[k] = (uint32_t)(qi < 0 ? 0 : (qi > 1023 ? 1023 : qi));
        }
        keys[i] = ((unsigned long long)host_morton30(q[0], q[1], q[2]) << 32) | i;
    }
    std::sort(keys.begin(), keys.end());
    for (uint32_t s = 0; s < n; s++) { sortedIndex[s] = (uint32_t)(keys[s] & 0xFFFFFFFFull); morton[s] = (uint32_t)(keys[s] >> 32); }
    memset(nodes, 0, sizeof(GpuNode) * std::max<uint32_t>(n - 1, 1));
    auto leafLo = [&](uint32_t s) { return &lo[3 * (size_t)sortedIndex[s]]; };
    auto leafHi = [&](uint32_t s) { return &hi[3 * (size_t)sortedIndex[s]]; };
    if (n == 1) {                                                        // G5, single leaf
        GpuNode &nd = nodes[0];
        nd.left = RT64_LEAF_BIT; nd.right = RT64_NO_CHILD; nd.parent = RT64_NO_CHILD;
        for (int k = 0; k < 3; k++) { nd.lmin[k] = leafLo(0)[k]; nd.lmax[k] = leafHi(0)[k]; nd.rmin[k] = INFINITY; nd.rmax[k] = -INFINITY; header->bmin[k] = nd.lmin[k]; header->bmax[k] = nd.lmax[k]; }
        header->count = 1; header->depth = 1;
        return;
    }
    const int N = (int)n;                                                // G4
    auto delta = [&](int i, int j) -> int { return (j < 0 || j >= N) ? -1 : __builtin_clzll(keys[(size_t)i] ^ keys[(size_t)j]); };
    nodes[0].parent = RT64_NO_CHILD;
    for (int i = 0; i < N - 1; i++) {
        const int d = (delta(i, i + 1) - delta(i, i - 1)) >= 0 ? 1 : -1, dmin = delta(i, i - d);
        int lmax = 2;
        while (delta(i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2) if (delta(i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d, dnode = delta(i, j);
        int s = 0;
        for (int div = 2;; div *= 2) { const int t = (l + div - 1) / div; if (delta(i, i + (s + t) * d) > dnode) s += t; if (t <= 1) break; }
        const int g = i + s * d + (d < 0 ? -1 : 0), lo_ = i < j ? i : j, hi_ = i < j ? j : i;
        if (lo_ == g) nodes[i].left = RT64_LEAF_BIT | (uint32_t)g; else { nodes[i].left = (uint32_t)g; nodes[g].parent = (uint32_t)i; }
        if (hi_ == g + 1) nodes[i].right = RT64_LEAF_BIT | (uint32_t)(g + 1); else { nodes[i].right = (uint32_t)(g + 1); nodes[g + 1].parent = (uint32_t)i; }
        nodes[i].pad = (uint32_t)j;
    }
    struct Fit { static uint32_t run(GpuNode *nodes, uint32_t i, const std::vector<float> &lo, const std::vector<float> &hi, const uint32_t *sorted, float *mn, float *mx) {
        GpuNode &nd = nodes[i]; uint32_t depth = 0;                      // G5: children's boxes in the parent; returns the subtree's depth in inner nodes
        for (int side = 0; side < 2; side++) {
            const uint32_t c = side ? nd.right : nd.left;
            float *cmn = side ? nd.rmin : nd.lmin, *cmx = side ? nd.rmax : nd.lmax;
            if (c & RT64_LEAF_BIT) { const size_t inst = sorted[c & 0x7FFFFFFFu]; for (int k = 0; k < 3; k++) { cmn[k] = lo[3 * inst + k]; cmx[k] = hi[3 * inst + k]; } }
            else depth = std::max(depth, run(nodes, c, lo, hi, sorted, cmn, cmx));
        }
        for (int k = 0; k < 3; k++) { mn[k] = gmin(nd.lmin[k], nd.rmin[k]); mx[k] = gmax(nd.lmax[k], nd.rmax[k]); }
        return depth + 1;
    } };
    float rmn[3], rmx[3];
    header->depth = Fit::run(nodes, 0, lo, hi, sortedIndex, rmn, rmx);
    for (int k = 0; k < 3; k++) { header->bmin[k] = rmn[k]; header->bmax[k] = rmx[k]; }
    header->count = n;
}

void View::update() {                          // View::update, rt64_view.cpp:1053-1178
    Device *dev = scene->device;
    memset(&prologue, 0, sizeof(prologue)); prologueCopyBytes = 0;       // (nothing is left from a frame that threw half-way)
    {   // View::createOutputBuffers: render size = lround(screen * resolutionScale) (rt64_view.cpp:138-139)
        const float scale = resolutionScale > 0.0f ? resolutionScale : 1.0f;
        int rw = std::max(1, (int)lroundf((float)dev->width * scale)), rh = std::max(1, (int)lroundf((float)dev->height * scale));
        // an upscaler decides the render size itself (rt64_view.cpp:114-136; upscalerResolutionOverride is off and not reachable through the C API)
        int uw = 0, uh = 0, phases = 1;
        upscaleActive = upscaler_info(upscaler, upscalerMode, dev->width, dev->height, uw, uh, phases);
        if (upscaleActive) { rw = uw; rh = uh; jitterPhases = phases; }
        if (imgW != rw || imgH != rh || finalW != dev->width || finalH != dev->height) createImages(rw, rh, dev->width, dev->height);
        if (upscaleActive && (upW != dev->width || upH != dev->height)) {
            for (auto &u : upscaled) { if (u) hipFree(u); u = nullptr; HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&u), (size_t)dev->width * dev->height * 16)); }
            upW = dev->width; upH = dev->height; upValid = false;
        }
        if (!upscaleActive) upValid = false;
    }
    usedTextures.clear();
    auto textureIndex = [&](Texture *t) -> int {
        if (!t) return -1;
        if (t->currentIndex < 0) { t->currentIndex = (int)usedTextures.size(); usedTextures.push_back(t); }
        return t->currentIndex;
    };
    int skyIndex = textureIndex(skyPlane);     // the sky plane gets slot 0 when set (rt64_view.cpp:1079)
    (void)skyIndex;
    rtInstances.clear(); rasterBg.clear(); rasterFg.clear();
    for (Instance *inst : scene->instances) {
        if (!inst->mesh || !inst->shader || !inst->diffuse) continue;          // asserted non-NULL in the reference (rt64_instance.cpp:152-155)
        textureIndex(inst->diffuse); textureIndex(inst->normal); textureIndex(inst->specular);
        RenderInstance ri = { inst };
        if (inst->mesh->blasCount > 0) rtInstances.push_back(ri);             // rt64_view.cpp:1138-1146
        else if (inst->flags & RT64_INSTANCE_RASTER_BACKGROUND) rasterBg.push_back(ri);
        else rasterFg.push_back(ri);
    }
    {
        const int sw = dev->width, sh = dev->height;
        rtViewport[0] = 0.0f; rtViewport[1] = 0.0f; rtViewport[2] = (float)sw; rtViewport[3] = (float)sh;
        rtScissor[0] = 0; rtScissor[1] = 0; rtScissor[2] = sw; rtScissor[3] = sh; rtRect = false;
        if (!rtInstances.empty()) {
            const Instance *i0 = rtInstances[0].instance;
            if (i0->scissorRect.w > 0 && i0->scissorRect.h > 0) { rtScissor[0] = i0->scissorRect.x; rtScissor[1] = sh - i0->scissorRect.y - i0->scissorRect.h; rtScissor[2] = i0->scissorRect.x + i0->scissorRect.w; rtScissor[3] = sh - i0->scissorRect.y; rtRect = true; }
            if (i0->viewportRect.w > 0 && i0->viewportRect.h > 0) { rtViewport[0] = (float)i0->viewportRect.x; rtViewport[1] = (float)(sh - i0->viewportRect.y - i0->viewportRect.h); rtViewport[2] = (float)i0->viewportRect.w; rtViewport[3] = (float)i0->viewportRect.h; rtRect = true; }
        }
    }
    if (separatePost() && (dev->stripCount > 1 || dev->tileY0 != 0 || dev->tileY1 != dev->height))
        throw std::runtime_error("RT64_DrawDevice: resolutionScale != 1, motion blur and a viewport / scissor on the ray-traced picture resample across rows; they need the whole frame on one device (no RT64_SetDeviceTile / RT64_SetDeviceInterleave).");
    if (usedTextures.size() > 512) throw std::runtime_error("More than 512 textures in one frame (SRV_TEXTURES_MAX).");

    // Tables: instances (transforms rt64_view.cpp:348-376, materials :388-410), textures, lights -> one staged upload.
    const size_t nInst = rtInstances.size(), nTex = usedTextures.size(), nLights = scene->lights.size();
    const size_t instBytes = nInst * sizeof(GpuInstance), texBytes = nTex * sizeof(GpuTexture), lightBytes = nLights * sizeof(RT64_LIGHT);
    tableScratch.resize(instBytes + texBytes + lightBytes + 64);
    uint8_t *stage = tableScratch.data();
    GpuInstance *hInst = reinterpret_cast<GpuInstance *>(stage);
    GpuTexture *hTex = reinterpret_cast<GpuTexture *>(stage + instBytes);
    RT64_LIGHT *hLights = reinterpret_cast<RT64_LIGHT *>(stage + instBytes + texBytes);
    // LDS scene cache (trace.h): instance records + TLAS nodes + every BLAS node array, if that is small enough to sit next to the
    // traversal stacks of a workgroup.  Offsets are in 16-byte words.
    std::vector<uint32_t> cacheOffset(nInst, 0);
    cacheWords = 0;
    if (dev->opt.ldsCache && nInst >= 1 && nInst <= 16) {
        uint64_t words = 4 * nInst + 4 * std::max<size_t>(nInst - 1, 1);
        for (size_t i = 0; i < nInst; i++) { cacheOffset[i] = (uint32_t)words; words += 4ull * std::max<uint32_t>(rtInstances[i].instance->mesh->blasCount - 1, 1); }
        // ... and the walk's stack (at most one entry per level: TLAS depth + the deepest BLAS) has to fit the LDS-only stack of the cached kernels
        uint32_t deepest = 0;
        if (words <= RT_CACHE_MAX_WORDS) for (size_t i = 0; i < nInst; i++) deepest = std::max(deepest, rtInstances[i].instance->mesh->treeDepth());
        if (words <= RT_CACHE_MAX_WORDS && std::max<size_t>(nInst - 1, 1) + deepest <= RT_STACK_LDS_CACHED) cacheWords = (uint32_t)words;
    }
    {   // The HBM spill slab behind the LDS stacks (84 entries per resident lane: 0.7 GB) is only needed when a walk can go deeper than the LDS
        // entries: TLAS depth (at most nInst - 1) + the deepest BLAS (the builders leave it in the header; 255 = "deep" for the multi-kernel path).
        uint32_t deepest = 0;
        for (size_t i = 0; i < nInst; i++) deepest = std::max(deepest, rtInstances[i].instance->mesh->treeDepth());
        needSpillSlab = nInst > 0 && std::max<size_t>(nInst - 1, 1) + deepest > RT_STACK_LDS;
    }
    maxDepthBias = nInst ? -INFINITY : 0.0f;
    anyNonOpaque = anyReflection = anyRefraction = anyFog = false;
    simpleFrame = dev->opt.simpleKernels && nInst > 0;
    for (size_t i = 0; i < nInst; i++) {
        Instance *inst = rtInstances[i].instance;
        GpuInstance &g = hInst[i];
        memset(&g, 0, sizeof(g));
        memcpy(g.objectToWorld, inst->transform.m, 64);
        memcpy(g.objectToWorldPrevious, subFrame > 0 ? inst->transform.m : inst->previousTransform.m, 64);      // (P2: nothing moves between the sub-frames of a primary_spp frame)
        Mat4 upper = inst->transform;           // rt64_view.cpp:358-368
        upper.m[3] = upper.m[7] = upper.m[11] = 0.0f; upper.m[12] = upper.m[13] = upper.m[14] = 0.0f; upper.m[15] = 1.0f;
        Mat4 nrm = mat_transpose(mat_inverse(upper));
        memcpy(g.objectToWorldNormal, nrm.m, 64);
        Mat4 w2o = mat_inverse(inst->transform);
        memcpy(g.worldToObject, w2o.m, 64);
        Mesh *mesh = inst->mesh;
        g.nodes = mesh->nodes.ptr; g.tris = mesh->tris.ptr; g.vertices = mesh->vertices.ptr; g.indices = mesh->indices.ptr; g.header = mesh->header.ptr;
        g.material = inst->material;
        g.cc = inst->shader->cc;
        g.texDiffuse = g.material.diffuseTexIndex = inst->diffuse->currentIndex;      // rt64_view.cpp:1110-1112
        g.texNormal = g.material.normalTexIndex = inst->normal ? inst->normal->currentIndex : -1;
        g.texSpecular = g.material.specularTexIndex = inst->specular ? inst->specular->currentIndex : -1;
        g.filter = inst->shader->filter; g.hAddr = inst->shader->hAddr; g.vAddr = inst->shader->vAddr;
        g.flags = 0;
        if (inst->flags & RT64_INSTANCE_DISABLE_BACKFACE_CULLING) g.flags |= GPU_INST_CULL_DISABLE;      // rt64_view.cpp:1109
        if (instance_is_opaque(inst, g.cc)) g.flags |= GPU_INST_OPAQUE;
        if (inst->shader->flags & RT64_SHADER_NORMAL_MAP_ENABLED) g.flags |= GPU_INST_NORMAL_MAP;
        if (inst->shader->flags & RT64_SHADER_SPECULAR_MAP_ENABLED) g.flags |= GPU_INST_SPECULAR_MAP;
        if (instance_is_shadow_opaque(inst, g.cc)) g.flags |= GPU_INST_SHADOW_OPAQUE;
        else simpleFrame = false;
        g.triCount = mesh->blasCount;
        g.meshVersion = mesh->version;
        g.cacheNodeOffset = cacheWords ? cacheOffset[i] : 0u;
        if (g.cc.vertexSize > mesh->vertexStride) throw std::runtime_error("Instance mesh vertex stride is smaller than the layout its shader reads.");
        maxDepthBias = std::max(maxDepthBias, inst->material.depthBias);
        if (!(g.flags & GPU_INST_OPAQUE)) anyNonOpaque = true;
        if (inst->material.reflectionFactor > 1e-6f) anyReflection = true;       // the passes below are no-ops otherwise
        if (inst->material.refractionFactor > 1e-6f) anyRefraction = true;
        if (inst->material.fogEnabled) anyFog = true;
    }
    for (size_t i = 0; i < nTex; i++) {
        Texture *t = usedTextures[i];
        GpuTexture &g = hTex[i];
        g.texels = t->texels.ptr; g.width = (uint32_t)t->width; g.height = (uint32_t)t->height; g.mips = (uint32_t)t->mips;
        g.pow2 = ((t->width & (t->width - 1)) == 0 && (t->height & (t->height - 1)) == 0) ? 1u : 0u;
        if (!g.pow2) simpleFrame = false;
        memcpy(g.mipOffset, t->mipOffset, sizeof(g.mipOffset));
    }
    if (nLights) memcpy(hLights, scene->lights.data(), lightBytes);
    // The tables (and with them the TLAS) only change when the host changed an instance, a mesh, a texture binding or a light.
    // Identical bytes => the device copies and the TLAS of the previous frame are still exact: skip upload and rebuild.
    const size_t tableBytes = instBytes + texBytes + lightBytes;
    const bool sameContent = tab[tabCur].uploadedTables.size() == tableBytes && tableBytes && memcmp(tab[tabCur].uploadedTables.data(), stage, tableBytes) == 0;
    const bool unchanged = sameContent && !dev->opt.alwaysRebuild;
    // A kept lean frame re-reads the tables, the TLAS and the cache image it was rendered with (View::materialise), and so does a frame that is still running on
    // another render stream: new contents go into ANOTHER slot and leave the last frame's alone.  Which one: a slot that only frames of THIS stream have read since its
    // last upload (stream order then protects it; an unused slot qualifies) -- with tables that change every frame each stream settles on a slot of its own, and a host
    // that waits for every frame alternates between two -- or, if there is none, the next one, behind a wait for the other streams.
    if (!unchanged) {
        int pick = -1;
        for (int k = 1; k < TABLE_SLOTS && pick < 0; k++) { const int c = (tabCur + k) % TABLE_SLOTS; if (!(tab[c].readers & ~(1u << dev->cur))) pick = c; }
        if (pick < 0) { pick = (tabCur + 1) % TABLE_SLOTS; dev->impure(); }
        tabCur = pick;
        tab[tabCur].readers = 0;
        // TLAS: full rebuild (rt64_view.cpp:412-452 rebuilds every frame, updateOnly = false) -- of a few instances on the host, into the same upload
        const uint32_t n = (uint32_t)nInst;
        const bool hostTlas = n >= 1 && n <= RT64_HOST_TLAS_MAX && dev->opt.hostTlas;
        const size_t tlasAt = (tableBytes + 63) & ~(size_t)63, nodeBytes = sizeof(GpuNode) * std::max<size_t>(n ? n - 1 : 0, 1);
        const bool hostCache = hostTlas && cacheWords != 0;
        const size_t cacheAt = (tlasAt + nodeBytes + sizeof(BlasHeader) + 2 * sizeof(uint32_t) * n + 63) & ~(size_t)63, headWords = hostCache ? 4 * (size_t)n + 4 * std::max<size_t>(n - 1, 1) : 0;
        const size_t uploadBytes = hostTlas ? (hostCache ? cacheAt + headWords * 16 : tlasAt + nodeBytes + sizeof(BlasHeader) + 2 * sizeof(uint32_t) * n) : tableBytes;
        {
            // (16-byte granules: the prologue kernel copies the upload as whole 16-byte words, and the ring rounds the region it reads them from to 256 bytes)
            const size_t need = (std::max<size_t>(hostCache ? cacheAt + (size_t)cacheWords * 16 : uploadBytes, 4096) + 15) & ~(size_t)15;
            if (need > tab[tabCur].dTables.count || (n && !hostTlas)) dev->impure();       // a slot that has to grow is freed and allocated again; a TLAS of more than 64 instances is built by kernels whose scratch is not per slot
            tab[tabCur].dTables.reserve(need);
        }
        tab[tabCur].dInstances.ptr = reinterpret_cast<GpuInstance *>(tab[tabCur].dTables.ptr); tab[tabCur].dTextures.ptr = reinterpret_cast<GpuTexture *>(tab[tabCur].dTables.ptr + instBytes);
        tab[tabCur].dLights.ptr = reinterpret_cast<RT64_LIGHT *>(tab[tabCur].dTables.ptr + instBytes + texBytes);
        if (uploadBytes) {       // staged in the pinned upload ring: no wait before the region is reused (a wrap-around of the ring drains the stream)
            uint8_t *pinnedStage = static_cast<uint8_t *>(dev->ringAlloc(uploadBytes));
            memcpy(pinnedStage, stage, tableBytes);
            if (hostTlas) {
                memset(pinnedStage + tableBytes, 0, tlasAt - tableBytes);
                GpuNode *hn = reinterpret_cast<GpuNode *>(pinnedStage + tlasAt);
                BlasHeader *hh = reinterpret_cast<BlasHeader *>(pinnedStage + tlasAt + nodeBytes);
                uint32_t *hi = reinterpret_cast<uint32_t *>(hh + 1), *hm = hi + n;
                float mn[RT64_HOST_TLAS_MAX][3], mx[RT64_HOST_TLAS_MAX][3];
                for (uint32_t i = 0; i < n; i++) { memcpy(mn[i], rtInstances[i].instance->mesh->hostBmin, 12); memcpy(mx[i], rtInstances[i].instance->mesh->hostBmax, 12); }
                host_build_tlas(hInst, mn, mx, n, hn, hi, hm, hh);
                if (hostCache) {        // head of the LDS scene cache image (layout: fill_scene_cache, passes.hip): one 64-byte record per TLAS leaf slot, then the TLAS nodes with 16-bit child references
                    memset(pinnedStage + tlasAt + nodeBytes + sizeof(BlasHeader) + 2 * sizeof(uint32_t) * n, 0, cacheAt - (tlasAt + nodeBytes + sizeof(BlasHeader) + 2 * sizeof(uint32_t) * n));
                    uint32_t *cw = reinterpret_cast<uint32_t *>(pinnedStage + cacheAt);
                    for (uint32_t k = 0; k < n; k++) {
                        const uint32_t inst = hi[k]; const GpuInstance &g = hInst[inst]; const float *M = g.worldToObject;
                        for (int c = 0; c < 3; c++) { const float row[4] = { M[c], M[4 + c], M[8 + c], M[12 + c] }; memcpy(cw + 16 * k + 4 * c, row, 16); }
                        const uint64_t tp = reinterpret_cast<uint64_t>(g.tris);
                        uint32_t info[4] = { inst | ((g.flags & 0xFFu) << 8) | (g.cacheNodeOffset << 16), 0u, (uint32_t)tp, (uint32_t)(tp >> 32) };
                        memcpy(&info[1], &g.material.depthBias, 4);
                        memcpy(cw + 16 * k + 12, info, 16);
                    }
                    auto childRef = [](uint32_t c) { return (c & RT64_LEAF_BIT) ? (c == RT64_NO_CHILD ? c : (0xFFFF8000u | (c & 0x7FFFu))) : c; };
                    const uint32_t tn = std::max<uint32_t>(n - 1, 1);
                    for (uint32_t t = 0; t < tn; t++) {
                        uint32_t *dst = cw + 16 * (size_t)n + 16 * (size_t)t;
                        memcpy(dst, &hn[t], 64);
                        dst[12] = childRef(hn[t].left); dst[13] = childRef(hn[t].right);
                    }
                }
                uint8_t *base = tab[tabCur].dTables.ptr + tlasAt;
                tab[tabCur].tlasNodesAt = reinterpret_cast<const GpuNode *>(base); tab[tabCur].tlasHeaderAt = reinterpret_cast<const BlasHeader *>(base + nodeBytes);
                tab[tabCur].tlasIndexAt = reinterpret_cast<const uint32_t *>(tab[tabCur].tlasHeaderAt + 1); tab[tabCur].tlasMortonAt = tab[tabCur].tlasIndexAt + n;
            }
            prologue.copySrc = pinnedStage; prologue.copyDst = tab[tabCur].dTables.ptr; prologueCopyBytes = uploadBytes;       // queued by flushPrologue / flushTableCopy
        }
        if (n && !hostTlas) {
            flushTableCopy();
            tab[tabCur].tlasNodes.reserve(std::max<size_t>(n - 1, 1)); tab[tabCur].tlasIndex.reserve(n); tab[tabCur].tlasMorton.reserve(n); tab[tabCur].tlasLeafParent.reserve(n); tab[tabCur].tlasHeader.reserve(1);
            LbvhArgs a = {};
            a.mode = LBVH_MODE_INSTANCES; a.refit = 0; a.n = n; a.instances = tab[tabCur].dInstances.ptr;
            a.nodes = tab[tabCur].tlasNodes.ptr; a.tris = nullptr; a.header = tab[tabCur].tlasHeader.ptr; a.sortedIndex = tab[tabCur].tlasIndex.ptr; a.morton = tab[tabCur].tlasMorton.ptr; a.leafParent = tab[tabCur].tlasLeafParent.ptr;
            if (n > LBVH_SMALL_MAX) { tab[tabCur].tlasScratch.reserve(lbvh_large_scratch_bytes(n)); a.scratch = tab[tabCur].tlasScratch.ptr; a.scratchBytes = tab[tabCur].tlasScratch.bytes(); }
            HIP_CHECK(lbvh_launch(a, dev->stream));
            tab[tabCur].tlasNodesAt = tab[tabCur].tlasNodes.ptr; tab[tabCur].tlasIndexAt = tab[tabCur].tlasIndex.ptr; tab[tabCur].tlasMortonAt = tab[tabCur].tlasMorton.ptr; tab[tabCur].tlasHeaderAt = tab[tabCur].tlasHeader.ptr;
        }
        tab[tabCur].cacheImageValid = false;
        if (hostCache) {
            // the BLAS node arrays behind the head: copied again only when one of them -- or the allocation they are copied into -- changed
            tab[tabCur].cacheImageAt = tab[tabCur].dTables.ptr + cacheAt;
            std::vector<uint64_t> key; key.reserve(2 + 4 * (size_t)n);
            key.push_back(reinterpret_cast<uint64_t>(tab[tabCur].cacheImageAt)); key.push_back(cacheWords);
            for (uint32_t i = 0; i < n; i++) { key.push_back(reinterpret_cast<uint64_t>(hInst[i].nodes)); key.push_back(hInst[i].meshVersion); key.push_back(hInst[i].cacheNodeOffset); key.push_back(hInst[i].triCount); }
            if (key != tab[tabCur].cacheBlasKey) {
                flushTableCopy();
                HIP_CHECK(launch_scene_cache_image(tab[tabCur].dInstances.ptr, tab[tabCur].tlasIndexAt, tab[tabCur].tlasNodesAt, n, const_cast<uint8_t *>(tab[tabCur].cacheImageAt), true, dev->stream));
                tab[tabCur].cacheBlasKey.swap(key);
            }
            tab[tabCur].cacheImageValid = true;
        }
        else tab[tabCur].cacheBlasKey.clear();
        tab[tabCur].uploadedTables.assign(stage, stage + tableBytes);
        dev->workSinceMark = true;
    }
    if (cacheWords && !tab[tabCur].cacheImageValid) {
        dev->impure();
        flushTableCopy();
        tab[tabCur].cacheImage.reserve((size_t)cacheWords * 16);
        HIP_CHECK(launch_scene_cache_image(tab[tabCur].dInstances.ptr, tab[tabCur].tlasIndexAt, tab[tabCur].tlasNodesAt, (uint32_t)nInst, tab[tabCur].cacheImage.ptr, false, dev->stream));
        tab[tabCur].cacheImageAt = tab[tabCur].cacheImage.ptr; tab[tabCur].cacheImageValid = true;
    }
    // Raster lists (background first, then foreground; rt64_view.cpp:1138-1147).  They are a handful of HUD instances: uploaded every frame.
    {
        const bool whole = separatePost();                     // the back buffer is screen size; device rows are screen rows
        const int sy0 = whole ? 0 : dev->tileY0, sy1 = whole ? finalH : dev->tileY1;
        prepareRasterList(rasterBg, rasterBgEnv, finalW, finalH, 0, finalH, false);               // gBackground: every rank needs all of it (env-map lookups)
        prepareRasterList((rtInstances.empty() || rtRect) ? rasterBg : std::vector<RenderInstance>(), rasterBgScreen, finalW, finalH, sy0, sy1, true);
        prepareRasterList(rasterFg, rasterFgScreen, finalW, finalH, sy0, sy1, true);
        flushPrologue();
        if (!rasterBg.empty() && (backgroundW != finalW || backgroundH != finalH)) { background.reserve((size_t)finalW * finalH * 4); backgroundW = finalW; backgroundH = finalH; rasterBgEnv.changed = true; }      // (a resize has dropped every kept frame: View::createImages)
        if (leanFrame && (rasterBgEnv.contentChanged || (!rasterBgEnv.ready && lastParams.background.texels))) { materialise(); dev->leanHoldoff = RT64_LEAN_HOLDOFF_FRAMES; }      // the kept frame's sky pixels read gBackground
        if (rasterBgEnv.ready && rasterBgEnv.changed) {        // gBackground: cleared to 0, drawn without scissors / viewports (rt64_view.cpp:1298-1319)
            dev->impure();
            const bool clearInDraw = dev->opt.framePrologue && rasterBgEnv.triTotal != 0;
            if (!clearInDraw) HIP_CHECK(hipMemsetAsync(background.ptr, 0, (size_t)finalW * finalH * 4, dev->stream));
            const int sr = dev->stripRank, sc = dev->stripCount; dev->stripRank = 0; dev->stripCount = 1;
            drawRasterList(rasterBgEnv, background.ptr, clearInDraw);
            dev->stripRank = sr; dev->stripCount = sc;
        }
    }
    for (Texture *t : usedTextures) t->currentIndex = -1;
}

void View::fillParams(FrameParams &P) {        // updateGlobalParamsBuffer, rt64_view.cpp:961-1028
    Device *dev = scene->device;
    memset(&P, 0, sizeof(P));
    const RT64_SCENE_DESC &d = scene->desc;
    auto set4 = [](float *dst, RT64_VECTOR3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = 0.0f; };
    set4(P.ambientBaseColor, d.ambientBaseColor); set4(P.ambientNoGIColor, d.ambientNoGIColor);
    set4(P.eyeLightDiffuseColor, d.eyeLightDiffuseColor); set4(P.eyeLightSpecularColor, d.eyeLightSpecularColor);
    set4(P.skyDiffuseMultiplier, d.skyDiffuseMultiplier); set4(P.skyHSLModifier, d.skyHSLModifier);
    P.skyYawOffset = d.skyYawOffset; P.giDiffuseStrength = d.giDiffuseStrength; P.giSkyStrength = d.giSkyStrength;

    projection = mat_perspective_fov_rh(fov, dev->aspect(), nearDist, farDist);      // setPerspective, :1766 (aspect of the window at draw time)
    // On the very first frame the reference's previous matrices are uninitialised memory; they are defined here as the current ones.
    const bool reproject = canReproject && matricesValid;
    if (reproject) { prevViewI = viewI; prevViewProj = viewProj; }
    viewI = mat_inverse(view); projectionI = mat_inverse(projection); viewProj = mat_mul(view, projection);
    if (!reproject) { prevViewI = viewI; prevViewProj = viewProj; }
    matricesValid = true;
    memcpy(P.view, view.m, 64); memcpy(P.viewI, viewI.m, 64); memcpy(P.prevViewI, prevViewI.m, 64); memcpy(P.projection, projection.m, 64);
    memcpy(P.projectionI, projectionI.m, 64); memcpy(P.viewProj, viewProj.m, 64); memcpy(P.prevViewProj, prevViewProj.m, 64);

    // Pinhole vectors, only consumed by the ray differentials (:992-1009; getViewDirection uses view-space +z, :1798).
    const float focal = (nearDist + farDist) / 2.0f, aspect = dev->aspect();
    V3 pos = mat_point(viewI, v3(0, 0, 0));
    V3 dir = mat_vector(viewI, v3(0, 0, 1)); { float l = vlen(dir); dir = v3(dir.x / l, dir.y / l, dir.z / l); }
    V3 target = pos + dir * focal;
    V3 W = vnorm(target - pos) * focal;
    V3 U = vnorm(vcross(W, v3(0, 1, 0)));
    V3 V = vnorm(vcross(U, W));
    const float ulen = focal * tanf(fov * 0.5f) * aspect, vlen_ = focal * tanf(fov * 0.5f);
    U = U * ulen; V = V * vlen_;
    P.cameraU[0] = U.x; P.cameraU[1] = U.y; P.cameraU[2] = U.z; P.cameraV[0] = V.x; P.cameraV[1] = V.y; P.cameraV[2] = V.z; P.cameraW[0] = W.x; P.cameraW[1] = W.y; P.cameraW[2] = W.z;
    for (int k = 0; k < 4; k++) { P.viewport[k] = rtViewport[k]; P.rtViewport[k] = rtViewport[k]; P.rtScissor[k] = rtScissor[k]; }      // gParams.viewport = rtViewport (rt64_view.cpp:1283-1286)
    P.resolution[0] = (float)imgW; P.resolution[1] = (float)imgH; P.resolution[2] = (float)dev->width; P.resolution[3] = (float)dev->height;
    // jitter only with an upscaler (:1273-1281): HaltonJitter(frameCount, phases), rt64_common.h:359-361
    pixelJitter[0] = pixelJitter[1] = 0.0f;
    if (upscaleActive) { const int fi = (int)(frameCount % (uint32_t)jitterPhases) + 1; pixelJitter[0] = halton_sequence(fi, 2) - 0.5f; pixelJitter[1] = halton_sequence(fi, 3) - 0.5f; }
    if (subFrames > 1) { pixelJitter[0] = halton_sequence(subFrame + 1, 2) - 0.5f; pixelJitter[1] = halton_sequence(subFrame + 1, 3) - 0.5f; }      // P2
    P.pixelJitter[0] = pixelJitter[0]; P.pixelJitter[1] = pixelJitter[1];
    P.motionBlurStrength = motionBlurStrength; P.motionBlurSamples = motionBlurSamples;
    P.skyPlaneTexIndex = skyPlane ? 0 : -1;
    P.skyTiled = nullptr; P.skyTiledLog2W = P.skyTiledLog2H = 0;
    // the tiled copy only pays for random lookups (bounce / reflection rays): frames without GI or reflections never make them
    if (skyPlane && (giSamples > 0 || anyReflection) && skyPlane->width >= 4 && skyPlane->height >= 4 &&
        (skyPlane->width & (skyPlane->width - 1)) == 0 && (skyPlane->height & (skyPlane->height - 1)) == 0) {
        if (skyTiledSerial != skyPlane->serial) {
            dev->impure();
            skyTiled.reserve((size_t)skyPlane->width * skyPlane->height);
            HIP_CHECK(tile_texture_launch(skyPlane->texels.ptr, skyTiled.ptr, (uint32_t)skyPlane->width, (uint32_t)skyPlane->height, dev->stream));
            skyTiledSerial = skyPlane->serial;
            skyTiledLog2[0] = (uint32_t)__builtin_ctz((unsigned)skyPlane->width); skyTiledLog2[1] = (uint32_t)__builtin_ctz((unsigned)skyPlane->height);
        }
        P.skyTiled = skyTiled.ptr; P.skyTiledLog2W = skyTiledLog2[0]; P.skyTiledLog2H = skyTiledLog2[1];
    }
    P.randomSeed = frameCount; P.frameCount = frameCount;
    P.diSamples = diSamples; P.giSamples = giSamples; P.maxLights = maxLights; P.giBounces = (uint32_t)dev->opt.giBounces;
    P.diReproject = 0;                                        // DI_REPROJECTION_SUPPORT undefined (:1012-1016)
    P.giReproject = (!skipReprojection && denoiserEnabled && giSamples > 0) ? 1u : 0u;
    P.binaryLockMask = upscaleActive ? 0u : 1u;               // rtUpscaleMode != FSR (:1018): the built-in stage stands where FSR does and takes the continuous mask
    P.visualizationMode = 0;
    P.width = imgW; P.height = imgH; P.tileY0 = dev->tileY0; P.tileY1 = dev->tileY1; P.stripRank = dev->stripRank; P.stripCount = dev->stripCount;
    P.cacheWords = cacheWords; P.cacheInstances = cacheWords ? (uint32_t)rtInstances.size() : 0u; P.cacheImage = cacheWords ? tab[tabCur].cacheImageAt : nullptr;
    P.separatePost = separatePost() ? 1u : 0u;
    P.simpleKernels = simpleFrame ? 1u : 0u;
    P.postSource = img.output; P.postSourceW = imgW; P.postSourceH = imgH;
    P.rasterFg = nullptr; P.rasterFgTris = nullptr; P.rasterFgCount = 0; P.rasterFgPad = 0; P.finalPacked = nullptr;
    memset(&P.background, 0, sizeof(P.background));
    if (rasterBgEnv.ready) { P.background.texels = background.ptr; P.background.width = (uint32_t)backgroundW; P.background.height = (uint32_t)backgroundH; P.background.mips = 1; P.background.pow2 = ((backgroundW & (backgroundW - 1)) == 0 && (backgroundH & (backgroundH - 1)) == 0) ? 1u : 0u; }
    if (P.separatePost) { P.tileY0 = 0; P.tileY1 = imgH; P.stripRank = 0; P.stripCount = 1; }       // device rows are screen rows; the render target has its own height
    P.maxDepthBias = maxDepthBias;
    {   // ComputeSkyPlaneUV (BgSky.hlsli:20-52): the view-only part, once per frame
        const float SCREEN_WIDTH = 320.0f, SCREEN_HEIGHT = 240.0f, SKYBOX_WIDTH = 4.0f * SCREEN_WIDTH, SKYBOX_HEIGHT = 4.0f * SCREEN_HEIGHT, PI = 3.14159265f, TWO_PI = PI * 2.0f;
        V3 vd = mat_vector(viewI, v3(0, 0, 1)); { float inv = 1.0f / vlen(vd); vd = v3(vd.x * inv, vd.y * inv, vd.z * inv); }
        auto hfmod = [](float x, float y) { return x - y * truncf(x / y); };
        float skyYawRadians = hfmod(d.skyYawOffset + atan2f(vd.x, -vd.z) + PI, TWO_PI);
        float baseX = SCREEN_WIDTH * 360.0f * (skyYawRadians - PI) / (90.0f * PI * 2.0f);
        float skyPitchRadians = atan2f(-vd.y, sqrtf(vd.x * vd.x + vd.z * vd.z));
        float pitchInDegrees = skyPitchRadians * 360.0f / (PI * 2.0f);
        float degreesToScale = 360.0f * pitchInDegrees / 90.0f;
        float baseY = degreesToScale + 5.0f * (SCREEN_HEIGHT / 2.0f);
        baseY = std::min(std::max(baseY, SCREEN_HEIGHT), SKYBOX_HEIGHT);
        float aspectRatio = P.viewport[2] / P.viewport[3];
        baseX += SCREEN_WIDTH / 2.0f;
        baseX -= (SCREEN_HEIGHT * aspectRatio) / 2.0f;
        baseX /= SKYBOX_WIDTH;
        baseY = (SKYBOX_HEIGHT - baseY) / SKYBOX_HEIGHT;
        float ratioDivision = aspectRatio / (4.0f / 3.0f);
        P.skyBase[0] = baseX; P.skyBase[1] = baseY; P.skyBase[2] = 0.25f * ratioDivision; P.skyBase[3] = 0.25f;
    }
    P.lightCount = (uint32_t)scene->lights.size(); P.instanceCount = (uint32_t)rtInstances.size();
    P.countTraversal = dev->opt.countTraversal ? 1u : 0u;
    { TableSlot &T = tab[tabCur]; P.instances = T.dInstances.ptr; P.tlasNodes = T.tlasNodesAt; P.tlasIndex = T.tlasIndexAt; P.textures = T.dTextures.ptr; P.lights = T.dLights.ptr; T.readers |= 1u << dev->cur; }
    {   // grows with the render size (no-op otherwise); never allocated for shallow scenes; one slab per render stream
        DevArray<uint32_t> &slab = dev->spillStack[dev->cur];
        const size_t words = needSpillSlab ? rt_stack_spill_bytes(imgW, imgH) / sizeof(uint32_t) : 0;
        if (words > slab.count) {
            dev->impure(); slab.reserve(words);
            // header in front of every lane's entries: where an overflowing walk reports (trace.h)
            void *flagDev = nullptr;
            HIP_CHECK(hipHostGetDevicePointer(&flagDev, dev->traversalOverflow, 0));
            HIP_CHECK(launch_stack_slab_init(slab.ptr, words / (RT_STACK_SPILL_HEADER + RT_STACK_SPILL), static_cast<const uint32_t *>(flagDev), dev->stream));
        }
        P.traversalStack = slab.ptr;
    }
    P.blueNoise = dev->blueNoise.ptr; P.counters = dev->counters.ptr;
    P.tileTiming = nullptr;
    if (dev->opt.tileTiming) { dev->tileTiming.reserve((size_t)RT_TIMING_WAVES * 3); P.tileTiming = dev->tileTiming.ptr; }      // two records per wave + one more in diagnostic builds
}

static void halo_exchange(Device *dev, const ViewImages &img, int W, int H, hipStream_t s);     // (after Gather, below)

void View::render() {                          // View::render, rt64_view.cpp:1180-1670
    Device *dev = scene->device;
    hipStream_t s = dev->stream;
    const bool prof = dev->profNow;
    const int slot = dev->cur;                   // per-stream storage of this frame: back buffer, tile order (Device::streams)
    img.final = dev->finalOverride ? dev->finalOverride : finalBuf[slot];
    static const bool hostTiming = getenv("RT64_HOST_TIMING") != nullptr;
    auto hostPrev = std::chrono::steady_clock::now();
    auto mark = [&](int ev) {
        if (hostTiming) { auto now = std::chrono::steady_clock::now(); dev->hostStageUs[ev] += std::chrono::duration<double, std::micro>(now - hostPrev).count(); hostPrev = now; }
        if (prof) {
            if (dev->workSinceMark) { HIP_CHECK(hipEventRecord(dev->events[ev], s)); dev->eventAlias[ev] = ev; }
            else dev->eventAlias[ev] = dev->eventAlias[dev->lastMark];
            dev->lastMark = ev; dev->workSinceMark = false;
        }
        if (hostTiming) { auto now = std::chrono::steady_clock::now(); dev->hostEventUs += std::chrono::duration<double, std::micro>(now - hostPrev).count(); hostPrev = now; }
    };
    auto L = [&](hipError_t e) { HIP_CHECK(e); dev->workSinceMark = true; };       // a launch between two marks
    if (!perspectiveSet) throw std::runtime_error("RT64_DrawDevice: RT64_SetViewPerspective was never called (fov must be > 0).");
    // Lean frame: nothing downstream reads the view direction, the reflection / refraction / transparent accumulators, motion
    // vectors, upscaler masks, history guides or a GI buffer.  A full frame after lean ones reads the previous frame's guides and
    // history (temporal reprojection), so what that lean frame skipped is produced first, while its hit records still exist.
    const bool leanNow = !rtInstances.empty() && dev->opt.leanFrames && !upscaleActive && !anyNonOpaque && !anyReflection && !anyRefraction && !anyFog && giSamples == 0 && motionBlurStrength <= 0.0f && dev->leanHoldoff == 0 && subFrames == 1;
    if (leanFrame && !leanNow) materialise();
    if (subFrames > 1 && (upscaleActive || rtRect || separatePost())) throw std::runtime_error("RT64_DrawDevice: primary_spp > 1 is not combined with an upscaler, a resolution scale, motion blur or a viewport rectangle.");
    FrameParams P;
    fillParams(P);
    const int cur = rtSwap ? 1 : 0;
    const size_t n = (size_t)imgW * imgH;
    mark(Device::EV_BUILD);
    bool fgFolded = false;
    if (!rtInstances.empty()) {
        if (anyNonOpaque && !img.klistA) {          // sorted per-pixel hit lists, only while some instance is not provably opaque
            const size_t slots = (size_t)(RT64_MAX_HIT_QUERIES + 1) * n;
            void *a = nullptr, *b = nullptr, *c = nullptr;
            HIP_CHECK(hipMalloc(&a, slots * sizeof(uint4))); allocations.push_back(a);
            HIP_CHECK(hipMalloc(&b, slots * sizeof(uint2))); allocations.push_back(b);
            HIP_CHECK(hipMalloc(&c, n * sizeof(uint32_t))); allocations.push_back(c);
            img.klistA = static_cast<uint4 *>(a); img.klistB = static_cast<uint2 *>(b); img.klistCount = static_cast<uint32_t *>(c);
        }
        if (!anyNonOpaque && giSamples > 0 && (!img.bounceRecords || bounceSamples < giSamples)) {     // records of the bounce_trace / bounce_shade pair
            void *a = nullptr;
            HIP_CHECK(hipMalloc(&a, (size_t)giSamples * n * 2 * sizeof(uint4))); allocations.push_back(a);
            img.bounceRecords = static_cast<uint4 *>(a); bounceSamples = giSamples;
            void *l = nullptr, *c = nullptr, *r = nullptr;
            // per-workgroup list segments: ceil(tiles / grid) tiles of 256 pixels each -> at most n + grid * 256 entries per list and sample
            const size_t tilesAll = (size_t)((imgW + 15) / 16) * (size_t)((imgH + 15) / 16);
            HIP_CHECK(hipMalloc(&l, (size_t)giSamples * (tilesAll + std::min<size_t>(tilesAll, RT_MAX_BOUNCE_GROUPS)) * 256 * 2 * sizeof(uint32_t))); allocations.push_back(l);
            HIP_CHECK(hipMalloc(&c, (size_t)RT_MAX_BOUNCE_GROUPS * 2 * sizeof(uint32_t))); allocations.push_back(c);
            HIP_CHECK(hipMalloc(&r, (size_t)giSamples * n * sizeof(BounceRadiance))); allocations.push_back(r);
            img.bounceLists = static_cast<uint32_t *>(l); img.bounceCounts = static_cast<uint32_t *>(c); img.bounceResults = static_cast<BounceRadiance *>(r);
        }
        const bool klist = anyNonOpaque;
        reflStatePending = false;
        if (anyReflection && dev->opt.maxReflections > 0) {
            if (!img.reflState0) {          // continuation state of the reflection passes (ViewImages::reflState0 / 1 / reflTag), 36 B per pixel, from the first frame with a reflective material on
                void *a = nullptr, *b = nullptr, *c = nullptr;
                HIP_CHECK(hipMalloc(&a, n * sizeof(uint4))); allocations.push_back(a);
                HIP_CHECK(hipMalloc(&b, n * sizeof(uint4))); allocations.push_back(b);
                HIP_CHECK(hipMalloc(&c, n * sizeof(uint32_t))); allocations.push_back(c);
                HIP_CHECK(hipMemsetAsync(c, 0, n * sizeof(uint32_t), s));
                img.reflState0 = static_cast<uint4 *>(a); img.reflState1 = static_cast<uint4 *>(b); img.reflTag = static_cast<uint32_t *>(c);
            }
            reflStatePending = true; reflStateTag = P.frameCount + 1u; reflStateY0 = P.tileY0; reflStateY1 = P.tileY1;
        }
        // Image-tile partition with a spatial filter downstream: the GI denoiser reads a neighbourhood of every row this device
        // owns (SVGF: 66 rows, the reference's five 3x3 Gaussians: 5), so the passes that feed it -- primary visibility, G-buffer,
        // GI bounce -- also cover a halo above and below the device's rows.  Pixel-local passes (direct light, reflection,
        // refraction, compose) stay on the owned rows.  By default the halo rows are recomputed -- no mid-frame collective; with a halo exchange set up (below) they arrive from the neighbouring bands.
        const bool denoiseGI = denoiserEnabled && giSamples > 0;
        bool guideByResolve = false, inputByResolve = false;
        FrameParams X = P;                                        // X: owned rows + halo
        bool haloExchange = false;
        if (denoiseGI && !P.separatePost && (P.tileY0 > 0 || P.tileY1 < imgH || P.stripCount > 1)) {
            if (P.stripCount > 1) throw std::runtime_error("RT64_DrawDevice: a frame with GI + denoiser filters across rows; partition it into contiguous bands (RT64_SetDeviceTile), not interleaved strips.");
            // ... or, with a halo exchange (SVGF only), the passes cover the band and the few rows the filter INPUT of the band needs (and the temporal
            // history margin); the filter input of the 62 halo rows arrives from the neighbouring bands in the middle of the frame (halo_exchange below).
            haloExchange = dev->opt.denoiserMode == 1 && dev->haloActive();
            const int halo = haloExchange ? std::max(dev->opt.haloMargin, SVGF_INPUT_HALO_ROWS) : (dev->opt.denoiserMode == 1 ? SVGF_HALO_ROWS : GAUSSIAN_HALO_ROWS);
            X.tileY0 = std::max(0, P.tileY0 - halo); X.tileY1 = std::min(imgH, P.tileY1 + halo);
        }
        const bool lean = leanNow;
        const bool perWave = dev->opt.perWaveFrame != 0;          // (only frames without the LDS scene cache take that form: launch_lean_frame)
        // A lean frame is pixel-local end to end: one kernel carries every pixel from the primary ray to the back buffer
        // (device option fused_lean = 0 keeps the three separate kernels; same back buffer bit for bit).
        const bool fused = lean && dev->opt.fusedLean;
        // The same one-kernel form serves full frames whose instances are all provably opaque (GI / denoiser / reflection frames):
        // it then writes the whole G-buffer over the rows with the denoiser halo (X) and DirectRayGen's images over the owned rows.
        const bool fusedFull = !lean && !klist && dev->opt.fusedLean;
        leanFrame = lean; fusedFrame = fused; fusedFullFrame = fusedFull; lastParams = P; lastCur = cur;
        // Only the one-kernel lean frame that stores nothing but its slot's back buffer may run beside its neighbours (Device::streams); every other kind of
        // frame reads or writes images that are not per slot and runs behind everything enqueued before it.
        if (!(fused && !P.separatePost && !rtRect && !dev->opt.leanRecords)) dev->impure();
        // The foreground (HUD) list is pixel-local too: the one-kernel frame blends it over each pixel before the store, no launch of its own.
        if (fused && !P.separatePost && !rtRect && rasterFgScreen.ready && rasterFgScreen.triTotal > 0 && dev->opt.foldForeground) {
            P.rasterFg = rasterFgScreen.table.ptr; P.rasterFgTris = rasterFgScreen.tris.ptr; P.rasterFgCount = rasterFgScreen.triTotal;
            fgFolded = true;
        }
        // ... and so is the copy into a gather's send buffer: when the frame's last writer of the back buffer is this kernel, it stores
        // the packed copy itself (RT64_SetDeviceGatherTarget).
        packedFinal = false;
        if (fused && !P.separatePost && !rtRect && (fgFolded || !(rasterFgScreen.ready && rasterFgScreen.triTotal > 0)) && dev->gatherTarget &&
            dev->gatherTargetBytes >= (size_t)dev->ownedRows() * (size_t)imgW * 4) {
            P.finalPacked = static_cast<uint32_t *>(dev->gatherTarget);
            packedFinal = true;
        }
        // Scenes that walk from HBM run the one-kernel frame as one-wave workgroups whose lengths differ by two orders of magnitude (a shadow ray that grazes a dense mesh
        // walks hundreds of dependent fetches): the launch is as long as its longest wave plus the time that wave waited to start.  Each tile's cost (the most visits any
        // of its lanes made) is recorded by the frame and the next frame starts its tiles most expensive first (tile_order_kernel; the scene changes little between frames).
        auto orderTiles = [&](FrameParams &F) -> unsigned {
            if (!(dev->opt.tileOrder && perWave && !F.cacheWords)) return 0u;
            const unsigned tiles = lean_frame_tiles(F);
            if (tiles == 0u) return 0u;
            if (tileOrderTiles != tiles) {
                dev->impure();
                for (int k = 0; k < dev->streamCount; k++) {
                    tileCost[k].reserve(tiles); tileOrder[k].reserve(tiles);
                    HIP_CHECK(hipMemsetAsync(tileCost[k].ptr, 0, (size_t)tiles * 4, s));
                    tileOrderValid[k] = false;
                }
                tileOrderTiles = tiles;
            }
            F.tileCost = tileCost[slot].ptr; F.tileOrder = tileOrderValid[slot] ? tileOrder[slot].ptr : nullptr;
            return tiles;
        };
        const bool reflectBeside = anyReflection && dev->opt.maxReflections > 0 && denoiseGI && dev->opt.denoiserMode == 1 && dev->opt.overlapReflection;
        auto reflectOnAux = [&]() {
            if (!dev->auxStream) {
                HIP_CHECK(hipStreamCreateWithFlags(&dev->auxStream, hipStreamNonBlocking));
                HIP_CHECK(hipEventCreateWithFlags(&dev->forkEvent, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&dev->joinEvent, hipEventDisableTiming));
            }
            HIP_CHECK(hipEventRecord(dev->forkEvent, s));
            HIP_CHECK(hipStreamWaitEvent(dev->auxStream, dev->forkEvent, 0));
            if (prof) { HIP_CHECK(hipEventRecord(dev->auxEvents[dev->evSet][0], dev->auxStream)); dev->auxTimed[dev->evSet] = true; }
            for (int r = 0; r < dev->opt.maxReflections; r++) L(launch_reflection(P, img, klist, r, r == dev->opt.maxReflections - 1, cur, dev->auxStream));
            if (prof) HIP_CHECK(hipEventRecord(dev->auxEvents[dev->evSet][1], dev->auxStream));
            HIP_CHECK(hipEventRecord(dev->joinEvent, dev->auxStream));
        };
        // ... and from the moment the G-buffer exists when nothing else of the frame shares storage with them: their continuation state is their own (ViewImages::reflState0 / 1),
        // but the per-pixel hit lists of a k-buffer frame and the HBM half of the traversal stacks are per launch position, one set per frame -- such frames keep the later fork.
        const bool reflectEarly = reflectBeside && !klist && !needSpillSlab && dev->opt.reflectionEarly;
        if (fused) {
            // nullptr: the frame stores its back buffer only (hit records and the direct-light image come back through materialise); option lean_records = 1 keeps them
            if (P.tileTiming) { HIP_CHECK(hipMemsetAsync(dev->tileTiming.ptr, 0, dev->tileTiming.bytes(), s)); }
            const unsigned ordered = orderTiles(P);
            L(launch_lean_frame(P, img, dev->opt.leanRecords ? hitInstance.ptr : nullptr, cur, false, 0, imgH, dev->opt.maxFrameGroups, perWave, s));
            if (ordered) { L(launch_tile_order(tileCost[slot].ptr, tileOrder[slot].ptr, ordered, s)); tileOrderValid[slot] = true; }
            fusedStoreless = !dev->opt.leanRecords;
            mark(Device::EV_PRIMARY_TRACE); mark(Device::EV_PRIMARY); mark(Device::EV_DIRECT);
        }
        else if (fusedFull) {
            if (P.stripCount > 1 && (X.tileY0 != P.tileY0 || X.tileY1 != P.tileY1)) throw std::runtime_error("RT64_DrawDevice: interleaved strips with a denoiser halo.");
            if (X.tileTiming) { HIP_CHECK(hipMemsetAsync(dev->tileTiming.ptr, 0, dev->tileTiming.bytes(), s)); }
            const unsigned ordered = orderTiles(X);
            L(launch_lean_frame(X, img, hitInstance.ptr, cur, true, P.tileY0, P.tileY1, dev->opt.maxFrameGroups, perWave, s));
            if (ordered) { L(launch_tile_order(tileCost[slot].ptr, tileOrder[slot].ptr, ordered, s)); tileOrderValid[slot] = true; }
            mark(Device::EV_PRIMARY_TRACE); mark(Device::EV_PRIMARY); mark(Device::EV_DIRECT);
            if (reflectEarly) reflectOnAux();              // beside the GI chain and the denoiser: joined before the composing iteration
        }
        else {
            L(launch_primary_trace(X, img, hitInstance.ptr, klist, s));
            mark(Device::EV_PRIMARY_TRACE);
            L(launch_primary_shade(X, img, hitInstance.ptr, cur, anyNonOpaque, lean, s));
            mark(Device::EV_PRIMARY);
            // DirectRayGen also writes the "filtered" copy: DI denoising is compiled out in the reference (rt64_view.cpp:1438-1463),
            // so rtFilteredDirectLight[1] is always a plain copy of the raw accumulation.
            L(launch_direct(P, img, cur, lean, s));
            mark(Device::EV_DIRECT);
            if (reflectEarly) reflectOnAux();
        }
        if (lean) {}                                                                  // constant ambient folded into Compose
        else if (giSamples == 0) L(launch_indirect_constant(P, img, cur, s));  // IndirectRayGen.hlsl:135: constant ambient
        else {
            bool refill = dev->opt.bounceRefill == 1;
            if (dev->opt.bounceRefill < 0) { size_t tri = 0; for (auto &ri : rtInstances) tri += ri.instance->mesh->blasCount; refill = tri >= 65536; }
            const bool split = dev->opt.bounceSplit == 1 || (dev->opt.bounceSplit < 0 && X.cacheWords != 0 && giSamples >= 2);
            const int walk = refill ? BOUNCE_WALK_REFILL : (split ? BOUNCE_WALK_SPLIT : BOUNCE_WALK_PLAIN);
            // grid: the two-phase walk takes one workgroup per tile (0); the plain walk of one-sample frames one resident round of persistent workgroups (4 per CU at its
            // 118 VGPRs): a tile of 256 rays is too little work to pay for a scene-cache fill of its own (C3 bounce kernels 0.235 -> 0.198 ms against 0.220 per tile)
            const unsigned groups = dev->opt.bounceGroups >= 0 ? (unsigned)dev->opt.bounceGroups : (walk == BOUNCE_WALK_PLAIN ? 1024u : 0u);
            // frames whose GI runs as the wavefront chain: its last kernel (the per-pixel resolve) writes the SVGF guide records of its rows as well
            guideByResolve = denoiseGI && dev->opt.denoiserMode == 1 && !klist && img.bounceRecords != nullptr && dev->opt.foldGuide;
            L(launch_indirect(X, img, cur, !denoiseGI, klist, walk, groups, guideByResolve ? (dev->opt.foldVariance ? 3 : 1) : 0, s));
            inputByResolve = guideByResolve && dev->opt.foldVariance;
        }
        mark(Device::EV_INDIRECT);
        // Refraction / reflection touch only pixels whose primary hit has a refraction / reflection factor (alpha > EPSILON).
        if (anyRefraction) L(launch_refraction(P, img, klist, s));
        // The reflection passes read and rewrite the G-buffer (position, view direction, normal, instance id) and their own image; the a-trous iterations
        // read and write the filter's ping-pong images and read the guide records: no image in common.  On frames that have both, the reflection launches
        // therefore go to a second stream once the filter's INPUT is made (the variance kernel reads the instance ids the reflection pass rewrites), beside the
        // five iterations, and the frame joins the two before Compose -- two latency-bound launches fill each other's idle issue slots (C5: DESIGN 8).
        // Frames with the SVGF denoiser: ComposePS runs inside the last a-trous iteration (svgf.hip), on the filtered value that iteration has just rounded
        if (subFrames > 1) sppSum.reserve(n * 4);
        SvgfComposeFold foldArgs = { img.diffuse, img.filteredDirect[1], img.reflection, img.refraction, img.transparent, img.output, img.final, P.tileY0, P.tileY1, (!P.separatePost && subFrames == 1) ? 1 : 0,
                                     subFrames > 1 ? sppSum.ptr : nullptr, subFrame, subFrames };
        const SvgfComposeFold *composeFold = (denoiseGI && dev->opt.denoiserMode == 1 && !lean && dev->opt.foldCompose && P.stripCount == 1) ? &foldArgs : nullptr;
        if (anyReflection && !reflectBeside) for (int r = 0; r < dev->opt.maxReflections; r++) L(launch_reflection(P, img, klist, r, r == dev->opt.maxReflections - 1, cur, s));
        mark(Device::EV_REFL);
        if (denoiseGI && dev->opt.denoiserMode == 1 && reflectBeside) {
            const int ay0 = haloExchange ? std::max(0, P.tileY0 - SVGF_ATROUS_HALO_ROWS) : X.tileY0, ay1 = haloExchange ? std::min(imgH, P.tileY1 + SVGF_ATROUS_HALO_ROWS) : X.tileY1;
            if (haloExchange) { L(launch_svgf_inputs(img, cur, imgW, imgH, std::max(0, P.tileY0 - 3), guideByResolve ? 0 : std::min(imgH, P.tileY1 + 3), P.tileY0, P.tileY1, inputByResolve, s)); }
            else L(launch_svgf_inputs(img, cur, imgW, imgH, X.tileY0, guideByResolve ? 0 : X.tileY1, X.tileY0, X.tileY1, inputByResolve, s));
            if (!reflectEarly) reflectOnAux();                   // (before the exchange: the reflection pass also runs beside the wait for the neighbours' rows)
            if (haloExchange) halo_exchange(dev, img, imgW, imgH, s);
            // (the folded Compose reads the reflection image: the join comes before the last iteration instead of behind it)
            L(launch_svgf_atrous(img, imgW, imgH, ay0, ay1, P.tileY0, P.tileY1, 0, composeFold ? 4 : 5, nullptr, s));
            HIP_CHECK(hipStreamWaitEvent(s, dev->joinEvent, 0));
            if (composeFold) L(launch_svgf_atrous(img, imgW, imgH, ay0, ay1, P.tileY0, P.tileY1, 4, 5, composeFold, s));
        }
        else if (denoiseGI && dev->opt.denoiserMode == 1 && haloExchange) {
            // filter input (variance image + guide records) of the band's own rows; the guide records of 3 rows around them feed the variance estimate
            L(launch_svgf_inputs(img, cur, imgW, imgH, std::max(0, P.tileY0 - 3), guideByResolve ? 0 : std::min(imgH, P.tileY1 + 3), P.tileY0, P.tileY1, inputByResolve, s));
            halo_exchange(dev, img, imgW, imgH, s);
            L(launch_svgf_atrous(img, imgW, imgH, std::max(0, P.tileY0 - SVGF_ATROUS_HALO_ROWS), std::min(imgH, P.tileY1 + SVGF_ATROUS_HALO_ROWS), P.tileY0, P.tileY1, 0, 5, composeFold, s));
        }
        else if (denoiseGI && dev->opt.denoiserMode == 1) { L(launch_svgf_inputs(img, cur, imgW, imgH, X.tileY0, guideByResolve ? 0 : X.tileY1, X.tileY0, X.tileY1, inputByResolve, s)); L(launch_svgf_atrous(img, imgW, imgH, X.tileY0, X.tileY1, P.tileY0, P.tileY1, 0, 5, composeFold, s)); }
        else if (denoiseGI) {
            L(hipMemcpyAsync(img.filteredIndirect[0], img.indirectLight[cur], n * 8, hipMemcpyDeviceToDevice, s));
            for (int k = 0; k < 5; k++)            // rt64_view.cpp:1512-1530
                L(launch_gaussian(img.filteredIndirect[k % 2], img.filteredIndirect[(k % 2) ^ 1], imgW, imgH, X.tileY0, X.tileY1, s));
        }
        mark(Device::EV_DENOISE);
        if (!lean && !composeFold) L(launch_compose_post(P, img, cur, false, subFrames == 1, s));       // a lean frame is composed by direct_kernel<false> itself
        if (subFrames > 1 && !composeFold) {        // P3: rtOutput of the sub-frames summed in order; the last one turns the sum into the mean and PostProcessPS of it into the back buffer
            sppSum.reserve(n * 4);                   // (frames with the SVGF denoiser: the composing a-trous iteration does this too)
            L(launch_spp_accumulate(P, img, sppSum.ptr, subFrame, subFrames, s));
        }
        if (rtRect) {            // the ray-traced picture covers only its rectangle: cleared buffer + background instances show around it (rt64_view.cpp:1292-1296)
            L(launch_clear_final(P, img, s));
            drawRasterList(rasterBgScreen, img.final);
        }
        if (upscaleActive) {        // Upscaler::upscale (rt64_view.cpp:1584-1618): rtOutput -> rtOutputUpscaled; PostProcessPS reads the latter (:800-801)
            if (skipReprojection) upValid = false;          // buffers were (re)created: the accumulation starts over
            L(launch_taa_upsample(img, cur, imgW, imgH, pixelJitter[0], pixelJitter[1], upscaled[upSwap ^ 1], upscaled[upSwap], finalW, finalH, upValid, s));
            P.postSource = upscaled[upSwap]; P.postSourceW = finalW; P.postSourceH = finalH;
            upValid = true; upSwap ^= 1;
        }
        if (P.separatePost) L(launch_post_process(P, img, s));
    }
    else {
        dev->impure();
        leanFrame = false; fusedFrame = false; fusedFullFrame = false; packedFinal = false;
        mark(Device::EV_PRIMARY_TRACE); mark(Device::EV_PRIMARY); mark(Device::EV_DIRECT); mark(Device::EV_INDIRECT); mark(Device::EV_REFL); mark(Device::EV_DENOISE);
        L(launch_clear_final(P, img, s));
        drawRasterList(rasterBgScreen, img.final);           // nothing ray traced covers the background instances (rt64_view.cpp:1292-1296)
    }
    if (!fgFolded && subFrame == subFrames - 1) drawRasterList(rasterFgScreen, img.final);   // foreground instances over the finished frame (rt64_view.cpp:1657-1661)
    // End of frame (rt64_view.cpp:1663-1667)
    rtSwap = !rtSwap; skipReprojection = false; frameCount++;
}

// The G-buffer as the reference leaves it after its reflection passes (ReflectionRayGen.hlsl:117-124 rewrites position / view direction / normal / instance id of
// every mirrored pixel): the passes here keep that state beside the G-buffer, a reader gets it folded back.
void View::applyReflectionState() {
    if (!reflStatePending || !img.reflState0) return;
    Device *dev = scene->device;
    dev->use(); dev->joinStreams();
    HIP_CHECK(launch_apply_reflection_state(img, imgW, reflStateY0, reflStateY1, reflStateTag, dev->stream));
    reflStatePending = false;
}

// A lean frame left some images untouched; produce them now from the frame's retained inputs (hit records, parameters).
void View::materialise() {
    if (!leanFrame) return;
    Device *dev = scene->device;
    dev->use();
    dev->joinStreams();
    if (fusedFrame && fusedStoreless) {
        // the frame kept no records: the FULL variant of the frame kernel traces the same rays again and writes the whole G-buffer and DirectRayGen's two images
        FrameParams Q = lastParams;
        Q.rasterFg = nullptr; Q.rasterFgTris = nullptr; Q.rasterFgCount = 0; Q.finalPacked = nullptr;
        Q.countTraversal = 0;        // the frame's rays were counted when the frame ran
        Q.tileTiming = nullptr;
        HIP_CHECK(launch_lean_frame(Q, img, hitInstance.ptr, lastCur, true, Q.tileY0, Q.tileY1, dev->opt.maxFrameGroups, dev->opt.perWaveFrame != 0, dev->stream));
        HIP_CHECK(launch_indirect_constant(lastParams, img, lastCur, dev->stream));
    }
    else {
        HIP_CHECK(launch_primary_shade(lastParams, img, hitInstance.ptr, lastCur, false, false, dev->stream));
        HIP_CHECK(launch_indirect_constant(lastParams, img, lastCur, dev->stream));
        const size_t rowBytes = (size_t)imgW * 8, off = (size_t)lastParams.tileY0 * rowBytes, bytes = (size_t)(lastParams.tileY1 - lastParams.tileY0) * rowBytes;
        HIP_CHECK(hipMemcpyAsync(reinterpret_cast<uint8_t *>(img.filteredDirect[1]) + off, reinterpret_cast<uint8_t *>(img.directLight[lastCur]) + off, bytes, hipMemcpyDeviceToDevice, dev->stream));
    }
    // the one-kernel lean frame kept the composed colour in registers: rtOutput from the images just rebuilt (the back buffer already has the foreground drawn over it)
    if (fusedFrame && !lastParams.separatePost) HIP_CHECK(launch_compose_post(lastParams, img, lastCur, true, false, dev->stream));
    HIP_CHECK(hipStreamSynchronize(dev->stream));
    leanFrame = false; fusedFrame = false;
}

void Device::draw(int, float) {                // Device::draw, rt64_device.cpp:1027-1083
    use();
    auto t0 = std::chrono::steady_clock::now();
    if (pendingWidth != width || pendingHeight != height) {      // updateSize (:199-231)
        width = pendingWidth; height = pendingHeight;
        if (!tileSet) { tileY0 = 0; tileY1 = height; }
    }
    if (tileY1 > height) tileY1 = height;
    if (tileY0 >= tileY1) { tileY0 = 0; tileY1 = height; }
    // Render streams (see Device::streams): an enqueued frame that follows a pure frame starts on the next stream, beside that frame's tail; it is joined
    // behind it the moment it turns out not to be pure itself (impure(): an upload, a build, a frame with history).  Frame counters, the tile-timing records and
    // the sub-frame accumulation are one per device: frames that use them stay in order.
    const bool mayOverlap = opt.overlapFrames && streamCount > 1 && !opt.syncPresent && !opt.countTraversal && !opt.tileTiming && opt.primarySpp <= 1;
    const bool flipped = mayOverlap && lastFramePure;
    const int curBefore = cur;
    if (flipped) switchStream();
    framePure = mayOverlap; lastFramePure = false;       // (a frame that ends in an exception leaves "not pure" behind)
    if (frameWait) { HIP_CHECK(hipStreamWaitEvent(stream, frameWait, 0)); frameWait = nullptr; }
    if (!mayOverlap) joinStreams();
    if (opt.countTraversal) HIP_CHECK(hipMemsetAsync(counters.ptr, 0, counters.bytes(), stream));
    // pass events on every profile_every-th frame only: each event is a barrier packet (~5 us of stream time; six of them are 5 % of a 0.6 ms GI frame)
    profNow = opt.profilePasses && (opt.profileEvery <= 1 || profCounter++ % (unsigned)opt.profileEvery == 0);
    if (profNow) beginEventSet(0);
    auto tu0 = std::chrono::steady_clock::now(), tu1 = tu0;
    try {
        flushMeshBuilds();
        for (Scene *sc : scenes) for (View *v : sc->views) v->update();
        tu1 = std::chrono::steady_clock::now();
        for (Scene *sc : scenes) for (View *v : sc->views) {
            // extension primary_spp = N: N complete sub-frames -- every pass up to Compose, history and frame count advancing after each -- with jittered primary rays;
            // the mean of their composed outputs is the frame (rules P1-P4 at oracle_render, oracle/oracle_render.c)
            v->subFrames = std::max(1, opt.primarySpp); v->subFrame = 0;
            v->render();
            for (int sub = 1; sub < v->subFrames; sub++) {
                if (profNow) { endEventSet(); beginEventSet(sub); }
                v->subFrame = sub; v->update(); v->render();
            }
            evSubFrames = v->subFrames; evSetsUsed = std::min(v->subFrames, (int)EV_SETS);
            v->subFrame = 0;
        }
    }
    catch (...) {
        // A frame the library refuses (or that fails half-way): the device keeps showing the last complete frame -- its stream and its back-buffer slot are current
        // again -- and whatever the failed frame has enqueued stays in order in front of everything that follows.
        if (cur != curBefore) { streamBusy[cur] = true; cur = curBefore; stream = streams[cur]; }
        try { joinStreams(); noteOrderedWork(); } catch (...) {}
        lastFramePure = false;
        for (Scene *sc : scenes) for (View *v : sc->views) v->img.final = finalOverride ? finalOverride : v->finalBuf[cur];
        throw;
    }
    if (leanHoldoff) leanHoldoff--;
    lastFramePure = framePure;
    auto tu2 = std::chrono::steady_clock::now();
    hostUpdateUs += std::chrono::duration<double, std::micro>(tu1 - tu0).count(); hostRenderUs += std::chrono::duration<double, std::micro>(tu2 - tu1).count(); hostFrames++;
    if (profNow) endEventSet();
    // postRender: Present + waitForGPU (:1006-1025).  Option sync_present = 0 turns RT64_DrawDevice into "enqueue the frame": the
    // host returns at once and orders its own work behind the frame on RT64_GetDeviceStream (pipelined multi-GPU gather in bench.py).
    if (opt.syncPresent) {
        if (opt.spinPresent) {       // poll instead of sleeping on the completion signal: the render thread is back ~10 us sooner
            hipError_t q;
            while ((q = hipStreamQuery(stream)) == hipErrorNotReady) { __builtin_ia32_pause(); }
            HIP_CHECK(q);
        }
        else HIP_CHECK(hipStreamSynchronize(stream));
    }
    auto t1 = std::chrono::steady_clock::now();
    const unsigned dropped = traversalOverflow ? *traversalOverflow : 0u;       // (enqueued frames: whatever earlier frames have reported by now; RT64_GetDeviceStats waits for the last one)

    RT64_FRAME_STATS st = {};
    st.structSize = sizeof(st); st.width = (unsigned)width; st.height = (unsigned)height; st.tileY0 = (unsigned)tileY0; st.tileY1 = (unsigned)tileY1;
    st.screenWidth = (unsigned)width; st.screenHeight = (unsigned)height;
    st.msHostWall = std::chrono::duration<float, std::milli>(t1 - t0).count();
    st.stripRank = (unsigned)stripRank; st.stripCount = (unsigned)stripCount; st.rowsRendered = (unsigned)ownedRows();
    drawn.tileY0 = tileY0; drawn.tileY1 = tileY1; drawn.stripRank = stripRank; drawn.stripCount = stripCount; drawn.valid = true;
    st.overlappedFrame = (flipped && framePure) ? 1u : 0u;
    bool haveView = false;
    for (Scene *sc : scenes) for (View *v : sc->views) {
        if (haveView) continue;
        haveView = true;
        st.instanceCount = (unsigned)v->rtInstances.size();
        st.packedFinal = v->packedFinal ? 1u : 0u; st.leanFrame = v->leanFrame ? 1u : 0u; st.fusedFrame = v->fusedFrame ? 1u : (v->fusedFullFrame ? 2u : 0u);
        st.width = (unsigned)v->imgW; st.height = (unsigned)v->imgH;          // render size ("Render buffer: WxH")
        unsigned tri = 0, nodeBytes = 0, triBytes = 0;
        for (auto &ri : v->rtInstances) { tri += ri.instance->mesh->blasCount; nodeBytes += (unsigned)(std::max<uint32_t>(ri.instance->mesh->blasCount - 1, 1) * sizeof(GpuNode)); triBytes += (unsigned)(ri.instance->mesh->blasCount * sizeof(GpuTri)); }
        st.triangleCount = tri; st.blasNodeBytes = nodeBytes; st.blasTriangleBytes = triBytes;
        st.tlasNodeBytes = (unsigned)(std::max<size_t>(v->rtInstances.size() ? v->rtInstances.size() - 1 : 0, 1) * sizeof(GpuNode));
    }
    st.traversalOverflow = dropped;
    stats = st; statsHaveView = haveView; statsPending = true; statsProfiled = profNow;
    if (opt.syncPresent) finishStats();
    checkTraversalOverflow();
}

// A walk that had to drop a stack entry has skipped a subtree: the image is wrong and the host must hear of it (RT64_DrawDevice keeps the message for RT64_GetLastError).
void Device::checkTraversalOverflow() {
    if (!traversalOverflow || *traversalOverflow == 0u) return;
    const unsigned n = *traversalOverflow; *traversalOverflow = 0;
    throw std::runtime_error("RT64_DrawDevice: " + std::to_string(n) + " traversal-stack entries were dropped (TLAS depth + BLAS depth exceed " + std::to_string(RT_STACK_LDS + RT_STACK_SPILL) +
                             " levels): the frame is missing geometry.");
}

// Timings and counters of the last frame: they need the frame to have finished on the GPU (lazy when sync_present = 0).
void Device::finishStats() {
    if (!statsPending) return;
    statsPending = false;
    RT64_FRAME_STATS st = stats;
    const bool haveView = statsHaveView;
    // The timings and counters below belong to the LAST frame handed to RT64_DrawDevice (one event set, re-recorded by every frame):
    // wait for it whatever sync_present says now -- the option may have been switched since the frame was enqueued.
    HIP_CHECK(hipStreamSynchronize(stream));
    if (traversalOverflow && *traversalOverflow) st.traversalOverflow = *traversalOverflow;
    if (statsProfiled && haveView) {
        // sums over the sub-frames of a primary_spp frame (one event set each; more than EV_SETS sub-frames: the measured ones scaled up)
        const float scale = (float)evSubFrames / (float)std::max(evSetsUsed, 1);
        for (int set = 0; set < evSetsUsed; set++) {
            hipEvent_t *E = eventSets[set]; const int *A = eventAliases[set];
            auto ms = [&](int a, int b) { float v = 0.0f; if (A[a] != A[b] && hipEventElapsedTime(&v, E[A[a]], E[A[b]]) != hipSuccess) { (void)hipGetLastError(); v = 0.0f; } return v * scale; };
            st.msTotal += ms(EV_BEGIN, EV_END); st.msBuild += ms(EV_BEGIN, EV_BUILD); st.msPrimary += ms(EV_BUILD, EV_PRIMARY);
            st.msPrimaryTrace += ms(EV_BUILD, EV_PRIMARY_TRACE); st.msPrimaryShade += ms(EV_PRIMARY_TRACE, EV_PRIMARY);
            st.msDirect += ms(EV_PRIMARY, EV_DIRECT); st.msIndirect += ms(EV_DIRECT, EV_INDIRECT);
            st.msDenoise += ms(EV_REFL, EV_DENOISE); st.msComposePost += ms(EV_DENOISE, EV_END);
            // reflection passes that ran on the second stream beside the a-trous iterations have their own pair of events: the time they took there (it
            // overlaps msDenoise; the two do not add up to wall time)
            float aux = 0.0f;
            if (auxTimed[set]) { if (hipEventElapsedTime(&aux, auxEvents[set][0], auxEvents[set][1]) != hipSuccess) { (void)hipGetLastError(); aux = 0.0f; } st.reflectionBesideDenoiser = 1; }
            st.msReflectRefract += ms(EV_INDIRECT, EV_REFL) + aux * scale;
        }
        accum.accumFrames++; accum.accumMsTotal += st.msTotal; accum.accumMsBuild += st.msBuild; accum.accumMsPrimaryTrace += st.msPrimaryTrace;
        accum.accumMsPrimaryShade += st.msPrimaryShade; accum.accumMsDirect += st.msDirect; accum.accumMsIndirect += st.msIndirect;
        accum.accumMsReflectRefract += st.msReflectRefract; accum.accumMsDenoise += st.msDenoise; accum.accumMsComposePost += st.msComposePost;
    }
    st.accumFrames = accum.accumFrames; st.accumMsTotal = accum.accumMsTotal; st.accumMsBuild = accum.accumMsBuild; st.accumMsPrimaryTrace = accum.accumMsPrimaryTrace;
    st.accumMsPrimaryShade = accum.accumMsPrimaryShade; st.accumMsDirect = accum.accumMsDirect; st.accumMsIndirect = accum.accumMsIndirect;
    st.accumMsReflectRefract = accum.accumMsReflectRefract; st.accumMsDenoise = accum.accumMsDenoise; st.accumMsComposePost = accum.accumMsComposePost;
    if (opt.countTraversal) {
        unsigned long long all[CTR_COUNT * RT_COUNTER_STRIPES], c[CTR_COUNT] = {};
        HIP_CHECK(hipMemcpy(all, counters.ptr, sizeof(all), hipMemcpyDeviceToHost));
        for (int sidx = 0; sidx < RT_COUNTER_STRIPES; sidx++) for (int k = 0; k < CTR_COUNT; k++) c[k] += all[sidx * CTR_COUNT + k];
        st.nodesVisited = c[CTR_NODES]; st.trianglesTested = c[CTR_TRIS]; st.primaryRays = c[CTR_PRIMARY]; st.shadowRays = c[CTR_SHADOW];
        st.indirectRays = c[CTR_INDIRECT]; st.reflectionRays = c[CTR_REFLECTION]; st.refractionRays = c[CTR_REFRACTION];
        st.nodesPrimary = c[CTR_PASS_BASE + 2 * PASS_PRIMARY_TRACE]; st.trianglesPrimary = c[CTR_PASS_BASE + 2 * PASS_PRIMARY_TRACE + 1];
        st.nodesDirect = c[CTR_PASS_BASE + 2 * PASS_DIRECT]; st.trianglesDirect = c[CTR_PASS_BASE + 2 * PASS_DIRECT + 1];
        st.nodesIndirect = c[CTR_PASS_BASE + 2 * PASS_INDIRECT]; st.trianglesIndirect = c[CTR_PASS_BASE + 2 * PASS_INDIRECT + 1];
    }
    stats = st;
}

// ---- readback -------------------------------------------------------------------------------------------------------------------

static float half_to_float(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, mant = h & 0x3FFu, bits;
    if (exp == 0) {
        if (mant == 0) bits = sign;
        else { int e = -1; do { e++; mant <<= 1; } while (!(mant & 0x400u)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3FFu) << 13); }
    }
    else if (exp == 0x1F) bits = sign | 0x7F800000u | (mant << 13);
    else bits = sign | ((exp - 15 + 127) << 23) | (mant << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

struct ImageInfo { const void *ptr; int srcBytes; int channels; int kind; };   // kind: 0 raw copy, 1 half->float, 2 unorm8->float

static bool image_info(View *v, int image, ImageInfo &info, size_t &dstPixelBytes) {
    const ViewImages &I = v->img;
    const int cur = v->rtSwap ? 0 : 1;        // rtSwap was flipped at the end of the frame: the last rendered set is the other one
    switch (image) {
    case RT64_IMAGE_FINAL_RGBA8: info = { I.final, 4, 4, 0 }; dstPixelBytes = 4; return true;
    case RT64_IMAGE_SHADING_POSITION: info = { I.shadingPosition, 16, 4, 0 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_SHADING_NORMAL: info = { I.shadingNormal, 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_SHADING_SPECULAR: info = { I.shadingSpecular, 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_DIFFUSE: info = { I.diffuse, 4, 4, 2 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_INSTANCE_ID: info = { I.instanceId, 4, 1, 0 }; dstPixelBytes = 4; return true;
    case RT64_IMAGE_FIRST_INSTANCE_ID: info = { I.firstInstanceId, 4, 1, 0 }; dstPixelBytes = 4; return true;
    case RT64_IMAGE_DIRECT_LIGHT_RAW: info = { I.directLight[cur], 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_DIRECT_LIGHT_FILTERED: info = { I.filteredDirect[1], 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_INDIRECT_LIGHT_RAW: info = { I.indirectLight[cur], 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_INDIRECT_LIGHT_FILTERED: info = { I.filteredIndirect[1], 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_REFLECTION: info = { I.reflection, 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_REFRACTION: info = { I.refraction, 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_TRANSPARENT: info = { I.transparent, 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_VIEW_DIRECTION: info = { I.viewDirection, 8, 4, 1 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_FLOW: info = { I.flow, 4, 2, 1 }; dstPixelBytes = 8; return true;
    case RT64_IMAGE_REACTIVE_MASK: info = { I.reactiveMask, 1, 1, 2 }; dstPixelBytes = 4; return true;
    case RT64_IMAGE_LOCK_MASK: info = { I.lockMask, 1, 1, 2 }; dstPixelBytes = 4; return true;
    case RT64_IMAGE_DEPTH: info = { I.depth[cur], 4, 1, 0 }; dstPixelBytes = 4; return true;
    case RT64_IMAGE_OUTPUT_RGBA32F: info = { I.output, 16, 4, 0 }; dstPixelBytes = 16; return true;
    case RT64_IMAGE_PRIMARY_HIT: info = { I.primaryHit, 16, 4, 0 }; dstPixelBytes = 16; return true;
    default: return false;
    }
}

static View *first_view(Device *dev) { for (Scene *sc : dev->scenes) for (View *v : sc->views) return v; return nullptr; }

static size_t readback(Device *dev, int image, void *dst, size_t dstBytes, bool toDevice) {
    dev->enter();
    // rows and layout are those of the frame that was rendered, whatever partition has been set since (restored on every way out)
    struct PartitionOfTheFrame {
        Device *d; int y0, y1, sr, sc;
        explicit PartitionOfTheFrame(Device *dev_) : d(dev_), y0(dev_->tileY0), y1(dev_->tileY1), sr(dev_->stripRank), sc(dev_->stripCount) {
            if (d->drawn.valid) { d->tileY0 = d->drawn.tileY0; d->tileY1 = d->drawn.tileY1; d->stripRank = d->drawn.stripRank; d->stripCount = d->drawn.stripCount; }
        }
        ~PartitionOfTheFrame() { d->tileY0 = y0; d->tileY1 = y1; d->stripRank = sr; d->stripCount = sc; }
    } frameRows(dev);
    View *v = first_view(dev);
    if (!v) throw std::runtime_error("RT64_ReadbackDevice: the device has no view.");
    if (image == RT64_IMAGE_BACKGROUND) {        // gBackground: whole screen on every device; zeros when the frame had no background instance
        const size_t need = (size_t)v->finalW * v->finalH * 4;
        if (dstBytes < need) throw std::runtime_error("RT64_ReadbackDevice: destination buffer is too small.");
        if (!v->rasterBgEnv.ready) { if (toDevice) HIP_CHECK(hipMemsetAsync(dst, 0, need, dev->stream)); else memset(dst, 0, need); }
        else HIP_CHECK(hipMemcpyAsync(dst, v->background.ptr, need, toDevice ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, dev->stream));
        HIP_CHECK(hipStreamSynchronize(dev->stream));
        return need;
    }
    if (image == RT64_IMAGE_UPSCALED) {          // rtOutputUpscaled of the last frame: screen size, whole frame
        const size_t need = (size_t)v->finalW * v->finalH * 16;
        if (!v->upscaleActive || !v->upValid) throw std::runtime_error("RT64_ReadbackDevice: no upscaled image (no upscaler is active).");
        if (dstBytes < need) throw std::runtime_error("RT64_ReadbackDevice: destination buffer is too small.");
        HIP_CHECK(hipMemcpyAsync(dst, v->upscaled[v->upSwap ^ 1], need, toDevice ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, dev->stream));
        HIP_CHECK(hipStreamSynchronize(dev->stream));
        return need;
    }
    if (image != RT64_IMAGE_FINAL_RGBA8 && (image != RT64_IMAGE_OUTPUT_RGBA32F || v->fusedFrame)) v->materialise();
    if (image == RT64_IMAGE_SHADING_POSITION || image == RT64_IMAGE_VIEW_DIRECTION || image == RT64_IMAGE_SHADING_NORMAL || image == RT64_IMAGE_INSTANCE_ID) v->applyReflectionState();
    ImageInfo info; size_t dstPixelBytes;
    if (!image_info(v, image, info, dstPixelBytes)) throw std::runtime_error("RT64_ReadbackDevice: unknown image id.");
    // Sizes: every image has the render size except the back buffer (screen size); they differ only with resolutionScale != 1,
    // which renders the whole frame on this device.
    const bool isFinal = image == RT64_IMAGE_FINAL_RGBA8, scaled = v->imgW != v->finalW || v->imgH != v->finalH;
    const size_t w = (size_t)(isFinal ? v->finalW : v->imgW);
    const size_t rows = scaled ? (size_t)(isFinal ? v->finalH : v->imgH) : (size_t)dev->ownedRows(), px = rows * w;
    const size_t need = px * dstPixelBytes;
    if (dstBytes < need) throw std::runtime_error("RT64_ReadbackDevice: destination buffer is too small.");
    const uint8_t *base = static_cast<const uint8_t *>(info.ptr);
    // Gather the owned strips (ascending rows) into one packed block of `srcBytes` per pixel.
    auto gather = [&](void *out, hipMemcpyKind kind, const void *srcBase, size_t srcPixelBytes) {
        size_t o = 0;
        if (dev->stripCount <= 1) {
            const size_t first = (size_t)dev->tileY0 * w;
            HIP_CHECK(hipMemcpyAsync(out, static_cast<const uint8_t *>(srcBase) + first * srcPixelBytes, px * srcPixelBytes, kind, dev->stream));
            return;
        }
        // strips are equally spaced: one 2-D copy for the full strips, one more for a ragged last strip
        const size_t stripBytes = 16 * w * srcPixelBytes, pitch = (size_t)dev->stripCount * stripBytes;
        const int firstRow = dev->tileY0 + dev->stripRank * 16;
        if (firstRow >= dev->tileY1) return;
        const size_t fullStrips = (size_t)((dev->tileY1 - firstRow) / (dev->stripCount * 16)) + (((dev->tileY1 - firstRow) % (dev->stripCount * 16)) >= 16 ? 1 : 0);
        const uint8_t *src = static_cast<const uint8_t *>(srcBase) + (size_t)firstRow * w * srcPixelBytes;
        if (fullStrips) { HIP_CHECK(hipMemcpy2DAsync(out, stripBytes, src, pitch, stripBytes, fullStrips, kind, dev->stream)); o = fullStrips * stripBytes; }
        const int lastRow = firstRow + (int)fullStrips * dev->stripCount * 16;
        if (lastRow < dev->tileY1) {
            const size_t rem = (size_t)(dev->tileY1 - lastRow) * w * srcPixelBytes;
            HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t *>(out) + o, src + fullStrips * pitch, rem, kind, dev->stream));
        }
    };
    if (toDevice) {
        if (info.kind != 0 || image == RT64_IMAGE_PRIMARY_HIT) throw std::runtime_error("RT64_CopyDeviceImage: only images stored in their API element type can be copied device-to-device.");
        gather(dst, hipMemcpyDeviceToDevice, base, (size_t)info.srcBytes);
        if (dev->opt.syncPresent) HIP_CHECK(hipStreamSynchronize(dev->stream));      // sync_present = 0: the copy is ordered on RT64_GetDeviceStream like the frame
        return need;
    }
    if (info.kind == 0 && image != RT64_IMAGE_PRIMARY_HIT) { gather(dst, hipMemcpyDeviceToHost, base, (size_t)info.srcBytes); HIP_CHECK(hipStreamSynchronize(dev->stream)); return need; }
    std::vector<uint8_t> tmp(px * info.srcBytes);
    gather(tmp.data(), hipMemcpyDeviceToHost, base, (size_t)info.srcBytes);
    HIP_CHECK(hipStreamSynchronize(dev->stream));
    if (image == RT64_IMAGE_PRIMARY_HIT) {
        std::vector<int32_t> inst(px);
        gather(inst.data(), hipMemcpyDeviceToHost, v->hitInstance.ptr, 4);
        HIP_CHECK(hipStreamSynchronize(dev->stream));
        const uint32_t *s = reinterpret_cast<const uint32_t *>(tmp.data()); uint32_t *d = static_cast<uint32_t *>(dst);
        for (size_t i = 0; i < px; i++) {
            d[4 * i] = s[4 * i]; d[4 * i + 1] = s[4 * i + 1]; d[4 * i + 2] = s[4 * i + 2];
            d[4 * i + 3] = inst[i] < 0 ? 0xFFFFFFFFu : (((uint32_t)inst[i] << 24) | (s[4 * i + 3] & 0xFFFFFFu));
        }
        return need;
    }
    float *d = static_cast<float *>(dst);
    const size_t count = px * info.channels;
    if (info.kind == 1) { const uint16_t *s = reinterpret_cast<const uint16_t *>(tmp.data()); for (size_t i = 0; i < count; i++) d[i] = half_to_float(s[i]); }
    else { const uint8_t *s = tmp.data(); for (size_t i = 0; i < count; i++) d[i] = (float)s[i] / 255.0f; }
    return need;
}

}  // namespace rt64

// ======================================================================================================================================
// extern "C" ABI.  One export per member of RT64_LIBRARY (include/rt64.h), each citing the reference export it replaces.
// ======================================================================================================================================
using namespace rt64;

#define RT64_TRY try {
#define RT64_CATCH(ret) } catch (const std::exception &e) { GlobalLastError = e.what(); fprintf(stderr, "%s\n", e.what()); return ret; }
#define RT64_CATCH_VOID } catch (const std::exception &e) { GlobalLastError = e.what(); fprintf(stderr, "%s\n", e.what()); }

RT64_EXPORT const char *RT64_GetLastError() { return GlobalLastError.c_str(); }                       // rt64_common.cpp:28

static int env_int(const char *name, int def) { const char *e = getenv(name); return e && *e ? atoi(e) : def; }

RT64_EXPORT RT64_DEVICE *RT64_CreateDeviceHeadless(int width, int height, int hipDevice) {
    RT64_TRY return reinterpret_cast<RT64_DEVICE *>(new Device(width, height, hipDevice)); RT64_CATCH(nullptr)
}
RT64_EXPORT RT64_DEVICE *RT64_CreateDevice(void *hwnd) {                                                 // rt64_device.cpp:1221
    (void)hwnd;   // no window system: the client-rect size source (rt64_device.cpp:199-231) is RT64_WIDTH x RT64_HEIGHT
    return RT64_CreateDeviceHeadless(env_int("RT64_WIDTH", 1280), env_int("RT64_HEIGHT", 720), env_int("RT64_HIP_DEVICE", -1));
}
RT64_EXPORT void RT64_DestroyDevice(RT64_DEVICE *device) { RT64_TRY delete reinterpret_cast<Device *>(device); RT64_CATCH_VOID }   // :1231
RT64_EXPORT void RT64_DrawDevice(RT64_DEVICE *device, int vsyncInterval, float deltaTimeMs) {            // :1239
    RT64_TRY if (!device) throw std::runtime_error("RT64_DrawDevice: NULL device."); reinterpret_cast<Device *>(device)->draw(vsyncInterval, deltaTimeMs); RT64_CATCH_VOID
}
RT64_EXPORT void RT64_SetDeviceSize(RT64_DEVICE *device, int width, int height) {
    Device *d = reinterpret_cast<Device *>(device); if (d && width > 0 && height > 0) { d->pendingWidth = width; d->pendingHeight = height; }
}
RT64_EXPORT void RT64_SetDeviceTile(RT64_DEVICE *device, int y0, int y1) {
    Device *d = reinterpret_cast<Device *>(device); if (!d) return;
    if (y0 < 0 || y1 <= y0) { d->tileSet = false; d->tileY0 = 0; d->tileY1 = d->height; } else { d->tileSet = true; d->tileY0 = y0; d->tileY1 = y1; }
}
RT64_EXPORT void RT64_SetDeviceInterleave(RT64_DEVICE *device, int rank, int count) {
    Device *d = reinterpret_cast<Device *>(device); if (!d) return;
    if (count <= 1 || rank < 0 || rank >= count) { d->stripRank = 0; d->stripCount = 1; } else { d->stripRank = rank; d->stripCount = count; }
}
RT64_EXPORT size_t RT64_ReadbackDevice(RT64_DEVICE *device, int image, void *dst, size_t dstBytes) {
    RT64_TRY if (!device || !dst) throw std::runtime_error("RT64_ReadbackDevice: NULL argument."); return readback(reinterpret_cast<Device *>(device), image, dst, dstBytes, false); RT64_CATCH(0)
}
RT64_EXPORT size_t RT64_CopyDeviceImage(RT64_DEVICE *device, int image, void *devicePtr, size_t dstBytes) {
    RT64_TRY if (!device || !devicePtr) throw std::runtime_error("RT64_CopyDeviceImage: NULL argument."); return readback(reinterpret_cast<Device *>(device), image, devicePtr, dstBytes, true); RT64_CATCH(0)
}
RT64_EXPORT void RT64_SetDeviceGatherTarget(RT64_DEVICE *device, void *devicePtr, size_t bytes) {
    Device *d = reinterpret_cast<Device *>(device);
    if (!d) return;
    d->gatherTarget = devicePtr; d->gatherTargetBytes = devicePtr ? bytes : 0;
}
RT64_EXPORT int RT64_GetDeviceStats(RT64_DEVICE *device, RT64_FRAME_STATS *stats) {
    Device *d = reinterpret_cast<Device *>(device);
    if (!d || !stats || stats->structSize < 5 * sizeof(unsigned int)) return 0;
    // The struct only ever grows at its end: a caller built against an older header passes its own (smaller) size and gets that prefix.
    d->use(); d->finishStats();
    RT64_FRAME_STATS full = d->stats; full.structSize = sizeof(RT64_FRAME_STATS);
    if (full.width == 0) { full.width = full.screenWidth = (unsigned)d->width; full.height = full.screenHeight = (unsigned)d->height; full.tileY0 = (unsigned)d->tileY0; full.tileY1 = (unsigned)d->tileY1; }
    const unsigned int n = stats->structSize < sizeof(RT64_FRAME_STATS) ? stats->structSize : (unsigned int)sizeof(RT64_FRAME_STATS);
    memcpy(stats, &full, n); stats->structSize = n;
    return 1;
}
RT64_EXPORT int RT64_SetDeviceOption(RT64_DEVICE *device, const char *key, double value) {
    Device *d = reinterpret_cast<Device *>(device); if (!d || !key) return 0;
    std::string k = key;
    if (k == "count_traversal") d->opt.countTraversal = value != 0.0;
    else if (k == "profile_passes") d->opt.profilePasses = value != 0.0;
    else if (k == "profile_every") { d->finishStats(); d->opt.profileEvery = std::max(1, (int)value); d->profCounter = 0; }       // pass timings (RT64_FRAME_STATS.ms*) from every n-th frame only; the other frames carry no event
    else if (k == "sync_present") d->opt.syncPresent = value != 0.0;
    else if (k == "denoiser_mode") d->opt.denoiserMode = (int)value;
    else if (k == "bounce_refill") d->opt.bounceRefill = (int)value;
    else if (k == "overlap_reflection") d->opt.overlapReflection = value != 0.0;
    else if (k == "reflection_early") d->opt.reflectionEarly = value != 0.0;
    else if (k == "overlap_frames") d->opt.overlapFrames = value != 0.0;       // 0: every frame on one render stream (a host that orders its own work behind frames on RT64_GetDeviceStream)
    else if (k == "tile_order") d->opt.tileOrder = value != 0.0;
    else if (k == "fold_variance") d->opt.foldVariance = value != 0.0;
    else if (k == "fold_guide") d->opt.foldGuide = value != 0.0;
    else if (k == "fold_compose") d->opt.foldCompose = value != 0.0;
    else if (k == "halo_exchange") d->opt.haloExchange = value != 0.0;
    else if (k == "halo_dry_run") d->opt.haloDryRun = value != 0.0;
    else if (k == "halo_margin") d->opt.haloMargin = std::max((int)value, SVGF_INPUT_HALO_ROWS);
    else if (k == "bounce_split") d->opt.bounceSplit = (int)value;
    else if (k == "bounce_groups") d->opt.bounceGroups = value >= 0.0 && value <= (double)RT_MAX_BOUNCE_GROUPS ? (int)value : -1;
    else if (k == "per_wave_frame") d->opt.perWaveFrame = (int)value;
    else if (k == "tile_timing") d->opt.tileTiming = value != 0.0;
    else if (k == "spin_present") d->opt.spinPresent = value != 0.0;
    else if (k == "lds_cache") d->opt.ldsCache = value != 0.0;
    else if (k == "lean_records") d->opt.leanRecords = value != 0.0;
    else if (k == "host_tlas") d->opt.hostTlas = value != 0.0;
    else if (k == "frame_prologue") d->opt.framePrologue = value != 0.0;
    else if (k == "simple_kernels") d->opt.simpleKernels = value != 0.0;       // 0: every frame runs the general kernels (A/B tests)
    else if (k == "reset_accum") { d->finishStats(); d->accum = RT64_FRAME_STATS(); }
    else if (k == "fold_foreground") d->opt.foldForeground = value != 0.0;        // 0: the foreground (HUD) list keeps its own raster_draw launch after a one-kernel frame
    else if (k == "fused_lean") d->opt.fusedLean = value != 0.0;                  // 0: a lean frame runs as primary_trace + primary_shade + direct instead of lean_frame_kernel
    else if (k == "lean_frames") d->opt.leanFrames = value != 0.0;                // 0: always write every image of the reference's G-buffer
    else if (k == "always_rebuild") d->opt.alwaysRebuild = value != 0.0;          // upload tables + rebuild the TLAS every frame like the reference
    else if (k == "max_reflections") d->opt.maxReflections = std::max(0, (int)value);
    else if (k == "gi_bounces") { if (value != 1.0 && value != 2.0) return 0; d->opt.giBounces = (int)value; }
    else if (k == "primary_spp") { if (!(value >= 1.0 && value <= 64.0)) return 0; d->opt.primarySpp = (int)value; }
    else if (k == "max_frame_groups") d->opt.maxFrameGroups = value >= 1.0 && value <= (double)RT_MAX_FRAME_GROUPS ? (unsigned)value : RT_MAX_FRAME_GROUPS;
    else return 0;
    return 1;
}
// Profiling aid (option tile_timing = 1): two 16-byte records per wave of the last one-kernel frame, in workgroup order -- at the wave's start { chip-wide 100 MHz
// clock, shader clock (low 32 bits each), HW_ID, 1 } and at its end { clock, shader clock, 0, 1 }.  Waves that did not run leave zeros.  Returns bytes written.
RT64_EXPORT size_t RT64_ReadbackTileTiming(RT64_DEVICE *device, void *dst, size_t dstBytes) {
    RT64_TRY
    Device *d = reinterpret_cast<Device *>(device);
    if (!d || !dst || !d->tileTiming.ptr) throw std::runtime_error("RT64_ReadbackTileTiming: set the device option tile_timing and draw a frame first.");
    d->enter();
    const size_t bytes = std::min(dstBytes, d->tileTiming.bytes());
    HIP_CHECK(hipMemcpyAsync(dst, d->tileTiming.ptr, bytes, hipMemcpyDeviceToHost, d->stream));
    HIP_CHECK(hipStreamSynchronize(d->stream));
    return bytes;
    RT64_CATCH(0)
}
RT64_EXPORT void *RT64_GetDeviceStream(RT64_DEVICE *device) { Device *d = reinterpret_cast<Device *>(device); return d ? d->stream : nullptr; }

// ---- view (rt64_view.cpp:2086-2201) ----
RT64_EXPORT RT64_VIEW *RT64_CreateView(RT64_SCENE *scenePtr) {
    RT64_TRY if (!scenePtr) throw std::runtime_error("RT64_CreateView: NULL scene."); return reinterpret_cast<RT64_VIEW *>(new View(reinterpret_cast<Scene *>(scenePtr))); RT64_CATCH(nullptr)
}
RT64_EXPORT void RT64_SetViewPerspective(RT64_VIEW *viewPtr, RT64_MATRIX4 viewMatrix, float fovRadians, float nearDist, float farDist, bool canReproject) {
    View *v = reinterpret_cast<View *>(viewPtr); if (!v) return;
    v->view = mat_from(viewMatrix); v->fov = fovRadians; v->nearDist = nearDist; v->farDist = farDist; v->canReproject = canReproject; v->perspectiveSet = fovRadians > 0.0f;
}
RT64_EXPORT void RT64_SetViewDescription(RT64_VIEW *viewPtr, RT64_VIEW_DESC viewDesc) {
    View *v = reinterpret_cast<View *>(viewPtr); if (!v) return;
    v->resolutionScale = viewDesc.resolutionScale; v->motionBlurStrength = viewDesc.motionBlurStrength; v->maxLights = viewDesc.maxLights;
    v->diSamples = viewDesc.diSamples; v->giSamples = viewDesc.giSamples; v->denoiserEnabled = viewDesc.denoiserEnabled;
    // upscaler / upscalerMode (rt64_view.cpp:2109-2163): AUTO prefers DLSS, then XeSS on Intel, then FSR; the vendor SDKs do not exist here, so AUTO and
    // FSR select the built-in temporal upscaler (upscale.hip) and DLSS / XeSS fall back to the bilinear resample like an uninitialised SDK does (:116,139-141).
    // upscalerSharpness is accepted and has no effect (no sharpening pass).
    v->upscaler = viewDesc.upscaler; v->upscalerMode = viewDesc.upscalerMode; v->upscalerSharpness = viewDesc.upscalerSharpness;
}
RT64_EXPORT void RT64_SetViewSkyPlane(RT64_VIEW *viewPtr, RT64_TEXTURE *texturePtr) { View *v = reinterpret_cast<View *>(viewPtr); if (v) v->skyPlane = reinterpret_cast<Texture *>(texturePtr); }
RT64_EXPORT RT64_INSTANCE *RT64_GetViewRaytracedInstanceAt(RT64_VIEW *viewPtr, int x, int y) {         // rt64_view.cpp:1932-1998
    RT64_TRY
    View *v = reinterpret_cast<View *>(viewPtr); if (!v) return nullptr;
    Device *dev = v->scene->device; dev->enter();
    const float xs = (float)v->imgW / (float)dev->width, ys = (float)v->imgH / (float)dev->height;
    x = (int)lroundf(x * xs); y = (int)lroundf(y * ys);
    if (x < 0 || x >= v->imgW || y < 0 || y >= v->imgH) return nullptr;
    int32_t id = -1;
    v->materialise();
    // ordered behind the frame on the renderer's (non-blocking) stream: a null-stream copy would not wait for an enqueued frame
    HIP_CHECK(hipMemcpyAsync(&id, v->img.firstInstanceId + (size_t)y * v->imgW + x, 4, hipMemcpyDeviceToHost, dev->stream));
    HIP_CHECK(hipStreamSynchronize(dev->stream));
    if (id >= 0 && (size_t)id < v->rtInstances.size()) return reinterpret_cast<RT64_INSTANCE *>(v->rtInstances[id].instance);
    return nullptr;
    RT64_CATCH(nullptr)
}
RT64_EXPORT bool RT64_GetViewUpscalerSupport(RT64_VIEW *viewPtr, int upscaler) {                       // rt64_view.cpp:2183 (declared (view, int) there)
    return viewPtr != nullptr && upscaler == RT64_UPSCALER_FSR;       // the built-in temporal upscaler answers for FSR; DLSS / XeSS: not initialised
}
RT64_EXPORT void RT64_DestroyView(RT64_VIEW *viewPtr) { RT64_TRY delete reinterpret_cast<View *>(viewPtr); RT64_CATCH_VOID }

// ---- scene (rt64_scene.cpp:170-187) ----
RT64_EXPORT RT64_SCENE *RT64_CreateScene(RT64_DEVICE *devicePtr) {
    RT64_TRY if (!devicePtr) throw std::runtime_error("RT64_CreateScene: NULL device."); return reinterpret_cast<RT64_SCENE *>(new Scene(reinterpret_cast<Device *>(devicePtr))); RT64_CATCH(nullptr)
}
RT64_EXPORT void RT64_SetSceneDescription(RT64_SCENE *scenePtr, RT64_SCENE_DESC sceneDesc) { Scene *s = reinterpret_cast<Scene *>(scenePtr); if (s) s->desc = sceneDesc; }
RT64_EXPORT void RT64_SetSceneLights(RT64_SCENE *scenePtr, RT64_LIGHT *lightArray, int lightCount) {   // rt64_scene.cpp:114-150
    Scene *s = reinterpret_cast<Scene *>(scenePtr); if (!s || lightCount < 0) return;
    s->lights.assign(lightCount, RT64_LIGHT());
    if (lightArray) {
        memcpy(s->lights.data(), lightArray, sizeof(RT64_LIGHT) * (size_t)lightCount);
        for (RT64_LIGHT &l : s->lights) {
            if (l.flickerIntensity > 0.0f) {
                const float mult = 1.0f + ((((float)rand() / (float)RAND_MAX) * 2.0f - 1.0f) * l.flickerIntensity);
                l.diffuseColor.x *= mult; l.diffuseColor.y *= mult; l.diffuseColor.z *= mult;
            }
        }
    }
}
RT64_EXPORT void RT64_DestroyScene(RT64_SCENE *scenePtr) { RT64_TRY delete reinterpret_cast<Scene *>(scenePtr); RT64_CATCH_VOID }

// ---- mesh (rt64_mesh.cpp:190-209) ----
RT64_EXPORT RT64_MESH *RT64_CreateMesh(RT64_DEVICE *devicePtr, int flags) {
    RT64_TRY if (!devicePtr) throw std::runtime_error("RT64_CreateMesh: NULL device."); return reinterpret_cast<RT64_MESH *>(new Mesh(reinterpret_cast<Device *>(devicePtr), flags)); RT64_CATCH(nullptr)
}
RT64_EXPORT void RT64_SetMesh(RT64_MESH *meshPtr, void *vertexArray, int vertexCount, int vertexStride, unsigned int *indexArray, int indexCount) {
    RT64_TRY if (!meshPtr) throw std::runtime_error("RT64_SetMesh: NULL mesh."); reinterpret_cast<Mesh *>(meshPtr)->set(vertexArray, vertexCount, vertexStride, indexArray, indexCount); RT64_CATCH_VOID
}
RT64_EXPORT void RT64_DestroyMesh(RT64_MESH *meshPtr) { RT64_TRY Mesh *m = reinterpret_cast<Mesh *>(meshPtr); if (m) { m->device->enter(); m->device->beforeSceneMutation(); hipStreamSynchronize(m->device->stream); } delete m; RT64_CATCH_VOID }

// ---- shader (rt64_shader.cpp:810-824) ----
RT64_EXPORT RT64_SHADER *RT64_CreateShader(RT64_DEVICE *devicePtr, unsigned int shaderId, unsigned int filter, unsigned int hAddr, unsigned int vAddr, int flags) {
    RT64_TRY
    if (!devicePtr) throw std::runtime_error("RT64_CreateShader: NULL device.");
    if (filter > 1 || hAddr > 2 || vAddr > 2) throw std::runtime_error("RT64_CreateShader: invalid sampler state.");
    Shader *s = new Shader{ reinterpret_cast<Device *>(devicePtr), shaderId, filter, hAddr, vAddr, flags, decode_combiner(shaderId) };
    return reinterpret_cast<RT64_SHADER *>(s);
    RT64_CATCH(nullptr)
}
RT64_EXPORT void RT64_DestroyShader(RT64_SHADER *shaderPtr) { delete reinterpret_cast<Shader *>(shaderPtr); }

// ---- instance (rt64_instance.cpp:145-173) ----
RT64_EXPORT RT64_INSTANCE *RT64_CreateInstance(RT64_SCENE *scenePtr) {
    RT64_TRY if (!scenePtr) throw std::runtime_error("RT64_CreateInstance: NULL scene."); return reinterpret_cast<RT64_INSTANCE *>(new Instance(reinterpret_cast<Scene *>(scenePtr))); RT64_CATCH(nullptr)
}
RT64_EXPORT void RT64_SetInstanceDescription(RT64_INSTANCE *instancePtr, RT64_INSTANCE_DESC d) {
    Instance *i = reinterpret_cast<Instance *>(instancePtr); if (!i) return;
    i->mesh = reinterpret_cast<Mesh *>(d.mesh); i->transform = mat_from(d.transform); i->previousTransform = mat_from(d.previousTransform);
    i->material = d.material; i->shader = reinterpret_cast<Shader *>(d.shader);
    i->diffuse = reinterpret_cast<Texture *>(d.diffuseTexture); i->normal = reinterpret_cast<Texture *>(d.normalTexture); i->specular = reinterpret_cast<Texture *>(d.specularTexture);
    i->flags = d.flags; i->scissorRect = d.scissorRect; i->viewportRect = d.viewportRect;
}
RT64_EXPORT void RT64_DestroyInstance(RT64_INSTANCE *instancePtr) { delete reinterpret_cast<Instance *>(instancePtr); }

// ---- texture (rt64_texture.cpp:207-233) ----
RT64_EXPORT RT64_TEXTURE *RT64_CreateTexture(RT64_DEVICE *devicePtr, RT64_TEXTURE_DESC textureDesc) {
    Texture *t = nullptr;
    RT64_TRY
    if (!devicePtr) throw std::runtime_error("RT64_CreateTexture: NULL device.");
    t = new Texture(reinterpret_cast<Device *>(devicePtr));
    switch (textureDesc.format) {
    case RT64_TEXTURE_FORMAT_RGBA8: t->setRGBA8(textureDesc.bytes, textureDesc.byteCount, textureDesc.width, textureDesc.height, textureDesc.rowPitch); break;
    case RT64_TEXTURE_FORMAT_DDS: t->setDDS(textureDesc.bytes, textureDesc.byteCount); break;
    default: throw std::runtime_error("RT64_CreateTexture: unknown texture format.");
    }
    return reinterpret_cast<RT64_TEXTURE *>(t);
    } catch (const std::exception &e) { GlobalLastError = e.what(); fprintf(stderr, "%s\n", e.what()); delete t; return nullptr; }
}
RT64_EXPORT void RT64_DestroyTexture(RT64_TEXTURE *texture) { RT64_TRY Texture *t = reinterpret_cast<Texture *>(texture); if (t) { t->device->enter(); t->device->beforeSceneMutation(); hipStreamSynchronize(t->device->stream); } delete t; RT64_CATCH_VOID }

// Debug readback of a texture's texels as the kernels sample them (additive): RGBA8, mip `mip`, rows top to bottom -- for a BC7 DDS what bc7_decode_kernel
// wrote at creation (rt64_texture.cpp:146-187 hands the blocks to the sampler hardware; here they are decoded once).  dst = NULL returns the size.
RT64_EXPORT size_t RT64_ReadbackTexture(RT64_TEXTURE *texturePtr, int mip, void *dst, size_t dstBytes) {
    RT64_TRY
    Texture *t = reinterpret_cast<Texture *>(texturePtr);
    if (!t || mip < 0 || mip >= t->mips) throw std::runtime_error("RT64_ReadbackTexture: NULL texture or no such mip level.");
    const int mw = std::max(t->width >> mip, 1), mh = std::max(t->height >> mip, 1);
    const size_t need = (size_t)mw * mh * 4;
    if (!dst) return need;
    if (dstBytes < need) throw std::runtime_error("RT64_ReadbackTexture: destination buffer is too small.");
    t->device->enter();
    HIP_CHECK(hipStreamSynchronize(t->device->stream));
    HIP_CHECK(hipMemcpy(dst, t->texels.ptr + (size_t)t->mipOffset[mip] * 4, need, hipMemcpyDeviceToHost));
    return need;
    RT64_CATCH(0)
}

// ---- debug readback of acceleration structures (additive) ----
static size_t accel_readback(Device *dev, int what, uint32_t n, const GpuNode *nodes, const GpuTri *tris, const uint32_t *sorted, const uint32_t *morton,
                             const BlasHeader *header, void *dst, size_t dstBytes) {
    dev->enter();
    const void *src = nullptr; size_t bytes = 0;
    switch (what) {
    case RT64_ACCEL_NODES: src = nodes; bytes = (size_t)std::max<uint32_t>(n > 0 ? n - 1 : 0, 1) * sizeof(GpuNode); break;
    case RT64_ACCEL_TRIANGLES: src = tris; bytes = (size_t)n * sizeof(GpuTri); break;
    case RT64_ACCEL_SORTED_INDEX: src = sorted; bytes = (size_t)n * 4; break;
    case RT64_ACCEL_MORTON: src = morton; bytes = (size_t)n * 4; break;
    case RT64_ACCEL_HEADER: src = header; bytes = sizeof(BlasHeader); break;
    default: throw std::runtime_error("RT64_Readback*Accel: unknown array id.");
    }
    if (!src || n == 0) throw std::runtime_error("RT64_Readback*Accel: no acceleration structure.");
    if (!dst) return bytes;
    if (dstBytes < bytes) throw std::runtime_error("RT64_Readback*Accel: destination buffer is too small.");
    HIP_CHECK(hipStreamSynchronize(dev->stream));
    HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return bytes;
}
RT64_EXPORT size_t RT64_ReadbackMeshAccel(RT64_MESH *meshPtr, int what, void *dst, size_t dstBytes) {
    RT64_TRY
    Mesh *m = reinterpret_cast<Mesh *>(meshPtr); if (!m) throw std::runtime_error("RT64_ReadbackMeshAccel: NULL mesh.");
    if (what == RT64_ACCEL_HOST_DEPTH) {
        if (m->blasCount == 0) throw std::runtime_error("RT64_ReadbackMeshAccel: no acceleration structure.");
        if (!dst) return sizeof(uint32_t);
        if (dstBytes < sizeof(uint32_t)) throw std::runtime_error("RT64_ReadbackMeshAccel: destination buffer is too small.");
        const uint32_t d = m->treeDepth(); memcpy(dst, &d, sizeof(d)); return sizeof(d);
    }
    m->device->flushMeshBuilds();
    return accel_readback(m->device, what, m->blasCount, m->nodes.ptr, m->tris.ptr, m->sortedIndex.ptr, m->morton.ptr, m->header.ptr, dst, dstBytes);
    RT64_CATCH(0)
}
// The depth RT64_SetMesh derives for the BLAS of these triangles, as a pure host function (no device): what RT64_ACCEL_HOST_DEPTH reports for a mesh.
RT64_EXPORT unsigned int RT64_MeshTreeDepth(const void *vertexArray, int vertexCount, int vertexStride, const unsigned int *indexArray, int indexCount) {
    if (!vertexArray || !indexArray || vertexCount <= 0 || vertexStride < 12 || indexCount < 3) return 0;
    for (int i = 0; i < indexCount; i++) if (indexArray[i] >= (unsigned int)vertexCount) return 0;
    return host_blas_depth(static_cast<const uint8_t *>(vertexArray), (size_t)vertexStride, indexArray, (uint32_t)indexCount / 3);
}
RT64_EXPORT size_t RT64_ReadbackViewAccel(RT64_VIEW *viewPtr, int what, void *dst, size_t dstBytes) {
    RT64_TRY
    View *v = reinterpret_cast<View *>(viewPtr); if (!v) throw std::runtime_error("RT64_ReadbackViewAccel: NULL view.");
    if (what == RT64_ACCEL_TRIANGLES) throw std::runtime_error("RT64_ReadbackViewAccel: a TLAS has no triangle array.");
    return accel_readback(v->scene->device, what, (uint32_t)v->rtInstances.size(), v->tab[v->tabCur].tlasNodesAt, nullptr, v->tab[v->tabCur].tlasIndexAt, v->tab[v->tabCur].tlasMortonAt, v->tab[v->tabCur].tlasHeaderAt, dst, dstBytes);
    RT64_CATCH(0)
}

// ---- multi-GPU: image-tile partition + RCCL gather of the composited back buffer (additive exports) -----------------------------
// One process per GPU.  Every rank creates a gather with the same unique id (rank 0 makes it with RT64_GetGatherUniqueId and hands it
// to the others over whatever channel the host has: a file, a pipe, MPI, torch.distributed's store); the gather sets the device's
// partition (interleaved 16-row strips, or contiguous bands for frames with a spatial filter) and owns two slots of send / receive
// buffers.  Per frame the host calls RT64_DrawDevice and RT64_SubmitGather: the frame's last kernel has written the rank's packed rows
// straight into the slot's send buffer (RT64_SetDeviceGatherTarget; frames that do not honour it are copied), the exchange -- grouped
// ncclSend / ncclRecv to rank 0, xGMI point to point -- and rank 0's reassembly run on the gather's own stream behind an event, so the
// gather of frame k overlaps the rendering of frame k + 1; the slot is handed back to the renderer two frames later.
// RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU host never loads it and librt64.so does not link it.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A build host without the RCCL headers (single-GPU deployments): the handful of declarations the gather binds with dlsym, as RCCL 2.x declares them.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
const char *ncclGetErrorString(ncclResult_t);
}
#endif

namespace rt64 {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr; decltype(&ncclCommInitRank) CommInitRank = nullptr; decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr; decltype(&ncclGroupEnd) GroupEnd = nullptr; decltype(&ncclSend) Send = nullptr; decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi &rccl() {
    static RcclApi api; static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = { getenv("RT64_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char *n : names) { if (n && *n && (api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break; }
        if (api.handle) {
#define RT64_RCCL_SYM(member, name) api.member = reinterpret_cast<decltype(api.member)>(dlsym(api.handle, name))
            RT64_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); RT64_RCCL_SYM(CommInitRank, "ncclCommInitRank"); RT64_RCCL_SYM(CommDestroy, "ncclCommDestroy");
            RT64_RCCL_SYM(GroupStart, "ncclGroupStart"); RT64_RCCL_SYM(GroupEnd, "ncclGroupEnd"); RT64_RCCL_SYM(Send, "ncclSend"); RT64_RCCL_SYM(Recv, "ncclRecv");
            RT64_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RT64_RCCL_SYM
        }
    }
    if (!api.handle || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv)
        throw std::runtime_error("RCCL (librccl.so.1) could not be loaded: the multi-GPU gather needs it.");
    return api;
}
#define RCCL_CHECK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    char msg_[512]; snprintf(msg_, sizeof(msg_), "RCCL call " #call " failed: %s", rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); throw std::runtime_error(msg_); } } while (0)

// Boundaries of `count` contiguous bands of about equal cost: cost(row) = width + RT64_BAND_HIT_WEIGHT * (pixels of the row that hit geometry).
// Greedy cut at the running cost's crossings of k / count of the total; every band keeps at least RT64_BAND_MIN_ROWS rows (while the frame has them).
#define RT64_BAND_HIT_WEIGHT 6
#define RT64_BAND_MIN_ROWS 16
static void gather_balanced_bands(const uint32_t *hitCounts, int width, int height, int count, int *starts) {
    std::vector<double> prefix((size_t)height + 1, 0.0);
    for (int y = 0; y < height; y++) prefix[(size_t)y + 1] = prefix[(size_t)y] + (double)width + (double)RT64_BAND_HIT_WEIGHT * (double)hitCounts[y];
    const double total = prefix[(size_t)height];
    const int minRows = height >= count * RT64_BAND_MIN_ROWS ? RT64_BAND_MIN_ROWS : std::max(height / count, 1);
    starts[0] = 0;
    for (int r = 1; r < count; r++) {
        const double target = total * (double)r / (double)count;
        int y = (int)(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());
        y = std::max(y, starts[r - 1] + minRows);                       // not thinner than the minimum ...
        y = std::min(y, height - (count - r) * minRows);                // ... and leave the minimum for every band still to come
        starts[r] = std::max(y, starts[r - 1]);
    }
    starts[count] = height;
}

// Feedback step of the band balance: given the bands a frame was cut into and what each rank's share cost (ms of GPU work per frame, measured), the bands of
// about equal cost under the assumption that a band's cost is spread evenly over its rows.  A fixed point when all costs are equal; the caller iterates
// (measure, rebalance) two or three times.  Moves are damped (below) and every band keeps RT64_BAND_MIN_ROWS rows.
static void gather_rebalanced_bands(int height, int count, const int *starts, const float *ms, int *out) {
    // what moves with the boundaries is the cost above the part every band pays whatever its height (launches, latency tails: about half of the cheapest
    // band's time on the sample scene); 0.4 of the cheapest band is taken as that part
    double fixed = (double)ms[0];
    for (int r = 1; r < count; r++) fixed = std::min(fixed, (double)ms[r]);
    fixed = 0.4 * std::max(fixed, 0.0);
    std::vector<double> cum((size_t)count + 1, 0.0);
    for (int r = 0; r < count; r++) cum[(size_t)r + 1] = cum[(size_t)r] + std::max((double)ms[r] - fixed, 1e-6);
    const double total = cum[(size_t)count];
    const int minRows = height >= count * RT64_BAND_MIN_ROWS ? RT64_BAND_MIN_ROWS : std::max(height / count, 1);
    // the even spread is a coarse model of a tall band (two ranks: a cut that moves into rows twice as dense as its band's average overshoots and the rounds
    // oscillate at full step) and a good one of a thin band: half the way for two bands, 0.8 of it for eight
    const double damp = 0.5 + 0.4 * (1.0 - 2.0 / (double)std::max(count, 2));
    out[0] = 0;
    for (int k = 1; k < count; k++) {
        const double target = total * (double)k / (double)count;
        int r = 0; while (r + 1 < count && cum[(size_t)r + 1] < target) r++;                  // the band the k-th cut falls into
        const double rows = (double)(starts[r + 1] - starts[r]), inside = (target - cum[(size_t)r]) / (cum[(size_t)r + 1] - cum[(size_t)r]);
        const double ideal = (double)starts[r] + rows * std::min(std::max(inside, 0.0), 1.0);
        int y = (int)std::lround((double)starts[k] + damp * (ideal - (double)starts[k]));
        y = std::max(y, out[k - 1] + minRows);
        y = std::min(y, height - (count - k) * minRows);
        out[k] = std::max(y, out[k - 1]);
    }
    out[count] = height;
}

static bool halo_starts_valid(int H, int count, const int *starts, bool strict = false);

struct Gather {
    Device *dev; int rank, count, bands; int W, H; size_t slotBytes; GatherLayout layout;
    ncclComm_t comm = nullptr; hipStream_t commStream = nullptr;
    struct Slot { uint8_t *local = nullptr, *bucket = nullptr, *frame = nullptr; hipEvent_t produced = nullptr, gathered = nullptr; bool pending = false; } slots[2];
    int next = 0, last = -1;
    // Direct mode (RT64_SetGatherDirect): no rows travel through RCCL.  Rank 0 owns DIRECT_SLOTS whole frames in fine-grained device memory and exports them over IPC;
    // every rank's frame kernel stores its rows straight into the slot of the frame in hand (Device::finalOverride) -- on the other ranks those are peer stores over
    // xGMI, 4 bytes per pixel riding inside the render kernel -- and the per-frame exchange shrinks to a 4-byte token per rank on the same communicator and stream
    // (ordering and flow control exactly as with the rows: a rank's token follows its frame, rank 0's `gathered` event follows every token).  Slot reuse: frame j is
    // written into slot j % DIRECT_SLOTS once this rank's token of frame j - DIRECT_LAG has been taken, i.e. rank 0 has reached that exchange and finished every
    // earlier one: the frame six submits back is complete, and on rank 0 the frames of the last three submits are never being overwritten.
    enum { DIRECT_SLOTS = 6, DIRECT_LAG = 3 };
    struct Direct {
        bool on = false, mapped = false; uint8_t *mem = nullptr; uint32_t *tokens = nullptr;
        hipEvent_t produced[DIRECT_SLOTS] = {}, gathered[DIRECT_SLOTS] = {}; bool pending[DIRECT_SLOTS] = {};
        long long frames = 0;
    } direct;
    size_t directHandle(void *handle, size_t bytes);
    void setDirect(const void *handle, size_t bytes, int enable);
    void prepareDirect(int slot);
    int submitDirect();
    Gather(Device *d, const ncclUniqueId &id, int rank_, int count_, int bands_);
    ~Gather();
    void prepare(int slot);
    int submit();
    void wait(int slot, bool host);
    void setBands(const int *starts);
};

Gather::Gather(Device *d, const ncclUniqueId &id, int rank_, int count_, int bands_) : dev(d), rank(rank_), count(count_), bands(bands_) {
    if (count < 1 || rank < 0 || rank >= count) throw std::runtime_error("RT64_CreateGather: rank / count out of range.");
    dev->enter();
    W = dev->pendingWidth; H = dev->pendingHeight;
    if (count > RT64_GATHER_MAX_RANKS) throw std::runtime_error("RT64_CreateGather: more ranks than RT64_GATHER_MAX_RANKS.");
    if (bands < 0 || bands > 2) throw std::runtime_error("RT64_CreateGather: bands must be 0 (interleaved strips), 1 (equal bands) or 2 (cost-balanced bands).");
    layout = gather_layout(H, count, bands);
    if (bands == 2) {
        // Cost-balanced contiguous bands.  Cost model of a row: its pixels, the ones whose primary ray hits geometry weighted RT64_BAND_HIT_WEIGHT
        // times (a GI + denoiser pixel costs several times a sky pixel), counted on the last frame this device rendered -- which has to be
        // a whole frame, the same on every rank (every rank renders the same scene), so that all ranks derive the same boundaries.
        View *v = first_view(dev);
        if (!v || v->frameCount == 0 || v->imgW != W || v->imgH != H || dev->tileY0 != 0 || dev->tileY1 != dev->height || dev->stripCount > 1)
            throw std::runtime_error("RT64_CreateGather: cost-balanced bands are cut from the device's last frame: render one whole frame (no tile / interleave) before creating the gather.");
        v->materialise();            // a pixel-local frame keeps no hit records: this brings them back (no-op after a frame that stored its G-buffer)
        DevArray<uint32_t> dCounts; dCounts.reserve((size_t)H);
        HIP_CHECK(launch_row_hit_count(v->hitInstance.ptr, dCounts.ptr, W, H, dev->stream));
        std::vector<uint32_t> counts((size_t)H);
        HIP_CHECK(hipMemcpyAsync(counts.data(), dCounts.ptr, (size_t)H * 4, hipMemcpyDeviceToHost, dev->stream));
        HIP_CHECK(hipStreamSynchronize(dev->stream));
        gather_balanced_bands(counts.data(), W, H, count, layout.starts);
    }
    slotBytes = (size_t)gather_max_owned_rows(layout) * (size_t)W * 4;
    // the device's share of the frame
    if (bands == 2) { dev->tileSet = true; dev->tileY0 = layout.starts[rank]; dev->tileY1 = layout.starts[rank + 1]; dev->stripRank = 0; dev->stripCount = 1; }
    else if (bands) { const int b = gather_band_rows(H, count); dev->tileSet = true; dev->tileY0 = std::min(rank * b, H); dev->tileY1 = std::min((rank + 1) * b, H); dev->stripRank = 0; dev->stripCount = 1; }
    else { dev->tileSet = false; dev->tileY0 = 0; dev->tileY1 = H; dev->stripRank = count > 1 ? rank : 0; dev->stripCount = count > 1 ? count : 1; }
    HIP_CHECK(hipStreamCreateWithFlags(&commStream, hipStreamNonBlocking));
    for (Slot &sl : slots) {
        HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&sl.local), std::max<size_t>(slotBytes, 16)));
        HIP_CHECK(hipMemsetAsync(sl.local, 0, std::max<size_t>(slotBytes, 16), dev->stream));
        if (rank == 0) {
            HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&sl.bucket), std::max<size_t>(slotBytes * (size_t)count, 16)));
            HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&sl.frame), (size_t)W * H * 4));
        }
        HIP_CHECK(hipEventCreateWithFlags(&sl.produced, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&sl.gathered, hipEventDisableTiming));
    }
    HIP_CHECK(hipStreamSynchronize(dev->stream));
    RCCL_CHECK(rccl().CommInitRank(&comm, count, id, rank));
    if (bands) dev->halo.gather = this;           // the bands can exchange their denoiser halos over this communicator (option halo_exchange)
    prepare(0);
}
Gather::~Gather() {
    hipSetDevice(dev->hipDevice);
    for (hipStream_t st : dev->streams) if (st) hipStreamSynchronize(st);
    if (commStream) hipStreamSynchronize(commStream);
    dev->finalOverride = nullptr; dev->frameWait = nullptr;
    if (direct.mem) { if (direct.mapped) hipIpcCloseMemHandle(direct.mem); else hipFree(direct.mem); }
    if (direct.tokens) hipFree(direct.tokens);
    for (hipEvent_t e : direct.produced) if (e) hipEventDestroy(e);
    for (hipEvent_t e : direct.gathered) if (e) hipEventDestroy(e);
    if (dev->gatherTarget == slots[0].local || dev->gatherTarget == slots[1].local) { dev->gatherTarget = nullptr; dev->gatherTargetBytes = 0; }
    if (dev->halo.gather == this) dev->halo.gather = nullptr;
    if (comm) rccl().CommDestroy(comm);
    for (Slot &sl : slots) { if (sl.local) hipFree(sl.local); if (sl.bucket) hipFree(sl.bucket); if (sl.frame) hipFree(sl.frame); if (sl.produced) hipEventDestroy(sl.produced); if (sl.gathered) hipEventDestroy(sl.gathered); }
    if (commStream) hipStreamDestroy(commStream);
}
// The renderer may write slot `slot` again: its previous exchange (two frames ago) is ordered before whatever the render stream does next.
void Gather::prepare(int slot) {
    Slot &sl = slots[slot];
    // (the next frame may start on any render stream: Device::draw makes the one it picks wait)
    if (sl.pending) { dev->frameWait = sl.gathered; sl.pending = false; }
    dev->gatherTarget = sl.local; dev->gatherTargetBytes = slotBytes;
}
// ---- direct mode ----
size_t Gather::directHandle(void *handle, size_t bytes) {
    if (rank != 0) throw std::runtime_error("RT64_GetGatherDirectHandle: rank 0 owns the frame slots.");
    if (!handle || bytes < sizeof(hipIpcMemHandle_t)) throw std::runtime_error("RT64_GetGatherDirectHandle: the handle buffer needs RT64_GATHER_DIRECT_HANDLE_BYTES bytes.");
    dev->enter();
    if (!direct.mem) {
        const size_t total = (size_t)DIRECT_SLOTS * (size_t)W * H * 4;
        HIP_CHECK(hipExtMallocWithFlags(reinterpret_cast<void **>(&direct.mem), total, hipDeviceMallocFinegrained));      // peers store into it while this device reads it: coherent at system scope
        HIP_CHECK(hipMemset(direct.mem, 0, total));
    }
    hipIpcMemHandle_t h;
    HIP_CHECK(hipIpcGetMemHandle(&h, direct.mem));
    memcpy(handle, &h, sizeof(h));
    return sizeof(h);
}
// enable: 0 back to the RCCL exchange of the rows; 1 the slots are rank 0's own memory (the other ranks map `handle`); 2 the slots belong to someone else -- a
// presenter process, say -- and EVERY rank, rank 0 included, maps `handle`.
void Gather::setDirect(const void *handle, size_t bytes, int enable) {
    dev->enter();
    for (hipStream_t st : dev->streams) if (st) HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipStreamSynchronize(commStream));          // nothing of the other mode is in flight
    if (!enable) {
        direct.on = false; dev->finalOverride = nullptr;
        for (Slot &sl : slots) sl.pending = false;
        next = 0; last = -1; prepare(next);
        return;
    }
    if (dev->width != W || dev->height != H) throw std::runtime_error("RT64_SetGatherDirect: the device was resized after RT64_CreateGather.");
    if (!direct.mem) {
        if (rank == 0 && enable != 2) { unsigned char tmp[sizeof(hipIpcMemHandle_t)]; directHandle(tmp, sizeof(tmp)); }
        else {
            if (!handle || bytes < sizeof(hipIpcMemHandle_t)) throw std::runtime_error("RT64_SetGatherDirect: ranks other than 0 need rank 0's handle (RT64_GetGatherDirectHandle).");
            hipIpcMemHandle_t h; memcpy(&h, handle, sizeof(h));
            void *p = nullptr;
            HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
            direct.mem = static_cast<uint8_t *>(p); direct.mapped = true;
        }
    }
    if (!direct.tokens) { HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&direct.tokens), sizeof(uint32_t) * (size_t)std::max(count, 16))); HIP_CHECK(hipMemset(direct.tokens, 0, sizeof(uint32_t) * (size_t)std::max(count, 16))); }
    for (int k = 0; k < DIRECT_SLOTS; k++) {
        if (!direct.produced[k]) { HIP_CHECK(hipEventCreateWithFlags(&direct.produced[k], hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&direct.gathered[k], hipEventDisableTiming)); }
        direct.pending[k] = false;
    }
    direct.on = true; direct.frames = 0; next = 0; last = -1;
    dev->gatherTarget = nullptr; dev->gatherTargetBytes = 0;      // no packed copy of the rows: they are stored once, into the frame itself
    prepareDirect(0);
}
// The renderer may write slot `slot`: this rank's token of the frame DIRECT_LAG submits back has been taken (see Gather::Direct).
void Gather::prepareDirect(int slot) {
    const int back = (slot + DIRECT_SLOTS - DIRECT_LAG) % DIRECT_SLOTS;
    if (direct.pending[back]) { dev->frameWait = direct.gathered[back]; direct.pending[back] = false; }          // (applied by Device::draw to the stream the frame starts on)
    dev->finalOverride = direct.mem + (size_t)slot * (size_t)W * H * 4;
}
int Gather::submitDirect() {
    dev->use();
    if (dev->width != W || dev->height != H) throw std::runtime_error("RT64_SubmitGather: the device was resized after RT64_CreateGather.");
    const int slot = next;
    HIP_CHECK(hipEventRecord(direct.produced[slot], dev->stream));          // the frame's last store into the slot is behind this
    HIP_CHECK(hipStreamWaitEvent(commStream, direct.produced[slot], 0));
    if (count > 1) {
        RcclApi &R = rccl();
        RCCL_CHECK(R.GroupStart());
        if (rank == 0) { for (int r = 1; r < count; r++) RCCL_CHECK(R.Recv(direct.tokens + r, 4, ncclUint8, r, comm, commStream)); }
        else RCCL_CHECK(R.Send(direct.tokens, 4, ncclUint8, 0, comm, commStream));
        RCCL_CHECK(R.GroupEnd());
    }
    HIP_CHECK(hipEventRecord(direct.gathered[slot], commStream));
    direct.pending[slot] = true; direct.frames++; last = slot; next = (slot + 1) % DIRECT_SLOTS;
    prepareDirect(next);
    return slot;
}

// After RT64_DrawDevice: exchange the frame just drawn.  Returns the slot it travels in.
int Gather::submit() {
    if (direct.on) return submitDirect();
    dev->use();
    if (dev->width != W || dev->height != H) throw std::runtime_error("RT64_SubmitGather: the device was resized after RT64_CreateGather.");
    const int slot = next; Slot &sl = slots[slot];
    View *v = first_view(dev);
    if (!v) throw std::runtime_error("RT64_SubmitGather: the device has no view.");
    const size_t mine = (size_t)gather_owned_rows(layout, rank) * (size_t)W * 4;
    if (!(v->packedFinal && dev->gatherTarget == sl.local)) {        // this kind of frame did not write the send buffer itself: pack the owned rows now (same layout)
        struct Enqueued { Device *d; bool sync; explicit Enqueued(Device *dv) : d(dv), sync(dv->opt.syncPresent) { d->opt.syncPresent = false; } ~Enqueued() { d->opt.syncPresent = sync; } };
        size_t got = 0;
        { Enqueued guard(dev); got = mine ? readback(dev, RT64_IMAGE_FINAL_RGBA8, sl.local, slotBytes, true) : 0; }      // the copy is ordered on the stream, not waited for; the option comes back even if the copy throws
        if (got != mine) throw std::runtime_error("RT64_SubmitGather: packing the owned rows failed.");
    }
    HIP_CHECK(hipEventRecord(sl.produced, dev->stream));
    HIP_CHECK(hipStreamWaitEvent(commStream, sl.produced, 0));
    if (count > 1) {
        RcclApi &R = rccl();
        RCCL_CHECK(R.GroupStart());
        if (rank == 0) { for (int r = 1; r < count; r++) { const size_t n = (size_t)gather_owned_rows(layout, r) * (size_t)W * 4; if (n) RCCL_CHECK(R.Recv(sl.bucket + (size_t)r * slotBytes, n, ncclUint8, r, comm, commStream)); } }
        else if (mine) RCCL_CHECK(R.Send(sl.local, mine, ncclUint8, 0, comm, commStream));
        RCCL_CHECK(R.GroupEnd());
    }
    if (rank == 0) HIP_CHECK(launch_gather_assemble(sl.local, sl.bucket, slotBytes, sl.frame, W, layout, commStream));
    HIP_CHECK(hipEventRecord(sl.gathered, commStream));
    sl.pending = true; last = slot; next = slot ^ 1;
    prepare(next);
    return slot;
}
// New boundaries for a gather of cost-balanced bands (RT64_SetGatherBands): every rank calls it with the same boundaries between the same two frames.
void Gather::setBands(const int *starts) {
    if (bands == 0) throw std::runtime_error("RT64_SetGatherBands: a gather of interleaved strips (bands = 0) has no boundaries; create it with bands = 1 or bands = 2.");
    if (!halo_starts_valid(H, count, starts, true)) throw std::runtime_error("RT64_SetGatherBands: starts[0 .. count] must rise strictly from 0 to the frame height (every rank keeps at least one row).");
    dev->enter();
    HIP_CHECK(hipStreamSynchronize(dev->stream)); HIP_CHECK(hipStreamSynchronize(commStream));        // nothing of the old layout is in flight
    for (Slot &sl : slots) sl.pending = false;
    bands = 2; layout.mode = 2;              // (a gather of equal bands becomes one of given boundaries: no whole frame has to be rendered for the first cut)
    for (int r = 0; r <= count; r++) layout.starts[r] = starts[r];
    const size_t need = (size_t)gather_max_owned_rows(layout) * (size_t)W * 4;
    if (need > slotBytes) {
        for (Slot &sl : slots) {
            if (dev->gatherTarget == sl.local) { dev->gatherTarget = nullptr; dev->gatherTargetBytes = 0; }
            HIP_CHECK(hipFree(sl.local)); sl.local = nullptr;
            HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&sl.local), need));
            HIP_CHECK(hipMemsetAsync(sl.local, 0, need, dev->stream));
            if (rank == 0) { HIP_CHECK(hipFree(sl.bucket)); sl.bucket = nullptr; HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&sl.bucket), need * (size_t)count)); }
        }
        slotBytes = need;
        HIP_CHECK(hipStreamSynchronize(dev->stream));
    }
    // Temporal state of the rows this band gains: a band keeps G-buffer, GI accumulation and luminance moments only for its rows and the halo it renders them with
    // (66 rows re-rendered, or halo_margin with the exchange); rows further out hold whatever this device rendered there last -- the first whole frame, or an older
    // cut -- and the next frame would reproject against that.  Their accumulation restarts instead (history length 0: IndirectRayGen.hlsl:43-56 then takes the new
    // sample alone), so a rebalanced partition converges like a freshly cut one; rows inside the old halo keep their history.
    if (View *v = first_view(dev)) {
        const int keep = dev->haloActive() ? std::max(dev->opt.haloMargin, SVGF_INPUT_HALO_ROWS) : (dev->opt.denoiserMode == 1 ? SVGF_HALO_ROWS : GAUSSIAN_HALO_ROWS);
        const int old0 = std::max(0, dev->tileY0 - keep), old1 = std::min(H, dev->tileY1 + keep);
        const int new0 = std::max(0, layout.starts[rank] - keep), new1 = std::min(H, layout.starts[rank + 1] + keep);
        auto restart = [&](int y0, int y1) {
            if (y1 <= y0 || v->imgW != W || v->imgH != H) return;
            for (int k = 0; k < 2; k++) {
                HIP_CHECK(hipMemsetAsync(reinterpret_cast<uint8_t *>(v->img.indirectLight[k]) + (size_t)y0 * W * 8, 0, (size_t)(y1 - y0) * W * 8, dev->stream));
                HIP_CHECK(hipMemsetAsync(reinterpret_cast<uint8_t *>(v->img.moments[k]) + (size_t)y0 * W * 8, 0, (size_t)(y1 - y0) * W * 8, dev->stream));
            }
        };
        restart(new0, std::min(new1, old0)); restart(std::max(new0, old1), new1);
    }
    dev->tileSet = true; dev->tileY0 = layout.starts[rank]; dev->tileY1 = layout.starts[rank + 1]; dev->stripRank = 0; dev->stripCount = 1;
    if (direct.on) { for (bool &p : direct.pending) p = false; prepareDirect(next); }       // (the frame slots are whole frames: new boundaries change nothing about them)
    else prepare(next);
}
void Gather::wait(int slot, bool host) {
    if (direct.on) {
        if (host) HIP_CHECK(hipEventSynchronize(direct.gathered[slot]));
        else for (hipStream_t st : dev->streams) if (st) HIP_CHECK(hipStreamWaitEvent(st, direct.gathered[slot], 0));
        return;
    }
    Slot &sl = slots[slot];
    if (host) HIP_CHECK(hipEventSynchronize(sl.gathered));
    else for (hipStream_t st : dev->streams) if (st) HIP_CHECK(hipStreamWaitEvent(st, sl.gathered, 0));
}

// ---- halo exchange of the SVGF filter input ("renders its rows plus a halo ..., or exchanges halos") -------------------------------------
// Schedule of rank `rank` for bands starts[0 .. count]: band q lies wholly above or below band r, so each pair of bands shares at most one interval
// per direction.  Regions are listed by peer, the send before the receive.  Returns how many there are (writes at most `cap`).
static int halo_plan(int H, int count, const int *starts, int rank, int halo, RT64_HALO_REGION *out, int cap) {
    auto needs = [&](int r, int q, int &a, int &b) {           // rows of band q among the `halo` rows above / below band r
        const int s0 = starts[r], e0 = starts[r + 1], qs = starts[q], qe = starts[q + 1];
        if (e0 <= s0 || qe <= qs) return false;
        a = std::max(std::max(0, s0 - halo), qs); b = std::min(s0, qe);
        if (b > a) return true;
        a = std::max(e0, qs); b = std::min(std::min(H, e0 + halo), qe);
        return b > a;
    };
    int n = 0;
    for (int q = 0; q < count; q++) {
        if (q == rank) continue;
        int a, b;
        if (needs(q, rank, a, b)) { if (n < cap) { RT64_HALO_REGION &g = out[n]; g.peer = q; g.send = 1; g.y0 = a; g.y1 = b; g.host = nullptr; g.bytes = 0; } n++; }
        if (needs(rank, q, a, b)) { if (n < cap) { RT64_HALO_REGION &g = out[n]; g.peer = q; g.send = 0; g.y0 = a; g.y1 = b; g.host = nullptr; g.bytes = 0; } n++; }
    }
    return n;
}
// strict: every band has at least one row -- what a LIVE partition needs (a device with an empty band would fall back to the whole frame while the layout says
// it owns nothing, and its neighbours would wait for rows it never sends); the schedule as a pure function (RT64_HaloPlan) also answers for empty bands.
static bool halo_starts_valid(int H, int count, const int *starts, bool strict) {
    if (!starts || count < 1 || count > RT64_GATHER_MAX_RANKS || starts[0] != 0 || starts[count] != H) return false;
    for (int r = 0; r < count; r++) if (strict ? starts[r + 1] <= starts[r] : starts[r + 1] < starts[r]) return false;
    return true;
}
bool Device::haloActive() const {
    if (halo.fn) return halo.count > 1;
    return opt.haloExchange && halo.gather && halo.gather->count > 1 && halo.gather->bands != 0;
}
// In the middle of a frame, on the render stream `s`: the filter input of this band's rows is in filteredIndirect[0] / svgfGuide; after this call so is
// the input of the RT64_HALO_ROWS rows above and below it (and the received colour rows are in the other ping-pong image too: the a-trous iterations
// skip sky pixels, which the variance kernel writes to both images).
static void halo_exchange(Device *dev, const ViewImages &img, int W, int H, hipStream_t s) {
    Device::HaloLink &h = dev->halo;
    int rank, count; std::vector<int> starts;
    if (h.fn) { rank = h.rank; count = h.count; starts = h.starts; }
    else {
        Gather *g = h.gather; rank = g->rank; count = g->count; starts.resize((size_t)count + 1);
        for (int r = 0; r <= count; r++) starts[(size_t)r] = g->bands == 2 ? g->layout.starts[r] : std::min(r * gather_band_rows(g->H, count), g->H);
    }
    if (!halo_starts_valid(H, count, starts.data()) || dev->tileY0 != starts[(size_t)rank] || dev->tileY1 != starts[(size_t)rank + 1])
        throw std::runtime_error("RT64_DrawDevice: the halo exchange's band layout does not match the device's rows (RT64_SetDeviceTile / frame size changed after it was set up).");
    RT64_HALO_REGION regs[2 * RT64_GATHER_MAX_RANKS];
    const int n = halo_plan(H, count, starts.data(), rank, SVGF_ATROUS_HALO_ROWS, regs, 2 * RT64_GATHER_MAX_RANKS);
    uint8_t *colour = reinterpret_cast<uint8_t *>(img.filteredIndirect[0]), *colourOther = reinterpret_cast<uint8_t *>(img.filteredIndirect[1]), *guide = reinterpret_cast<uint8_t *>(img.svgfGuide);
    const size_t rowC = (size_t)W * 8, rowG = (size_t)W * 16;
    if (dev->opt.haloDryRun) return;         // timing aid (tools/band_costs.py): the frame of an exchanging band without the transfer itself -- the halo rows keep whatever they held
    if (h.fn) {
        size_t total = 0;
        for (int k = 0; k < n; k++) { regs[k].bytes = (size_t)(regs[k].y1 - regs[k].y0) * (rowC + rowG); total += regs[k].bytes; }
        if (total > h.pinnedBytes) {
            if (h.pinned) hipHostFree(h.pinned);
            h.pinnedBytes = total; HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h.pinned), h.pinnedBytes, hipHostMallocDefault));
        }
        size_t at = 0;
        for (int k = 0; k < n; k++) {
            RT64_HALO_REGION &g = regs[k]; g.host = h.pinned + at; at += g.bytes;
            if (!g.send) continue;
            const size_t rows = (size_t)(g.y1 - g.y0);
            HIP_CHECK(hipMemcpyAsync(g.host, colour + (size_t)g.y0 * rowC, rows * rowC, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t *>(g.host) + rows * rowC, guide + (size_t)g.y0 * rowG, rows * rowG, hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
        h.fn(h.user, regs, n);
        for (int k = 0; k < n; k++) {
            const RT64_HALO_REGION &g = regs[k];
            if (g.send) continue;
            const size_t rows = (size_t)(g.y1 - g.y0);
            HIP_CHECK(hipMemcpyAsync(colour + (size_t)g.y0 * rowC, g.host, rows * rowC, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(guide + (size_t)g.y0 * rowG, static_cast<const uint8_t *>(g.host) + rows * rowC, rows * rowG, hipMemcpyHostToDevice, s));
        }
    }
    else {
        // RCCL: one group of point-to-point transfers between the images themselves (rows are contiguous), on the gather's communication stream --
        // behind the previous frame's gather, in the same order on every rank -- and the render stream waits for the group.
        Gather *g = h.gather; RcclApi &R = rccl();
        if (!h.ready) { HIP_CHECK(hipEventCreateWithFlags(&h.ready, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&h.done, hipEventDisableTiming)); }
        HIP_CHECK(hipEventRecord(h.ready, s));
        HIP_CHECK(hipStreamWaitEvent(g->commStream, h.ready, 0));
        RCCL_CHECK(R.GroupStart());
        for (int k = 0; k < n; k++) {
            const RT64_HALO_REGION &q = regs[k];
            const size_t rows = (size_t)(q.y1 - q.y0);
            if (q.send) {
                RCCL_CHECK(R.Send(colour + (size_t)q.y0 * rowC, rows * rowC, ncclUint8, q.peer, g->comm, g->commStream));
                RCCL_CHECK(R.Send(guide + (size_t)q.y0 * rowG, rows * rowG, ncclUint8, q.peer, g->comm, g->commStream));
            }
            else {
                RCCL_CHECK(R.Recv(colour + (size_t)q.y0 * rowC, rows * rowC, ncclUint8, q.peer, g->comm, g->commStream));
                RCCL_CHECK(R.Recv(guide + (size_t)q.y0 * rowG, rows * rowG, ncclUint8, q.peer, g->comm, g->commStream));
            }
        }
        RCCL_CHECK(R.GroupEnd());
        HIP_CHECK(hipEventRecord(h.done, g->commStream));
        HIP_CHECK(hipStreamWaitEvent(s, h.done, 0));
    }
    for (int k = 0; k < n; k++) {
        const RT64_HALO_REGION &g = regs[k];
        if (!g.send) HIP_CHECK(hipMemcpyAsync(colourOther + (size_t)g.y0 * rowC, colour + (size_t)g.y0 * rowC, (size_t)(g.y1 - g.y0) * rowC, hipMemcpyDeviceToDevice, s));
    }
}

}  // namespace rt64

RT64_EXPORT int RT64_GetGatherUniqueId(void *id, size_t idBytes) {
    RT64_TRY
    if (!id || idBytes < sizeof(ncclUniqueId)) throw std::runtime_error("RT64_GetGatherUniqueId: the id buffer needs RT64_GATHER_ID_BYTES bytes.");
    ncclUniqueId u; RCCL_CHECK(rccl().GetUniqueId(&u)); memcpy(id, &u, sizeof(u)); return 1;
    RT64_CATCH(0)
}
RT64_EXPORT RT64_GATHER *RT64_CreateGather(RT64_DEVICE *device, const void *id, size_t idBytes, int rank, int count, int bands) {
    RT64_TRY
    if (!device || !id || idBytes < sizeof(ncclUniqueId)) throw std::runtime_error("RT64_CreateGather: NULL device or id.");
    ncclUniqueId u; memcpy(&u, id, sizeof(u));
    return reinterpret_cast<RT64_GATHER *>(new Gather(reinterpret_cast<Device *>(device), u, rank, count, bands));
    RT64_CATCH(nullptr)
}
RT64_EXPORT int RT64_SubmitGather(RT64_GATHER *gather) {
    RT64_TRY if (!gather) throw std::runtime_error("RT64_SubmitGather: NULL gather."); return reinterpret_cast<Gather *>(gather)->submit(); RT64_CATCH(-1)
}
RT64_EXPORT size_t RT64_ReadbackGather(RT64_GATHER *gather, int slot, void *dst, size_t dstBytes, int toDevice) {
    RT64_TRY
    Gather *g = reinterpret_cast<Gather *>(gather);
    if (!g) throw std::runtime_error("RT64_ReadbackGather: NULL gather.");
    if (slot < 0) slot = g->last;
    const int slotCount = g->direct.on ? (int)Gather::DIRECT_SLOTS : 2;
    if (slot < 0 || slot >= slotCount || (g->direct.on && g->direct.frames == 0)) throw std::runtime_error("RT64_ReadbackGather: no frame has been submitted.");
    g->dev->use();
    g->wait(slot, true);                                         // every rank: its part of the exchange has completed
    if (g->rank != 0) return 0;
    const size_t need = (size_t)g->W * g->H * 4;
    if (!dst) return need;
    if (dstBytes < need) throw std::runtime_error("RT64_ReadbackGather: destination buffer is too small.");
    const uint8_t *src = g->direct.on ? g->direct.mem + (size_t)slot * need : g->slots[slot].frame;
    HIP_CHECK(hipMemcpy(dst, src, need, toDevice ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return need;
    RT64_CATCH(0)
}
RT64_EXPORT void *RT64_GetGatherFrame(RT64_GATHER *gather, int slot) {           // rank 0: device pointer of the assembled RGBA8 frame of `slot` (valid once its gather has run)
    Gather *g = reinterpret_cast<Gather *>(gather);
    if (!g || g->rank != 0) return nullptr;
    if (slot < 0) slot = g->last;
    if (g->direct.on) return (slot >= 0 && slot < (int)Gather::DIRECT_SLOTS) ? g->direct.mem + (size_t)slot * (size_t)g->W * g->H * 4 : nullptr;
    return (slot == 0 || slot == 1) ? g->slots[slot].frame : nullptr;
}
// Direct mode of a gather (see Gather::Direct): rank 0 exports its frame slots, every rank switches over between the same two frames.
RT64_EXPORT size_t RT64_GetGatherDirectHandle(RT64_GATHER *gather, void *handle, size_t handleBytes) {
    RT64_TRY Gather *g = reinterpret_cast<Gather *>(gather); if (!g) throw std::runtime_error("RT64_GetGatherDirectHandle: NULL gather."); return g->directHandle(handle, handleBytes); RT64_CATCH(0)
}
RT64_EXPORT int RT64_SetGatherDirect(RT64_GATHER *gather, const void *handle, size_t handleBytes, int enable) {
    RT64_TRY Gather *g = reinterpret_cast<Gather *>(gather); if (!g) throw std::runtime_error("RT64_SetGatherDirect: NULL gather."); g->setDirect(handle, handleBytes, enable); return 1; RT64_CATCH(0)
}
RT64_EXPORT void RT64_DestroyGather(RT64_GATHER *gather) { RT64_TRY delete reinterpret_cast<Gather *>(gather); RT64_CATCH_VOID }
// Partition layout, as pure functions (no device needed): which rank owns frame row y and where the row sits in that rank's packed buffer.
// bands = 0 (interleaved strips) and 1 (equal bands) are functions of (height, count) alone.  bands = 2 (cost-balanced) is not: its boundaries are
// cut from a frame, so these three answer -1 for it and the *Of forms below take the boundaries (RT64_GetGatherBands / RT64_BalanceGatherBands).
RT64_EXPORT int RT64_GatherRowOwner(int height, int count, int bands, int y, int *packedRow) {
    if (bands < 0 || bands > 1 || height < 1 || count < 1 || y < 0 || y >= height) return -1;
    int p = 0; const int r = gather_row_owner(gather_layout(height, count, bands), y, &p); if (packedRow) *packedRow = p; return r;
}
RT64_EXPORT int RT64_GatherOwnedRows(int height, int count, int bands, int rank) { return (bands < 0 || bands > 1 || height < 1 || count < 1 || rank < 0 || rank >= count) ? -1 : gather_owned_rows(gather_layout(height, count, bands), rank); }
RT64_EXPORT int RT64_GatherSlotRows(int height, int count, int bands) { return (bands < 0 || bands > 1 || height < 1 || count < 1) ? -1 : gather_max_owned_rows(gather_layout(height, count, bands)); }
// The same three for a layout given by its band boundaries starts[0 .. count] (cost-balanced bands; also valid for equal bands).
static bool gather_layout_of(int height, int count, const int *starts, GatherLayout &L) {
    if (!starts || height < 1 || count < 1 || count > RT64_GATHER_MAX_RANKS || starts[0] != 0 || starts[count] != height) return false;
    L = gather_layout(height, count, 2);
    for (int r = 0; r <= count; r++) { if (r && starts[r] < starts[r - 1]) return false; L.starts[r] = starts[r]; }
    return true;
}
RT64_EXPORT int RT64_GatherRowOwnerOf(int height, int count, const int *starts, int y, int *packedRow) {
    GatherLayout L; if (!gather_layout_of(height, count, starts, L) || y < 0 || y >= height) return -1;
    int p = 0; const int r = gather_row_owner(L, y, &p); if (packedRow) *packedRow = p; return r;
}
RT64_EXPORT int RT64_GatherOwnedRowsOf(int height, int count, const int *starts, int rank) { GatherLayout L; return (!gather_layout_of(height, count, starts, L) || rank < 0 || rank >= count) ? -1 : gather_owned_rows(L, rank); }
RT64_EXPORT int RT64_GatherSlotRowsOf(int height, int count, const int *starts) { GatherLayout L; return !gather_layout_of(height, count, starts, L) ? -1 : gather_max_owned_rows(L); }
RT64_EXPORT int RT64_HaloPlan(int height, int count, const int *starts, int rank, int haloRows, RT64_HALO_REGION *regions, int capacity) {
    if (!halo_starts_valid(height, count, starts) || rank < 0 || rank >= count || haloRows < 0 || (capacity > 0 && !regions)) return -1;
    return halo_plan(height, count, starts, rank, haloRows, regions, capacity < 0 ? 0 : capacity);
}
RT64_EXPORT int RT64_SetDeviceHaloExchange(RT64_DEVICE *device, RT64_HALO_EXCHANGE exchange, void *user, const int *starts, int rank, int count) {
    Device *d = reinterpret_cast<Device *>(device); if (!d) return 0;
    if (!exchange) { d->halo.fn = nullptr; d->halo.user = nullptr; d->halo.count = 0; d->halo.starts.clear(); return 1; }
    if (!halo_starts_valid(d->pendingHeight, count, starts, true) || rank < 0 || rank >= count) return 0;       // (strictly rising: every band has a row)
    d->halo.fn = exchange; d->halo.user = user; d->halo.rank = rank; d->halo.count = count; d->halo.starts.assign(starts, starts + count + 1);
    return 1;
}
// The band boundaries of an existing gather (bands = 1 or 2): starts[0 .. count], rank r owns rows [starts[r], starts[r + 1]).  Returns count, 0 for interleaved strips.
RT64_EXPORT int RT64_GetGatherBands(RT64_GATHER *gather, int *starts, int capacity) {
    Gather *g = reinterpret_cast<Gather *>(gather);
    if (!g || !starts || g->bands == 0 || capacity < g->count + 1) return 0;
    for (int r = 0; r <= g->count; r++) starts[r] = g->bands == 2 ? g->layout.starts[r] : std::min(r * gather_band_rows(g->H, g->count), g->H);
    return g->count;
}
// The same cut as a pure function: boundaries of `count` cost-balanced bands from per-row hit counts (what RT64_CreateGather(bands = 2) computes from its last frame).
RT64_EXPORT int RT64_RebalanceGatherBands(int height, int count, const int *starts, const float *msPerRank, int *newStarts) {
    if (!halo_starts_valid(height, count, starts) || !msPerRank || !newStarts) return 0;
    gather_rebalanced_bands(height, count, starts, msPerRank, newStarts);
    return 1;
}
RT64_EXPORT int RT64_SetGatherBands(RT64_GATHER *gather, const int *starts) {
    RT64_TRY Gather *g = reinterpret_cast<Gather *>(gather); if (!g || !starts) return 0; g->setBands(starts); return 1; RT64_CATCH(0)
}
RT64_EXPORT void RT64_BalanceGatherBands(const unsigned int *hitCounts, int width, int height, int count, int *starts) {
    if (hitCounts && starts && width > 0 && height > 0 && count >= 1 && count <= RT64_GATHER_MAX_RANKS) gather_balanced_bands(hitCounts, width, height, count, starts);
}

// ---- inspector (rt64_inspector.cpp:469-515): the ImGui/Im3d debug UI is Win32-only; the exports exist so that hosts resolve all 33 symbols ----
struct InspectorStub { Device *device; };
RT64_EXPORT RT64_INSPECTOR *RT64_CreateInspector(RT64_DEVICE *devicePtr) { return reinterpret_cast<RT64_INSPECTOR *>(new InspectorStub{ reinterpret_cast<Device *>(devicePtr) }); }
RT64_EXPORT bool RT64_HandleMessageInspector(RT64_INSPECTOR *, RT64_UINT, RT64_WPARAM, RT64_LPARAM) { return false; }
RT64_EXPORT void RT64_PrintClearInspector(RT64_INSPECTOR *) {}
RT64_EXPORT void RT64_PrintMessageInspector(RT64_INSPECTOR *, const char *message) { if (message) fprintf(stdout, "%s\n", message); }
RT64_EXPORT void RT64_SetSceneInspector(RT64_INSPECTOR *, RT64_SCENE_DESC *) {}
RT64_EXPORT void RT64_SetMaterialInspector(RT64_INSPECTOR *, RT64_MATERIAL *, const char *) {}
RT64_EXPORT void RT64_SetLightsInspector(RT64_INSPECTOR *, RT64_LIGHT *, int *, int) {}
RT64_EXPORT void RT64_DestroyInspector(RT64_INSPECTOR *inspectorPtr) { delete reinterpret_cast<InspectorStub *>(inspectorPtr); }
