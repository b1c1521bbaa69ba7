// lbvh.hip -- LBVH construction / refit kernels (BLAS per mesh, TLAS per frame).
//
// Replaces the DXR driver's BuildRaytracingAccelerationStructure calls the reference issues from
//   rt64_mesh.cpp:114-158 (BottomLevelASGenerator.cpp:245) and rt64_view.cpp:412-452 (TopLevelASGenerator.cpp:244).
// Algorithm: Karras 2012 LBVH -- 30-bit Morton codes of the leaf-box centres, stable radix sort, binary radix tree by
// longest common prefix of (code << 32 | leaf), bottom-up box fit.  The arithmetic follows the "Geometry spec" in
// DESIGN.md operation by operation, so the node and triangle arrays are reproducible bit for bit.
//
// Small trees (n <= LBVH_SMALL_MAX leaves): ONE workgroup of 1024 threads does the whole build with the sort keys, the
// parent links and the fit counters staged in LDS (1-bit split radix passes with wave-ballot-free prefix sums); only
// the final GpuNode / GpuTri arrays touch HBM.  One launch per RT64_SetMesh and one per frame for the TLAS.
// Large trees: multi-kernel path (global Morton pass, multi-block LSD radix sort, Karras pass, level-synchronous fit).
#include <algorithm>
#include "kernels.h"
#include "device_math.h"

#define LBVH_THREADS 1024
#define LBVH_COUNTING_MAX 1024u

namespace {

DEV uint32_t float_to_ordered(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
DEV float ordered_to_float(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

DEV uint32_t morton30(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t v[3] = { x & 1023u, y & 1023u, z & 1023u };
    uint32_t code = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t t = v[k];
        t = (t | (t << 16)) & 0x030000FFu;
        t = (t | (t << 8)) & 0x0300F00Fu;
        t = (t | (t << 4)) & 0x030C30C3u;
        t = (t | (t << 2)) & 0x09249249u;
        code |= t << k;
    }
    return code;
}

struct Box { float mn[3], mx[3]; };

DEV void load_positions(const LbvhArgs &a, uint32_t prim, float v[3][3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t idx = a.indices[3 * prim + k];
        const float *p = reinterpret_cast<const float *>(a.vertices + (size_t)idx * a.vertexStride);   // position = first 12 bytes (rt64_shader.cpp:88)
        v[k][0] = p[0]; v[k][1] = p[1]; v[k][2] = p[2];
    }
}

// G1: box of leaf i (leaf-index order).
DEV Box leaf_box(const LbvhArgs &a, uint32_t i) {
    Box b;
    if (a.mode == LBVH_MODE_TRIANGLES) {
        float v[3][3]; load_positions(a, i, v);
#pragma unroll
        for (int k = 0; k < 3; k++) { b.mn[k] = fminf(fminf(v[0][k], v[1][k]), v[2][k]); b.mx[k] = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]); }
    }
    else {   // LBVH_MODE_INSTANCES: the 8 corners of the mesh box through objectToWorld
        const GpuInstance &in = a.instances[i];
        const BlasHeader h = *in.header;
#pragma unroll
        for (int k = 0; k < 3; k++) { b.mn[k] = INFINITY; b.mx[k] = -INFINITY; }
        for (int c = 0; c < 8; c++) {
            float p[3] = { (c & 1) ? h.bmax[0] : h.bmin[0], (c & 2) ? h.bmax[1] : h.bmin[1], (c & 4) ? h.bmax[2] : h.bmin[2] }, w[3];
            g_xform_point(in.objectToWorld, p, w);
#pragma unroll
            for (int k = 0; k < 3; k++) { b.mn[k] = fminf(b.mn[k], w[k]); b.mx[k] = fmaxf(b.mx[k], w[k]); }
        }
    }
    return b;
}

DEV uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
    return v;
}

DEV int delta64(const uint32_t *key, const uint32_t *val, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long a = ((unsigned long long)key[i] << 32) | val[i], b = ((unsigned long long)key[j] << 32) | val[j];
    return __clzll((long long)(a ^ b));
}

// LDS carve (uint32 words): keyA[n] valA[n] keyB[n] valB[n] parentLeaf[n] parentNode[n] counters[n] scratch[64]
DEV void lbvh_small_body(const LbvhArgs &a, uint32_t *lds) {
    const uint32_t n = a.n, tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t *keyA = lds, *valA = lds + n, *keyB = lds + 2 * n, *valB = lds + 3 * n;
    uint32_t *parentLeaf = lds + 4 * n, *parentNode = lds + 5 * n, *counters = lds + 6 * n, *scratch = lds + 7 * n;
    // scratch[0..5] ordered bounds, [8..23] wave totals, [24] total, [25] flag

    if (!a.refit) {
        // ---- G2: scene box ------------------------------------------------------------------------------------
        if (tid < 6) scratch[tid] = tid < 3 ? 0xFFFFFFFFu : 0u;
        __syncthreads();
        {
            uint32_t mn[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, mx[3] = { 0, 0, 0 };
            for (uint32_t i = tid; i < n; i += T) {
                Box b = leaf_box(a, i);
#pragma unroll
                for (int k = 0; k < 3; k++) { mn[k] = min(mn[k], float_to_ordered(b.mn[k])); mx[k] = max(mx[k], float_to_ordered(b.mx[k])); }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { atomicMin(&scratch[k], mn[k]); atomicMax(&scratch[3 + k], mx[k]); }
        }
        __syncthreads();
        float bmin[3], scale[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            bmin[k] = ordered_to_float(scratch[k]);
            float ext = ordered_to_float(scratch[3 + k]) - bmin[k];
            scale[k] = ext > 0.0f ? 1024.0f / ext : 0.0f;
        }
        // ---- Morton keys ---------------------------------------------------------------------------------------
        for (uint32_t i = tid; i < n; i += T) {
            Box b = leaf_box(a, i);
            uint32_t q[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float c = (b.mn[k] + b.mx[k]) * 0.5f;
                float f = (c - bmin[k]) * scale[k];
                int qi = (int)f;
                q[k] = (uint32_t)(qi < 0 ? 0 : (qi > 1023 ? 1023 : qi));
            }
            keyA[i] = morton30(q[0], q[1], q[2]);
            valA[i] = i;
        }
        __syncthreads();
        // ---- G3: sort by (code, leaf) ----------------------------------------------------------------------------
        // Small inputs: rank by counting.  Every lane reads the same keyA[j] (LDS broadcast), n*n/T compares per thread,
        // no barriers inside -- 1-2 us for the few hundred leaves of a game mesh or the instances of a frame.
        if (n <= LBVH_COUNTING_MAX) {
            for (uint32_t i = tid; i < n; i += T) {
                const uint32_t k = keyA[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < n; j++) { const uint32_t kj = keyA[j]; rank += (kj < k || (kj == k && j < i)) ? 1u : 0u; }
                keyB[rank] = k; valB[rank] = i;
            }
            __syncthreads();
            for (uint32_t i = tid; i < n; i += T) { keyA[i] = keyB[i]; valA[i] = valB[i]; }
            __syncthreads();
        }
        else {
        // Larger inputs: stable LSD radix sort on the 30 code bits, one bit per pass, staged in LDS.
        // Blocked arrangement: thread t owns items [t*ipt, (t+1)*ipt), so rank order == index order (stable).
        const uint32_t ipt = (n + T - 1) / T;
        const uint32_t first = min(tid * ipt, n), last = min(first + ipt, n);
        uint32_t *ksrc = keyA, *vsrc = valA, *kdst = keyB, *vdst = valB;
        for (int bit = 0; bit < 30; bit++) {
            uint32_t c0 = 0;
            for (uint32_t i = first; i < last; i++) c0 += ((ksrc[i] >> bit) & 1u) ^ 1u;
            uint32_t incl = wave_inclusive_scan(c0, lane);
            if (lane == 63) scratch[8 + wave] = incl;
            __syncthreads();
            if (wave == 0) {
                uint32_t v = lane < (T / 64) ? scratch[8 + lane] : 0u;
                uint32_t s = wave_inclusive_scan(v, lane);
                if (lane < (T / 64)) scratch[8 + lane] = s - v;
                if (lane == 63) scratch[24] = s;
            }
            __syncthreads();
            const uint32_t zeros = scratch[24];
            if (zeros != 0 && zeros != n) {
                uint32_t r0 = scratch[8 + wave] + incl - c0;
                uint32_t r1 = zeros + (first - r0);
                for (uint32_t i = first; i < last; i++) {
                    uint32_t k = ksrc[i], v = vsrc[i];
                    uint32_t pos = ((k >> bit) & 1u) ? r1++ : r0++;
                    kdst[pos] = k; vdst[pos] = v;
                }
                uint32_t *t = ksrc; ksrc = kdst; kdst = t; t = vsrc; vsrc = vdst; vdst = t;
            }
            __syncthreads();
        }
        if (ksrc != keyA) {   // uniform: every thread took the same branches
            for (uint32_t i = tid; i < n; i += T) { keyA[i] = keyB[i]; valA[i] = valB[i]; }
            __syncthreads();
        }
        }
        for (uint32_t s = tid; s < n; s += T) { a.sortedIndex[s] = valA[s]; a.morton[s] = keyA[s]; }
        // ---- G4: Karras radix tree -------------------------------------------------------------------------------
        if (n == 1) {
            if (tid == 0) { a.nodes[0].left = RT64_LEAF_BIT; a.nodes[0].right = RT64_NO_CHILD; a.nodes[0].parent = RT64_NO_CHILD; a.nodes[0].pad = 0; parentLeaf[0] = 0; a.leafParent[0] = 0; }
        }
        else {
            if (tid == 0) { parentNode[0] = RT64_NO_CHILD; a.nodes[0].parent = RT64_NO_CHILD; }
            const int N = (int)n;
            for (int i = (int)tid; i < N - 1; i += T) {
                int d = (delta64(keyA, valA, N, i, i + 1) - delta64(keyA, valA, N, i, i - 1)) >= 0 ? 1 : -1;
                int dmin = delta64(keyA, valA, N, i, i - d);
                int lmax = 2;
                while (delta64(keyA, valA, N, i, i + lmax * d) > dmin) lmax *= 2;
                int l = 0;
                for (int t = lmax / 2; t >= 1; t /= 2)
                    if (delta64(keyA, valA, N, i, i + (l + t) * d) > dmin) l += t;
                int j = i + l * d;
                int dnode = delta64(keyA, valA, N, i, j);
                int s = 0;
                for (int div = 2;; div *= 2) {
                    int t = (l + div - 1) / div;
                    if (delta64(keyA, valA, N, i, i + (s + t) * d) > dnode) s += t;
                    if (t <= 1) break;
                }
                int g = i + s * d + (d < 0 ? -1 : 0);
                int lo = i < j ? i : j, hi = i < j ? j : i;
                uint32_t left, right;
                if (lo == g) { left = RT64_LEAF_BIT | (uint32_t)g; parentLeaf[g] = (uint32_t)i; a.leafParent[g] = (uint32_t)i; }
                else { left = (uint32_t)g; parentNode[g] = (uint32_t)i; a.nodes[g].parent = (uint32_t)i; }
                if (hi == g + 1) { right = RT64_LEAF_BIT | (uint32_t)(g + 1); parentLeaf[g + 1] = (uint32_t)i; a.leafParent[g + 1] = (uint32_t)i; }
                else { right = (uint32_t)(g + 1); parentNode[g + 1] = (uint32_t)i; a.nodes[g + 1].parent = (uint32_t)i; }
                a.nodes[i].left = left; a.nodes[i].right = right; a.nodes[i].pad = (uint32_t)j;      // the other end of the leaf range (G4)
            }
        }
    }
    else {
        // G6 refit: topology from the previous build (written by an earlier launch).
        for (uint32_t s = tid; s < n; s += T) parentLeaf[s] = a.leafParent[s];
        for (uint32_t i = tid; i + 1 < n; i += T) parentNode[i] = a.nodes[i].parent;
    }
    for (uint32_t i = tid; i < n; i += T) counters[i] = 0;
    __syncthreads();   // also orders this workgroup's global child-link stores before the loads below (one CU, shared L1)

    // ---- G5: leaves in Morton order + bottom-up box fit ------------------------------------------------------------
    for (uint32_t s = tid; s < n; s += T) {
        const uint32_t leaf = a.sortedIndex[s];
        Box b;
        if (a.mode == LBVH_MODE_TRIANGLES) {
            float v[3][3]; load_positions(a, leaf, v);
            GpuTri t;
#pragma unroll
            for (int k = 0; k < 3; k++) { t.v0[k] = v[0][k]; t.v1[k] = v[1][k]; t.v2[k] = v[2][k]; b.mn[k] = fminf(fminf(v[0][k], v[1][k]), v[2][k]); b.mx[k] = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]); }
            t.prim = leaf; t.pad1 = 0; t.pad2 = 0;
            a.tris[s] = t;
        }
        else b = leaf_box(a, leaf);

        if (n == 1) {
            GpuNode &nd = a.nodes[0];
#pragma unroll
            for (int k = 0; k < 3; k++) { nd.lmin[k] = b.mn[k]; nd.lmax[k] = b.mx[k]; nd.rmin[k] = INFINITY; nd.rmax[k] = -INFINITY; a.header->bmin[k] = b.mn[k]; a.header->bmax[k] = b.mx[k]; }
            a.header->count = 1; a.header->depth = 1;
            break;
        }
        uint32_t child = RT64_LEAF_BIT | s, p = parentLeaf[s];
        for (;;) {
            GpuNode &nd = a.nodes[p];
            const bool isLeft = nd.left == child;
            float *dmn = isLeft ? nd.lmin : nd.rmin, *dmx = isLeft ? nd.lmax : nd.rmax;
#pragma unroll
            for (int k = 0; k < 3; k++) { dmn[k] = b.mn[k]; dmx[k] = b.mx[k]; }
            __threadfence_block();
            uint32_t arrived = atomicAdd(&counters[p], 1u);
            if (arrived == 0) break;                        // the sibling subtree finishes this node
            __threadfence_block();
            const float *smn = isLeft ? nd.rmin : nd.lmin, *smx = isLeft ? nd.rmax : nd.lmax;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float omn = *(volatile const float *)&smn[k], omx = *(volatile const float *)&smx[k];
                b.mn[k] = fminf(b.mn[k], omn); b.mx[k] = fmaxf(b.mx[k], omx);
            }
            if (p == 0) {
#pragma unroll
                for (int k = 0; k < 3; k++) { a.header->bmin[k] = b.mn[k]; a.header->bmax[k] = b.mx[k]; }
                a.header->count = n;
                break;
            }
            child = p; p = parentNode[p];
        }
    }
    // tree depth (inner nodes on the longest root-to-leaf path): the host sizes the traversal stack of the LDS-cached kernels by it
    if (n > 1) {
        if (tid == 0) scratch[26] = 0;
        __syncthreads();
        uint32_t deepest = 0;
        for (uint32_t s2 = tid; s2 < n; s2 += T) {
            uint32_t d = 1, p = parentLeaf[s2];
            while (p != 0) { p = parentNode[p]; d++; }
            deepest = max(deepest, d);
        }
        atomicMax(&scratch[26], deepest);
        __syncthreads();
        if (tid == 0) a.header->depth = scratch[26];
    }
}

__global__ __launch_bounds__(LBVH_THREADS) void lbvh_small_kernel(LbvhArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    lbvh_small_body(a, lds);
}

// One workgroup per tree: the meshes a host updated between two frames (RT64_SetMesh records them, RT64_DrawDevice builds them
// together -- the reference executes its recorded uploads and BLAS builds at the next frame too, rt64_device.cpp:979-983).
__global__ __launch_bounds__(LBVH_THREADS) void lbvh_small_batch_kernel(const LbvhArgs *args) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const LbvhArgs a = args[blockIdx.x];
    lbvh_small_body(a, lds);
}

}  // namespace

size_t lbvh_small_lds_bytes(uint32_t n) { return ((size_t)7 * n + 64) * sizeof(uint32_t); }

// More than 64 KB of dynamic LDS has to be allowed per kernel AND per device: remembered per (kernel, current device), so a second
// Device on another GPU of the same process gets the attribute too.
static hipError_t allow_big_lds(const void *kernel, int which) {
    static unsigned long long done[2] = { 0ull, 0ull };
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && (done[which] >> dev) & 1ull) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lbvh_small_lds_bytes(LBVH_SMALL_MAX));
    if (e == hipSuccess && dev >= 0 && dev < 64) done[which] |= 1ull << dev;
    return e;
}

hipError_t lbvh_launch(const LbvhArgs &args, hipStream_t stream) {
    if (args.n == 0) return hipErrorInvalidValue;
    if (args.n <= LBVH_SMALL_MAX) {
        size_t lds = lbvh_small_lds_bytes(args.n);
        hipError_t e = allow_big_lds(reinterpret_cast<const void *>(lbvh_small_kernel), 0);
        if (e != hipSuccess) return e;
        const uint32_t threads = args.n >= LBVH_THREADS ? LBVH_THREADS : ((args.n + 63u) / 64u) * 64u;
        hipLaunchKernelGGL(lbvh_small_kernel, dim3(1), dim3(threads), lds, stream, args);
        return hipGetLastError();
    }
    return lbvh_launch_large(args, stream);
}

// `count` trees of at most `maxN` <= LBVH_SMALL_MAX leaves each, argument blocks in device memory.
hipError_t lbvh_launch_batch(const LbvhArgs *deviceArgs, uint32_t count, uint32_t maxN, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (maxN == 0 || maxN > LBVH_SMALL_MAX) return hipErrorInvalidValue;
    hipError_t e = allow_big_lds(reinterpret_cast<const void *>(lbvh_small_batch_kernel), 1);
    if (e != hipSuccess) return e;
    const uint32_t threads = maxN >= LBVH_THREADS ? LBVH_THREADS : ((maxN + 63u) / 64u) * 64u;
    hipLaunchKernelGGL(lbvh_small_batch_kernel, dim3(count), dim3(threads), lbvh_small_lds_bytes(maxN), stream, deviceArgs);
    return hipGetLastError();
}

// ---- large trees (n > LBVH_SMALL_MAX) -----------------------------------------------------------------------------------------
// Same Geometry spec (G1-G6), spread over the whole chip:
//   bounds (ordered-uint atomics) -> Morton keys -> 4 x 8-bit stable LSD radix passes {histogram, scan, scatter} -> Karras
//   nodes -> leaves in Morton order -> bottom-up box fit in two launches (chunk-local nodes by one workgroup per chunk, the nodes that
//   span chunks by a single workgroup: no fence ever has to cross an XCD, see lg_fit_local_kernel).
// Scratch (caller-allocated, lbvh_large_scratch_bytes): keys/vals x2, histogram, bounds, leaf boxes, node boxes, done stamps.
namespace {

#define LG_THREADS 256
#define LG_TILE 2048            // keys per workgroup in the radix passes
#define LG_CHUNK 2048           // leaves (in Morton order) per workgroup of the box fit
#define LG_ROOT_SLOTS 192       // list entries per chunk: the maximal subtrees inside an interval of leaves number at most 2 x depth (depth <= 62 key bits + 1)
// Why a list can never overflow: the maximal subtrees of a radix tree that lie inside one interval of leaves hang off the two root-to-leaf
// paths that bound the interval, at most one per level and side, and a path of the tree over LG_KEY_BITS-bit unique keys has at most
// LG_KEY_BITS inner nodes.  A wider key needs more slots: the assertion keeps the two numbers together (the `k < LG_ROOT_SLOTS` tests in the
// fit kernels only keep a wrong bound inside the arrays).
#define LG_KEY_BITS 62           // 30-bit Morton code << 32 | 32-bit leaf number: bits 61..0 can differ
static_assert(LG_ROOT_SLOTS >= 2 * (LG_KEY_BITS + 1), "LG_ROOT_SLOTS must hold 2 x (key bits + 1) roots per chunk");

struct LargeScratch {
    uint32_t *keyA, *valA, *keyB, *valB, *hist, *bounds, *done, *rootList, *rootCount;
    float *leafBox, *nodeBox;   // [n][6]
    uint32_t blocks;
};

#define LG_FAN_LISTS 64          // = LG_FAN (defined with the fit kernels): lists one workgroup of a level takes from the level below
__host__ __device__ inline size_t lg_lists_all_levels(uint32_t n) {      // root lists of every level of the fit, back to back: chunks, groups of chunks, ... , 1 (+ the unused list of the last level)
    size_t total = 0, lists = ((size_t)n + LG_CHUNK - 1) / LG_CHUNK;
    for (;;) { total += lists; if (lists == 1) break; lists = (lists + LG_FAN_LISTS - 1) / LG_FAN_LISTS; }
    return total + 1;
}
__host__ __device__ inline LargeScratch carve(void *base, uint32_t n) {
    LargeScratch L;
    const uint32_t blocks = (n + LG_TILE - 1) / LG_TILE;
    uint8_t *p = static_cast<uint8_t *>(base);
    auto take = [&](size_t bytes) { uint8_t *r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    L.keyA = reinterpret_cast<uint32_t *>(take((size_t)n * 4)); L.valA = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
    L.keyB = reinterpret_cast<uint32_t *>(take((size_t)n * 4)); L.valB = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
    L.hist = reinterpret_cast<uint32_t *>(take(((size_t)blocks + 1) * 256 * 4));      // [digit][block] counts, then the 256 row totals
    L.bounds = reinterpret_cast<uint32_t *>(take(64));
    L.done = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
    L.leafBox = reinterpret_cast<float *>(take((size_t)n * 24)); L.nodeBox = reinterpret_cast<float *>(take((size_t)n * 24));
    const size_t listsAllLevels = lg_lists_all_levels(n);
    L.rootList = reinterpret_cast<uint32_t *>(take(listsAllLevels * LG_ROOT_SLOTS * 4)); L.rootCount = reinterpret_cast<uint32_t *>(take(listsAllLevels * 4));
    L.blocks = blocks;
    return L;
}

__global__ void lg_init_bounds(uint32_t *bounds) { if (threadIdx.x < 6) bounds[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u; }

__global__ __launch_bounds__(LG_THREADS) void lg_bounds_kernel(LbvhArgs a, uint32_t *bounds) {
    __shared__ uint32_t sb[6];
    if (threadIdx.x < 6) sb[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    uint32_t mn[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, mx[3] = { 0, 0, 0 };
    for (uint32_t i = blockIdx.x * LG_THREADS + threadIdx.x; i < a.n; i += gridDim.x * LG_THREADS) {
        Box b = leaf_box(a, i);
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = min(mn[k], float_to_ordered(b.mn[k])); mx[k] = max(mx[k], float_to_ordered(b.mx[k])); }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { atomicMin(&sb[k], mn[k]); atomicMax(&sb[3 + k], mx[k]); }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], sb[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], sb[threadIdx.x]);
}

__global__ __launch_bounds__(LG_THREADS) void lg_morton_kernel(LbvhArgs a, const uint32_t *bounds, uint32_t *key, uint32_t *val) {
    float bmin[3], scale[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        bmin[k] = ordered_to_float(bounds[k]);
        float ext = ordered_to_float(bounds[3 + k]) - bmin[k];
        scale[k] = ext > 0.0f ? 1024.0f / ext : 0.0f;
    }
    for (uint32_t i = blockIdx.x * LG_THREADS + threadIdx.x; i < a.n; i += gridDim.x * LG_THREADS) {
        Box b = leaf_box(a, i);
        uint32_t q[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float c = (b.mn[k] + b.mx[k]) * 0.5f;
            float f = (c - bmin[k]) * scale[k];
            int qi = (int)f;
            q[k] = (uint32_t)(qi < 0 ? 0 : (qi > 1023 ? 1023 : qi));
        }
        key[i] = morton30(q[0], q[1], q[2]); val[i] = i;
    }
}

// hist[digit * blocks + block] = number of keys of this block's tile with that digit
__global__ __launch_bounds__(LG_THREADS) void lg_hist_kernel(const uint32_t *key, uint32_t n, int shift, uint32_t *hist, uint32_t blocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * LG_TILE;
    for (uint32_t r = 0; r < LG_TILE / LG_THREADS; r++) {
        uint32_t i = base + r * LG_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(key[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * blocks + blockIdx.x] = h[threadIdx.x];
}

// Exclusive scan of the histogram in (digit, block) order, in two levels: workgroup d scans row d (the `blocks` counts of digit d) in
// place and leaves the row's total in hist[256 * blocks + d]; the scatter kernel adds the exclusive scan of those 256 totals itself.
// (One workgroup over all 256 * blocks entries -- 1.3 M for 5 M triangles -- took 308 us per radix pass.)
__global__ __launch_bounds__(LG_THREADS) void lg_scan_rows_kernel(uint32_t *hist, uint32_t blocks) {
    __shared__ uint32_t waveTot[LG_THREADS / 64];
    __shared__ uint32_t carry;
    uint32_t *row = hist + (size_t)blockIdx.x * blocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < blocks; base += LG_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < blocks ? row[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v, lane);
        if (lane == 63) waveTot[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < wave; w++) before += waveTot[w];
        if (i < blocks) row[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == LG_THREADS - 1) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) hist[(size_t)256 * blocks + blockIdx.x] = carry;
}

// stable scatter of one tile: rounds of 256 keys in index order; rank inside a wave by ballot matching
__global__ __launch_bounds__(LG_THREADS) void lg_scatter_kernel(const uint32_t *keyIn, const uint32_t *valIn, uint32_t *keyOut, uint32_t *valOut,
                                                                uint32_t n, int shift, const uint32_t *hist, uint32_t blocks) {
    __shared__ uint32_t digitOffset[256];
    __shared__ uint32_t waveCount[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // where this tile's keys of digit d go: keys of smaller digits (exclusive scan of the 256 row totals) + keys of digit d in earlier tiles
        const uint32_t total = hist[(size_t)256 * blocks + threadIdx.x];
        const uint32_t incl = wave_inclusive_scan(total, lane);
        if (lane == 63) waveCount[0][wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += waveCount[0][w];
        digitOffset[threadIdx.x] = before + incl - total + hist[(size_t)threadIdx.x * blocks + blockIdx.x];
        __syncthreads();
    }
    const uint32_t base = blockIdx.x * LG_TILE;
    for (uint32_t r = 0; r < LG_TILE / LG_THREADS; r++) {
        for (int w = 0; w < 4; w++) waveCount[w][threadIdx.x] = 0;
        __syncthreads();
        const uint32_t i = base + r * LG_THREADS + threadIdx.x;
        const bool valid = i < n;
        const uint32_t k = valid ? keyIn[i] : 0u, v = valid ? valIn[i] : 0u;
        const uint32_t d = (k >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            unsigned long long b = __ballot((d >> bit) & 1u);
            peers &= ((d >> bit) & 1u) ? b : ~b;
        }
        const unsigned long long lower = peers & ((1ull << lane) - 1ull);
        const uint32_t rank = (uint32_t)__popcll(lower);
        if (valid && lower == 0) waveCount[wave][d] = (uint32_t)__popcll(peers);      // one leader per (wave, digit)
        __syncthreads();
        if (valid) {
            uint32_t pos = digitOffset[d] + rank;
            for (int w = 0; w < wave; w++) pos += waveCount[w][d];
            keyOut[pos] = k; valOut[pos] = v;
        }
        __syncthreads();
        digitOffset[threadIdx.x] += waveCount[0][threadIdx.x] + waveCount[1][threadIdx.x] + waveCount[2][threadIdx.x] + waveCount[3][threadIdx.x];
        __syncthreads();
    }
}

__global__ __launch_bounds__(LG_THREADS) void lg_karras_kernel(LbvhArgs a, const uint32_t *key, const uint32_t *val) {
    const int N = (int)a.n;
    for (int i = (int)(blockIdx.x * LG_THREADS + threadIdx.x); i < N; i += (int)(gridDim.x * LG_THREADS)) {
        a.sortedIndex[i] = val[i]; a.morton[i] = key[i];
        if (i >= N - 1) continue;
        if (i == 0) a.nodes[0].parent = RT64_NO_CHILD;
        int d = (delta64(key, val, N, i, i + 1) - delta64(key, val, N, i, i - 1)) >= 0 ? 1 : -1;
        int dmin = delta64(key, val, N, i, i - d);
        int lmax = 2;
        while (delta64(key, val, N, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta64(key, val, N, i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d;
        int dnode = delta64(key, val, N, i, j);
        int s = 0;
        for (int div = 2;; div *= 2) {
            int t = (l + div - 1) / div;
            if (delta64(key, val, N, i, i + (s + t) * d) > dnode) s += t;
            if (t <= 1) break;
        }
        int g = i + s * d + (d < 0 ? -1 : 0);
        int lo = i < j ? i : j, hi = i < j ? j : i;
        uint32_t left, right;
        if (lo == g) { left = RT64_LEAF_BIT | (uint32_t)g; a.leafParent[g] = (uint32_t)i; } else { left = (uint32_t)g; a.nodes[g].parent = (uint32_t)i; }
        if (hi == g + 1) { right = RT64_LEAF_BIT | (uint32_t)(g + 1); a.leafParent[g + 1] = (uint32_t)i; } else { right = (uint32_t)(g + 1); a.nodes[g + 1].parent = (uint32_t)i; }
        a.nodes[i].left = left; a.nodes[i].right = right; a.nodes[i].pad = (uint32_t)j;      // pad = the other end of the node's leaf range [min(i, j), max(i, j)]: the fit asks whether a node lies inside one chunk
    }
}

// leaves in Morton order (+ their boxes) and reset of the fit stamps
__global__ __launch_bounds__(LG_THREADS) void lg_leaves_kernel(LbvhArgs a, float *leafBox, uint32_t *done) {
    for (uint32_t s = blockIdx.x * LG_THREADS + threadIdx.x; s < a.n; s += gridDim.x * LG_THREADS) {
        const uint32_t leaf = a.sortedIndex[s];
        Box b;
        if (a.mode == LBVH_MODE_TRIANGLES) {
            float v[3][3]; load_positions(a, leaf, v);
            GpuTri t;
#pragma unroll
            for (int k = 0; k < 3; k++) { t.v0[k] = v[0][k]; t.v1[k] = v[1][k]; t.v2[k] = v[2][k]; b.mn[k] = fminf(fminf(v[0][k], v[1][k]), v[2][k]); b.mx[k] = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]); }
            t.prim = leaf; t.pad1 = 0; t.pad2 = 0;
            a.tris[s] = t;
        }
        else b = leaf_box(a, leaf);
#pragma unroll
        for (int k = 0; k < 3; k++) { leafBox[6 * s + k] = b.mn[k]; leafBox[6 * s + 3 + k] = b.mx[k]; }
        done[s] = 0;
    }
}

// Box fit in a few launches instead of one per tree level.  The leaves (Morton order) are cut into chunks of LG_CHUNK, chunks into groups
// of LG_FAN chunks, groups into groups of groups ...; an inner node is LOCAL at granularity G when its leaf range [min(i, pad), max(i, pad)]
// lies inside one interval of G leaves, and then both of its subtrees do.
//   lg_fit_local_kernel : one workgroup per chunk fits the chunk's local nodes bottom-up -- a lane climbs from its leaf, the first arrival
//                         at a node leaves its box there and stops, the second merges and goes on (arrival counters in LDS: a local node's
//                         index lies inside the chunk; workgroup-scope fences: both arrivals belong to this workgroup).  A climb stops below
//                         the first node that is not local and puts the subtree it finished (a leaf or an inner node: a "root") on the
//                         chunk's list.
//   lg_fit_group_kernel : one workgroup per group of LG_FAN lists climbs from every listed root through the nodes local at the group's
//                         granularity, same protocol (counters in HBM now), and lists what it finished for the next level; the last
//                         level is one workgroup for which every node is local.
// No fence ever has to cross an XCD: a node is fitted by exactly one workgroup, and what that workgroup reads from earlier levels was
// written by earlier launches.  Unions of boxes do not depend on the order of arrival: the boxes are the ones the level-by-level fit
// produced (105 launches, 2.1 ms for the 5.4 M-triangle stress scene; a single workgroup for everything above the chunks: 3.2 ms).
static_assert(LG_FAN_LISTS == 64, "LG_FAN");
#define LG_FAN LG_FAN_LISTS
DEV bool lg_is_local(const GpuNode &nd, uint32_t i, uint32_t granule) {
    const uint32_t lo = i < nd.pad ? i : nd.pad, hi = i < nd.pad ? nd.pad : i;
    return lo / granule == hi / granule;
}
// climbs from `child` (box b) through the nodes local at `granule`; ended = it stopped below a node that is not (child / b then describe a root)
template <class Counter>
DEV void lg_climb(LbvhArgs a, float *nodeBox, Counter &&counter, uint32_t granule, uint32_t &child, uint32_t p, Box &b, bool &ended) {
    ended = false;
    for (;;) {
        GpuNode &nd = a.nodes[p];
        if (!lg_is_local(nd, p, granule)) { ended = true; return; }
        const bool isLeft = nd.left == child;
        float *dmn = isLeft ? nd.lmin : nd.rmin, *dmx = isLeft ? nd.lmax : nd.rmax;
#pragma unroll
        for (int k = 0; k < 3; k++) { dmn[k] = b.mn[k]; dmx[k] = b.mx[k]; }
        __threadfence_block();
        if (atomicAdd(counter(p), 1u) == 0) return;             // the sibling subtree finishes this node
        __threadfence_block();
        const float *smn = isLeft ? nd.rmin : nd.lmin, *smx = isLeft ? nd.rmax : nd.lmax;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float omn = *(volatile const float *)&smn[k], omx = *(volatile const float *)&smx[k];
            b.mn[k] = fminf(b.mn[k], omn); b.mx[k] = fmaxf(b.mx[k], omx);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { nodeBox[6 * (size_t)p + k] = b.mn[k]; nodeBox[6 * (size_t)p + 3 + k] = b.mx[k]; }
        if (p == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) { a.header->bmin[k] = b.mn[k]; a.header->bmax[k] = b.mx[k]; }
            a.header->count = a.n; a.header->depth = 255;
            return;
        }
        child = p; p = nd.parent;
    }
}
#define LG_FIT_THREADS 1024      // two leaves per lane: a climb is a chain of L2 round trips, so the chunk wants lanes, not trips
__global__ __launch_bounds__(LG_FIT_THREADS) void lg_fit_local_kernel(LbvhArgs a, const float *leafBox, float *nodeBox, uint32_t *rootList, uint32_t *rootCount) {
    __shared__ uint32_t arrivals[LG_CHUNK];
    __shared__ uint32_t listed;
    for (uint32_t k = threadIdx.x; k < LG_CHUNK; k += LG_FIT_THREADS) arrivals[k] = 0;
    if (threadIdx.x == 0) listed = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * LG_CHUNK;
    for (uint32_t s = first + threadIdx.x; s < first + LG_CHUNK && s < a.n; s += LG_FIT_THREADS) {
        Box b;
#pragma unroll
        for (int k = 0; k < 3; k++) { b.mn[k] = leafBox[6 * (size_t)s + k]; b.mx[k] = leafBox[6 * (size_t)s + 3 + k]; }
        uint32_t child = RT64_LEAF_BIT | s;
        bool ended;
        lg_climb(a, nodeBox, [&](uint32_t p) { return &arrivals[p - first]; }, LG_CHUNK, child, a.leafParent[s], b, ended);
        if (ended) { const uint32_t k = atomicAdd(&listed, 1u); if (k < LG_ROOT_SLOTS) rootList[(size_t)blockIdx.x * LG_ROOT_SLOTS + k] = child; }
    }
    __syncthreads();
    if (threadIdx.x == 0) rootCount[blockIdx.x] = listed < LG_ROOT_SLOTS ? listed : LG_ROOT_SLOTS;
}
// lists [blockIdx.x * LG_FAN, + LG_FAN) of the level below -> the nodes local at `granule` (0xFFFFFFFF: all of them) -> this group's list
__global__ __launch_bounds__(1024) void lg_fit_group_kernel(LbvhArgs a, const float *leafBox, float *nodeBox, uint32_t *counter, const uint32_t *inList, const uint32_t *inCount,
                                                            uint32_t inLists, uint32_t granule, uint32_t *outList, uint32_t *outCount) {
    __shared__ uint32_t listed;
    if (threadIdx.x == 0) listed = 0;
    __syncthreads();
    const uint32_t firstList = blockIdx.x * LG_FAN, lists = inLists - firstList < LG_FAN ? inLists - firstList : LG_FAN;
    for (uint32_t e = threadIdx.x; e < lists * LG_ROOT_SLOTS; e += 1024) {
        const uint32_t list = firstList + e / LG_ROOT_SLOTS, k = e % LG_ROOT_SLOTS;
        if (k >= inCount[list]) continue;
        uint32_t child = inList[(size_t)list * LG_ROOT_SLOTS + k];
        const float *src = (child & RT64_LEAF_BIT) ? leafBox + 6 * (size_t)(child & 0x7FFFFFFFu) : nodeBox + 6 * (size_t)child;
        Box b;
#pragma unroll
        for (int q = 0; q < 3; q++) { b.mn[q] = src[q]; b.mx[q] = src[3 + q]; }
        const uint32_t p = (child & RT64_LEAF_BIT) ? a.leafParent[child & 0x7FFFFFFFu] : a.nodes[child].parent;
        bool ended;
        lg_climb(a, nodeBox, [&](uint32_t q) { return &counter[q]; }, granule, child, p, b, ended);
        if (ended) { const uint32_t slot = atomicAdd(&listed, 1u); if (slot < LG_ROOT_SLOTS) outList[(size_t)blockIdx.x * LG_ROOT_SLOTS + slot] = child; }
    }
    __syncthreads();
    if (threadIdx.x == 0) outCount[blockIdx.x] = listed < LG_ROOT_SLOTS ? listed : LG_ROOT_SLOTS;
}

}  // namespace

size_t lbvh_large_scratch_bytes(uint32_t n) {
    const size_t blocks = ((size_t)n + LG_TILE - 1) / LG_TILE;
    return (size_t)n * (4 * 4 + 4 + 24 + 24) + (blocks + 1) * 256 * 4 + 64 + 16 * 256 + lg_lists_all_levels(n) * (LG_ROOT_SLOTS + 1) * 4 + 2 * 256;
}

hipError_t lbvh_launch_large(const LbvhArgs &args, hipStream_t stream) {
    if (!args.scratch || args.scratchBytes < lbvh_large_scratch_bytes(args.n)) return hipErrorInvalidValue;
    const uint32_t n = args.n;
    LargeScratch L = carve(args.scratch, n);
    const uint32_t grid = std::min<uint32_t>((n + LG_THREADS - 1) / LG_THREADS, 2048u);
    if (!args.refit) {
        hipLaunchKernelGGL(lg_init_bounds, dim3(1), dim3(64), 0, stream, L.bounds);
        hipLaunchKernelGGL(lg_bounds_kernel, dim3(grid), dim3(LG_THREADS), 0, stream, args, L.bounds);
        hipLaunchKernelGGL(lg_morton_kernel, dim3(grid), dim3(LG_THREADS), 0, stream, args, L.bounds, L.keyA, L.valA);
        uint32_t *kin = L.keyA, *vin = L.valA, *kout = L.keyB, *vout = L.valB;
        for (int shift = 0; shift < 32; shift += 8) {      // 30 code bits: four 8-bit digits
            hipLaunchKernelGGL(lg_hist_kernel, dim3(L.blocks), dim3(LG_THREADS), 0, stream, kin, n, shift, L.hist, L.blocks);
            hipLaunchKernelGGL(lg_scan_rows_kernel, dim3(256), dim3(LG_THREADS), 0, stream, L.hist, L.blocks);
            hipLaunchKernelGGL(lg_scatter_kernel, dim3(L.blocks), dim3(LG_THREADS), 0, stream, kin, vin, kout, vout, n, shift, L.hist, L.blocks);
            std::swap(kin, kout); std::swap(vin, vout);
        }
        hipLaunchKernelGGL(lg_karras_kernel, dim3(grid), dim3(LG_THREADS), 0, stream, args, kin, vin);
    }
    hipLaunchKernelGGL(lg_leaves_kernel, dim3(grid), dim3(LG_THREADS), 0, stream, args, L.leafBox, L.done);
    uint32_t lists = (n + LG_CHUNK - 1) / LG_CHUNK;
    uint32_t *inList = L.rootList, *inCount = L.rootCount;
    hipLaunchKernelGGL(lg_fit_local_kernel, dim3(lists), dim3(LG_FIT_THREADS), 0, stream, args, L.leafBox, L.nodeBox, inList, inCount);
    for (uint64_t granule = (uint64_t)LG_CHUNK * LG_FAN;; granule *= LG_FAN) {         // levels above the chunks: 2 560 chunks -> 40 groups -> 1
        const uint32_t groups = (lists + LG_FAN - 1) / LG_FAN;
        uint32_t *outList = inList + (size_t)lists * LG_ROOT_SLOTS, *outCount = inCount + lists;
        hipLaunchKernelGGL(lg_fit_group_kernel, dim3(groups), dim3(1024), 0, stream, args, L.leafBox, L.nodeBox, L.done, inList, inCount, lists,
                           groups == 1 ? 0xFFFFFFFFu : (uint32_t)granule, outList, outCount);
        if (groups == 1) break;
        lists = groups; inList = outList; inCount = outCount;
    }
    return hipGetLastError();
}
