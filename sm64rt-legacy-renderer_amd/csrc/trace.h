// trace.h -- two-level LBVH traversal + Moller-Trumbore for one ray per lane (wave64), node stack in LDS.
//
// Replaces the DXR TraceRay black box behind PrimaryRayGen.hlsl:71, Lights.hlsli:50, IndirectRayGen.hlsl:79,
// ReflectionRayGen.hlsl:63 and RefractionRayGen.hlsl:59.  DXR semantics kept: un-normalised directions (t in units of
// |dir|), TMin < t < TMax, object-space traversal per instance, object-space facing (front iff dot(cross(e1,e2),d) < 0),
// per-instance cull disable, InstanceIndex = TLAS build position, PrimitiveIndex = triangle number.
//
// Geometry spec (every operation below is part of the bit-exact contract with the scalar reference tracer):
//   R1 ds = |d| < 1e-20 ? copysign(1e-20, d) : d ; inv = 1/ds ; oi = -(o*inv)
//   R2 slab test with fmaf(bound, inv, oi); tfar scaled by 1.0000004f; hit iff tnear <= tfar
//   R3 depth-first, near child first (ties: left), far child pushed; a TLAS leaf walks its BLAS to exhaustion
//   R4 Moller-Trumbore with the g_dot3 / g_cross3 fma chains; u,v >= 0, u+v <= 1, tmin < t < tmax
//   R5 the hit handler may lower tmax or end the walk
//
// Stack: RT_STACK_LDS entries per lane in LDS, laid out [wave][entry][lane] so a wave's push/pop is one conflict-free
// ds_write_b32/ds_read_b32; deeper entries spill to a per-resident-lane slab in HBM (never touched on the sample scene:
// LBVH depth <= 30 + log2(n) per level).
#pragma once
#include "rt64_gpu.h"
#include "device_math.h"

// Frame constants as the device functions see them: a reference into the constant address space (the kernel-argument segment),
// so that a field is read where it is used (scalar load, scalar cache) instead of being held -- or spilled -- from kernel entry.
typedef const FrameParams __attribute__((address_space(4))) &PRef;
typedef const FrameParams __attribute__((address_space(4))) *PPtr;
struct Mat16c { float m[16]; };
DEV Mat16c cmat(const float __attribute__((address_space(4))) *p) { Mat16c r; __builtin_memcpy(&r, p, sizeof(r)); return r; }
typedef const ViewImages __attribute__((address_space(4))) &IRef;
typedef const ViewImages __attribute__((address_space(4))) *IPtr;
// The image table is the second parameter of every ray kernel: it follows the frame constants in the kernel-argument segment.
#define RT_KERNARG_IMAGES_OFFSET ((sizeof(FrameParams) + alignof(ViewImages) - 1) / alignof(ViewImages) * alignof(ViewImages))
DEV IPtr kernel_images() { return (IPtr)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + RT_KERNARG_IMAGES_OFFSET); }
DEV IPtr kernel_images_here() { uint32_t z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return (IPtr)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + RT_KERNARG_IMAGES_OFFSET + z); }
// The kernel's FrameParams argument is its first parameter: offset 0 of the kernel-argument segment.
DEV PPtr kernel_params() { return (PPtr)__builtin_amdgcn_kernarg_segment_ptr(); }
// Same, through an offset the compiler cannot see through (always 0): loads that depend on it stay inside the loop iteration that made it.
DEV PPtr kernel_params_here() { uint32_t z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return (PPtr)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + z); }
#define RT_BLOCK 256                 // threads per workgroup of the ray kernels (the per-wave form of the one-kernel frame runs 64)
#define RT_LANES 64                  // lanes of a wave: the per-lane LDS columns (node stack, light candidates) are laid out [wave][entry][lane], so a
                                     // wave's push / pop is one conflict-free ds access and the layout does not depend on the workgroup size
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 24                 // (kernels.h defines it for the host too)
#endif
#ifndef RT_STACK_SPILL
#define RT_STACK_SPILL 84               // (kernels.h defines both for the host too) entries per lane in the HBM slab behind the LDS entries
#define RT_STACK_SPILL_HEADER 2         // uint32 words in front of EVERY lane's entries: the address of the overflow word (host-pinned memory)
#endif
#ifndef RT_STACK_LDS_CACHED
#define RT_STACK_LDS_CACHED 16           // (kernels.h defines it for the host too) kernels that also hold the LDS scene cache: the host enables the cache only when TLAS depth + the deepest BLAS fit in these
                                        // entries (BlasHeader::depth), so their push / pop are plain LDS accesses -- no spill branch in the node loop
#endif

struct RaySpace { float o[3], d[3], inv[3], oi[3]; };

DEV void make_ray_space(const float o[3], const float d[3], RaySpace &r) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r.o[k] = o[k]; r.d[k] = d[k];
        float ds = fabsf(d[k]) < 1e-20f ? copysignf(1e-20f, d[k]) : d[k];
        r.inv[k] = 1.0f / ds;
        r.oi[k] = -(o[k] * r.inv[k]);
    }
}

DEV bool box_hit(const RaySpace &r, const float lo[3], const float hi[3], float tmin, float tmax, float &tnear) {
    float ax = fmaf(lo[0], r.inv[0], r.oi[0]), bx = fmaf(hi[0], r.inv[0], r.oi[0]);
    float ay = fmaf(lo[1], r.inv[1], r.oi[1]), by = fmaf(hi[1], r.inv[1], r.oi[1]);
    float az = fmaf(lo[2], r.inv[2], r.oi[2]), bz = fmaf(hi[2], r.inv[2], r.oi[2]);
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax)) * 1.0000004f;
    tnear = tn;
    return tn <= tf;
}

DEV bool tri_hit(const RaySpace &r, const GpuTri &tri, bool cull, float tmin, float tmax, float &t, float &u, float &v) {
    float e1[3], e2[3], p[3], q[3], tv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { e1[k] = tri.v1[k] - tri.v0[k]; e2[k] = tri.v2[k] - tri.v0[k]; }
    g_cross3(r.d, e2, p);
    float det = g_dot3(e1, p);
    // one exit instead of two: the determinant test joins the final condition (a rejected determinant only feeds inf / NaN into
    // comparisons that are then ignored), so a wave does not pay an exec-mask round trip before the division
    const bool detRejected = cull ? !(det > 0.0f) : (det == 0.0f || det != det);
    float inv = 1.0f / det;
#pragma unroll
    for (int k = 0; k < 3; k++) tv[k] = r.o[k] - tri.v0[k];
    float uu = g_dot3(tv, p) * inv;
    g_cross3(tv, e1, q);
    float vv = g_dot3(r.d, q) * inv;
    float tt = g_dot3(e2, q) * inv;
    if (detRejected || !(uu >= 0.0f) || !(vv >= 0.0f) || !(uu + vv <= 1.0f) || !(tt > tmin) || !(tt < tmax)) return false;
    t = tt; u = uu; v = vv;
    return true;
}

typedef uint32_t u32x4_lds __attribute__((ext_vector_type(4)));      // a 16-byte word of the LDS scene cache

// The two halves of the stack carry their address spaces in the pointer types: with plain pointers the compiler merges the two
// branches of push / pop into ONE flat access through a selected pointer (flat_load / flat_store: both wait counters, the texture
// addresser's queue) -- on every pop of the node loop.  Typed, the LDS half is a ds_read_b32 / ds_write_b32.
typedef uint32_t __attribute__((address_space(3))) *LdsU32Ptr;
typedef uint32_t __attribute__((address_space(1))) *GlobalU32Ptr;
typedef int16_t __attribute__((address_space(3))) *LdsI16Ptr;
// Child references inside the LDS scene cache are 16-bit values, sign-extended: inner node = its index, leaf = 0xFFFF8000 | index,
// no child = 0xFFFFFFFF (scene_cache_image_kernel rewrites the node arrays that way; the host enables the cache only below 2^15 - 1
// leaves and nodes per tree).  Bit 31 still says "leaf" and RT64_NO_CHILD is unchanged, the index is `& RT_CACHE_INDEX_MASK`, and
// the stack of the cached kernels holds int16 entries: ds_write_b16 truncates, ds_read_i16 sign-extends -- half the LDS of a uint32 stack.
#define RT_CACHE_INDEX_MASK 0x7FFFu
struct TraceStack {
    LdsU32Ptr lds;        // this lane's column in its wave's block of the stack array, stride RT_LANES
    LdsI16Ptr lds16;      // the same block as int16 entries (cached kernels), stride RT_LANES
    GlobalU32Ptr spill;   // per-lane slab of RT_STACK_SPILL entries
    const u32x4_lds *cache;   // LDS scene cache (see fill_scene_cache), nullptr when the scene does not fit
    int ldsEntries;       // entries of this lane's stack that live in LDS (RT_STACK_LDS or RT_STACK_LDS_CACHED)
    DEV void use_cache(const u32x4_lds *c) { cache = c; ldsEntries = RT_STACK_LDS_CACHED; }
    // An entry that fits neither half is dropped -- the walk then misses a subtree -- and the frame says so: the two words in front of every lane's entries
    // (RT_STACK_SPILL_HEADER; written once when the slab is allocated) hold the address of a word in pinned host memory, so the report needs nothing but the
    // lane's own slab pointer and costs nothing where no overflow happens.  The host reads the word after the frame (Device::checkTraversalOverflow).
    DEV void report_overflow() const {
        typedef uint32_t *__attribute__((address_space(1))) *FlagSlot;
        uint32_t *flag = *reinterpret_cast<FlagSlot>(reinterpret_cast<uintptr_t>(spill - RT_STACK_SPILL_HEADER));
        if (flag) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    template <bool LDS_ONLY = false> DEV void push(int &sp, uint32_t v) const {
        if (LDS_ONLY) { lds16[(sp & (RT_STACK_LDS_CACHED - 1)) * RT_LANES] = (int16_t)v; sp++; return; }      // depth checked by the host (the mask only keeps a wrong depth inside the array)
        if (sp < ldsEntries) lds[sp * RT_LANES] = v;
        else if (sp < ldsEntries + RT_STACK_SPILL) spill[sp - ldsEntries] = v;
        else { report_overflow(); return; }      // deeper than TLAS depth + BLAS depth of any tree these builders make from fewer than 2^23 leaves and 2^13 instances with 30-bit keys; never silent
        sp++;
    }
    template <bool LDS_ONLY = false> DEV uint32_t pop(int &sp) const {
        sp--;
        if (LDS_ONLY) return (uint32_t)(int32_t)lds16[(sp & (RT_STACK_LDS_CACHED - 1)) * RT_LANES];
        return sp < ldsEntries ? lds[sp * RT_LANES] : spill[sp - ldsEntries];
    }
};

#ifdef RT_PROFILE_TRIPS       // diagnostic build (never shipped): how many trips of the node loop / leaf step a WAVE made, counted in SGPRs whatever the exec mask is
struct TraceCounts { uint32_t nodes, tris, tripsNode, tripsLeaf, spills; };
#define RT_TRIP_DECL(name) __shared__ uint32_t name##Lds[16]; if ((threadIdx.x & 63u) == 0u) name##Lds[threadIdx.x >> 6] = 0u; uint32_t name = 0
#define RT_TRIP(name) do { if ((int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(1)) - 1) name##Lds[threadIdx.x >> 6]++; } while (0)
#define RT_TRIP_END(name) name = name##Lds[threadIdx.x >> 6]
#else
struct TraceCounts { uint32_t nodes, tris; };
#define RT_TRIP_DECL(name)
#define RT_TRIP(name)
#define RT_TRIP_END(name)
#endif

// Pointers that were loaded from memory are "generic" to the compiler (flat_load + both wait counters).  Every BVH array
// lives in HBM, so fetch through address space 1: global_load_dwordx4, vmcnt only.
template <class T> DEV T load_global(const T *p) {              // scalars / pointers
    typedef const T __attribute__((address_space(1))) *GP;
    return *reinterpret_cast<GP>(reinterpret_cast<uintptr_t>(p));
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // builtin vector: loads through address space 1 directly
typedef const u32x4 __attribute__((address_space(1))) *GlobalU4;
DEV GpuNode load_node(const GpuNode *p) {                         // 4 x global_load_dwordx4
    GlobalU4 q = reinterpret_cast<GlobalU4>(reinterpret_cast<uintptr_t>(p));
    union { u32x4 w[4]; GpuNode n; } u;
    u.w[0] = q[0]; u.w[1] = q[1]; u.w[2] = q[2]; u.w[3] = q[3];
    return u.n;
}
DEV GpuTri load_tri(const GpuTri *p) {                            // 3 x global_load_dwordx4
    GlobalU4 q = reinterpret_cast<GlobalU4>(reinterpret_cast<uintptr_t>(p));
    union { u32x4 w[3]; GpuTri t; } u;
    u.w[0] = q[0]; u.w[1] = q[1]; u.w[2] = q[2];
    return u.t;
}

// OnHit: bool operator()(float t, float u, float v, uint32_t instance, uint32_t prim, float &tmax, uint32_t instanceFlags,
//                        float instanceDepthBias) -> true = end search.   (RayWalk::run below keeps the six-argument form.)
//
// Loop shape ("while-while"): the wave first walks inner nodes until every live lane holds a leaf (or has finished), then all
// lanes process their leaf together (ray/triangle test, or the object-space switch at a TLAS leaf).  Lanes never execute the
// node path and the leaf path in the same trip, which is what a one-ray-per-lane walk loses most to on wave64.  The order of
// operations of each individual ray is unchanged (R3), so hits and visit counts stay bit-identical to the scalar tracer.
//
// CACHED: the scene's nodes and per-instance records sit in LDS (stk.cache, layout in fill_scene_cache): a node visit and an
// instance entry are ds_read_b128s (~100 cycles) instead of dependent global loads (~1 us each under load); only triangles still come
// from HBM/L2 (with them in LDS too -- 36 KB per workgroup instead of 21 -- the longer fill of every workgroup cost more than the L2
// round trips it saved: C2 0.158 against 0.153 ms, profiles/r02_experiments/lds_triangles_*).  Same data, same arithmetic, same order:
// results and visit counts are unchanged.
DEV GpuNode load_node_lds(const u32x4_lds *q) {
    union { u32x4_lds w[4]; GpuNode n; } u;
    u.w[0] = q[0]; u.w[1] = q[1]; u.w[2] = q[2]; u.w[3] = q[3];
    return u.n;
}


// ---- the one-step-per-trip walk (scenes that walk from HBM / L2) ---------------------------------------------------------------------------------
// The "while-while" loop of trace_ray below makes a wave's lanes take turns by kind of step: the node loop runs until EVERY live lane holds a leaf,
// then all leaves are processed.  That is the right shape while the vector ALU is what a wave waits for (the LDS-cached walk of small scenes).  A
// walk that fetches every node from HBM waits for memory instead, one dependent round trip of ~1 300 cycles per step, and there the turns cost
// dearly: a lane that reaches its leaf early idles through the other lanes' node steps, so a wave makes 2-3 x as many trips as its busiest lane has
// steps (measured on the 5.4 M-triangle stress scene, tools/tile_timing.py: 846 node trips + 48 leaf trips against 327 visits of the busiest lane in
// the waves that set the frame time).  Here every live lane takes exactly ONE step per trip, whatever its kind: the lane's record -- a 64-byte node,
// or a 48-byte triangle read as 64 bytes (the triangle arrays carry 16 bytes of padding behind the last record) -- is fetched by the same four
// 16-byte loads from a per-lane address, so node lanes and triangle lanes wait for memory together, and the two kinds of arithmetic follow under
// their exec masks (~130 vector instructions against the round trip).  A wave then makes as many trips as its busiest lane has steps.  The order
// of operations of each individual ray is that of trace_ray (R3), so hits, visit counts and every bit of the results are unchanged.
// (Measured and not kept: touching the record of every pushed entry with a one-dword load, so that its pop would find the line in L2.  Loads
// return in order, so the next trip's record -- often an L2 hit -- then waits behind the prefetch's HBM miss: the stress frame went from 0.667
// to 0.743 ms, gpurun_out/r03_tt_stress_pf2.txt.)
union NodeOrTri { u32x4 w[4]; GpuNode n; GpuTri t; };
template <class OnHit>
DEV void trace_ray_stepwise(PRef P, const float o[3], const float d[3], float tmin, float tmax, bool cullBackFaces,
                            const TraceStack &stk, OnHit &&onHit, TraceCounts &cnt) {
    RaySpace W, R;
    make_ray_space(o, d, W);
    R = W;
    const GpuNode *nodes = P.tlasNodes;
    const GpuTri *tris = nullptr;
    int sp = 0, blasBase = -1;
    uint32_t inst = 0, instFlags = 0;
    float instDepthBias = 0.0f;
    bool cull = false;
    uint32_t cur = 0;
    bool alive = true;
    RT_TRIP_DECL(tripsNode); RT_TRIP_DECL(tripsLeaf);
    auto popNext = [&]() -> bool {
        if (blasBase >= 0 && sp == blasBase) { blasBase = -1; R = W; nodes = P.tlasNodes; }      // BLAS exhausted: resume the TLAS walk in world space
        if (sp == 0) return false;
        cur = stk.pop(sp);
        return true;
    };
    while (alive) {
        RT_TRIP(tripsNode);
        const bool isNode = !(cur & RT64_LEAF_BIT), isTri = !isNode && cur != RT64_NO_CHILD && blasBase >= 0;
        NodeOrTri rec;
        if (isNode || isTri) {       // one fetch for both kinds of lane
            const uintptr_t at = isNode ? reinterpret_cast<uintptr_t>(nodes + cur) : reinterpret_cast<uintptr_t>(tris + (cur & 0x7FFFFFFFu));
            GlobalU4 q = reinterpret_cast<GlobalU4>(at);
            rec.w[0] = q[0]; rec.w[1] = q[1]; rec.w[2] = q[2]; rec.w[3] = q[3];
        }
        if (isNode) {
            cnt.nodes++;
            float tl, tr;
            const bool hl = box_hit(R, rec.n.lmin, rec.n.lmax, tmin, tmax, tl);
            const bool hr = box_hit(R, rec.n.rmin, rec.n.rmax, tmin, tmax, tr);
            const bool both = hl && hr, rightFirst = tr < tl;
            const uint32_t nearChild = both ? (rightFirst ? rec.n.right : rec.n.left) : (hl ? rec.n.left : rec.n.right);
            if (both) stk.push(sp, rightFirst ? rec.n.left : rec.n.right);
            if (hl || hr) cur = nearChild;
            else alive = popNext();
        }
        else if (isTri) {
            cnt.tris++;
            float t, u, v;
            if (tri_hit(R, rec.t, cull, tmin, tmax, t, u, v) && onHit(t, u, v, inst, rec.t.prim, tmax, instFlags, instDepthBias)) alive = false;
            else alive = popNext();
        }
        else if (cur != RT64_NO_CHILD) {
            // TLAS leaf: enter the instance (G8)
            inst = load_global(P.tlasIndex + (cur & 0x7FFFFFFFu));
            const GpuInstance *in = P.instances + inst;
            const float *M = in->worldToObject;
            float oo[3], dd[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float m0 = load_global(M + c), m1 = load_global(M + 4 + c), m2 = load_global(M + 8 + c), m3 = load_global(M + 12 + c);
                oo[c] = fmaf(W.o[2], m2, fmaf(W.o[1], m1, fmaf(W.o[0], m0, m3)));
                dd[c] = fmaf(W.d[2], m2, fmaf(W.d[1], m1, W.d[0] * m0));
            }
            nodes = load_global(&in->nodes); tris = load_global(&in->tris);
            const uint32_t flags = load_global(&in->flags); instDepthBias = load_global(&in->material.depthBias);
            make_ray_space(oo, dd, R);
            instFlags = flags;
            cull = cullBackFaces && !(flags & GPU_INST_CULL_DISABLE);
            blasBase = sp;
            cur = 0;
        }
        else alive = popNext();
    }
#ifdef RT_PROFILE_TRIPS
    RT_TRIP_END(tripsNode); RT_TRIP_END(tripsLeaf);
    cnt.tripsNode += tripsNode; cnt.tripsLeaf += tripsLeaf;
#endif
}

#ifndef RT_STEPWISE_WALK
#define RT_STEPWISE_WALK 1       // 0: scenes without the LDS scene cache keep the while-while loop too (A/B builds)
#endif

template <bool CACHED = false, class OnHit>
DEV void trace_ray(PRef P, const float o[3], const float d[3], float tmin, float tmax, bool cullBackFaces,
                   const TraceStack &stk, OnHit &&onHit, TraceCounts &cnt) {
    if (P.instanceCount == 0) return;
    if (!CACHED && RT_STEPWISE_WALK) { trace_ray_stepwise(P, o, d, tmin, tmax, cullBackFaces, stk, onHit, cnt); return; }
    RaySpace W, R;
    make_ray_space(o, d, W);
    R = W;
    const GpuNode *nodes = P.tlasNodes;
    const uint32_t tlasOff = 4u * P.cacheInstances;          // CACHED: node arrays are addressed by their word offset in the cache
    uint32_t nodeOff = tlasOff;
    const GpuTri *tris = nullptr;
    const uint32_t indexMask = CACHED ? RT_CACHE_INDEX_MASK : 0x7FFFFFFFu;
    int sp = 0, blasBase = -1;
    uint32_t inst = 0, instFlags = 0;
    float instDepthBias = 0.0f;            // of the instance being walked: handed to the hit handler (no table lookup per hit)
    bool cull = false;
    uint32_t cur = 0;
    bool alive = true;
    RT_TRIP_DECL(tripsNode); RT_TRIP_DECL(tripsLeaf);
    auto popNext = [&]() -> bool {
        if (blasBase >= 0 && sp == blasBase) {          // BLAS exhausted: resume the TLAS walk in world space
            blasBase = -1; R = W; nodes = P.tlasNodes; nodeOff = tlasOff;
        }
        if (sp == 0) return false;
#ifdef RT_PROFILE_TRIPS
        if (!CACHED && sp > stk.ldsEntries) cnt.spills++;
#endif
        cur = stk.template pop<CACHED>(sp);
        return true;
    };
    while (alive) {
        // ---- inner nodes ----
        while (alive && !(cur & RT64_LEAF_BIT)) {
            RT_TRIP(tripsNode);
            const GpuNode nd = CACHED ? load_node_lds(stk.cache + nodeOff + 4u * cur) : load_node(nodes + cur);
            cnt.nodes++;
            float tl, tr;
            const bool hl = box_hit(R, nd.lmin, nd.lmax, tmin, tmax, tl);
            const bool hr = box_hit(R, nd.rmin, nd.rmax, tmin, tmax, tr);
            // same decisions as "both: push the farther, go to the nearer; one: go there; none: pop", as value selects with two
            // predicated regions (push, pop) instead of a five-way branch nest: fewer exec-mask round trips per visited node
            const bool both = hl && hr, rightFirst = tr < tl;
            const uint32_t nearChild = both ? (rightFirst ? nd.right : nd.left) : (hl ? nd.left : nd.right);
            if (both) stk.template push<CACHED>(sp, rightFirst ? nd.left : nd.right);
            if (hl || hr) cur = nearChild;
            else alive = popNext();
        }
        if (!alive) break;
        RT_TRIP(tripsLeaf);
        // ---- leaf ----
        if (cur != RT64_NO_CHILD) {
            if (blasBase < 0) {
                // TLAS leaf: enter the instance (G8)
                float oo[3], dd[3];
                uint32_t flags;
                if (CACHED) {
                    // 64-byte record of leaf slot `cur`: three words (M[c], M[4+c], M[8+c], M[12+c]), then (instance | flags << 8 | node offset << 16, depth bias, tris pointer)
                    const u32x4_lds *rec = stk.cache + 4u * (cur & indexMask);
                    const u32x4_lds info = rec[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const u32x4_lds mw = rec[c];
                        const float m0 = __uint_as_float(mw.x), m1 = __uint_as_float(mw.y), m2 = __uint_as_float(mw.z), m3 = __uint_as_float(mw.w);
                        oo[c] = fmaf(W.o[2], m2, fmaf(W.o[1], m1, fmaf(W.o[0], m0, m3)));
                        dd[c] = fmaf(W.d[2], m2, fmaf(W.d[1], m1, W.d[0] * m0));
                    }
                    inst = info.x & 0xFFu; flags = (info.x >> 8) & 0xFFu; nodeOff = info.x >> 16; instDepthBias = __uint_as_float(info.y);
                    tris = reinterpret_cast<const GpuTri *>(((uint64_t)info.w << 32) | (uint64_t)info.z);
                }
                else {
                    inst = load_global(P.tlasIndex + (cur & 0x7FFFFFFFu));
                    const GpuInstance *in = P.instances + inst;
                    // p * M with M row-major 4x4: column c of rows 0..3 = M[c], M[4+c], M[8+c], M[12+c]
                    const float *M = in->worldToObject;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float m0 = load_global(M + c), m1 = load_global(M + 4 + c), m2 = load_global(M + 8 + c), m3 = load_global(M + 12 + c);
                        oo[c] = fmaf(W.o[2], m2, fmaf(W.o[1], m1, fmaf(W.o[0], m0, m3)));
                        dd[c] = fmaf(W.d[2], m2, fmaf(W.d[1], m1, W.d[0] * m0));
                    }
                    nodes = load_global(&in->nodes); tris = load_global(&in->tris);
                    flags = load_global(&in->flags); instDepthBias = load_global(&in->material.depthBias);
                }
                make_ray_space(oo, dd, R);
                instFlags = flags;
                cull = cullBackFaces && !(flags & GPU_INST_CULL_DISABLE);
                blasBase = sp;
                cur = 0;
                continue;
            }
            const GpuTri tri = load_tri(tris + (cur & indexMask));
            cnt.tris++;
            float t, u, v;
            if (tri_hit(R, tri, cull, tmin, tmax, t, u, v))
                if (onHit(t, u, v, inst, tri.prim, tmax, instFlags, instDepthBias)) alive = false;
            if (!alive) break;
        }
        alive = popNext();
    }
#ifdef RT_PROFILE_TRIPS
    RT_TRIP_END(tripsNode); RT_TRIP_END(tripsLeaf);
    cnt.tripsNode += tripsNode; cnt.tripsLeaf += tripsLeaf;
#endif
}

// RayWalk: the same walk as trace_ray (same operations per ray, kept textually parallel) with its state in a struct so that it
// can be PAUSED.  RayWalk holds the state of one ray's walk in registers so that a walk can be PAUSED: run() returns either when the ray has
// finished (alive == false) or -- with MIN_LIVE > 0 -- at a leaf boundary as soon as fewer than MIN_LIVE lanes of the wave are
// still walking.  The caller then hands fresh rays to the finished lanes and calls run() again (persistent-thread traversal
// with wave-ballot refill, used for incoherent secondary rays where one long ray would otherwise hold 63 finished lanes).
struct RayWalk {
    RaySpace W, R;
    const GpuNode *nodes; const GpuTri *tris;
    float tmin, tmax;
    int sp, blasBase;
    uint32_t inst, cur;
    bool cull, alive;

    DEV void begin(PRef P, const float o[3], const float d[3], float tmin_, float tmax_) {
        make_ray_space(o, d, W);
        R = W;
        nodes = P.tlasNodes; tris = nullptr;
        tmin = tmin_; tmax = tmax_;
        sp = 0; blasBase = -1; inst = 0; cull = false; cur = 0;
        alive = P.instanceCount != 0;
    }

    DEV bool pop_next(PRef P, const TraceStack &stk) {
        if (blasBase >= 0 && sp == blasBase) {          // BLAS exhausted: resume the TLAS walk in world space
            blasBase = -1; R = W; nodes = P.tlasNodes;
        }
        if (sp == 0) return false;
        cur = stk.pop(sp);
        return true;
    }

    template <int MIN_LIVE = 0, class OnHit>
    DEV void run(PRef P, bool cullBackFaces, const TraceStack &stk, OnHit &&onHit, TraceCounts &cnt, bool mayPause = false) {
        while (alive) {
            // ---- inner nodes ----
            while (alive && !(cur & RT64_LEAF_BIT)) {
                const GpuNode nd = load_node(nodes + cur);
                cnt.nodes++;
                float tl, tr;
                const bool hl = box_hit(R, nd.lmin, nd.lmax, tmin, tmax, tl);
                const bool hr = box_hit(R, nd.rmin, nd.rmax, tmin, tmax, tr);
                const bool both = hl && hr, rightFirst = tr < tl;          // (value selects, as in trace_ray)
                const uint32_t nearChild = both ? (rightFirst ? nd.right : nd.left) : (hl ? nd.left : nd.right);
                if (both) stk.push(sp, rightFirst ? nd.left : nd.right);
                if (hl || hr) cur = nearChild;
                else alive = pop_next(P, stk);
            }
            if (!alive) break;
            // ---- leaf ----
            if (cur != RT64_NO_CHILD) {
                if (blasBase < 0) {
                    // TLAS leaf: enter the instance (G8)
                    inst = load_global(P.tlasIndex + (cur & 0x7FFFFFFFu));
                    const GpuInstance *in = P.instances + inst;
                    float oo[3], dd[3];
                    // p * M with M row-major 4x4: column c of rows 0..3 = M[c], M[4+c], M[8+c], M[12+c]
                    const float *M = in->worldToObject;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float m0 = load_global(M + c), m1 = load_global(M + 4 + c), m2 = load_global(M + 8 + c), m3 = load_global(M + 12 + c);
                        oo[c] = fmaf(W.o[2], m2, fmaf(W.o[1], m1, fmaf(W.o[0], m0, m3)));
                        dd[c] = fmaf(W.d[2], m2, fmaf(W.d[1], m1, W.d[0] * m0));
                    }
                    make_ray_space(oo, dd, R);
                    nodes = load_global(&in->nodes); tris = load_global(&in->tris);
                    cull = cullBackFaces && !(load_global(&in->flags) & GPU_INST_CULL_DISABLE);
                    blasBase = sp;
                    cur = 0;
                    continue;
                }
                const GpuTri tri = load_tri(tris + (cur & 0x7FFFFFFFu));
                cnt.tris++;
                float t, u, v;
                if (tri_hit(R, tri, cull, tmin, tmax, t, u, v))
                    if (onHit(t, u, v, inst, tri.prim, tmax)) { alive = false; return; }
            }
            alive = pop_next(P, stk);
            if (MIN_LIVE > 0 && mayPause && alive && __popcll(__ballot(1)) < MIN_LIVE) return;      // pause: let the caller refill finished lanes
        }
    }
};

// Wave-level add of the per-lane traversal counters into the frame counters (only when instrumentation is on).
DEV void flush_counts(PRef P, const TraceCounts &c, int pass) {
    if (!P.countTraversal) return;
    unsigned long long n = c.nodes, t = c.tris;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { n += __shfl_down(n, d, 64); t += __shfl_down(t, d, 64); }
    if ((threadIdx.x & 63) == 0 && (n | t)) {
        unsigned long long *ctr = P.counters + (size_t)(blockIdx.x % RT_COUNTER_STRIPES) * CTR_COUNT;
        atomicAdd(&ctr[CTR_NODES], n); atomicAdd(&ctr[CTR_TRIS], t);
        atomicAdd(&ctr[CTR_PASS_BASE + 2 * pass], n); atomicAdd(&ctr[CTR_PASS_BASE + 2 * pass + 1], t);
    }
}
