/* oracle_trace.c -- two-level LBVH traversal + ray/triangle test of the CPU oracle (TEST INFRASTRUCTURE).
 *
 * Replaces the DXR TraceRay black box at ref:shaders/PrimaryRayGen.hlsl:71, ref:shaders/Lights.hlsli:50,
 * ref:shaders/IndirectRayGen.hlsl:79, ref:shaders/ReflectionRayGen.hlsl:63, ref:shaders/RefractionRayGen.hlsl:59.
 * DXR semantics restated from the public DXR functional spec:
 *   - rays are NOT normalised; t is in units of |direction|; a hit needs TMin < t < TMax;
 *   - the ray is transformed into object space per instance (direction not renormalised, t unchanged);
 *   - facing is decided in object space: front-facing iff dot(cross(v1-v0, v2-v0), dir) < 0;
 *     RAY_FLAG_CULL_BACK_FACING_TRIANGLES drops back faces unless the instance has TRIANGLE_CULL_DISABLE
 *     (ref:private/rt64_view.cpp:1109, ref:contrib/nv_helpers_dx12/TopLevelASGenerator.cpp:189-202);
 *   - InstanceIndex() = position in the TLAS build list, PrimitiveIndex() = triangle number, barycentrics (u, v)
 *     weight v1 and v2;
 *   - any-hit order along the ray is unspecified by DXR; here it is the traversal order defined below.
 *
 * Geometry spec, part 2 (bit-exact contract with the HIP traversal kernel):
 *   R1 safe direction: ds_k = |d_k| < 1e-20f ? copysignf(1e-20f, d_k) : d_k;  inv_k = 1.0f / ds_k;  oi_k = -(o_k * inv_k).
 *   R2 box test against [lo, hi]: a_k = fmaf(lo_k, inv_k, oi_k), b_k = fmaf(hi_k, inv_k, oi_k);
 *      tn = max(max(min(a_x,b_x), min(a_y,b_y)), max(min(a_z,b_z), tmin));
 *      tf = min(min(max(a_x,b_x), max(a_y,b_y)), min(max(a_z,b_z), tmax)) * 1.0000004f;   hit iff tn <= tf.
 *   R3 order: depth first with an explicit stack.  At an inner node test left then right; if both hit, continue
 *      with the one with the smaller tn (ties: left) and push the other; if one hits continue with it; else pop.
 *      A TLAS leaf switches to object space (G8: o' = g_xform_point(worldToObject, o), d' = g_xform_vector(...))
 *      and walks the mesh's BLAS to exhaustion before the TLAS walk resumes.
 *   R4 triangle (Moller-Trumbore with fixed fma chains): e1 = v1-v0, e2 = v2-v0, p = g_cross3(d, e2),
 *      det = g_dot3(e1, p); culled if (cull ? det <= 0 : det == 0); inv = 1.0f/det; tv = o - v0;
 *      u = g_dot3(tv, p) * inv; q = g_cross3(tv, e1); v = g_dot3(d, q) * inv; t = g_dot3(e2, q) * inv;
 *      hit iff u >= 0 && v >= 0 && u + v <= 1 && t > tmin && t < tmax.
 *   R5 the any-hit callback may lower tmax for the rest of the walk (closest-hit shortening behind opaque hits:
 *      tmax = min(tmax, (t - depthBias_instance) + maxDepthBias), see oracle_render.c) or terminate it.
 */
#include <math.h>
#include <string.h>
#include "oracle_internal.h"

#define STACK_MAX 256

typedef struct {
    float o[3], d[3], inv[3], oi[3];
} RaySpace;

static void ray_space(const float o[3], const float d[3], RaySpace *r) {
    for (int k = 0; k < 3; k++) {
        r->o[k] = o[k]; r->d[k] = d[k];
        float ds = fabsf(d[k]) < 1e-20f ? copysignf(1e-20f, d[k]) : d[k];
        r->inv[k] = 1.0f / ds;
        r->oi[k] = -(o[k] * r->inv[k]);
    }
}

static inline int box_hit(const RaySpace *r, const float lo[3], const float hi[3], float tmin, float tmax, float *tnear) {
    float ax = fmaf(lo[0], r->inv[0], r->oi[0]), bx = fmaf(hi[0], r->inv[0], r->oi[0]);
    float ay = fmaf(lo[1], r->inv[1], r->oi[1]), by = fmaf(hi[1], r->inv[1], r->oi[1]);
    float az = fmaf(lo[2], r->inv[2], r->oi[2]), bz = fmaf(hi[2], r->inv[2], r->oi[2]);
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax)) * 1.0000004f;
    *tnear = tn;
    return tn <= tf;
}

static inline int tri_hit(const RaySpace *r, const OTri *tri, int cull, float tmin, float tmax, float *t, float *u, float *v) {
    float e1[3], e2[3], p[3], q[3], tv[3];
    for (int k = 0; k < 3; k++) { e1[k] = tri->v1[k] - tri->v0[k]; e2[k] = tri->v2[k] - tri->v0[k]; }
    g_cross3(r->d, e2, p);
    float det = g_dot3(e1, p);
    if (cull ? !(det > 0.0f) : (det == 0.0f || det != det)) return 0;
    float inv = 1.0f / det;
    for (int k = 0; k < 3; k++) tv[k] = r->o[k] - tri->v0[k];
    float uu = g_dot3(tv, p) * inv;
    g_cross3(tv, e1, q);
    float vv = g_dot3(r->d, q) * inv;
    float tt = g_dot3(e2, q) * inv;
    if (!(uu >= 0.0f) || !(vv >= 0.0f) || !(uu + vv <= 1.0f) || !(tt > tmin) || !(tt < tmax)) return 0;
    *t = tt; *u = uu; *v = vv;
    return 1;
}

/* Walk one BLAS.  Returns 1 when the walk must terminate the whole trace. */
static int walk_blas(const OMesh *mesh, const RaySpace *r, int cull, float tmin, float *tmax, uint32_t instance,
                     OAnyHitFn fn, void *user, OTraceCounters *ctr) {
    const OBvh *b = &mesh->bvh;
    uint32_t stack[STACK_MAX]; int sp = 0;
    uint32_t cur = 0;                                   /* inner node index, or leaf with bit 31 */
    for (;;) {
        if (cur & 0x80000000u) {
            if (cur != 0xFFFFFFFFu) {
                const OTri *tri = &mesh->tris[cur & 0x7FFFFFFFu];
                float t, u, v;
                if (ctr) ctr->tris++;
                if (tri_hit(r, tri, cull, tmin, *tmax, &t, &u, &v)) {
                    OHit h = { t, u, v, instance, tri->prim };
                    int terminate = 0;
                    fn(user, &h, tmax, &terminate);
                    if (terminate) return 1;
                }
            }
            if (sp == 0) return 0;
            cur = stack[--sp];
            continue;
        }
        const ONode *nd = &b->nodes[cur];
        if (ctr) ctr->nodes++;
        float tl, tr;
        int hl = box_hit(r, nd->lmin, nd->lmax, tmin, *tmax, &tl);
        int hr = box_hit(r, nd->rmin, nd->rmax, tmin, *tmax, &tr);
        if (hl && hr) {
            if (tr < tl) { stack[sp++] = nd->left; cur = nd->right; }
            else { stack[sp++] = nd->right; cur = nd->left; }
        }
        else if (hl) cur = nd->left;
        else if (hr) cur = nd->right;
        else {
            if (sp == 0) return 0;
            cur = stack[--sp];
        }
    }
}

static int visit_instance(const OScene *s, uint32_t instance, const ORay *ray, float *tmax, int bruteForce,
                          OAnyHitFn fn, void *user, OTraceCounters *ctr) {
    const OInst *in = &s->rt[instance];
    float o[3], d[3];
    g_xform_point(&in->worldToObject, ray->o, o);
    g_xform_vector(&in->worldToObject, ray->d, d);
    RaySpace r; ray_space(o, d, &r);
    int cull = ray->cullBackFaces && !in->cullDisable;
    const OMesh *mesh = in->desc.mesh;
    if (!bruteForce) return walk_blas(mesh, &r, cull, ray->tmin, tmax, instance, fn, user, ctr);
    for (uint32_t k = 0; k < mesh->bvh.count; k++) {
        float t, u, v;
        if (ctr) ctr->tris++;
        if (tri_hit(&r, &mesh->tris[k], cull, ray->tmin, *tmax, &t, &u, &v)) {
            OHit h = { t, u, v, instance, mesh->tris[k].prim };
            int terminate = 0;
            fn(user, &h, tmax, &terminate);
            if (terminate) return 1;
        }
    }
    return 0;
}

void otrace(const OScene *s, const ORay *ray, int bruteForce, OAnyHitFn fn, void *user, OTraceCounters *ctr) {
    float tmax = ray->tmax;
    if (s->rtCount == 0) return;
    if (bruteForce) {
        for (int i = 0; i < s->rtCount; i++)
            if (visit_instance(s, (uint32_t)i, ray, &tmax, 1, fn, user, ctr)) return;
        return;
    }
    RaySpace r; ray_space(ray->o, ray->d, &r);
    const OBvh *b = &s->tlas;
    uint32_t stack[STACK_MAX]; int sp = 0;
    uint32_t cur = 0;
    for (;;) {
        if (cur & 0x80000000u) {
            if (cur != 0xFFFFFFFFu) {
                uint32_t instance = b->sortedIndex[cur & 0x7FFFFFFFu];
                if (visit_instance(s, instance, ray, &tmax, 0, fn, user, ctr)) return;
            }
            if (sp == 0) return;
            cur = stack[--sp];
            continue;
        }
        const ONode *nd = &b->nodes[cur];
        if (ctr) ctr->nodes++;
        float tl, tr;
        int hl = box_hit(&r, nd->lmin, nd->lmax, ray->tmin, tmax, &tl);
        int hr = box_hit(&r, nd->rmin, nd->rmax, ray->tmin, tmax, &tr);
        if (hl && hr) {
            if (tr < tl) { stack[sp++] = nd->left; cur = nd->right; }
            else { stack[sp++] = nd->right; cur = nd->left; }
        }
        else if (hl) cur = nd->left;
        else if (hr) cur = nd->right;
        else {
            if (sp == 0) return;
            cur = stack[--sp];
        }
    }
}
