/* oracle_render.c -- frame orchestration and ray-generation passes of the CPU oracle (TEST INFRASTRUCTURE).
 *
 * Restates, pass by pass and with the reference's storage formats as quantisation points:
 *   View::update               ref:private/rt64_view.cpp:1053-1178   (instance partition, TLAS, instance buffers)
 *   updateGlobalParamsBuffer   ref:private/rt64_view.cpp:961-1028    (matrices, camera vectors, reprojection flags)
 *   View::render               ref:private/rt64_view.cpp:1321-1667   (pass order, ping-pong, frameCount)
 *   PrimaryRayGen              ref:shaders/PrimaryRayGen.hlsl:31-198
 *   DirectRayGen               ref:shaders/DirectRayGen.hlsl:14-65
 *   IndirectRayGen             ref:shaders/IndirectRayGen.hlsl:31-137
 *   RefractionRayGen           ref:shaders/RefractionRayGen.hlsl:19-117
 *   ReflectionRayGen           ref:shaders/ReflectionRayGen.hlsl:25-143
 *   GaussianFilterRGB3x3CS     ref:shaders/GaussianFilterRGB3x3CS.hlsl:21-82 (driver ref:rt64_view.cpp:1488-1530)
 *   ComposePS / PostProcessPS  ref:shaders/ComposePS.hlsl:18-37, ref:shaders/PostProcessPS.hlsl:13-36
 * Image formats (ref:rt64_view.cpp:152-241): see the q_* calls at every store.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle_shade.h"

void osvgf_filter(OScene *s, const OFrameParams *p, int cur);   /* oracle_svgf.c */

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

/* ---- matrices -------------------------------------------------------------------------------------------------- */

/* General 4x4 inverse by cofactors in double precision, rounded once to float.  (The reference uses
 * DirectXMath's float XMMatrixInverse, ref:rt64_view.cpp:981-984,368; its exact rounding is not reproducible
 * without DirectXMath.)  The HIP host side uses the same formula text so both produce identical floats. */
int omatrix_inverse_d(const om4 *M, om4 *out) {
    double m[16], inv[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m[i * 4 + j] = (double)M->m[i][j];
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
    if (det == 0.0) { memset(out, 0, sizeof(*out)); return 0; }
    double r = 1.0 / det;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out->m[i][j] = (float)(inv[i * 4 + j] * r);
    return 1;
}
int oracle_matrix_inverse(const om4 *m, om4 *out) { return omatrix_inverse_d(m, out); }

/* XMMatrixPerspectiveFovRH(fov, aspect, zn, zf) (called at ref:rt64_view.cpp:1766); rows per SURVEY appendix A1. */
void oracle_perspective_fov_rh(float fov, float aspect, float zn, float zf, om4 *P) {
    float s = sinf(0.5f * fov), c = cosf(0.5f * fov);
    float h = c / s, w = h / aspect, range = zf / (zn - zf);
    memset(P, 0, sizeof(*P));
    P->m[0][0] = w; P->m[1][1] = h; P->m[2][2] = range; P->m[2][3] = -1.0f; P->m[3][2] = range * zn;
}

/* ---- scene object ---------------------------------------------------------------------------------------------- */

OScene *oracle_scene_create(void) {
    OScene *s = (OScene *)calloc(1, sizeof(OScene));
    /* Scene constructor defaults, ref:private/rt64_scene.cpp:19-27 */
    s->desc.eyeLightDiffuseColor = v3s(0.08f); s->desc.eyeLightSpecularColor = v3s(0.04f);
    s->desc.skyDiffuseMultiplier = v3s(1.0f); s->desc.giDiffuseStrength = 0.7f; s->desc.giSkyStrength = 0.35f;
    return s;
}

static void free_images(OScene *s) {
    free(s->outputRGBA32F); free(s->shadingPosition); free(s->shadingNormal); free(s->shadingSpecular);
    free(s->diffuse); free(s->instanceId); free(s->reflection); free(s->refraction); free(s->transparent); free(s->viewDirection);
    free(s->flow); free(s->reactiveMask); free(s->lockMask); free(s->primaryHit);
    for (int i = 0; i < 2; i++) {
        free(s->directLight[i]); free(s->indirectLight[i]); free(s->filteredDirect[i]); free(s->filteredIndirect[i]);
        free(s->normal[i]); free(s->depth[i]); free(s->moments[i]);
    }
}

void oracle_scene_destroy(OScene *s) {
    if (!s) return;
    free_images(s); free(s->upscaled[0]); free(s->upscaled[1]); free(s->finalRGBA8); free(s->backgroundRGBA8); free(s->blueNoise); free(s->instances); free(s->rt); obvh_free(&s->tlas); free(s);
}

void oracle_scene_set_desc(OScene *s, const OSceneDesc *d) { s->desc = *d; }
void oracle_scene_set_lights(OScene *s, const OLight *l, int n) { if (n > 64) n = 64; memcpy(s->lights, l, sizeof(OLight) * (size_t)n); s->lightCount = n; }
void oracle_scene_set_bluenoise(OScene *s, const uint8_t *p) { free(s->blueNoise); s->blueNoise = (uint8_t *)malloc(512 * 512 * 4); memcpy(s->blueNoise, p, 512 * 512 * 4); }
void oracle_scene_set_sky(OScene *s, OTexture *t) { s->sky = t; }
int oracle_scene_add_instance(OScene *s, const OInstanceDesc *d) {
    if (s->instanceCount == s->instanceCap) { s->instanceCap = s->instanceCap ? s->instanceCap * 2 : 16; s->instances = (OInstanceDesc *)realloc(s->instances, sizeof(OInstanceDesc) * (size_t)s->instanceCap); }
    s->instances[s->instanceCount] = *d;
    return s->instanceCount++;
}
void oracle_scene_set_instance(OScene *s, int i, const OInstanceDesc *d) { s->instances[i] = *d; }
const OBvh *oracle_scene_tlas(const OScene *s) { return &s->tlas; }
uint32_t oracle_scene_frame_count(const OScene *s) { return s->frameCount; }

static void alloc_images(OScene *s, int w, int h) {
    if (s->imgW == w && s->imgH == h) return;
    free_images(s);
    size_t n = (size_t)w * (size_t)h;
#define A4(p) p = (float *)calloc(n * 4, sizeof(float))
    A4(s->outputRGBA32F); A4(s->shadingPosition); A4(s->shadingNormal); A4(s->shadingSpecular); A4(s->diffuse);
    A4(s->reflection); A4(s->refraction); A4(s->transparent); A4(s->viewDirection);
    s->instanceId = (int32_t *)calloc(n, sizeof(int32_t));
    s->flow = (float *)calloc(n * 2, sizeof(float));
    s->reactiveMask = (float *)calloc(n, sizeof(float)); s->lockMask = (float *)calloc(n, sizeof(float));
    s->primaryHit = (uint32_t *)calloc(n * 4, sizeof(uint32_t));
    for (int i = 0; i < 2; i++) {
        A4(s->directLight[i]); A4(s->indirectLight[i]); A4(s->filteredDirect[i]); A4(s->filteredIndirect[i]); A4(s->normal[i]);
        A4(s->moments[i]);
        s->depth[i] = (float *)calloc(n, sizeof(float));
    }
#undef A4
    s->imgW = w; s->imgH = h;
    s->haveHistory = 0;                                            /* rtSkipReprojection = true, ref:rt64_view.cpp:143 */
}

/* ---- View::update ------------------------------------------------------------------------------------------------ */

/* Static opacity rule O1 (shared with the HIP host side): an instance is "opaque" when every hit it can produce
 * stores alpha 255 in the RGBA8 hit colour, so nothing behind it can contribute (PrimaryRayGen.hlsl:150,174).
 * Decided from bounds on the combiner's alpha sources; only used when cullBehindOpaque is on. */
static void alpha_source_bounds(const OInst *in, int item, float *lo, float *hi) {
    const OCombiner *cc = &in->cc;
    const OMesh *mesh = in->desc.mesh;
    switch (item) {
    default: case 0: *lo = *hi = 0.0f; return;
    case 1: case 2: case 3: case 4: {
        if (!cc->opt_alpha) { *lo = *hi = 1.0f; return; }
        float mn = INFINITY, mx = -INFINITY;
        for (int v = 0; v < mesh->vertexCount; v++) {
            float a; memcpy(&a, mesh->vertices + (size_t)v * (size_t)cc->vertexSize + (size_t)cc->inputOffset[item - 1] + 12, 4);
            if (!(a >= mn)) mn = a;                                  /* NaN poisons both bounds */
            if (!(a <= mx)) mx = a;
        }
        *lo = mn; *hi = mx; return;
    }
    case 5: case 6: {
        const OTexture *t = in->desc.diffuse;
        uint8_t mn = 255, mx = 0;
        for (int m = 0; m < t->mips; m++) {
            size_t n = (size_t)t->w[m] * (size_t)t->h[m];
            for (size_t i = 0; i < n; i++) { uint8_t a = t->rgba[m][4 * i + 3]; if (a < mn) mn = a; if (a > mx) mx = a; }
        }
        *lo = (float)mn / 255.0f; *hi = (float)mx / 255.0f; return;
    }
    case 7: *lo = *hi = 1.0f; return;
    }
}

/* Lower bound of the combiner's alpha over the whole instance, or a negative value when nothing is proven. */
static float instance_alpha_lower_bound(const OInst *in) {
    const OCombiner *cc = &in->cc;
    if (cc->opt_noise || cc->opt_texture_edge) return -1.0f;
    float lo;
    if (!cc->opt_alpha) lo = 1.0f;                                  /* float4(..., 1.0f) everywhere, rt64_shader.cpp:232-256 */
    else {
        float l[4], h[4];
        for (int k = 0; k < 4; k++) alpha_source_bounds(in, cc->c[1][k], &l[k], &h[k]);
        if (cc->do_single[1]) lo = l[3];
        else if (cc->do_multiply[1]) lo = fminf(fminf(l[0] * l[2], l[0] * h[2]), fminf(h[0] * l[2], h[0] * h[2]));
        else return -1.0f;                                          /* mix / general formula: not proven, use the k-buffer path */
    }
    return lo;
}

static int instance_is_opaque(const OInst *in) {
    float lo = instance_alpha_lower_bound(in);
    return lo >= 0.0f && in->desc.material.solidAlphaMultiplier * lo >= 0.999f;     /* to_unorm8(0.999) == 255 */
}

/* Rule O2 (shared with the HIP host side): the shadow any-hit (rt64_shader.cpp:611-659) subtracts
 * clamp(alpha * shadowAlphaMultiplier) from payload.shadowHit and ends the search at 0.  When that product is provably
 * >= 0.999 for every hit of the instance, the first hit is treated as saturating (deviation from the reference <= 1e-3 in the
 * shadow factor, and only for alphas within 1e-3 of 1).  A combiner without opt_alpha always saturates (:661). */
static int instance_is_shadow_opaque(const OInst *in) {
    if (!in->cc.opt_alpha) return 1;
    float lo = instance_alpha_lower_bound(in);
    return lo >= 0.0f && in->desc.material.shadowAlphaMultiplier * lo >= 0.999f;
}

/* `still`: a later sub-frame of a primarySpp frame (rule P2): nothing has moved since the sub-frame before it */
static void update_view(OScene *s, const OFrameParams *p, int still) {
    free(s->rt); s->rt = (OInst *)calloc((size_t)(s->instanceCount > 0 ? s->instanceCount : 1), sizeof(OInst)); s->rtCount = 0;
    for (int i = 0; i < s->instanceCount; i++) {
        const OInstanceDesc *d = &s->instances[i];
        if (!d->mesh || !(d->mesh->flags & 0x1) || d->mesh->bvh.count == 0) continue;   /* BLAS present -> RT instance, ref:rt64_view.cpp:1138 */
        OInst *in = &s->rt[s->rtCount++];
        in->desc = *d; in->sceneIndex = i;
        ocombiner_decode(d->shaderId, &in->cc);
        in->objectToWorld = d->transform; in->objectToWorldPrevious = still ? d->transform : d->previousTransform;
        om4 upper = d->transform;                                   /* ref:rt64_view.cpp:358-368 */
        upper.m[0][3] = upper.m[1][3] = upper.m[2][3] = 0.0f; upper.m[3][0] = upper.m[3][1] = upper.m[3][2] = 0.0f; upper.m[3][3] = 1.0f;
        om4 inv; omatrix_inverse_d(&upper, &inv);
        m4_transpose(&inv, &in->objectToWorldNormal);
        omatrix_inverse_d(&d->transform, &in->worldToObject);
        in->cullDisable = (d->flags & 0x2) != 0;                    /* RT64_INSTANCE_DISABLE_BACKFACE_CULLING */
        in->opaque = p->cullBehindOpaque ? instance_is_opaque(in) : 0;
        in->shadowOpaque = p->cullBehindOpaque ? instance_is_shadow_opaque(in) : !in->cc.opt_alpha;
    }
    /* TLAS over world boxes of the instances (G1/G7). */
    obvh_free(&s->tlas);
    if (s->rtCount > 0) {
        float *bmin = (float *)malloc(sizeof(float) * 3 * (size_t)s->rtCount), *bmax = (float *)malloc(sizeof(float) * 3 * (size_t)s->rtCount);
        for (int i = 0; i < s->rtCount; i++) {
            const OBvh *b = &s->rt[i].desc.mesh->bvh;
            float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (int c = 0; c < 8; c++) {
                float pnt[3] = { (c & 1) ? b->bmax[0] : b->bmin[0], (c & 2) ? b->bmax[1] : b->bmin[1], (c & 4) ? b->bmax[2] : b->bmin[2] }, w[3];
                g_xform_point(&s->rt[i].objectToWorld, pnt, w);
                for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], w[k]); mx[k] = fmaxf(mx[k], w[k]); }
            }
            memcpy(bmin + 3 * i, mn, 12); memcpy(bmax + 3 * i, mx, 12);
        }
        obvh_build(&s->tlas, (uint32_t)s->rtCount, bmin, bmax, NULL, NULL);
        free(bmin); free(bmax);
    }
}

static void update_global_params(OScene *s, const OFrameParams *p, int screenW, int screenH, OShadeCtx *c) {
    float aspect = (float)screenW / (float)screenH;                /* Device::getAspectRatio, ref:rt64_device.cpp:211 */
    s->view = p->view;
    oracle_perspective_fov_rh(p->fovRadians, aspect, p->nearDist, p->farDist, &s->projection);
    /* ref:rt64_view.cpp:975-990.  On the very first frame the reference's "previous" matrices are uninitialised
     * memory; the oracle defines them as the current ones. */
    int reproject = p->canReproject && s->matricesValid;
    if (reproject) { s->prevViewI = s->viewI; s->prevViewProj = s->viewProj; }
    omatrix_inverse_d(&s->view, &s->viewI);
    omatrix_inverse_d(&s->projection, &s->projectionI);
    m4_mul(&s->view, &s->projection, &s->viewProj);
    if (!reproject) { s->prevViewI = s->viewI; s->prevViewProj = s->viewProj; }
    s->matricesValid = 1;

    memset(c, 0, sizeof(*c));
    c->scene = s; c->rt = s->rt; c->rtCount = s->rtCount; c->lights = s->lights; c->lightCount = s->lightCount;
    c->desc = s->desc; c->blueNoise = s->blueNoise; c->sky = s->sky; c->background = NULL;
    c->view = s->view; c->viewI = s->viewI; c->prevViewI = s->prevViewI; c->projection = s->projection; c->projectionI = s->projectionI;
    c->viewProj = s->viewProj; c->prevViewProj = s->prevViewProj;
    /* Pinhole vectors for the ray differentials only, ref:rt64_view.cpp:992-1009 (getViewDirection uses view-space +z, :1798). */
    float focal = (p->nearDist + p->farDist) / 2.0f;
    of3 pos = m4_point(&s->viewI, v3s(0.0f));
    of3 dir = m4_vector(&s->viewI, v3(0.0f, 0.0f, 1.0f));
    { float l = v3len(dir); dir = v3(dir.x / l, dir.y / l, dir.z / l); }
    of3 target = v3add(pos, v3scale(dir, focal));
    of3 tw = v3sub(target, pos); { float l = v3len(tw); if (l > 0.0f) tw = v3(tw.x / l, tw.y / l, tw.z / l); }
    of3 W = v3scale(tw, focal);
    of3 U = v3cross(W, v3(0.0f, 1.0f, 0.0f)); { float l = v3len(U); if (l > 0.0f) U = v3(U.x / l, U.y / l, U.z / l); }
    of3 V = v3cross(U, W); { float l = v3len(V); if (l > 0.0f) V = v3(V.x / l, V.y / l, V.z / l); }
    float ulen = focal * tanf(p->fovRadians * 0.5f) * aspect, vlen = focal * tanf(p->fovRadians * 0.5f);
    c->cameraU = v3scale(U, ulen); c->cameraV = v3scale(V, vlen); c->cameraW = W;
    c->viewportW = (float)screenW; c->viewportH = (float)screenH;
    c->width = p->width; c->height = p->height; c->screenW = screenW; c->screenH = screenH;     /* resolution.xy = render size, .zw = screen size (ref:rt64_view.cpp:145-148) */
    c->pixelJitter.x = c->pixelJitter.y = 0.0f;                     /* jitter only with an upscaler, ref:rt64_view.cpp:1273-1281 */
    c->frameCount = s->frameCount; c->diSamples = p->diSamples; c->giSamples = p->giSamples; c->maxLights = p->maxLights; c->giBounces = p->giBounces;
    c->diReproject = 0;                                             /* DI_REPROJECTION_SUPPORT undefined, ref:rt64_view.cpp:1012-1016 */
    c->giReproject = (s->haveHistory && p->denoiserEnabled && p->giSamples > 0) ? 1 : 0;   /* :1017 */
    c->binaryLockMask = 1;                                          /* rtUpscaleMode != FSR, :1018 */
    c->bruteForce = p->bruteForce; c->cullBehindOpaque = p->cullBehindOpaque;
    float mb = -INFINITY;
    for (int i = 0; i < s->rtCount; i++) mb = fmaxf(mb, s->rt[i].desc.material.depthBias);
    c->maxDepthBias = s->rtCount ? mb : 0.0f;
}

/* ---- the per-pixel hit list (k-buffer) ----------------------------------------------------------------------------- */

typedef struct {
    OShadeCtx *c;
    of3 rayDir; ORayDiff rayDiff;
    uint32_t px, py;
    uint32_t nhits;
    OHitRecord list[O_MAX_HIT_QUERIES + 1];                         /* MaxQueries = 16 + 1, ref:rt64_view.cpp:25 */
} SurfacePayload;

/* ref:rt64_shader.cpp:547-581 (sorted insertion, strict '<' keeps first-come order on ties). */
static int surface_cb(void *user, const OHit *hit, float *tmax, int *terminate) {
    SurfacePayload *p = (SurfacePayload *)user;
    (void)terminate;
    OHitRecord rec;
    if (!oshade_surface_anyhit(p->c, hit, p->rayDir, p->rayDiff, p->px, p->py, &rec)) return 0;
    uint32_t hi = p->nhits < O_MAX_HIT_QUERIES ? p->nhits : O_MAX_HIT_QUERIES;
    while (hi > 0 && rec.dist < p->list[hi - 1].dist) { p->list[hi] = p->list[hi - 1]; hi--; }
    int accepted = 0;
    if (hi < O_MAX_HIT_QUERIES) {
        p->list[hi] = rec;
        ++p->nhits;
        if (hi == O_MAX_HIT_QUERIES - 1) { if (hit->t < *tmax) *tmax = hit->t; accepted = 1; }   /* not IgnoreHit(): the driver commits t */
    }
    if (p->c->cullBehindOpaque && p->c->rt[hit->instance].opaque) {
        /* R5: nothing with a larger sort key can contribute behind a fully opaque record. */
        float lim = rec.dist + p->c->maxDepthBias;
        if (lim < *tmax) *tmax = lim;
        accepted = 1;
    }
    return accepted;
}

static void trace_surface(OShadeCtx *c, of3 o, of3 d, ORayDiff rd, uint32_t px, uint32_t py, SurfacePayload *pl, uint64_t *nodes, uint64_t *tris) {
    pl->c = c; pl->rayDir = d; pl->rayDiff = rd; pl->px = px; pl->py = py; pl->nhits = 0;
    ORay ray = { { o.x, o.y, o.z }, { d.x, d.y, d.z }, O_RAY_MIN_DISTANCE, O_RAY_MAX_DISTANCE, 1 };
    OTraceCounters ctr = { 0, 0 };
    otrace(c->scene, &ray, c->bruteForce, surface_cb, pl, &ctr);
    *nodes += ctr.nodes; *tris += ctr.tris;
}

/* Reads of the hit list beyond the 17 allocated slots return zeros (out-of-bounds typed UAV load). */
static const OHitRecord *hit_slot(const SurfacePayload *pl, uint32_t hit) {
    static const OHitRecord zero;
    return hit <= O_MAX_HIT_QUERIES ? &pl->list[hit] : &zero;
}
static of4 rec_color(const OHitRecord *r) { of4 c = { from_unorm8(r->color[0]), from_unorm8(r->color[1]), from_unorm8(r->color[2]), from_unorm8(r->color[3]) }; return c; }
static of3 rec_normal(const OHitRecord *r) { return v3(from_snorm16(r->normal[0]), from_snorm16(r->normal[1]), from_snorm16(r->normal[2])); }
static of3 rec_specular(const OHitRecord *r) { return v3(from_unorm8(r->specular[0]), from_unorm8(r->specular[1]), from_unorm8(r->specular[2])); }

static of2 world_to_screen(const om4 *viewProj, of3 p) {           /* PrimaryRayGen.hlsl:19-23 */
    of4 v = { p.x, p.y, p.z, 1.0f };
    of4 clip = m4_mul_vec(viewProj, v);
    of2 r = { 0.5f + (clip.x / clip.w) / 2.0f, 0.5f + (clip.y / clip.w) / 2.0f };
    return r;
}

static float fresnel_reflect_amount(of3 normal, of3 incident, float reflectivity, float fresnelMultiplier) {   /* :25-29 */
    float ret = powf(fclampf(1.0f + v3dot(normal, incident), O_EPSILON, 1.0f), 5.0f);
    return reflectivity + ((1.0f - reflectivity) * ret * fresnelMultiplier);
}

static void primary_ray(const OShadeCtx *c, uint32_t px, uint32_t py, of3 *origin, of3 *dir, of2 *dOut) {   /* :33-39 */
    of2 d = { (((float)px + 0.5f + c->pixelJitter.x) / (float)c->width) * 2.0f - 1.0f,
              (((float)py + 0.5f + c->pixelJitter.y) / (float)c->height) * 2.0f - 1.0f };
    of4 tin = { d.x, -d.y, 1.0f, 1.0f };
    of4 target = m4_mul_vec(&c->projectionI, tin);
    *origin = m4_point(&c->viewI, v3s(0.0f));
    *dir = m4_vector(&c->viewI, v3(target.x, target.y, target.z));
    if (dOut) *dOut = d;
}

static inline void st4(float *img, size_t i, float x, float y, float z, float w) { img[4 * i] = x; img[4 * i + 1] = y; img[4 * i + 2] = z; img[4 * i + 3] = w; }
static inline of3 ld3i(const float *img, size_t i) { return v3(img[4 * i], img[4 * i + 1], img[4 * i + 2]); }

/* ---- PrimaryRayGen ----------------------------------------------------------------------------------------------- */

static void pass_primary(OScene *s, OShadeCtx *c, uint32_t px, uint32_t py, int cur) {
    size_t i = (size_t)py * (size_t)c->width + px;
    of3 rayOrigin, rayDirection; of2 d;
    primary_ray(c, px, py, &rayOrigin, &rayDirection, &d);
    of3 nonNormRayDir = v3add(v3add(v3scale(c->cameraU, d.x), v3scale(c->cameraV, d.y)), c->cameraW);
    st4(s->viewDirection, i, q_f16(rayDirection.x), q_f16(rayDirection.y), q_f16(rayDirection.z), 0.0f);
    float reflRGBA[4] = { 0, 0, 0, 0 }, refrRGBA[4] = { 0, 0, 0, 0 };

    of2 screenUV = { ((float)px + c->pixelJitter.x) / (float)c->width, ((float)py + c->pixelJitter.y) / (float)c->height };
    of3 bgColor = oshade_sample_background_2d(c, screenUV);
    of4 skyColor = oshade_sample_sky_2d(c, screenUV);
    of3 bgPosition = v3add(rayOrigin, v3scale(rayDirection, O_RAY_MAX_DISTANCE));
    of2 prevBgPos = world_to_screen(&c->prevViewProj, bgPosition), curBgPos = world_to_screen(&c->viewProj, bgPosition);
    bgColor = v3lerp(bgColor, v3(skyColor.x, skyColor.y, skyColor.z), skyColor.w);

    ORayDiff rayDiff; memset(&rayDiff, 0, sizeof(rayDiff));
    of2 vpDims = { (float)c->screenW, (float)c->screenH };
    oshade_compute_ray_diffs(nonNormRayDir, c->cameraU, c->cameraV, vpDims, &rayDiff.dDdx, &rayDiff.dDdy);

    SurfacePayload pl;
    trace_surface(c, rayOrigin, rayDirection, rayDiff, px, py, &pl, &c->nodesPrimary, &c->trisPrimary);
    c->primaryRays++;

    of3 resPosition = v3s(0.0f), resNormal = v3neg(rayDirection), resSpecular = v3s(0.0f), resTransparent = v3s(0.0f), resTransparentLight = v3s(0.0f);
    int resTransparentLightComputed = 0;
    of4 resColor = { 0, 0, 0, 1 };
    of2 resFlow = { (curBgPos.x - prevBgPos.x) * (float)c->width, (curBgPos.y - prevBgPos.y) * (float)c->height };
    float resReactiveMask = 0.0f, resLockMask = 0.0f, resDepth = 1.0f;
    int resInstanceId = -1;
    uint32_t firstHit[4] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu };
    if (pl.nhits > 0) {
        const OHit *g = &pl.list[0].geo;
        firstHit[0] = f2u(g->t); firstHit[1] = f2u(g->u); firstHit[2] = f2u(g->v); firstHit[3] = (g->instance << 24) | (g->prim & 0xFFFFFFu);
    }
    of3 ambient = v3add(c->desc.ambientBaseColor, c->desc.ambientNoGIColor);
    for (uint32_t hit = 0; hit < pl.nhits; hit++) {
        const OHitRecord *r = hit_slot(&pl, hit);
        of4 hitColor = rec_color(r);
        float alphaContrib = resColor.w * hitColor.w;
        if (alphaContrib >= O_EPSILON) {
            uint32_t instanceId = r->instanceId;
            const OMaterial *m = &c->rt[instanceId].desc.material;
            resLockMask += m->lockMask * alphaContrib;
            int usesLighting = m->lightGroupMaskBits > 0;
            int applyLighting = usesLighting && (hitColor.w > O_APPLY_LIGHTS_MINIMUM_ALPHA);
            of3 vertexPosition = v3add(rayOrigin, v3scale(rayDirection, r->dist + m->depthBias));   /* WithoutDistanceBias */
            of3 vertexNormal = rec_normal(r);
            of3 specular = v3mul(m->specularColor, rec_specular(r));
            int storeHit = 0;
            if (m->fogEnabled) {
                of4 fog = oshade_fog_from_camera(c, instanceId, vertexPosition);
                resTransparent = v3add(resTransparent, v3scale(v3(fog.x, fog.y, fog.z), fog.w * alphaContrib));
                alphaContrib *= (1.0f - fog.w);
            }
            if (m->reflectionFactor > O_EPSILON) {
                float fresnelAmount = fresnel_reflect_amount(vertexNormal, rayDirection, m->reflectionFactor, m->reflectionFresnelFactor);
                float reflectAmount = fresnelAmount * alphaContrib;
                reflRGBA[3] = reflectAmount;
                alphaContrib *= (1.0f - fresnelAmount);
                storeHit = 1;
                resLockMask += reflectAmount;
            }
            of3 resColorAdd = v3scale(v3(hitColor.x, hitColor.y, hitColor.z), alphaContrib);
            if (applyLighting) {
                storeHit = 1;
                resColor.x += resColorAdd.x; resColor.y += resColorAdd.y; resColor.z += resColorAdd.z;
            }
            else if (usesLighting) {
                if (!resTransparentLightComputed) {
                    resTransparentLight = oshade_lights_random(c, px, py, rayDirection, instanceId, vertexPosition, vertexNormal, specular, 1, 1);
                    resTransparentLightComputed = 1;
                }
                resTransparent = v3add(resTransparent, v3mul(resColorAdd, v3add(v3add(ambient, m->selfLight), resTransparentLight)));
            }
            else resTransparent = v3add(resTransparent, v3mul(resColorAdd, v3add(ambient, m->selfLight)));
            resColor.w *= (1.0f - hitColor.w);
            if (m->refractionFactor > O_EPSILON) {
                storeHit = 1;
                refrRGBA[3] = resColor.w;
                resColor.w = 0.0f;
            }
            if (storeHit && resInstanceId < 0) {
                of2 prevPos = world_to_screen(&c->prevViewProj, v3sub(vertexPosition, r->flow));
                of2 curPos = world_to_screen(&c->viewProj, vertexPosition);
                of4 vp4 = { vertexPosition.x, vertexPosition.y, vertexPosition.z, 1.0f };
                of4 projPos = m4_mul_vec(&c->viewProj, vp4);
                resPosition = vertexPosition; resNormal = vertexNormal; resSpecular = specular; resInstanceId = (int)instanceId;
                resFlow.x = (curPos.x - prevPos.x) * (float)c->width; resFlow.y = (curPos.y - prevPos.y) * (float)c->height;
                resDepth = projPos.z / projPos.w;
            }
        }
        if (resColor.w <= O_EPSILON) break;
    }
    resReactiveMask += fmaxf(resTransparent.x, fmaxf(resTransparent.y, resTransparent.z));
    resColor.x += bgColor.x * resColor.w; resColor.y += bgColor.y * resColor.w; resColor.z += bgColor.z * resColor.w;
    resColor.w = 1.0f - resColor.w;

    st4(s->reflection, i, q_f16(reflRGBA[0]), q_f16(reflRGBA[1]), q_f16(reflRGBA[2]), q_f16(reflRGBA[3]));
    st4(s->refraction, i, q_f16(refrRGBA[0]), q_f16(refrRGBA[1]), q_f16(refrRGBA[2]), q_f16(refrRGBA[3]));
    st4(s->shadingPosition, i, resPosition.x, resPosition.y, resPosition.z, 0.0f);                       /* RGBA32F */
    st4(s->shadingNormal, i, q_f16(resNormal.x), q_f16(resNormal.y), q_f16(resNormal.z), 0.0f);         /* RGBA16F */
    st4(s->shadingSpecular, i, q_f16(resSpecular.x), q_f16(resSpecular.y), q_f16(resSpecular.z), 0.0f);
    st4(s->diffuse, i, q_unorm8(resColor.x), q_unorm8(resColor.y), q_unorm8(resColor.z), q_unorm8(resColor.w));   /* RGBA8 */
    s->instanceId[i] = resInstanceId;
    st4(s->transparent, i, q_f16(resTransparent.x), q_f16(resTransparent.y), q_f16(resTransparent.z), 1.0f);
    s->flow[2 * i] = q_f16(-resFlow.x); s->flow[2 * i + 1] = q_f16(resFlow.y);                           /* RG16F */
    s->reactiveMask[i] = q_unorm8(fminf(resReactiveMask, 0.9f));                                         /* R8 */
    s->lockMask[i] = q_unorm8(c->binaryLockMask ? (resLockMask >= 0.5f ? 1.0f : 0.0f) : fminf(resLockMask, 1.0f));
    st4(s->normal[cur], i, q_f16(resNormal.x), q_f16(resNormal.y), q_f16(resNormal.z), 0.0f);
    s->depth[cur][i] = resDepth;                                                                         /* R32F */
    memcpy(s->primaryHit + 4 * i, firstHit, 16);
}

/* ---- DirectRayGen ------------------------------------------------------------------------------------------------- */

static float history_weight(const OScene *s, const OShadeCtx *c, size_t i, uint32_t px, uint32_t py, of3 normal, int cur, int *prevIndexOut) {
    /* DirectRayGen.hlsl:31-45 / IndirectRayGen.hlsl:43-56 */
    float fx = s->flow[2 * i], fy = s->flow[2 * i + 1];
    int ix = (int)((float)px + 0.5f + fx), iy = (int)((float)py + 0.5f + fy);
    int prev = cur ^ 1;
    float prevDepth = 0.0f; of3 prevNormal = v3s(0.0f);
    *prevIndexOut = -1;
    if (ix >= 0 && iy >= 0 && ix < c->width && iy < c->height) {                     /* out-of-bounds loads return 0 */
        size_t j = (size_t)iy * (size_t)c->width + (size_t)ix;
        prevDepth = s->depth[prev][j]; prevNormal = ld3i(s->normal[prev], j);
        *prevIndexOut = (int)j;
    }
    float weightDepth = fabsf(s->depth[cur][i] - prevDepth) / 0.01f;
    float weightNormal = powf(fmaxf(0.0f, v3dot(prevNormal, normal)), 128.0f);
    return expf(-weightDepth) * weightNormal;
}

static void pass_direct(OScene *s, OShadeCtx *c, uint32_t px, uint32_t py, int cur) {
    size_t i = (size_t)py * (size_t)c->width + px;
    int instanceId = s->instanceId[i];
    if (instanceId < 0) { st4(s->directLight[cur], i, 1.0f, 1.0f, 1.0f, 0.0f); return; }
    of3 o, rayDirection; primary_ray(c, px, py, &o, &rayDirection, NULL);
    of3 position = ld3i(s->shadingPosition, i), normal = ld3i(s->shadingNormal, i), specular = ld3i(s->shadingSpecular, i);
    of3 newDirect = v3s(0.0f); float historyLength = 0.0f;
    if (c->diReproject) {
        int j; float w = history_weight(s, c, i, px, py, normal, cur, &j);
        of4 prevAccum = { 0, 0, 0, 0 };
        if (j >= 0) { const float *q = s->directLight[cur ^ 1] + 4 * (size_t)j; prevAccum.x = q[0]; prevAccum.y = q[1]; prevAccum.z = q[2]; prevAccum.w = q[3]; }
        newDirect = v3(prevAccum.x, prevAccum.y, prevAccum.z); historyLength = prevAccum.w * w;
    }
    const OMaterial *m = &c->rt[instanceId].desc.material;
    of3 resDirect = oshade_lights_random(c, px, py, rayDirection, (uint32_t)instanceId, position, normal, specular, c->maxLights, 1);
    resDirect = v3add(resDirect, m->selfLight);
    float eyeLambert = fmaxf(v3dot(normal, v3neg(rayDirection)), 0.0f);
    of3 eyeReflected = v3reflect(rayDirection, normal);
    float eyeSpec = powf(fmaxf(fsaturate(v3dot(eyeReflected, v3neg(rayDirection))), 0.0f), m->specularExponent);
    resDirect = v3add(resDirect, v3add(v3scale(c->desc.eyeLightDiffuseColor, eyeLambert), v3mul(c->desc.eyeLightSpecularColor, v3scale(specular, eyeSpec))));
    historyLength = fminf(historyLength + 1.0f, 64.0f);
    newDirect = v3lerp(newDirect, resDirect, 1.0f / historyLength);
    st4(s->directLight[cur], i, q_f16(newDirect.x), q_f16(newDirect.y), q_f16(newDirect.z), q_f16(historyLength));
}

/* ---- shared resolve for bounce rays ---------------------------------------------------------------------------------- */

typedef struct { of3 position, normal, specular, transparent; of4 color; int instanceId; float newReflectionAlpha; } BounceResult;

/* ---- IndirectRayGen ------------------------------------------------------------------------------------------------- */

/* Radiance one GI ray brings back (IndirectRayGen.hlsl:58-131): front-to-back resolve of its hit list, one light sample at the resolved surface, the sky behind what
 * is left.  `more` = bounces still allowed behind this one.  The reference has more == 0: what a surface receives besides its direct light is the constant
 * ambient term.  Extension giBounces = 2 (rt64_oracle.h):
 *   B1  a ray that resolves to a surface sends ONE further ray from the resolved position (depth bias included, like every bounce origin), cosine-weighted about
 *       the resolved normal (RGBA16 SNORM hit-record normal), direction from blue-noise slice noiseFrame + noiseStep (noiseStep = half the distance between the
 *       slices of two GI samples, 1 when they are adjacent) at the PIXEL's coordinates;
 *   B2  the radiance that ray returns (this function, more - 1) stands where `ambient` stands in the reference's line 118: (ambient + directLight) becomes
 *       (incoming + directLight); everything else -- giDiffuseStrength, the ambientBase start value, the sky term -- is per ray, unchanged;
 *   B3  the further ray counts as an indirect ray, its node / triangle visits with the indirect pass's. */
static of3 gi_ray_radiance(OShadeCtx *c, of3 rayOrigin, of3 rayDirection, uint32_t px, uint32_t py, uint32_t noiseFrame, uint32_t noiseStep, int more) {
    const of3 ambient = v3add(c->desc.ambientBaseColor, c->desc.ambientNoGIColor);
    ORayDiff rd; memset(&rd, 0, sizeof(rd));
    SurfacePayload pl;
    trace_surface(c, rayOrigin, rayDirection, rd, px, py, &pl, &c->nodesOther, &c->trisOther);
    c->indirectRays++;
    of3 bgColor = oshade_sample_background_envmap(c, rayDirection);
    of4 sky = oshade_sample_sky_plane(c, rayDirection);
    bgColor = v3lerp(bgColor, v3(sky.x, sky.y, sky.z), sky.w);
    of3 resPosition = v3s(0.0f), resNormal = v3s(0.0f), resSpecular = v3s(0.0f); of4 resColor = { 0, 0, 0, 1 }; int resInstanceId = -1;
    for (uint32_t hit = 0; hit < pl.nhits; hit++) {
        const OHitRecord *r = hit_slot(&pl, hit);
        of4 hitColor = rec_color(r);
        float alphaContrib = resColor.w * hitColor.w;
        if (alphaContrib >= O_EPSILON) {
            uint32_t id = r->instanceId; const OMaterial *m = &c->rt[id].desc.material;
            resPosition = v3add(rayOrigin, v3scale(rayDirection, r->dist + m->depthBias));
            resNormal = rec_normal(r); resSpecular = v3mul(m->specularColor, rec_specular(r));
            resColor.x += hitColor.x * alphaContrib; resColor.y += hitColor.y * alphaContrib; resColor.z += hitColor.z * alphaContrib;
            resColor.w *= (1.0f - hitColor.w);
            resInstanceId = (int)id;
        }
        if (resColor.w <= O_EPSILON) break;
    }
    of3 resIndirect = c->desc.ambientBaseColor;
    if (resInstanceId >= 0) {
        of3 directLight = v3add(oshade_lights_random(c, px, py, rayDirection, (uint32_t)resInstanceId, resPosition, resNormal, resSpecular, 1, 1),
                                c->rt[resInstanceId].desc.material.selfLight);
        of3 incoming = ambient;
        if (more > 0) {                                                                                  /* B1, B2 */
            of3 nextDirection = oshade_cos_hemisphere_blue_noise(c, px, py, noiseFrame + noiseStep, resNormal);
            incoming = gi_ray_radiance(c, resPosition, nextDirection, px, py, noiseFrame + noiseStep, noiseStep, more - 1);
        }
        of3 indirectLight = v3scale(v3mul(v3scale(v3(resColor.x, resColor.y, resColor.z), 1.0f - resColor.w), v3add(incoming, directLight)), c->desc.giDiffuseStrength);
        resIndirect = v3add(resIndirect, indirectLight);
    }
    resIndirect = v3add(resIndirect, v3scale(bgColor, c->desc.giSkyStrength * resColor.w));
    return resIndirect;
}

static void pass_indirect(OScene *s, OShadeCtx *c, uint32_t px, uint32_t py, int cur) {
    size_t i = (size_t)py * (size_t)c->width + px;
    int instanceId = s->instanceId[i];
    of3 ambient = v3add(c->desc.ambientBaseColor, c->desc.ambientNoGIColor);
    if (!(instanceId >= 0 && c->giSamples > 0)) {
        st4(s->indirectLight[cur], i, q_f16(ambient.x), q_f16(ambient.y), q_f16(ambient.z), 0.0f);
        s->moments[cur][2 * i] = s->moments[cur][2 * i + 1] = 0.0f;
        return;
    }
    of3 rayOrigin = ld3i(s->shadingPosition, i), shadingNormal = ld3i(s->shadingNormal, i);
    of3 newIndirect = v3s(0.0f); float historyLength = 0.0f;
    float prevM1 = 0.0f, prevM2 = 0.0f, sumL = 0.0f, sumL2 = 0.0f;      /* SVGF luminance moments (oracle_svgf.c) */
    if (c->giReproject) {
        int j; float w = history_weight(s, c, i, px, py, shadingNormal, cur, &j);
        of4 prevAccum = { 0, 0, 0, 0 };
        if (j >= 0) {
            const float *q = s->indirectLight[cur ^ 1] + 4 * (size_t)j; prevAccum.x = q[0]; prevAccum.y = q[1]; prevAccum.z = q[2]; prevAccum.w = q[3];
            prevM1 = s->moments[cur ^ 1][2 * (size_t)j]; prevM2 = s->moments[cur ^ 1][2 * (size_t)j + 1];
        }
        newIndirect = v3(prevAccum.x, prevAccum.y, prevAccum.z); historyLength = prevAccum.w * w;
    }
    uint32_t maxSamples = c->giSamples; const uint32_t blueNoiseMult = 64u / c->giSamples;
    while (maxSamples > 0) {
        const uint32_t noiseFrame = c->frameCount + maxSamples * blueNoiseMult;
        of3 rayDirection = oshade_cos_hemisphere_blue_noise(c, px, py, noiseFrame, shadingNormal);
        of3 resIndirect = gi_ray_radiance(c, rayOrigin, rayDirection, px, py, noiseFrame, blueNoiseMult > 1u ? blueNoiseMult / 2u : 1u, c->giBounces >= 2 ? 1 : 0);
        historyLength = fminf(historyLength + 1.0f, 64.0f);
        newIndirect = v3lerp(newIndirect, resIndirect, 1.0f / historyLength);
        { float l = 0.2126f * resIndirect.x + 0.7152f * resIndirect.y + 0.0722f * resIndirect.z; sumL += l; sumL2 += l * l; }
        maxSamples--;
    }
    st4(s->indirectLight[cur], i, q_f16(newIndirect.x), q_f16(newIndirect.y), q_f16(newIndirect.z), q_f16(historyLength));
    {
        float n = (float)c->giSamples, alphaM = fminf(n / historyLength, 1.0f);
        s->moments[cur][2 * i] = flerp(prevM1, sumL / n, alphaM);
        s->moments[cur][2 * i + 1] = flerp(prevM2, sumL2 / n, alphaM);
    }
}

/* ---- RefractionRayGen ------------------------------------------------------------------------------------------------ */

static of3 hlsl_refract(of3 i, of3 n, float eta) {
    float cosi = v3dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
    if (k < 0.0f) return v3s(0.0f);
    return v3sub(v3scale(i, eta), v3scale(n, eta * cosi + sqrtf(k)));
}

static void pass_refraction(OScene *s, OShadeCtx *c, uint32_t px, uint32_t py) {
    size_t i = (size_t)py * (size_t)c->width + px;
    int instanceId = s->instanceId[i];
    float refractionAlpha = s->refraction[4 * i + 3];
    if (instanceId < 0 || refractionAlpha <= O_EPSILON) return;
    of3 rayOrigin = ld3i(s->shadingPosition, i), viewDirection = ld3i(s->viewDirection, i), shadingNormal = ld3i(s->shadingNormal, i);
    of3 rayDirection = hlsl_refract(viewDirection, shadingNormal, c->rt[instanceId].desc.material.refractionFactor);
    of2 screenUV = { ((float)px + c->pixelJitter.x) / (float)c->width, ((float)py + c->pixelJitter.y) / (float)c->height };
    of3 bgColor = oshade_sample_background_2d(c, screenUV);
    of4 sky = oshade_sample_sky_2d(c, screenUV);
    bgColor = v3lerp(bgColor, v3(sky.x, sky.y, sky.z), sky.w);
    ORayDiff rd; memset(&rd, 0, sizeof(rd));
    SurfacePayload pl;
    trace_surface(c, rayOrigin, rayDirection, rd, px, py, &pl, &c->nodesOther, &c->trisOther);
    c->refractionRays++;
    of3 ambient = v3add(c->desc.ambientBaseColor, c->desc.ambientNoGIColor);
    of3 resPosition = v3s(0.0f), resNormal = v3s(0.0f), resSpecular = v3s(0.0f), resTransparent = v3s(0.0f); of4 resColor = { 0, 0, 0, 1 }; int resInstanceId = -1;
    for (uint32_t hit = 0; hit < pl.nhits; hit++) {
        const OHitRecord *r = hit_slot(&pl, hit);
        of4 hitColor = rec_color(r);
        float alphaContrib = resColor.w * hitColor.w;
        if (alphaContrib >= O_EPSILON) {
            uint32_t id = r->instanceId; const OMaterial *m = &c->rt[id].desc.material;
            int usesLighting = m->lightGroupMaskBits > 0;
            of3 vertexPosition = v3add(rayOrigin, v3scale(rayDirection, r->dist + m->depthBias));
            if (m->fogEnabled) {
                of4 fog = oshade_fog_from_camera(c, id, vertexPosition);
                resTransparent = v3add(resTransparent, v3scale(v3(fog.x, fog.y, fog.z), fog.w * alphaContrib));
                alphaContrib *= (1.0f - fog.w);
            }
            if (usesLighting) {
                resColor.x += hitColor.x * alphaContrib; resColor.y += hitColor.y * alphaContrib; resColor.z += hitColor.z * alphaContrib;
                resPosition = vertexPosition; resNormal = rec_normal(r); resSpecular = v3mul(m->specularColor, rec_specular(r)); resInstanceId = (int)id;
            }
            else resTransparent = v3add(resTransparent, v3mul(v3scale(v3(hitColor.x, hitColor.y, hitColor.z), alphaContrib), v3add(ambient, m->selfLight)));
            resColor.w *= (1.0f - hitColor.w);
        }
        if (resColor.w <= O_EPSILON) break;
    }
    of3 rgb = v3(resColor.x, resColor.y, resColor.z);
    if (resInstanceId >= 0) {
        of3 directLight = v3add(oshade_lights_random(c, px, py, rayDirection, (uint32_t)resInstanceId, resPosition, resNormal, resSpecular, 1, 1), c->rt[resInstanceId].desc.material.selfLight);
        rgb = v3mul(rgb, v3add(ambient, directLight));
    }
    rgb = v3add(rgb, v3add(v3scale(bgColor, resColor.w), resTransparent));
    float *o = s->refraction + 4 * i;
    o[0] = q_f16(o[0] + rgb.x * refractionAlpha); o[1] = q_f16(o[1] + rgb.y * refractionAlpha); o[2] = q_f16(o[2] + rgb.z * refractionAlpha);
}

/* ---- ReflectionRayGen ------------------------------------------------------------------------------------------------ */

static void pass_reflection(OScene *s, OShadeCtx *c, uint32_t px, uint32_t py) {
    size_t i = (size_t)py * (size_t)c->width + px;
    int instanceId = s->instanceId[i];
    float reflectionAlpha = s->reflection[4 * i + 3];
    if (instanceId < 0 || reflectionAlpha <= O_EPSILON) return;
    of3 shadingPosition = ld3i(s->shadingPosition, i), viewDirection = ld3i(s->viewDirection, i), shadingNormal = ld3i(s->shadingNormal, i);
    of3 rayDirection = v3reflect(viewDirection, shadingNormal);
    float newReflectionAlpha = 0.0f;
    of3 bgColor = oshade_sample_background_envmap(c, rayDirection);
    of4 sky = oshade_sample_sky_plane(c, rayDirection);
    bgColor = v3lerp(bgColor, v3(sky.x, sky.y, sky.z), sky.w);
    ORayDiff rd; memset(&rd, 0, sizeof(rd));
    SurfacePayload pl;
    trace_surface(c, shadingPosition, rayDirection, rd, px, py, &pl, &c->nodesOther, &c->trisOther);
    c->reflectionRays++;
    of3 ambient = v3add(c->desc.ambientBaseColor, c->desc.ambientNoGIColor);
    of3 resPosition = v3s(0.0f), resNormal = v3s(0.0f), resSpecular = v3s(0.0f), resTransparent = v3s(0.0f); of4 resColor = { 0, 0, 0, 1 }; int resInstanceId = -1;
    const OMaterial *pm = &c->rt[instanceId].desc.material;
    for (uint32_t hit = 0; hit < pl.nhits; hit++) {
        const OHitRecord *r = hit_slot(&pl, hit);
        of4 hitColor = rec_color(r);
        float alphaContrib = resColor.w * hitColor.w;
        if (alphaContrib >= O_EPSILON) {
            uint32_t id = r->instanceId; const OMaterial *m = &c->rt[id].desc.material;
            int usesLighting = m->lightGroupMaskBits > 0;
            of3 vertexPosition = v3add(shadingPosition, v3scale(rayDirection, r->dist + m->depthBias));
            if (m->fogEnabled) {
                of4 fog = oshade_fog_from_origin(c, id, vertexPosition, shadingPosition);
                resTransparent = v3add(resTransparent, v3scale(v3(fog.x, fog.y, fog.z), fog.w * alphaContrib));
                alphaContrib *= (1.0f - fog.w);
            }
            of3 vertexNormal = rec_normal(r);
            of3 specular = v3mul(m->specularColor, rec_specular(r));
            if (m->reflectionFactor > O_EPSILON) {
                float fresnelAmount = fresnel_reflect_amount(vertexNormal, rayDirection, m->reflectionFactor, pm->reflectionFresnelFactor);   /* sic: [instanceId], :98 */
                newReflectionAlpha += fresnelAmount * alphaContrib * reflectionAlpha;
            }
            if (usesLighting) { resColor.x += hitColor.x * alphaContrib; resColor.y += hitColor.y * alphaContrib; resColor.z += hitColor.z * alphaContrib; }
            else resTransparent = v3add(resTransparent, v3mul(v3scale(v3(hitColor.x, hitColor.y, hitColor.z), alphaContrib), v3add(ambient, m->selfLight)));
            resPosition = vertexPosition; resNormal = vertexNormal; resSpecular = specular; resInstanceId = (int)id;
            resColor.w *= (1.0f - hitColor.w);
        }
        if (resColor.w <= O_EPSILON) break;
    }
    of3 rgb = v3(resColor.x, resColor.y, resColor.z);
    if (resInstanceId >= 0) {
        of3 directLight = v3add(oshade_lights_random(c, px, py, rayDirection, (uint32_t)resInstanceId, resPosition, resNormal, resSpecular, 1, 0), c->rt[resInstanceId].desc.material.selfLight);
        rgb = v3mul(rgb, v3add(ambient, directLight));
        st4(s->shadingPosition, i, resPosition.x, resPosition.y, resPosition.z, 0.0f);
        st4(s->viewDirection, i, q_f16(rayDirection.x), q_f16(rayDirection.y), q_f16(rayDirection.z), 0.0f);
        st4(s->shadingNormal, i, q_f16(resNormal.x), q_f16(resNormal.y), q_f16(resNormal.z), 0.0f);
        s->instanceId[i] = resInstanceId;
    }
    rgb = v3add(rgb, v3add(v3scale(bgColor, resColor.w), resTransparent));
    const of3 HighlightColor = { 1.0f, 1.05f, 1.2f }, ShadowColor = { 0.1f, 0.05f, 0.0f };
    float shine = pm->reflectionShineFactor;
    rgb = v3lerp(rgb, HighlightColor, powf(fmaxf(rayDirection.y, 0.0f) * shine, 3.0f));
    rgb = v3lerp(rgb, ShadowColor, powf(fmaxf(-rayDirection.y, 0.0f) * shine, 3.0f));
    float k = reflectionAlpha * fsaturate(1.0f - newReflectionAlpha);
    float *o = s->reflection + 4 * i;
    o[0] = q_f16(o[0] + rgb.x * k); o[1] = q_f16(o[1] + rgb.y * k); o[2] = q_f16(o[2] + rgb.z * k);
    o[3] = q_f16(fsaturate(newReflectionAlpha));
}

/* ---- GaussianFilterRGB3x3CS ------------------------------------------------------------------------------------------ */

static of3 bilinear_clamp_rgb(const float *img, int w, int h, float u, float v) {   /* LINEAR + CLAMP static sampler, ref:rt64_device.cpp:737-742 */
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y), fx = x - x0f, fy = y - y0f;
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 >= w ? w - 1 : x0); x1 = x1 < 0 ? 0 : (x1 >= w ? w - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 >= h ? h - 1 : y0); y1 = y1 < 0 ? 0 : (y1 >= h ? h - 1 : y1);
    of3 c00 = ld3i(img, (size_t)y0 * w + x0), c10 = ld3i(img, (size_t)y0 * w + x1), c01 = ld3i(img, (size_t)y1 * w + x0), c11 = ld3i(img, (size_t)y1 * w + x1);
    of3 top = v3lerp(c00, c10, fx), bot = v3lerp(c01, c11, fx);
    return v3lerp(top, bot, fy);
}

static void gaussian_pass(const float *in, float *out, int w, int h, int y0, int y1) {
    const float texel[2] = { 1.0f / (float)w, 1.0f / (float)h };
    const float k00 = 0.077847f, k01 = 0.123317f, k11 = 0.195346f;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < w; x++) {
            float wt[4];
            int xl = x == 0, xr = x == w - 1, yt = y == 0, yb = y == h - 1;
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
            of3 smp[4];
            for (int k = 0; k < 3; k++) smp[k] = bilinear_clamp_rgb(in, w, h, ((float)x + off[k][0]) * texel[0], ((float)y + off[k][1]) * texel[1]);
            smp[3] = (x + 1 < w && y + 1 < h) ? ld3i(in, (size_t)(y + 1) * w + (x + 1)) : v3s(0.0f);     /* gInput[DTid + 1], OOB load = 0 */
            size_t i = (size_t)y * w + x;
            out[4 * i] = q_f16(smp[0].x * wt[0] + smp[1].x * wt[1] + smp[2].x * wt[2] + smp[3].x * wt[3]);
            out[4 * i + 1] = q_f16(smp[0].y * wt[0] + smp[1].y * wt[1] + smp[2].y * wt[2] + smp[3].y * wt[3]);
            out[4 * i + 2] = q_f16(smp[0].z * wt[0] + smp[1].z * wt[1] + smp[2].z * wt[2] + smp[3].z * wt[3]);
        }
}

/* ---- ComposePS + PostProcessPS ----------------------------------------------------------------------------------------- */

static void pass_compose_post(OScene *s, const OShadeCtx *c, uint32_t px, uint32_t py) {
    size_t i = (size_t)py * (size_t)c->width + px;
    const float *d = s->diffuse + 4 * i;
    of3 result;
    if (d[3] > O_EPSILON) {
        of3 diffuse = v3(d[0], d[1], d[2]);
        of3 direct = ld3i(s->filteredDirect[1], i), indirect = ld3i(s->filteredIndirect[1], i);
        result = v3mul(diffuse, v3add(direct, indirect));
        result = v3lerp(diffuse, result, d[3]);
        result = v3add(result, ld3i(s->reflection, i));
        result = v3add(result, ld3i(s->refraction, i));
        result = v3add(result, ld3i(s->transparent, i));
    }
    else result = v3(d[0], d[1], d[2]);
    st4(s->outputRGBA32F, i, result.x, result.y, result.z, 1.0f);
    /* PostProcessPS: motionBlurStrength == 0 and render size == screen size -> passthrough of the same texel. */
    if (c->separatePost) return;
    s->finalRGBA8[4 * i] = to_unorm8(result.x); s->finalRGBA8[4 * i + 1] = to_unorm8(result.y); s->finalRGBA8[4 * i + 2] = to_unorm8(result.z); s->finalRGBA8[4 * i + 3] = 255;
}

/* ---- frame ------------------------------------------------------------------------------------------------------------- */

static void reduce_ctx(OShadeCtx *dst, const OShadeCtx *src) {
    dst->primaryRays += src->primaryRays; dst->shadowRays += src->shadowRays; dst->indirectRays += src->indirectRays;
    dst->reflectionRays += src->reflectionRays; dst->refractionRays += src->refractionRays;
    dst->nodesPrimary += src->nodesPrimary; dst->trisPrimary += src->trisPrimary; dst->nodesShadow += src->nodesShadow;
    dst->trisShadow += src->trisShadow; dst->nodesOther += src->nodesOther; dst->trisOther += src->trisOther;
}

typedef void (*PixelPass)(OScene *, OShadeCtx *, uint32_t, uint32_t, int);

static void run_pass(OScene *s, OShadeCtx *total, const OFrameParams *p, int cur, int which) {
    int y0 = p->tileY0, y1 = p->tileY1, w = p->width;
#ifdef _OPENMP
    if (p->threads > 0) omp_set_num_threads(p->threads);
#endif
#pragma omp parallel
    {
        OShadeCtx local = *total;
        local.primaryRays = local.shadowRays = local.indirectRays = local.reflectionRays = local.refractionRays = 0;
        local.nodesPrimary = local.trisPrimary = local.nodesShadow = local.trisShadow = local.nodesOther = local.trisOther = 0;
#pragma omp for schedule(dynamic, 4)
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < w; x++) {
                switch (which) {
                case 0: pass_primary(s, &local, (uint32_t)x, (uint32_t)y, cur); break;
                case 1: pass_direct(s, &local, (uint32_t)x, (uint32_t)y, cur); break;
                case 2: pass_indirect(s, &local, (uint32_t)x, (uint32_t)y, cur); break;
                case 3: pass_refraction(s, &local, (uint32_t)x, (uint32_t)y); break;
                case 4: pass_reflection(s, &local, (uint32_t)x, (uint32_t)y); break;
                case 5: pass_compose_post(s, &local, (uint32_t)x, (uint32_t)y); break;
                }
            }
#pragma omp critical
        reduce_ctx(total, &local);
    }
}

/* ---- PostProcessPS as its own pass (resolution scale and / or motion blur), ref:shaders/PostProcessPS.hlsl:13-36 ------------- */
/* gSampler is the static sampler of ref:rt64_device.cpp:958-973: MIN_MAG_MIP_LINEAR, WRAP.  Filtering follows the texture spec
 * (texel centres at +0.5, fp32 weights). */
static int wrapi(int i, int n) { int j = i % n; return j < 0 ? j + n : j; }
static void sample_linear_wrap(const float *img, int ch, int w, int h, float u, float v, float out[4]) {
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y), fx = x - x0f, fy = y - y0f;
    int x0 = wrapi((int)x0f, w), x1 = wrapi((int)x0f + 1, w), y0 = wrapi((int)y0f, h), y1 = wrapi((int)y0f + 1, h);
    for (int k = 0; k < ch; k++) {
        float c00 = img[((size_t)y0 * w + x0) * ch + k], c10 = img[((size_t)y0 * w + x1) * ch + k];
        float c01 = img[((size_t)y1 * w + x0) * ch + k], c11 = img[((size_t)y1 * w + x1) * ch + k];
        float top = c00 + fx * (c10 - c00), bot = c01 + fx * (c11 - c01);
        out[k] = top + fy * (bot - top);
    }
}
/* `vp` = viewport (x, y, w, h) and `sc` = scissor (left, top, right, bottom) the full-screen triangle is drawn with: the screen unless
 * the first ray-traced instance carries its own rectangles (ref:rt64_view.cpp:1258-1271,1624-1626). */
static void pass_post(OScene *s, const OFrameParams *p, const float *src, int srcW, int srcH, int rtW, int rtH, int screenW, int screenH, const float vp[4], const int sc[4]) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < screenH; y++)
        for (int x = 0; x < screenW; x++) {
            float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
            if (x < sc[0] || x >= sc[2] || y < sc[1] || y >= sc[3]) continue;
            if (!(cx >= vp[0]) || !(cx < vp[0] + vp[2]) || !(cy >= vp[1]) || !(cy < vp[1] + vp[3])) continue;
            float u = (cx - vp[0]) / vp[2], v = (cy - vp[1]) / vp[3];                                  /* FullScreenVS interpolant at the pixel centre */
            float color[4];
            int blurred = 0;
            if (p->motionBlurStrength > 0.0f && p->motionBlurSamples > 0) {
                float fl[4];
                sample_linear_wrap(s->flow, 2, rtW, rtH, u, v, fl);
                float flx = fl[0] / (float)rtW, fly = fl[1] / (float)rtH;
                float flowLength = sqrtf(flx * flx + fly * fly);
                if (flowLength > 1e-6f) {
                    const float sampleStep = p->motionBlurStrength / (float)p->motionBlurSamples;
                    float sum[3] = { 0.0f, 0.0f, 0.0f }, sumWeight = 0.0f;
                    float su = u - (flx * p->motionBlurStrength / 2.0f), sv = v - (fly * p->motionBlurStrength / 2.0f);
                    for (unsigned int k = 0; k < p->motionBlurSamples; k++) {
                        float uu = su + flx * (float)k * sampleStep, vv = sv + fly * (float)k * sampleStep;
                        uu = fminf(fmaxf(uu, 0.0f), 1.0f); vv = fminf(fmaxf(vv, 0.0f), 1.0f);
                        float c4[4];
                        sample_linear_wrap(src, 4, srcW, srcH, uu, vv, c4);
                        sum[0] += c4[0] * 1.0f; sum[1] += c4[1] * 1.0f; sum[2] += c4[2] * 1.0f; sumWeight += 1.0f;
                    }
                    color[0] = sum[0] / sumWeight; color[1] = sum[1] / sumWeight; color[2] = sum[2] / sumWeight;
                    blurred = 1;
                }
            }
            if (!blurred) sample_linear_wrap(src, 4, srcW, srcH, u, v, color);
            uint8_t *f = s->finalRGBA8 + 4 * ((size_t)y * (size_t)screenW + (size_t)x);
            f[0] = to_unorm8(color[0]); f[1] = to_unorm8(color[1]); f[2] = to_unorm8(color[2]); f[3] = 255;
        }
}

int oracle_render(OScene *s, const OFrameParams *pIn, OFrameResult *out) {
    if (pIn->width <= 0 || pIn->height <= 0 || pIn->tileY0 < 0 || pIn->tileY1 > pIn->height || pIn->tileY0 >= pIn->tileY1) return 0;
    /* Screen size vs render size (View::createOutputBuffers, ref:rt64_view.cpp:138-139). */
    const int screenW = pIn->width, screenH = pIn->height;
    const float scale = pIn->resolutionScale > 0.0f ? pIn->resolutionScale : 1.0f;
    OFrameParams local = *pIn;
    local.width = (int)lroundf((float)screenW * scale); local.height = (int)lroundf((float)screenH * scale);
    /* An upscaler decides the render size itself (View::createOutputBuffers, ref:rt64_view.cpp:114-136; upscalerResolutionOverride is off) */
    int upW = 0, upH = 0, phases = 1;
    const int upscale = oracle_upscaler_info(pIn->upscaler, pIn->upscalerMode, screenW, screenH, &upW, &upH, &phases);
    if (upscale) { local.width = upW; local.height = upH; }
    if (local.width < 1) local.width = 1;
    if (local.height < 1) local.height = 1;
    int separatePost = upscale || local.width != screenW || local.height != screenH || (pIn->motionBlurStrength > 0.0f && pIn->motionBlurSamples > 0);
    /* Viewport / scissor of the ray-traced content: the rectangles of the first ray-traced instance, when it has any (ref:rt64_view.cpp:1258-1271). */
    float rtVp[4] = { 0.0f, 0.0f, (float)screenW, (float)screenH };
    int rtSc[4] = { 0, 0, screenW, screenH }, rtRect = 0;
    for (int i = 0; i < s->instanceCount; i++) {
        const OInstanceDesc *d = &s->instances[i];
        if (!d->mesh || !d->diffuse || !((d->mesh->flags & 0x1) && d->mesh->bvh.count > 0)) continue;
        if (d->scissorRect[2] > 0 && d->scissorRect[3] > 0) {
            rtSc[0] = d->scissorRect[0]; rtSc[1] = screenH - d->scissorRect[1] - d->scissorRect[3]; rtSc[2] = d->scissorRect[0] + d->scissorRect[2]; rtSc[3] = screenH - d->scissorRect[1];
            rtRect = 1;
        }
        if (d->viewportRect[2] > 0 && d->viewportRect[3] > 0) {
            rtVp[0] = (float)d->viewportRect[0]; rtVp[1] = (float)(screenH - d->viewportRect[1] - d->viewportRect[3]); rtVp[2] = (float)d->viewportRect[2]; rtVp[3] = (float)d->viewportRect[3];
            rtRect = 1;
        }
        break;
    }
    if (rtRect) separatePost = 1;
    if (separatePost) { local.tileY0 = 0; local.tileY1 = local.height; }       /* the resample reads neighbours: whole frame */
    const OFrameParams *p = &local;
    double t0 = now_s();
    alloc_images(s, p->width, p->height);
    if (s->finalW != screenW || s->finalH != screenH) { free(s->finalRGBA8); s->finalRGBA8 = (uint8_t *)calloc((size_t)screenW * (size_t)screenH * 4, 1); s->finalW = screenW; s->finalH = screenH; }
    /* Extension primarySpp = N (rt64_oracle.h): the frame as N complete sub-frames.
     *   P1  sub-frame k = 0 .. N - 1 runs every pass of a frame up to ComposePS; frameCount, the ping-pong index and the temporal history advance after each, as if
     *       the host had drawn N frames without moving anything;
     *   P2  its primary rays (and the screen-space sky / background lookups) are jittered by (Halton(k + 1, 2) - 0.5, Halton(k + 1, 3) - 0.5); for k >= 1 every
     *       instance's previous transform is its transform and the previous camera is the camera;
     *   P3  rtOutput = (((out_0 + out_1) + ...) + out_{N-1}) * (1.0f / N) in fp32, per channel; the back buffer is PostProcessPS of that image; the other images a
     *       reader sees afterwards are those of the last sub-frame; ray and visit counters are sums over the sub-frames;
     *   P4  not combined with an upscaler, a resolution scale, motion blur or a viewport rectangle (oracle_render returns 0). */
    const int spp = pIn->primarySpp > 1 ? pIn->primarySpp : 1;
    if (spp > 1 && (separatePost || upscale)) return 0;
    float *sppSum = spp > 1 ? (float *)malloc((size_t)local.width * (size_t)local.height * 4 * sizeof(float)) : NULL;
    OShadeCtx sums; memset(&sums, 0, sizeof(sums));
    update_view(s, p, 0);
    OShadeCtx ctx;
    update_global_params(s, p, screenW, screenH, &ctx);
    if (upscale) {                  /* jitter only with an upscaler (ref:rt64_view.cpp:1273-1281); FSR takes a continuous lock mask (:1018) */
        const int fi = (int)(s->frameCount % (uint32_t)phases) + 1;
        ctx.pixelJitter.x = oracle_halton(fi, 2) - 0.5f; ctx.pixelJitter.y = oracle_halton(fi, 3) - 0.5f;
        ctx.binaryLockMask = 0;
        if (s->upW != screenW || s->upH != screenH) {
            for (int k = 0; k < 2; k++) { free(s->upscaled[k]); s->upscaled[k] = (float *)calloc((size_t)screenW * (size_t)screenH * 4, sizeof(float)); }
            s->upW = screenW; s->upH = screenH; s->upValid = 0;
        }
        if (!s->haveHistory) s->upValid = 0;           /* buffers were (re)created: the accumulation starts over */
    }
    else s->upValid = 0;
    ctx.separatePost = separatePost;
    ctx.viewportW = rtVp[2]; ctx.viewportH = rtVp[3];                 /* gParams.viewport.zw, ref:rt64_view.cpp:1283-1286 */
    /* Raster instances: everything that is not ray traced, background-flagged ones first (View::update, ref:rt64_view.cpp:1138-1147). */
    int *bgList = (int *)malloc(sizeof(int) * (size_t)(s->instanceCount + 1)), *fgList = (int *)malloc(sizeof(int) * (size_t)(s->instanceCount + 1));
    int bgCount = 0, fgCount = 0;
    for (int i = 0; i < s->instanceCount; i++) {
        const OInstanceDesc *d = &s->instances[i];
        if (!d->mesh || !d->diffuse) continue;
        if ((d->mesh->flags & 0x1) && d->mesh->bvh.count > 0) continue;
        if (d->flags & 0x1) bgList[bgCount++] = i; else fgList[fgCount++] = i;       /* RT64_INSTANCE_RASTER_BACKGROUND */
    }
    if (bgCount > 0) {      /* gBackground: cleared to 0, background instances without their scissors / viewports (ref:rt64_view.cpp:1298-1319) */
        if (s->bgW != screenW || s->bgH != screenH) { free(s->backgroundRGBA8); s->backgroundRGBA8 = (uint8_t *)malloc((size_t)screenW * (size_t)screenH * 4); s->bgW = screenW; s->bgH = screenH; }
        memset(s->backgroundRGBA8, 0, (size_t)screenW * (size_t)screenH * 4);
        oraster_draw(s, bgList, bgCount, s->backgroundRGBA8, screenW, screenH, 0, screenH, 0);
        memset(&s->bgTex, 0, sizeof(s->bgTex));
        s->bgTex.mips = 1; s->bgTex.w[0] = screenW; s->bgTex.h[0] = screenH; s->bgTex.rgba[0] = s->backgroundRGBA8;
        ctx.background = &s->bgTex;
    }
    double t1 = now_s();
    int cur = s->rtSwap;
    size_t n = (size_t)p->width * (size_t)p->height;
    if (s->rtCount > 0) {
      for (int sub = 0; sub < spp; sub++) {
        if (sub > 0) {                                  /* P1: the sub-frame before this one is over; P2: nothing moved */
            s->rtSwap ^= 1; s->haveHistory = 1; s->frameCount++;
            cur = s->rtSwap;
            const OTexture *bg = ctx.background;
            reduce_ctx(&sums, &ctx);
            update_view(s, p, 1);
            update_global_params(s, p, screenW, screenH, &ctx);
            ctx.viewportW = rtVp[2]; ctx.viewportH = rtVp[3]; ctx.background = bg;
        }
        if (spp > 1) {
            ctx.pixelJitter.x = oracle_halton(sub + 1, 2) - 0.5f; ctx.pixelJitter.y = oracle_halton(sub + 1, 3) - 0.5f;
            ctx.separatePost = 1;                       /* ComposePS writes rtOutput only; the back buffer comes from the mean (P3) */
        }
        run_pass(s, &ctx, p, cur, 0);
        run_pass(s, &ctx, p, cur, 1);
        run_pass(s, &ctx, p, cur, 2);
        run_pass(s, &ctx, p, cur, 3);
        for (int r = 0; r < p->maxReflections; r++) run_pass(s, &ctx, p, cur, 4);
        /* raw -> filtered copies, ref:rt64_view.cpp:1438-1509 (DI denoising compiled out: destination index 1) */
        memcpy(s->filteredDirect[1], s->directLight[cur], n * 4 * sizeof(float));
        int denoiseGI = p->denoiserEnabled && p->giSamples > 0;
        if (!denoiseGI) memcpy(s->filteredIndirect[1], s->indirectLight[cur], n * 4 * sizeof(float));
        else if (p->denoiserMode == 0) {
            memcpy(s->filteredIndirect[0], s->indirectLight[cur], n * 4 * sizeof(float));
            for (int k = 0; k < 5; k++)                                      /* ref:rt64_view.cpp:1512-1530 */
                gaussian_pass(s->filteredIndirect[k % 2], s->filteredIndirect[(k % 2) ^ 1], p->width, p->height, 0, p->height);
        }
        else osvgf_filter(s, p, cur);
        run_pass(s, &ctx, p, cur, 5);
        if (spp > 1) {                                  /* P3 */
            for (int y = p->tileY0; y < p->tileY1; y++)
                for (size_t q = (size_t)y * p->width * 4; q < (size_t)(y + 1) * p->width * 4; q++) sppSum[q] = sub == 0 ? s->outputRGBA32F[q] : sppSum[q] + s->outputRGBA32F[q];
        }
      }
      if (spp > 1) {
            const float inv = 1.0f / (float)spp;
            for (int y = p->tileY0; y < p->tileY1; y++)
                for (size_t q = (size_t)y * p->width; q < (size_t)(y + 1) * p->width; q++) {
                    float *o = s->outputRGBA32F + 4 * q;
                    for (int k = 0; k < 4; k++) o[k] = sppSum[4 * q + k] * inv;
                    uint8_t *f = s->finalRGBA8 + 4 * q;
                    f[0] = to_unorm8(o[0]); f[1] = to_unorm8(o[1]); f[2] = to_unorm8(o[2]); f[3] = 255;
                }
      }
        if (rtRect) {       /* the ray-traced picture covers only its rectangle: cleared buffer + background instances show around it (ref:rt64_view.cpp:1292-1296) */
            const size_t ns = (size_t)screenW * (size_t)screenH;
            memset(s->finalRGBA8, 0, ns * 4);
            for (size_t i = 0; i < ns; i++) s->finalRGBA8[4 * i + 3] = 255;
            oraster_draw(s, bgList, bgCount, s->finalRGBA8, screenW, screenH, 0, screenH, 1);
        }
        const float *postSrc = s->outputRGBA32F; int postW = p->width, postH = p->height;
        if (upscale) {              /* Upscaler::upscale, ref:rt64_view.cpp:1584-1618: jitterX/Y = -pixelJitter is the sample offset convention of the SDKs */
            const int uc = s->upSwap;
            oupscale_frame(s->outputRGBA32F, s->flow, s->reactiveMask, s->lockMask, s->depth[cur], p->width, p->height, ctx.pixelJitter.x, ctx.pixelJitter.y,
                           s->upscaled[uc ^ 1], s->upscaled[uc], screenW, screenH, s->upValid);
            postSrc = s->upscaled[uc]; postW = screenW; postH = screenH;
            s->upValid = 1; s->upSwap ^= 1;
        }
        if (separatePost) pass_post(s, p, postSrc, postW, postH, p->width, p->height, screenW, screenH, rtVp, rtSc);
    }
    else {
        const size_t ns = (size_t)screenW * (size_t)screenH;
        memset(s->finalRGBA8, 0, ns * 4);
        for (size_t i = 0; i < ns; i++) s->finalRGBA8[4 * i + 3] = 255;      /* cleared back buffer, ref:rt64_device.cpp:996-997 */
        /* Without ray-traced content nothing covers the background instances on the back buffer (ref:rt64_view.cpp:1292-1296). */
        oraster_draw(s, bgList, bgCount, s->finalRGBA8, screenW, screenH, separatePost ? 0 : pIn->tileY0, separatePost ? screenH : pIn->tileY1, 1);
    }
    /* Foreground instances over the finished frame (ref:rt64_view.cpp:1657-1661). */
    oraster_draw(s, fgList, fgCount, s->finalRGBA8, screenW, screenH, separatePost ? 0 : pIn->tileY0, separatePost ? screenH : pIn->tileY1, 1);
    free(bgList); free(fgList); free(sppSum);
    double t2 = now_s();
    if (out) {
        memset(out, 0, sizeof(*out));
        out->width = p->width; out->height = p->height; out->screenWidth = screenW; out->screenHeight = screenH;
        out->backgroundRGBA8 = bgCount > 0 ? s->backgroundRGBA8 : NULL;
        out->upscaledRGBA32F = (upscale && s->rtCount > 0) ? s->upscaled[s->upSwap ^ 1] : NULL;
        out->pixelJitter[0] = ctx.pixelJitter.x; out->pixelJitter[1] = ctx.pixelJitter.y;
        out->finalRGBA8 = s->finalRGBA8; out->outputRGBA32F = s->outputRGBA32F; out->shadingPosition = s->shadingPosition;
        out->shadingNormal = s->shadingNormal; out->shadingSpecular = s->shadingSpecular; out->diffuse = s->diffuse;
        out->instanceId = s->instanceId; out->directLight = s->directLight[cur]; out->indirectLight = s->indirectLight[cur];
        out->filteredDirect = s->filteredDirect[1]; out->filteredIndirect = s->filteredIndirect[1];
        out->reflection = s->reflection; out->refraction = s->refraction; out->transparent = s->transparent;
        out->viewDirection = s->viewDirection; out->normal = s->normal[cur]; out->flow = s->flow; out->reactiveMask = s->reactiveMask;
        out->lockMask = s->lockMask; out->depth = s->depth[cur]; out->primaryHit = s->primaryHit;
        reduce_ctx(&ctx, &sums);                        /* (all zero without primarySpp) */
        out->primaryRays = ctx.primaryRays; out->shadowRays = ctx.shadowRays; out->indirectRays = ctx.indirectRays;
        out->reflectionRays = ctx.reflectionRays; out->refractionRays = ctx.refractionRays;
        out->nodesVisitedPrimary = ctx.nodesPrimary; out->trianglesTestedPrimary = ctx.trisPrimary;
        out->nodesVisitedShadow = ctx.nodesShadow; out->trianglesTestedShadow = ctx.trisShadow;
        out->nodesVisited = ctx.nodesPrimary + ctx.nodesShadow + ctx.nodesOther; out->trianglesTested = ctx.trisPrimary + ctx.trisShadow + ctx.trisOther;
        out->secondsBuild = t1 - t0; out->secondsRender = t2 - t1;
    }
    /* End of frame, ref:rt64_view.cpp:1663-1667 */
    s->rtSwap ^= 1; s->haveHistory = 1; s->frameCount++;
    return 1;
}
