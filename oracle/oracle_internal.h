/* oracle_internal.h -- private structures of the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h). */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H

#include "rt64_oracle.h"
#include "oracle_math.h"

#define O_MAX_MIPS 16
#define O_MAX_HIT_QUERIES 16               /* ref:shaders/GlobalHitBuffers.hlsli:8 */
#define O_MAX_LIGHTS 16                    /* ref:shaders/Lights.hlsli:25 */

struct OTexture {
    int mips;
    int w[O_MAX_MIPS], h[O_MAX_MIPS];
    uint8_t *rgba[O_MAX_MIPS];             /* tightly packed RGBA8 per mip */
};

struct OMesh {
    int flags;
    uint8_t *vertices; int vertexCount, vertexStride;
    uint32_t *indices; int indexCount;
    OBvh bvh;
    OTri *tris;                            /* sorted leaf order */
    uint32_t version;
};

/* Decoded colour combiner + vertex layout, ref:private/rt64_shader.cpp:32-96. */
typedef struct {
    int c[2][4];
    int inputCount;
    int useTextures[2];
    int do_single[2], do_multiply[2], do_mix[2];
    int color_alpha_same, opt_alpha, opt_texture_edge, opt_noise;
    int vertexSize, positionOffset, normalOffset, uvOffset, inputOffset[4];
    int vertexUV;
} OCombiner;

void ocombiner_decode(uint32_t shaderId, OCombiner *cc);

/* One render instance of the frame (ref:private/rt64_view.h RenderInstance). */
typedef struct {
    OInstanceDesc desc;
    int sceneIndex;
    OCombiner cc;
    om4 objectToWorld, objectToWorldNormal, objectToWorldPrevious;   /* ref:rt64_view.cpp:348-376 */
    om4 worldToObject;
    int cullDisable;
    int opaque;                             /* every hit stores alpha 255 (see oracle_render.c: instance_is_opaque) */
    int shadowOpaque;                       /* rule O2: every shadow any-hit saturates payload.shadowHit */
} OInst;

typedef struct {
    float t, u, v;
    uint32_t instance, prim;               /* instance = index into rtInstances; prim = original triangle number */
} OHit;

typedef struct {                           /* Igehy ray differentials, ref:shaders/Ray.hlsli:14-19 */
    of3 dOdx, dOdy, dDdx, dDdy;
} ORayDiff;

typedef struct {                           /* one entry of the per-pixel hit list, ref:rt64_view.cpp:237-241,573-594 */
    float dist; of3 flow;                  /* gHitDistAndFlow  RGBA32F */
    uint8_t color[4];                      /* gHitColor        RGBA8 UNORM */
    int16_t normal[4];                     /* gHitNormal       RGBA16 SNORM */
    uint8_t specular[4];                   /* gHitSpecular     RGBA8 UNORM */
    uint16_t instanceId;                   /* gHitInstanceId   R16 UINT */
    OHit geo;                              /* oracle extra: geometric hit behind the record */
} OHitRecord;

typedef struct {
    uint64_t nodes, tris;
} OTraceCounters;

struct OScene {
    OSceneDesc desc;
    OLight lights[64]; int lightCount;
    uint8_t *blueNoise;                    /* 512x512 RGBA8 */
    OTexture *sky;
    OInstanceDesc *instances; int instanceCount, instanceCap;

    /* per-frame state built by oracle_render (View::update) */
    OInst *rt; int rtCount;
    OBvh tlas;

    /* View state that survives frames (ref:rt64_view.cpp:961-1028,1664-1667) */
    uint32_t frameCount;
    int haveHistory;                       /* !rtSkipReprojection */
    int histW, histH;
    om4 viewI, prevViewI, viewProj, prevViewProj, view, projection, projectionI;
    int matricesValid;

    /* images (allocated for width x height) */
    int imgW, imgH, finalW, finalH;
    uint8_t *finalRGBA8;
    uint8_t *backgroundRGBA8; int bgW, bgH; struct OTexture bgTex;   /* gBackground: raster background instances, screen size */
    float *outputRGBA32F, *shadingPosition, *shadingNormal, *shadingSpecular, *diffuse;
    int32_t *instanceId;
    float *directLight[2], *indirectLight[2], *filteredDirect[2], *filteredIndirect[2];
    float *reflection, *refraction, *transparent, *viewDirection, *normal[2];
    float *flow, *reactiveMask, *lockMask, *depth[2];
    uint32_t *primaryHit;
    float *moments[2];                     /* SVGF: luminance moments + history */
    float *upscaled[2]; int upW, upH, upValid, upSwap;      /* rtOutputUpscaled ping-pong (display size), oracle_upscale.c */
    int rtSwap;
};

/* oracle_bvh.c */
void obvh_free(OBvh *b);
void obvh_build(OBvh *b, uint32_t n, const float *boxMin, const float *boxMax, float **outMin, float **outMax);
void obvh_fit(OBvh *b, const float *leafMin, const float *leafMax);

/* oracle_trace.c */
typedef struct {
    float o[3], d[3];
    float tmin, tmax;
    int cullBackFaces;                     /* RAY_FLAG_CULL_BACK_FACING_TRIANGLES */
} ORay;

/* Any-hit callback, called for every ray/triangle intersection in traversal order.  It may lower *tmax (commit)
 * and/or set *terminate to end the search.  The return value is informational (1 = accepted). */
typedef int (*OAnyHitFn)(void *user, const OHit *hit, float *tmax, int *terminate);
void otrace(const OScene *s, const ORay *ray, int bruteForce, OAnyHitFn fn, void *user, OTraceCounters *ctr);

/* oracle_texture.c */
void otex_sample_level(const OTexture *t, float u, float v, int level, int filter, int hAddr, int vAddr, float out[4]);
void otex_sample_grad(const OTexture *t, float u, float v, of2 ddx, of2 ddy, int filter, int hAddr, int vAddr, float out[4]);

/* oracle_raster.c */
void oraster_draw(const OScene *s, const int *list, int count, uint8_t *target, int w, int h, int y0, int y1, int applyScissorsAndViewports);

/* oracle_upscale.c */
void oupscale_frame(const float *color, const float *flow, const float *reactive, const float *lock, const float *depth, int rw, int rh,
                    float jx, float jy, const float *prev, float *out, int dw, int dh, int haveHistory);

/* oracle_render.c / oracle_shade.c */
int omatrix_inverse_d(const om4 *m, om4 *out);

#endif
