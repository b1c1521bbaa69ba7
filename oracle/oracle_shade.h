/* oracle_shade.h -- shading context of the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h).
 * Mirrors the constant block the reference shaders read: ref:shaders/GlobalParams.hlsli:8-43
 * (filled by ref:private/rt64_view.cpp:961-1028). */
#ifndef ORACLE_SHADE_H
#define ORACLE_SHADE_H

#include "oracle_internal.h"

typedef struct {
    const OScene *scene;
    const OInst *rt; int rtCount;
    const OLight *lights; int lightCount;
    OSceneDesc desc;
    const uint8_t *blueNoise;
    const OTexture *sky;
    const OTexture *background;            /* raster background target, NULL = transparent black */
    om4 view, viewI, prevViewI, projection, projectionI, viewProj, prevViewProj;
    of3 cameraU, cameraV, cameraW;
    float viewportW, viewportH;            /* viewport.zw */
    int width, height;                     /* resolution.xy (render size) */
    int screenW, screenH;                  /* resolution.zw */
    int separatePost;                      /* PostProcessPS runs as its own pass (resolution scale / motion blur) */
    of2 pixelJitter;
    uint32_t frameCount, diSamples, giSamples, maxLights;
    int giBounces;                         /* extension: 2 = a second bounce (oracle_render.c, rules B1-B3); otherwise the reference's one */
    int giReproject, diReproject, binaryLockMask;
    float maxDepthBias;
    int bruteForce, cullBehindOpaque;
    /* counters (per thread copy, reduced by the caller) */
    uint64_t primaryRays, shadowRays, indirectRays, reflectionRays, refractionRays;
    uint64_t nodesPrimary, trisPrimary, nodesShadow, trisShadow, nodesOther, trisOther;
} OShadeCtx;

int oshade_surface_anyhit(const OShadeCtx *c, const OHit *hit, of3 rayDirW, ORayDiff payloadDiff, uint32_t px, uint32_t py, OHitRecord *rec);
float oshade_shadow_anyhit_alpha(const OShadeCtx *c, const OHit *hit, uint32_t px, uint32_t py);
float oshade_trace_shadow(OShadeCtx *c, of3 origin, of3 dir, float tmin, float tmax, uint32_t px, uint32_t py);
of3 oshade_lights_random(OShadeCtx *c, uint32_t px, uint32_t py, of3 rayDirection, uint32_t instanceId, of3 position, of3 normal,
                         of3 specular, uint32_t maxLightCount, int checkShadows);
of3 oshade_blue_noise(const OShadeCtx *c, uint32_t px, uint32_t py, uint32_t frame);
of3 oshade_cos_hemisphere_blue_noise(const OShadeCtx *c, uint32_t px, uint32_t py, uint32_t frame, of3 hitNorm);
void oshade_compute_ray_diffs(of3 nonNormDir, of3 right, of3 up, of2 viewportDims, of3 *dDdx, of3 *dDdy);
of4 oshade_sample_sky_2d(const OShadeCtx *c, of2 screenUV);
of4 oshade_sample_sky_plane(const OShadeCtx *c, of3 rayDirection);
void oshade_raster_pixel(const OCombiner *cc, const of4 inputs[4], of4 texVal0, float out[4]);
of3 oshade_sample_background_2d(const OShadeCtx *c, of2 screenUV);
of3 oshade_sample_background_envmap(const OShadeCtx *c, of3 rayDirection);
of4 oshade_fog_from_camera(const OShadeCtx *c, uint32_t instanceId, of3 position);
of4 oshade_fog_from_origin(const OShadeCtx *c, uint32_t instanceId, of3 position, of3 origin);

#endif
