/*
 * rt64_oracle.h -- CPU ORACLE for the RT64 ray-traced render path.   *** TEST INFRASTRUCTURE ***
 *
 * A scalar C11 restatement of the reference's per-frame render path (View::update + View::render and
 * every shader they dispatch), used ONLY as the checker for the HIP implementation: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the product (librt64.so) links,
 * includes or calls this code.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer vectors for this path and
 * cannot be built here (Win32 + D3D12 + DXR + DXC; the BVH build / traversal / ray-triangle arithmetic
 * live inside the DXR driver, not in the repository).  The oracle is therefore pinned only by
 *   (1) its line-by-line correspondence with the HLSL / C++ cited at every function,
 *   (2) self-consistency checks (BVH traversal == brute force over all triangles; closed-form answers for
 *       the sample camera, the floor plane and the analytic sphere; BC7 decode == an independent decoder).
 * Citations: "ref:<file>:<lines>" are relative to /root/reference/src/rt64lib/ (shaders/, private/, public/).
 *
 * Two families of results:
 *   - GEOMETRY (Morton codes, sort order, LBVH topology, node boxes, traversal hits: instance, primitive,
 *     t, u, v).  The arithmetic is specified operation by operation (explicit fmaf, no contraction) so that
 *     the HIP kernels reproduce it BIT-EXACTLY.  See "Geometry spec" in oracle_bvh.c / oracle_trace.c.
 *   - SHADING (everything after the hit).  Same formulas and the same storage-format quantisation points
 *     as the reference (RGBA8 / RGBA16F / SNORM16 images); libm vs GPU transcendental functions differ in
 *     the last bits, so parity is checked within a tolerance stated in the tests.
 */
#ifndef RT64_ORACLE_H
#define RT64_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- plain data mirrors of the ABI structs (same layouts as ref:public/rt64.h:98-205) ------------- */

typedef struct { float x, y, z; } ov3;
typedef struct { float x, y, z, w; } ov4;
typedef struct { float m[4][4]; } om4;            /* row-major, row-vector convention: p' = p * M */

typedef struct {                                  /* == RT64_MATERIAL, 132 bytes */
    int diffuseTexIndex, normalTexIndex, specularTexIndex;
    float ignoreNormalFactor, uvDetailScale;
    float reflectionFactor, reflectionFresnelFactor, reflectionShineFactor, refractionFactor;
    ov3 specularColor;
    float specularExponent, solidAlphaMultiplier, shadowAlphaMultiplier, depthBias, shadowRayBias;
    ov3 selfLight;
    unsigned int lightGroupMaskBits;
    ov3 fogColor;
    ov4 diffuseColorMix;
    float fogMul, fogOffset;
    unsigned int fogEnabled;
    float lockMask;
    int enabledAttributes;
} OMaterial;

typedef struct {                                  /* == RT64_LIGHT, 60 bytes */
    ov3 position, diffuseColor;
    float attenuationRadius, pointRadius;
    ov3 specularColor;
    float shadowOffset, attenuationExponent, flickerIntensity;
    unsigned int groupBits;
} OLight;

typedef struct {                                  /* == RT64_SCENE_DESC, 84 bytes */
    ov3 ambientBaseColor, ambientNoGIColor, eyeLightDiffuseColor, eyeLightSpecularColor;
    ov3 skyDiffuseMultiplier, skyHSLModifier;
    float skyYawOffset, giDiffuseStrength, giSkyStrength;
} OSceneDesc;

/* ---- oracle objects ------------------------------------------------------------------------------- */

typedef struct OTexture OTexture;
typedef struct OMesh OMesh;
typedef struct OScene OScene;

/* LBVH node, 64 bytes.  Children: bit 31 set = leaf (low bits: sorted leaf slot), else inner node index.
 * 0xFFFFFFFF = no child (single-leaf trees). */
typedef struct {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t left, right;
    uint32_t parent, pad;
} ONode;

typedef struct {                                  /* BLAS leaf payload in sorted order, 48 bytes */
    float v0[3]; uint32_t prim;
    float v1[3]; uint32_t pad1;
    float v2[3]; uint32_t pad2;
} OTri;

typedef struct {
    uint32_t count;                               /* leaves */
    ONode *nodes;                                 /* max(count-1, 1) inner nodes, root = 0 */
    uint32_t *sortedIndex;                        /* leaf slot -> primitive (BLAS) / instance (TLAS) */
    uint32_t *morton;                             /* per leaf slot, after sort */
    float bmin[3], bmax[3];                       /* bounds of all leaves */
} OBvh;

typedef struct {
    OMesh *mesh;
    om4 transform, previousTransform;
    OTexture *diffuse, *normal, *specular;        /* normal/specular may be NULL */
    uint32_t shaderId, filter, hAddr, vAddr;
    int shaderFlags;                              /* RT64_SHADER_* */
    OMaterial material;
    unsigned int flags;                           /* RT64_INSTANCE_* */
    int scissorRect[4], viewportRect[4];          /* RT64_RECT x, y, w, h (origin bottom-left, ref:rt64_view.cpp:1114-1136); w or h <= 0 = unset */
} OInstanceDesc;

typedef struct {
    int width, height;                            /* full frame */
    int tileY0, tileY1;                           /* rows to render */
    om4 view;
    float fovRadians, nearDist, farDist;
    int canReproject;
    /* RT64_VIEW_DESC subset + inspector-only knobs */
    unsigned int diSamples, giSamples, maxLights;
    int denoiserEnabled;
    int denoiserMode;                             /* 0 = reference 5x Gaussian (ref:rt64_view.cpp:1512-1530), 1 = SVGF a-trous */
    float motionBlurStrength; unsigned int motionBlurSamples;
    int maxReflections;
    /* oracle-only switches */
    int bruteForce;                               /* 1: test every triangle instead of walking the LBVH */
    int cullBehindOpaque;                         /* 1: shorten tmax behind fully opaque hits (GPU behaviour); 0: visit all (reference) */
    int threads;                                  /* OpenMP threads, 0 = default */
    /* RT64_VIEW_DESC.resolutionScale (ref:rt64_view.cpp:138-139): width/height above are the SCREEN size, every image except
     * finalRGBA8 is lround(screen * scale); 0 or 1 = off.  With a scale != 1 (or motion blur) the whole frame is rendered. */
    float resolutionScale;
    /* RT64_VIEW_DESC.upscaler / upscalerMode (RT64_UPSCALER_*, RT64_UPSCALER_MODE_*): AUTO / FSR select the built-in temporal upscaler
     * (oracle_upscale.c): the render size comes from the quality mode, primary rays are jittered, PostProcessPS reads the upscaled image. */
    int upscaler, upscalerMode;
    /* Extensions beyond the reference (which has one bounce per GI ray, ref:IndirectRayGen.hlsl:58-131, and one primary sample per pixel,
     * ref:rt64.h:172-182); both mirror device options of the HIP library ("Path-tracing extensions"):
     *   giBounces  0 / 1 = the reference's single bounce; 2 = a bounce ray that hits a surface sends a second cosine-weighted ray from there (rule B1-B3
     *              at pass_indirect) and the radiance that ray finds stands where the constant ambient term stands at the first hit;
     *   primarySpp 0 / 1 = off; N = the frame is N complete sub-frames (primary + direct + GI + filter + Compose each) whose primary rays are jittered by
     *              Halton(k + 1; 2, 3) - 0.5, k = 0 .. N - 1, and the composed RGBA32F outputs are averaged before PostProcess (rule P1-P4 at oracle_render). */
    int giBounces, primarySpp;
} OFrameParams;

/* Output images of one frame, full-frame row-major arrays owned by the scene (valid until next render). */
typedef struct {
    int width, height;                            /* render size (all images below except finalRGBA8) */
    const uint8_t *finalRGBA8;                    /* [screenHeight][screenWidth][4] */
    const float *outputRGBA32F;                   /* [h][w][4] ComposePS */
    const float *shadingPosition, *shadingNormal, *shadingSpecular, *diffuse;   /* [h][w][4], values after storage quantisation */
    const int32_t *instanceId;                    /* [h][w] */
    const float *directLight, *indirectLight, *filteredDirect, *filteredIndirect; /* [h][w][4] */
    const float *reflection, *refraction, *transparent, *viewDirection, *normal;  /* [h][w][4] */
    const float *flow;                            /* [h][w][2] */
    const float *reactiveMask, *lockMask, *depth; /* [h][w] */
    const uint32_t *primaryHit;                   /* [h][w][4]: t bits, u bits, v bits, inst<<24|prim ; all 0xFFFFFFFF on miss */
    /* counters */
    uint64_t primaryRays, shadowRays, indirectRays, reflectionRays, refractionRays;
    uint64_t nodesVisited, trianglesTested;       /* over all rays of the frame */
    uint64_t nodesVisitedPrimary, trianglesTestedPrimary, nodesVisitedShadow, trianglesTestedShadow;
    double secondsBuild, secondsRender;
    int screenWidth, screenHeight;                /* size of finalRGBA8 */
    const uint8_t *backgroundRGBA8;               /* [screenHeight][screenWidth][4]: gBackground (raster background instances), NULL when there are none */
    const float *upscaledRGBA32F;                 /* [screenHeight][screenWidth][4]: rtOutputUpscaled (rgb, accumulated frames), NULL without an upscaler */
    float pixelJitter[2];
} OFrameResult;

/* ---- API --------------------------------------------------------------------------------------------- */

OScene *oracle_scene_create(void);
void oracle_scene_destroy(OScene *s);
void oracle_scene_set_desc(OScene *s, const OSceneDesc *d);
void oracle_scene_set_lights(OScene *s, const OLight *lights, int count);
void oracle_scene_set_bluenoise(OScene *s, const uint8_t *rgba8_512x512);
void oracle_scene_set_sky(OScene *s, OTexture *t);
/* Instances in creation order, exactly the list the host would hand to RT64 (RT + raster ones). */
int oracle_scene_add_instance(OScene *s, const OInstanceDesc *d);
void oracle_scene_set_instance(OScene *s, int index, const OInstanceDesc *d);

OTexture *oracle_texture_create_rgba8(const uint8_t *bytes, int width, int height, int rowPitch);
OTexture *oracle_texture_create_dds(const uint8_t *bytes, size_t byteCount);   /* BC7 / uncompressed RGBA8 DDS */
void oracle_texture_destroy(OTexture *t);
int oracle_texture_info(const OTexture *t, int *width, int *height, int *mips);
const uint8_t *oracle_texture_mip(const OTexture *t, int mip, int *w, int *h);
void oracle_texture_sample(const OTexture *t, float u, float v, float ddxu, float ddxv, float ddyu, float ddyv,
                           int filter, int hAddr, int vAddr, float out[4]);

OMesh *oracle_mesh_create(int flags);
/* Mirrors RT64_SetMesh (ref:private/rt64_mesh.cpp:195-205): copies the arrays; builds the BLAS, or refits it
 * when the mesh is UPDATABLE and counts are unchanged. */
void oracle_mesh_set(OMesh *m, const void *vertices, int vertexCount, int vertexStride, const uint32_t *indices, int indexCount);
void oracle_mesh_destroy(OMesh *m);
const OBvh *oracle_mesh_bvh(const OMesh *m);
const OTri *oracle_mesh_tris(const OMesh *m);

/* Render one frame (View::update + View::render).  History (previous depth/normal/accumulation, previous
 * matrices, frameCount) lives in the scene, as in the reference's View. */
int oracle_render(OScene *s, const OFrameParams *p, OFrameResult *out);
const OBvh *oracle_scene_tlas(const OScene *s);
uint32_t oracle_scene_frame_count(const OScene *s);

/* Small known-answer helpers exported for unit tests. */
uint32_t oracle_init_rand(uint32_t v0, uint32_t v1, uint32_t backoff);      /* ref:shaders/Random.hlsli:14-26 */
float oracle_next_rand(uint32_t *s);                                        /* ref:shaders/Random.hlsli:28-37 */
float oracle_halton(int i, int b);                                          /* ref:private/rt64_common.h:347-357 */
/* render size + jitter phase count of the built-in upscaler for a display size; 0 = this (upscaler, mode) selects no upscaler */
int oracle_upscaler_info(int upscaler, int mode, int displayW, int displayH, int *renderW, int *renderH, int *phases);
void oracle_rgb_to_hsl(const float rgb[3], float hsl[3]);                   /* ref:shaders/Color.hlsli:36-42 */
void oracle_hsl_to_rgb(const float hsl[3], float rgb[3]);                   /* ref:shaders/Color.hlsli:29-34 */
void oracle_fake_envmap_uv(const float dir[3], float yaw, float uv[2]);     /* ref:shaders/BgSky.hlsli:14-18 */
void oracle_perspective_fov_rh(float fov, float aspect, float zn, float zf, om4 *out);  /* XMMatrixPerspectiveFovRH */
int oracle_matrix_inverse(const om4 *m, om4 *out);
uint32_t oracle_morton30(uint32_t x, uint32_t y, uint32_t z);
uint16_t oracle_f32_to_f16(float f);
float oracle_f16_to_f32(uint16_t h);
void oracle_decode_bc7_block(const uint8_t block[16], uint8_t rgba[64]);
/* Colour-combiner decode, ref:private/rt64_shader.cpp:32-96.  out[0..7] = c[0][0..3], c[1][0..3]; out[8]=inputCount,
 * out[9]=useTex0, out[10]=useTex1, out[11..12]=do_single, [13..14]=do_multiply, [15..16]=do_mix, [17]=color_alpha_same,
 * [18]=opt_alpha, [19]=opt_texture_edge, [20]=opt_noise, [21]=vertexSize, [22]=normalOffset, [23]=uvOffset, [24..27]=inputOffset */
void oracle_decode_combiner(uint32_t shaderId, int out[28]);

#ifdef __cplusplus
}
#endif
#endif
