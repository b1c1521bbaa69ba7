/*
 * rt64.h -- portable C ABI of the RT64 render path, MI355X (HIP) implementation.
 *
 * This header is the drop-in boundary.  It keeps every public identifier, constant value, POD
 * layout and export name of the reference's plugin header
 *     /root/reference/src/rt64lib/public/rt64.h            (cited below as ref:NNN)
 * so that a C host which today does  LoadLibrary("rt64lib.dll") + GetProcAddress x33  can
 * instead  dlopen("librt64.so") + dlsym x33  and keep calling through the same RT64_LIBRARY
 * function-pointer table.  Differences, all additive or platform shims:
 *   - Win32 types are shimmed when <Windows.h> is absent (HMODULE -> void*, UINT/WPARAM/LPARAM).
 *   - `bool` comes from <stdbool.h> in C (the reference relies on <Windows.h>/C++ for it).
 *   - RT64_LoadLibrary() uses dlopen/dlsym on non-Windows hosts.
 *   - A block of additive RT64_* exports at the end ("MI355X extensions") gives a headless host
 *     what the Win32 window gave the reference: a size source, readback, tiles, timings.
 *     No existing signature changes; hosts that ignore the extensions behave as before.
 *
 * The function list is written once (RT64_API_LIST) and expanded three times: pointer typedefs,
 * RT64_LIBRARY members, loader.  Order of members == ref:305-342.
 */
#ifndef RT64_H_INCLUDED
#define RT64_H_INCLUDED

#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>
#include <stdint.h>

#if defined(_WIN32)
#   include <Windows.h>
    typedef HMODULE RT64_MODULE;
    typedef UINT RT64_UINT; typedef WPARAM RT64_WPARAM; typedef LPARAM RT64_LPARAM;
#   define RT64_DLSYM(h, n) GetProcAddress((h), (n))
#else
#   include <dlfcn.h>
    typedef void *RT64_MODULE;                  /* ref:306 HMODULE handle */
    typedef unsigned int RT64_UINT;             /* ref:296 UINT msg */
    typedef uintptr_t RT64_WPARAM;              /* ref:296 WPARAM */
    typedef intptr_t RT64_LPARAM;               /* ref:296 LPARAM */
#   define RT64_DLSYM(h, n) dlsym((h), (n))
#endif

#if !defined(__cplusplus)
#   include <stdbool.h>
#   define RT64_INLINE static inline
#else
#   define RT64_INLINE inline
#endif

/* ---- constants (values identical to ref:11-86) -------------------------------------------- */

enum { /* material filter / addressing / colour-combiner sources, ref:11-24 */
    RT64_MATERIAL_FILTER_POINT = 0, RT64_MATERIAL_FILTER_LINEAR = 1,
    RT64_MATERIAL_ADDR_WRAP = 0, RT64_MATERIAL_ADDR_MIRROR = 1, RT64_MATERIAL_ADDR_CLAMP = 2,
    RT64_MATERIAL_CC_SHADER_0 = 0, RT64_MATERIAL_CC_SHADER_INPUT_1 = 1, RT64_MATERIAL_CC_SHADER_INPUT_2 = 2,
    RT64_MATERIAL_CC_SHADER_INPUT_3 = 3, RT64_MATERIAL_CC_SHADER_INPUT_4 = 4, RT64_MATERIAL_CC_SHADER_TEXEL0 = 5,
    RT64_MATERIAL_CC_SHADER_TEXEL0A = 6, RT64_MATERIAL_CC_SHADER_TEXEL1 = 7
};

/* Material attribute bits, ref:26-42.  X(name, bit, member) -- also drives RT64_ApplyMaterialAttributes. */
#define RT64_ATTRIBUTE_LIST(X) \
    X(IGNORE_NORMAL_FACTOR,      0x0001, ignoreNormalFactor) \
    X(UV_DETAIL_SCALE,           0x0002, uvDetailScale) \
    X(REFLECTION_FACTOR,         0x0004, reflectionFactor) \
    X(REFLECTION_FRESNEL_FACTOR, 0x0008, reflectionFresnelFactor) \
    X(REFLECTION_SHINE_FACTOR,   0x0010, reflectionShineFactor) \
    X(REFRACTION_FACTOR,         0x0020, refractionFactor) \
    X(SPECULAR_COLOR,            0x0040, specularColor) \
    X(SPECULAR_EXPONENT,         0x0080, specularExponent) \
    X(SOLID_ALPHA_MULTIPLIER,    0x0100, solidAlphaMultiplier) \
    X(SHADOW_ALPHA_MULTIPLIER,   0x0200, shadowAlphaMultiplier) \
    X(DEPTH_BIAS,                0x0400, depthBias) \
    X(SHADOW_RAY_BIAS,           0x0800, shadowRayBias) \
    X(SELF_LIGHT,                0x1000, selfLight) \
    X(LIGHT_GROUP_MASK_BITS,     0x2000, lightGroupMaskBits) \
    X(DIFFUSE_COLOR_MIX,         0x4000, diffuseColorMix)

enum {
    RT64_ATTRIBUTE_NONE = 0x0000,
#define RT64_X(name, bit, member) RT64_ATTRIBUTE_##name = bit,
    RT64_ATTRIBUTE_LIST(RT64_X)
#undef RT64_X
    RT64_ATTRIBUTE_ALL_ = 0x7FFF
};

enum { /* ref:44-63 */
    RT64_MESH_RAYTRACE_ENABLED = 0x1, RT64_MESH_RAYTRACE_UPDATABLE = 0x2,
    RT64_MESH_RAYTRACE_FAST_TRACE = 0x4, RT64_MESH_RAYTRACE_COMPACT = 0x8,
    RT64_SHADER_FILTER_POINT = 0x0, RT64_SHADER_FILTER_LINEAR = 0x1,
    RT64_SHADER_ADDRESSING_WRAP = 0x0, RT64_SHADER_ADDRESSING_MIRROR = 0x1, RT64_SHADER_ADDRESSING_CLAMP = 0x2,
    RT64_SHADER_RASTER_ENABLED = 0x1, RT64_SHADER_RAYTRACE_ENABLED = 0x2,
    RT64_SHADER_NORMAL_MAP_ENABLED = 0x4, RT64_SHADER_SPECULAR_MAP_ENABLED = 0x8,
    RT64_INSTANCE_RASTER_BACKGROUND = 0x1, RT64_INSTANCE_DISABLE_BACKFACE_CULLING = 0x2
};

#define RT64_LIGHT_GROUP_MASK_ALL   0xFFFFFFFFu   /* ref:66 */
#define RT64_LIGHT_GROUP_DEFAULT    0x1u          /* ref:67 */
#define RT64_LIGHT_MAX_SAMPLES      128           /* ref:68 */

enum { /* ref:70-86 */
    RT64_UPSCALER_OFF = 0x0, RT64_UPSCALER_AUTO = 0x1, RT64_UPSCALER_DLSS = 0x2, RT64_UPSCALER_FSR = 0x3, RT64_UPSCALER_XESS = 0x4,
    RT64_UPSCALER_MODE_AUTO = 0x0, RT64_UPSCALER_MODE_ULTRA_PERFORMANCE = 0x1, RT64_UPSCALER_MODE_PERFORMANCE = 0x2,
    RT64_UPSCALER_MODE_BALANCED = 0x3, RT64_UPSCALER_MODE_QUALITY = 0x4, RT64_UPSCALER_MODE_ULTRA_QUALITY = 0x5,
    RT64_UPSCALER_MODE_NATIVE = 0x6,
    RT64_TEXTURE_FORMAT_RGBA8 = 0x1, RT64_TEXTURE_FORMAT_DDS = 0x2
};

/* The reference spells every constant above as a preprocessor macro (ref:11-86), so a host may test one with #ifdef / #if.  They are enumerators here
   (typed, visible to a debugger) AND macros of the same value, defined after the enumerations so that the two never meet in one declaration;
   tests/test_boundary_cpu.py holds the two spellings -- and the reference's -- together. */
#define RT64_MATERIAL_FILTER_POINT               0
#define RT64_MATERIAL_FILTER_LINEAR              1
#define RT64_MATERIAL_ADDR_WRAP                  0
#define RT64_MATERIAL_ADDR_MIRROR                1
#define RT64_MATERIAL_ADDR_CLAMP                 2
#define RT64_MATERIAL_CC_SHADER_0                0
#define RT64_MATERIAL_CC_SHADER_INPUT_1          1
#define RT64_MATERIAL_CC_SHADER_INPUT_2          2
#define RT64_MATERIAL_CC_SHADER_INPUT_3          3
#define RT64_MATERIAL_CC_SHADER_INPUT_4          4
#define RT64_MATERIAL_CC_SHADER_TEXEL0           5
#define RT64_MATERIAL_CC_SHADER_TEXEL0A          6
#define RT64_MATERIAL_CC_SHADER_TEXEL1           7
#define RT64_ATTRIBUTE_NONE                      0x0000
#define RT64_MESH_RAYTRACE_ENABLED               0x1
#define RT64_MESH_RAYTRACE_UPDATABLE             0x2
#define RT64_MESH_RAYTRACE_FAST_TRACE            0x4
#define RT64_MESH_RAYTRACE_COMPACT               0x8
#define RT64_SHADER_FILTER_POINT                 0x0
#define RT64_SHADER_FILTER_LINEAR                0x1
#define RT64_SHADER_ADDRESSING_WRAP              0x0
#define RT64_SHADER_ADDRESSING_MIRROR            0x1
#define RT64_SHADER_ADDRESSING_CLAMP             0x2
#define RT64_SHADER_RASTER_ENABLED               0x1
#define RT64_SHADER_RAYTRACE_ENABLED             0x2
#define RT64_SHADER_NORMAL_MAP_ENABLED           0x4
#define RT64_SHADER_SPECULAR_MAP_ENABLED         0x8
#define RT64_INSTANCE_RASTER_BACKGROUND          0x1
#define RT64_INSTANCE_DISABLE_BACKFACE_CULLING   0x2
#define RT64_UPSCALER_OFF                        0x0
#define RT64_UPSCALER_AUTO                       0x1
#define RT64_UPSCALER_DLSS                       0x2
#define RT64_UPSCALER_FSR                        0x3
#define RT64_UPSCALER_XESS                       0x4
#define RT64_UPSCALER_MODE_AUTO                  0x0
#define RT64_UPSCALER_MODE_ULTRA_PERFORMANCE     0x1
#define RT64_UPSCALER_MODE_PERFORMANCE           0x2
#define RT64_UPSCALER_MODE_BALANCED              0x3
#define RT64_UPSCALER_MODE_QUALITY               0x4
#define RT64_UPSCALER_MODE_ULTRA_QUALITY         0x5
#define RT64_UPSCALER_MODE_NATIVE                0x6
#define RT64_TEXTURE_FORMAT_RGBA8                0x1
#define RT64_TEXTURE_FORMAT_DDS                  0x2
#define RT64_ATTRIBUTE_IGNORE_NORMAL_FACTOR      0x0001
#define RT64_ATTRIBUTE_UV_DETAIL_SCALE           0x0002
#define RT64_ATTRIBUTE_REFLECTION_FACTOR         0x0004
#define RT64_ATTRIBUTE_REFLECTION_FRESNEL_FACTOR 0x0008
#define RT64_ATTRIBUTE_REFLECTION_SHINE_FACTOR   0x0010
#define RT64_ATTRIBUTE_REFRACTION_FACTOR         0x0020
#define RT64_ATTRIBUTE_SPECULAR_COLOR            0x0040
#define RT64_ATTRIBUTE_SPECULAR_EXPONENT         0x0080
#define RT64_ATTRIBUTE_SOLID_ALPHA_MULTIPLIER    0x0100
#define RT64_ATTRIBUTE_SHADOW_ALPHA_MULTIPLIER   0x0200
#define RT64_ATTRIBUTE_DEPTH_BIAS                0x0400
#define RT64_ATTRIBUTE_SHADOW_RAY_BIAS           0x0800
#define RT64_ATTRIBUTE_SELF_LIGHT                0x1000
#define RT64_ATTRIBUTE_LIGHT_GROUP_MASK_BITS     0x2000
#define RT64_ATTRIBUTE_DIFFUSE_COLOR_MIX         0x4000

/* ---- opaque handles, ref:88-96 ------------------------------------------------------------- */

typedef struct RT64_DEVICE RT64_DEVICE;
typedef struct RT64_VIEW RT64_VIEW;
typedef struct RT64_SCENE RT64_SCENE;
typedef struct RT64_INSTANCE RT64_INSTANCE;
typedef struct RT64_MESH RT64_MESH;
typedef struct RT64_TEXTURE RT64_TEXTURE;
typedef struct RT64_SHADER RT64_SHADER;
typedef struct RT64_INSPECTOR RT64_INSPECTOR;

/* ---- POD descriptors.  Sizes/offsets are the x86-64 layouts of ref:98-205 (see static asserts) */

typedef struct { float x, y; } RT64_VECTOR2;
typedef struct { float x, y, z; } RT64_VECTOR3;
typedef struct { float x, y, z, w; } RT64_VECTOR4;
typedef struct { float m[4][4]; } RT64_MATRIX4;      /* row-major, row-vector convention: p' = p * M */
typedef struct { int x, y, w, h; } RT64_RECT;

typedef struct {                                     /* ref:118-145, 132 bytes */
    int diffuseTexIndex, normalTexIndex, specularTexIndex;
    float ignoreNormalFactor, uvDetailScale;
    float reflectionFactor, reflectionFresnelFactor, reflectionShineFactor, refractionFactor;
    RT64_VECTOR3 specularColor;
    float specularExponent, solidAlphaMultiplier, shadowAlphaMultiplier, depthBias, shadowRayBias;
    RT64_VECTOR3 selfLight;
    unsigned int lightGroupMaskBits;
    RT64_VECTOR3 fogColor;
    RT64_VECTOR4 diffuseColorMix;
    float fogMul, fogOffset;
    unsigned int fogEnabled;
    float lockMask;
    int enabledAttributes;                           /* which members a modifier material overrides */
} RT64_MATERIAL;

typedef struct {                                     /* ref:148-158, 60 bytes */
    RT64_VECTOR3 position, diffuseColor;
    float attenuationRadius, pointRadius;
    RT64_VECTOR3 specularColor;
    float shadowOffset, attenuationExponent, flickerIntensity;
    unsigned int groupBits;
} RT64_LIGHT;

typedef struct {                                     /* ref:160-170, 84 bytes */
    RT64_VECTOR3 ambientBaseColor, ambientNoGIColor, eyeLightDiffuseColor, eyeLightSpecularColor;
    RT64_VECTOR3 skyDiffuseMultiplier, skyHSLModifier;
    float skyYawOffset, giDiffuseStrength, giSkyStrength;
} RT64_SCENE_DESC;

typedef struct {                                     /* ref:172-182, 32 bytes */
    float resolutionScale, motionBlurStrength;
    unsigned int diSamples, giSamples, maxLights;
    unsigned char upscaler, upscalerMode;
    float upscalerSharpness;
    bool denoiserEnabled;
} RT64_VIEW_DESC;

typedef struct {                                     /* ref:184-196, 336 bytes */
    RT64_MESH *mesh;
    RT64_MATRIX4 transform, previousTransform;
    RT64_TEXTURE *diffuseTexture, *normalTexture, *specularTexture;
    RT64_SHADER *shader;
    RT64_MATERIAL material;
    RT64_RECT scissorRect, viewportRect;
    unsigned int flags;
} RT64_INSTANCE_DESC;

typedef struct {                                     /* ref:198-205, 32 bytes */
    void *bytes;
    int byteCount, format, width, height, rowPitch;
} RT64_TEXTURE_DESC;

/* ref:207-267: copy every member flagged in src->enabledAttributes from src into dst. */
RT64_INLINE void RT64_ApplyMaterialAttributes(RT64_MATERIAL *dst, RT64_MATERIAL *src) {
#define RT64_X(name, bit, member) if (src->enabledAttributes & (bit)) { dst->member = src->member; }
    RT64_ATTRIBUTE_LIST(RT64_X)
#undef RT64_X
}

/* ---- the function table -------------------------------------------------------------------- */

/* X(Member, ExportedSymbol, ReturnType, (args)) in the member order of ref:305-342.
 * CORE = present in RT64_MINIMAL builds too (ref:307-309); FULL = the rest (ref:311-340).       */
#define RT64_API_LIST_CORE(X) \
    X(GetLastError,  RT64_GetLastError,  const char *,  (void)) \
    X(CreateDevice,  RT64_CreateDevice,  RT64_DEVICE *, (void *hwnd)) \
    X(DestroyDevice, RT64_DestroyDevice, void,          (RT64_DEVICE *device))

#define RT64_API_LIST_FULL(X) \
    X(DrawDevice, RT64_DrawDevice, void, (RT64_DEVICE *device, int vsyncInterval, float deltaTimeMs)) \
    X(CreateView, RT64_CreateView, RT64_VIEW *, (RT64_SCENE *scenePtr)) \
    X(SetViewPerspective, RT64_SetViewPerspective, void, (RT64_VIEW *viewPtr, RT64_MATRIX4 viewMatrix, float fovRadians, float nearDist, float farDist, bool canReproject)) \
    X(SetViewDescription, RT64_SetViewDescription, void, (RT64_VIEW *viewPtr, RT64_VIEW_DESC viewDesc)) \
    X(SetViewSkyPlane, RT64_SetViewSkyPlane, void, (RT64_VIEW *viewPtr, RT64_TEXTURE *texturePtr)) \
    X(GetViewRaytracedInstanceAt, RT64_GetViewRaytracedInstanceAt, RT64_INSTANCE *, (RT64_VIEW *viewPtr, int x, int y)) \
    X(GetViewUpscalerSupport, RT64_GetViewUpscalerSupport, bool, (RT64_VIEW *viewPtr, char upscaler)) \
    X(DestroyView, RT64_DestroyView, void, (RT64_VIEW *viewPtr)) \
    X(CreateScene, RT64_CreateScene, RT64_SCENE *, (RT64_DEVICE *devicePtr)) \
    X(SetSceneDescription, RT64_SetSceneDescription, void, (RT64_SCENE *scenePtr, RT64_SCENE_DESC sceneDesc)) \
    X(SetSceneLights, RT64_SetSceneLights, void, (RT64_SCENE *scenePtr, RT64_LIGHT *lightArray, int lightCount)) \
    X(DestroyScene, RT64_DestroyScene, void, (RT64_SCENE *scenePtr)) \
    X(CreateMesh, RT64_CreateMesh, RT64_MESH *, (RT64_DEVICE *devicePtr, int flags)) \
    X(SetMesh, RT64_SetMesh, void, (RT64_MESH *meshPtr, void *vertexArray, int vertexCount, int vertexStride, unsigned int *indexArray, int indexCount)) \
    X(DestroyMesh, RT64_DestroyMesh, void, (RT64_MESH *meshPtr)) \
    X(CreateShader, RT64_CreateShader, RT64_SHADER *, (RT64_DEVICE *devicePtr, unsigned int shaderId, unsigned int filter, unsigned int hAddr, unsigned int vAddr, int flags)) \
    X(DestroyShader, RT64_DestroyShader, void, (RT64_SHADER *shaderPtr)) \
    X(CreateInstance, RT64_CreateInstance, RT64_INSTANCE *, (RT64_SCENE *scenePtr)) \
    X(SetInstanceDescription, RT64_SetInstanceDescription, void, (RT64_INSTANCE *instancePtr, RT64_INSTANCE_DESC instanceDesc)) \
    X(DestroyInstance, RT64_DestroyInstance, void, (RT64_INSTANCE *instancePtr)) \
    X(CreateTexture, RT64_CreateTexture, RT64_TEXTURE *, (RT64_DEVICE *devicePtr, RT64_TEXTURE_DESC textureDesc)) \
    X(DestroyTexture, RT64_DestroyTexture, void, (RT64_TEXTURE *texture)) \
    X(CreateInspector, RT64_CreateInspector, RT64_INSPECTOR *, (RT64_DEVICE *devicePtr)) \
    X(HandleMessageInspector, RT64_HandleMessageInspector, bool, (RT64_INSPECTOR *inspectorPtr, RT64_UINT msg, RT64_WPARAM wParam, RT64_LPARAM lParam)) \
    X(PrintClearInspector, RT64_PrintClearInspector, void, (RT64_INSPECTOR *inspectorPtr)) \
    X(PrintMessageInspector, RT64_PrintMessageInspector, void, (RT64_INSPECTOR *inspectorPtr, const char *message)) \
    X(SetSceneInspector, RT64_SetSceneInspector, void, (RT64_INSPECTOR *inspectorPtr, RT64_SCENE_DESC *sceneDesc)) \
    X(SetMaterialInspector, RT64_SetMaterialInspector, void, (RT64_INSPECTOR *inspectorPtr, RT64_MATERIAL *material, const char *materialName)) \
    X(SetLightsInspector, RT64_SetLightsInspector, void, (RT64_INSPECTOR *inspectorPtr, RT64_LIGHT *lights, int *lightCount, int maxLightCount)) \
    X(DestroyInspector, RT64_DestroyInspector, void, (RT64_INSPECTOR *inspectorPtr))

#ifdef RT64_MINIMAL
#   define RT64_API_LIST(X) RT64_API_LIST_CORE(X)
#else
#   define RT64_API_LIST(X) RT64_API_LIST_CORE(X) RT64_API_LIST_FULL(X)
#endif

/* Pointer typedefs: GetLastErrorPtr, CreateDevicePtr, ... (ref:269-302). */
#define RT64_X(member, symbol, ret, args) typedef ret (*member##Ptr) args;
RT64_API_LIST_CORE(RT64_X)
RT64_API_LIST_FULL(RT64_X)
#undef RT64_X

/* ref:305-342 */
typedef struct {
    RT64_MODULE handle;
#define RT64_X(member, symbol, ret, args) member##Ptr member;
    RT64_API_LIST(RT64_X)
#undef RT64_X
} RT64_LIBRARY;

/* Library file names searched by RT64_LoadLibrary (ref:349-355 picks rt64libm/rt64libd/rt64lib.dll). */
#if defined(_WIN32)
#   define RT64_LIBRARY_FILE "rt64lib.dll"
#else
#   define RT64_LIBRARY_FILE "librt64.so"
#endif

/* ref:346-402.  Unresolved symbols are left NULL exactly like the reference (no check).
 * The file can be overridden with the RT64_LIBRARY_PATH environment variable (additive). */
RT64_INLINE RT64_LIBRARY RT64_LoadLibraryFrom(const char *path) {
    RT64_LIBRARY lib;
    const char *file = path ? path : RT64_LIBRARY_FILE;
#if defined(_WIN32)
    lib.handle = LoadLibraryA(file);
#else
    lib.handle = dlopen(file, RTLD_NOW | RTLD_LOCAL);
#endif
    if (lib.handle != 0) {
#define RT64_X(member, symbol, ret, args) lib.member = (member##Ptr)(RT64_DLSYM(lib.handle, #symbol));
        RT64_API_LIST(RT64_X)
#undef RT64_X
    }
    else {
#if defined(_WIN32)
        fprintf(stderr, "Error when loading library: %lu\n", (unsigned long)GetLastError());
#else
        fprintf(stderr, "Error when loading library: %s\n", dlerror());
#endif
#define RT64_X(member, symbol, ret, args) lib.member = 0;
        RT64_API_LIST(RT64_X)
#undef RT64_X
    }
    return lib;
}

RT64_INLINE RT64_LIBRARY RT64_LoadLibrary(void) {
    const char *env = 0;
#if !defined(_WIN32)
    env = getenv("RT64_LIBRARY_PATH");
#endif
    return RT64_LoadLibraryFrom(env);
}

RT64_INLINE void RT64_UnloadLibrary(RT64_LIBRARY lib) {   /* ref:404-406 */
    if (lib.handle != 0) {
#if defined(_WIN32)
        FreeLibrary(lib.handle);
#else
        dlclose(lib.handle);
#endif
    }
}

/* ============================================================================================
 * MI355X extensions (additive exports of librt64.so; resolved with dlsym like the rest).
 * The reference takes its frame size from the Win32 window (rt64_device.cpp:199-231) and presents
 * to a swap chain; a headless HIP device needs an explicit size source and a way to read pixels.
 * ============================================================================================ */

/* Images that RT64_ReadbackDevice / RT64_CopyDeviceImage can return.  Element type in brackets.
 * 0 is the presented frame; 1..16 follow the reference's debug view order (GlobalParams.hlsli:45-61).
 * Both calls return the rows of the LAST RENDERED frame -- the tile / strip partition that frame was drawn with (RT64_FRAME_STATS
 * tileY0 / tileY1 / stripRank / stripCount / rowsRendered), packed in ascending row order -- whatever RT64_SetDeviceTile /
 * RT64_SetDeviceInterleave / RT64_SetGatherBands have set for the next one. */
enum {
    RT64_IMAGE_FINAL_RGBA8 = 0,        /* [u8 x4]  back buffer after PostProcessPS                  */
    RT64_IMAGE_SHADING_POSITION = 1,   /* [f32 x4] */
    RT64_IMAGE_SHADING_NORMAL = 2,     /* [f32 x4] values as stored (RGBA16F precision)             */
    RT64_IMAGE_SHADING_SPECULAR = 3,   /* [f32 x4] */
    RT64_IMAGE_DIFFUSE = 4,            /* [f32 x4] values as stored (RGBA8 precision)               */
    RT64_IMAGE_INSTANCE_ID = 5,        /* [i32]    */
    RT64_IMAGE_DIRECT_LIGHT_RAW = 6,   /* [f32 x4] */
    RT64_IMAGE_DIRECT_LIGHT_FILTERED = 7,
    RT64_IMAGE_INDIRECT_LIGHT_RAW = 8,
    RT64_IMAGE_INDIRECT_LIGHT_FILTERED = 9,
    RT64_IMAGE_REFLECTION = 10,
    RT64_IMAGE_REFRACTION = 11,
    RT64_IMAGE_TRANSPARENT = 12,
    RT64_IMAGE_FLOW = 13,              /* [f32 x2] */
    RT64_IMAGE_REACTIVE_MASK = 14,     /* [f32]    */
    RT64_IMAGE_LOCK_MASK = 15,         /* [f32]    */
    RT64_IMAGE_DEPTH = 16,             /* [f32]    */
    RT64_IMAGE_OUTPUT_RGBA32F = 17,    /* [f32 x4] ComposePS result (rtOutput)                       */
    RT64_IMAGE_PRIMARY_HIT = 18,       /* [u32 x4] first-hit record: t bits, u bits, v bits, (instance<<24 | primitive), 0xFFFFFFFF = miss */
    RT64_IMAGE_VIEW_DIRECTION = 19,    /* [f32 x4] */
    RT64_IMAGE_FIRST_INSTANCE_ID = 20, /* [i32]    copy used by GetViewRaytracedInstanceAt           */
    RT64_IMAGE_BACKGROUND = 21,        /* [u8 x 4] gBackground: raster background instances, screen size (zeros when there are none) */
    RT64_IMAGE_UPSCALED = 22,          /* [f32 x4] rtOutputUpscaled, screen size: colour + accumulated frames (only behind an upscaler) */
    RT64_IMAGE_COUNT_ = 23
};

/* Arrays returned by RT64_ReadbackMeshAccel / RT64_ReadbackViewAccel. */
enum {
    RT64_ACCEL_NODES = 0,              /* 64-byte LBVH nodes: lmin[3] lmax[3] rmin[3] rmax[3] left right parent pad, max(n-1,1) of them */
    RT64_ACCEL_TRIANGLES = 1,          /* 48-byte leaves in Morton order: v0[3] prim v1[3] pad v2[3] pad (BLAS only) */
    RT64_ACCEL_SORTED_INDEX = 2,       /* u32[n]: leaf slot -> primitive (BLAS) / instance (TLAS) */
    RT64_ACCEL_MORTON = 3,             /* u32[n]: 30-bit Morton code per leaf slot */
    RT64_ACCEL_HEADER = 4,             /* bmin[3] count bmax[3] depth (inner nodes on the longest root-to-leaf path; 255 = built by the multi-kernel path) */
    RT64_ACCEL_HOST_DEPTH = 5          /* u32 (BLAS only): the depth the host computed from the triangles at RT64_SetMesh -- what sizes the traversal stacks; equals the header's */
};

/* Per-frame counters and GPU timings of the last RT64_DrawDevice (milliseconds, HIP events on the device stream). */
typedef struct {
    unsigned int structSize;           /* caller sets to sizeof(RT64_FRAME_STATS) */
    unsigned int width, height;        /* render size: lround(screen size x resolutionScale) */
    unsigned int tileY0, tileY1;       /* rows this device rendered */
    unsigned long long primaryRays, shadowRays, indirectRays, reflectionRays, refractionRays;
    unsigned long long nodesVisited, trianglesTested;      /* only when option "count_traversal" = 1 */
    float msTotal;                     /* whole frame on the GPU stream */
    float msBuild;                     /* BLAS (re)builds/refits executed this frame + TLAS build */
    float msPrimary, msDirect, msIndirect, msReflectRefract, msDenoise, msComposePost;
    float msHostWall;                  /* host wall clock of RT64_DrawDevice */
    unsigned int blasNodeBytes, blasTriangleBytes, tlasNodeBytes, instanceCount, triangleCount;
    float msPrimaryTrace, msPrimaryShade;   /* split of msPrimary: pure traversal kernel / shading kernel */
    unsigned int stripRank, stripCount;     /* interleaved 16-row strips (RT64_SetDeviceInterleave), count 1 = off */
    unsigned int rowsRendered;              /* rows of the frame this device rendered */
    unsigned int leanFrame;                 /* 1: images no pass consumed were skipped this frame (produced on readback) */
    /* per-pass split of nodesVisited / trianglesTested (count_traversal = 1) */
    unsigned long long nodesPrimary, trianglesPrimary, nodesDirect, trianglesDirect, nodesIndirect, trianglesIndirect;
    unsigned int screenWidth, screenHeight; /* back-buffer size; width/height above are the render size (screen x RT64_VIEW_DESC.resolutionScale) */
    /* Sums of the per-frame timings above over every frame since RT64_SetDeviceOption("reset_accum", 1): a host that times many
       frames reads them once instead of calling RT64_GetDeviceStats inside its frame loop. */
    unsigned int accumFrames;
    float accumMsTotal, accumMsBuild, accumMsPrimaryTrace, accumMsPrimaryShade, accumMsDirect, accumMsIndirect, accumMsReflectRefract, accumMsDenoise, accumMsComposePost;
    unsigned int fusedFrame;                /* 1: the lean frame ran as ONE kernel (primary visibility + resolve + direct light + compose); 2: a full frame's primary visibility + G-buffer + direct light ran as one kernel; the kernel's time is reported as msPrimaryTrace */
    unsigned int packedFinal;               /* 1: the frame wrote its owned back-buffer rows to the RT64_SetDeviceGatherTarget memory */
    unsigned int overlappedFrame;           /* 1: the frame was enqueued on the device's second render stream, beside the frame before it (sync_present = 0, option overlap_frames; pixel-local frames on unchanged tables only) */
    unsigned int reflectionBesideDenoiser;  /* 1: the reflection passes ran on a second stream beside the denoiser's iterations; msReflectRefract is their own time there and overlaps msDenoise */
    unsigned int traversalOverflow;         /* traversal-stack entries dropped by the frame's rays (count_traversal = 1); anything but 0 is an error RT64_DrawDevice also reports through RT64_GetLastError */
} RT64_FRAME_STATS;

#define RT64_EXT_API_LIST(X) \
    /* Headless device on HIP device `hipDevice` (-1: current) rendering width x height. \
       RT64_CreateDevice(NULL) is the same with RT64_WIDTH/RT64_HEIGHT/RT64_HIP_DEVICE env vars (default 1280x720, device 0). */ \
    X(CreateDeviceHeadless, RT64_CreateDeviceHeadless, RT64_DEVICE *, (int width, int height, int hipDevice)) \
    /* Stands in for a window resize; takes effect at the next RT64_DrawDevice like rt64_device.cpp:1039. */ \
    X(SetDeviceSize, RT64_SetDeviceSize, void, (RT64_DEVICE *device, int width, int height)) \
    /* Image-tile partition: this device renders rows [y0, y1) of the frame (default: all). */ \
    X(SetDeviceTile, RT64_SetDeviceTile, void, (RT64_DEVICE *device, int y0, int y1)) \
    /* Interleaved partition for load balance: the frame is cut into 16-row strips and this device renders strips \
       rank, rank+count, rank+2*count, ... (count <= 1 turns it off).  Combines with the row range of SetDeviceTile. */ \
    X(SetDeviceInterleave, RT64_SetDeviceInterleave, void, (RT64_DEVICE *device, int rank, int count)) \
    /* Copy image `image` (RT64_IMAGE_*) of the first view to host memory. Returns bytes written, 0 on error. \
       Only the rows this device rendered are returned, tightly packed in ascending row order. */ \
    X(ReadbackDevice, RT64_ReadbackDevice, size_t, (RT64_DEVICE *device, int image, void *dst, size_t dstBytes)) \
    /* Same, device-to-device into a caller-owned device pointer (e.g. a torch tensor feeding an RCCL gather), \
       ordered on the device's stream; the call returns after the copy has completed. */ \
    X(CopyDeviceImage, RT64_CopyDeviceImage, size_t, (RT64_DEVICE *device, int image, void *devicePtr, size_t dstBytes)) \
    /* Device memory (e.g. the send buffer of an RCCL gather) that the frames from now on ALSO write the back-buffer pixels of the rows \
       this device owns to, tightly packed in the order RT64_CopyDeviceImage(RT64_IMAGE_FINAL_RGBA8) returns them -- by the frame's own \
       last kernel, so no copy is queued after it.  Honoured by frames whose last pass is the one-kernel frame (RT64_FRAME_STATS.packedFinal \
       = 1 tells; any other frame leaves the memory alone and the caller copies as before).  devicePtr = NULL turns it off. */ \
    X(SetDeviceGatherTarget, RT64_SetDeviceGatherTarget, void, (RT64_DEVICE *device, void *devicePtr, size_t bytes)) \
    X(GetDeviceStats, RT64_GetDeviceStats, int, (RT64_DEVICE *device, RT64_FRAME_STATS *stats)) \
    /* Named numeric knobs ("count_traversal", "profile_passes", "sync_present", ...). Returns 0 when the key is unknown. */ \
    X(SetDeviceOption, RT64_SetDeviceOption, int, (RT64_DEVICE *device, const char *key, double value)) \
    /* Profiling aid (RT64_SetDeviceOption("tile_timing", 1)): two 16-byte records per wave of the last one-kernel frame, in workgroup order: at its \
       start { chip-wide 100 MHz clock, shader clock (low 32 bits each), HW_ID, 1 }, at its end { clock, shader clock, 0, 1 }; zeros for waves that did not run. */ \
    X(ReadbackTileTiming, RT64_ReadbackTileTiming, size_t, (RT64_DEVICE *device, void *dst, size_t dstBytes)) \
    /* hipStream_t the device submitted its LAST frame on, as void*.  The device has several render streams and enqueued pixel-local frames alternate over them \
       (option overlap_frames): a host that orders work of its own behind frames on ONE stream sets overlap_frames = 0 first. */ \
    X(GetDeviceStream, RT64_GetDeviceStream, void *, (RT64_DEVICE *device)) \
    /* Debug readback of acceleration structures (RT64_ACCEL_*): the mesh's BLAS / the view's TLAS of the last frame. \
       Returns bytes written (0 on error); pass dst = NULL to query the size. */ \
    X(ReadbackMeshAccel, RT64_ReadbackMeshAccel, size_t, (RT64_MESH *mesh, int what, void *dst, size_t dstBytes)) \
    /* Debug readback of a texture's texels as the kernels sample them: RGBA8 of mip level `mip`, rows top to bottom (a BC7 DDS: what the library's \
       decoder made of the blocks at RT64_CreateTexture).  Returns bytes written (0 on error); dst = NULL queries the size. */ \
    X(ReadbackTexture, RT64_ReadbackTexture, size_t, (RT64_TEXTURE *texture, int mip, void *dst, size_t dstBytes)) \
    X(ReadbackViewAccel, RT64_ReadbackViewAccel, size_t, (RT64_VIEW *view, int what, void *dst, size_t dstBytes)) \
    /* Depth of the BLAS RT64_SetMesh will build over these triangles (pure host function, no device; 0 = invalid arguments, 255 = a tree \
       of the multi-kernel builder): the library sizes its traversal stacks by it without waiting for the build. */ \
    X(MeshTreeDepth, RT64_MeshTreeDepth, unsigned int, (const void *vertexArray, int vertexCount, int vertexStride, const unsigned int *indexArray, int indexCount)) \
    /* ---- multi-GPU: one process per GPU, the frame's rows partitioned over `count` devices, ONE gather of the composited RGBA8 back buffer \
       to rank 0 per frame over RCCL / xGMI (the reference is single-GPU: NodeMask 0, rt64_device.cpp:753).  Rank 0 fills an id with \
       RT64_GetGatherUniqueId and passes it to the other ranks (file, pipe, MPI, ...); every rank then calls RT64_CreateGather on its own \
       device (all devices the same size): bands = 0 partitions into interleaved 16-row strips (pixel-local frames), bands = 1 into contiguous \
       bands (frames with GI + denoiser: the library renders the filter's halo around the band).  Per frame: RT64_DrawDevice, then \
       RT64_SubmitGather (returns the slot, 0 / 1, the frame travels in; the exchange runs beside the next frame's rendering). \
       RT64_ReadbackGather waits for a slot's exchange (slot < 0: the last submitted) and, on rank 0, copies the assembled frame out \
       (returns its size; 0 on other ranks); RT64_GetGatherFrame is its device pointer on rank 0.  RCCL is loaded on first use. */ \
    X(GetGatherUniqueId, RT64_GetGatherUniqueId, int, (void *id, size_t idBytes)) \
    X(CreateGather, RT64_CreateGather, RT64_GATHER *, (RT64_DEVICE *device, const void *id, size_t idBytes, int rank, int count, int bands)) \
    X(SubmitGather, RT64_SubmitGather, int, (RT64_GATHER *gather)) \
    X(ReadbackGather, RT64_ReadbackGather, size_t, (RT64_GATHER *gather, int slot, void *dst, size_t dstBytes, int toDevice)) \
    X(GetGatherFrame, RT64_GetGatherFrame, void *, (RT64_GATHER *gather, int slot)) \
    X(DestroyGather, RT64_DestroyGather, void, (RT64_GATHER *gather)) \
    /* DIRECT mode of a gather: no rows travel through RCCL.  Rank 0 owns six whole frame slots and exports them (RT64_GetGatherDirectHandle fills an IPC handle of \
       RT64_GATHER_DIRECT_HANDLE_BYTES; the host hands it to the other ranks like the unique id); after RT64_SetGatherDirect(gather, handle, bytes, 1) on every rank -- between \
       the same two frames -- each rank's frames store their rows straight into the slot of the frame in hand (on ranks other than 0: peer stores over xGMI into rank 0's \
       memory, inside the render kernels) and RT64_SubmitGather exchanges a 4-byte token per rank instead of the rows: same ordering, same flow control, no transfer stage \
       and no reassembly.  RT64_SubmitGather then returns slots 0 .. 5 in turn; on rank 0 the frames of the last three submits are never being overwritten.  enable = 0 \
       returns to the RCCL exchange; enable = 2 makes EVERY rank, 0 included, map `handle` (the slots live in another process: a presenter).  Returns 1, or 0 with \
       RT64_GetLastError set (no IPC, no peer access): the gather then stays in the mode it was in. */ \
    X(GetGatherDirectHandle, RT64_GetGatherDirectHandle, size_t, (RT64_GATHER *gather, void *handle, size_t handleBytes)) \
    X(SetGatherDirect, RT64_SetGatherDirect, int, (RT64_GATHER *gather, const void *handle, size_t handleBytes, int enable)) \
    /* the partition's layout as pure functions (usable without a device): owner rank of frame row y and the row's place in that rank's \
       packed buffer; rows a rank owns; rows every rank's slot is sized for */ \
    X(GatherRowOwner, RT64_GatherRowOwner, int, (int height, int count, int bands, int y, int *packedRow)) \
    X(GatherOwnedRows, RT64_GatherOwnedRows, int, (int height, int count, int bands, int rank)) \
    X(GatherSlotRows, RT64_GatherSlotRows, int, (int height, int count, int bands)) \
    /* (bands = 0 or 1 only; they return -1 for bands = 2, whose boundaries are not a function of height and count.)  The same three for a \
       layout given by its band boundaries starts[0 .. count] -- what RT64_GetGatherBands / RT64_BalanceGatherBands produce: */ \
    X(GatherRowOwnerOf, RT64_GatherRowOwnerOf, int, (int height, int count, const int *starts, int y, int *packedRow)) \
    X(GatherOwnedRowsOf, RT64_GatherOwnedRowsOf, int, (int height, int count, const int *starts, int rank)) \
    X(GatherSlotRowsOf, RT64_GatherSlotRowsOf, int, (int height, int count, const int *starts)) \
    /* bands = 2 in RT64_CreateGather: contiguous bands of about equal COST instead of equal height -- cost of a row = its pixels, those whose \
       primary ray hit geometry in the device's last (whole) frame counted 6 times; every rank must have rendered that frame, so that all \
       derive the same boundaries.  RT64_GetGatherBands reads the boundaries of a gather (starts[0..count]); RT64_BalanceGatherBands is the cut \
       itself as a pure function of per-row hit counts. */ \
    X(GetGatherBands, RT64_GetGatherBands, int, (RT64_GATHER *gather, int *starts, int capacity)) \
    X(BalanceGatherBands, RT64_BalanceGatherBands, void, (const unsigned int *hitCounts, int width, int height, int count, int *starts)) \
    /* Feedback on the measured cost: the modelled cut does not know which rows are expensive (a band over the sphere costs 1.5 x a band over the floor of the \
       sample scene).  Each rank measures the GPU time of its band, the host shares the figures (one all-gather of `count` floats), every rank computes the \
       same new boundaries with RT64_RebalanceGatherBands (pure function: equal cost above a fixed part (0.4 of the cheapest band) under an even spread inside each measured band, moves damped to 0.5 (two bands) ... 0.8 (eight), 16-row \
       minimum; returns 0 on invalid boundaries) and hands them to its gather with RT64_SetGatherBands -- a gather of bands = 1 or 2 (equal bands need no whole \
       frame for the first cut), between the same two frames on every rank; the communicator stays, send buffers grow when a band does.  Two or three rounds level the bands (tools/band_costs.py --rebalance). \
       Boundaries must rise strictly (every rank keeps at least one row).  Temporal state: a band holds GI history only for its rows and the halo it renders them with; the \
       accumulation of rows a band GAINS beyond that restarts at the next frame (history length 0), so for a few frames after a rebalance those rows are as noisy as in a \
       freshly cut partition and differ from the single-device frame until their history has grown back. */ \
    X(RebalanceGatherBands, RT64_RebalanceGatherBands, int, (int height, int count, const int *starts, const float *msPerRank, int *newStarts)) \
    X(SetGatherBands, RT64_SetGatherBands, int, (RT64_GATHER *gather, const int *starts)) \
    /* ---- halo EXCHANGE for band partitions of frames with GI + the SVGF denoiser.  The filter result of a row depends on the filter INPUT (noisy GI + \
       variance, guide record: 24 bytes per pixel) of RT64_HALO_ROWS rows above and below.  By default a band re-renders those rows (primary visibility, \
       G-buffer, GI bounce: the expensive passes); with the exchange a band makes the filter input of its own rows only and, in the middle of the frame, \
       ships its edge rows to the neighbouring bands and receives theirs.  RT64_HaloPlan is the schedule as a pure function: the regions rank `rank` \
       sends (send = 1: rows [y0, y1) of its own band, to `peer`) and receives (send = 0: rows of `peer`'s band it needs), for bands starts[0 .. count]; \
       returns the number of regions (writes at most `capacity`; `host` / `bytes` are left 0).  Two transports: \
         - RCCL (the device's gather, bands = 1 or 2): RT64_SetDeviceOption(device, "halo_exchange", 1) -- grouped ncclSend / ncclRecv between the \
           devices' images on the gather's communication stream, the render stream waits for it; \
         - the host's own (MPI, sockets, shared memory; the gloo rehearsal of tests/): RT64_SetDeviceHaloExchange(device, fn, user, starts, rank, count) \
           -- in the middle of RT64_DrawDevice the library stages the outgoing rows in pinned host memory, calls fn(user, regions, n) -- every region \
           with its `host` buffer of `bytes` = rows x width x 24: fn sends the send regions' bytes to their peers and fills the others from theirs -- \
           and uploads what arrived.  fn = NULL turns it off.  Returns 0 on invalid arguments. \
       Both leave the frame bit-identical to the single-device frame (tests/test_gpu_halo.py).  The temporal history of a band then covers its rows + \
       RT64_SetDeviceOption(device, "halo_margin", rows) (default 4, the minimum): a host whose camera moves more than that per frame raises it. */ \
    X(HaloPlan, RT64_HaloPlan, int, (int height, int count, const int *starts, int rank, int haloRows, RT64_HALO_REGION *regions, int capacity)) \
    X(SetDeviceHaloExchange, RT64_SetDeviceHaloExchange, int, (RT64_DEVICE *device, RT64_HALO_EXCHANGE exchange, void *user, const int *starts, int rank, int count))

typedef struct RT64_GATHER RT64_GATHER;
#define RT64_GATHER_ID_BYTES 128       /* size of the rendezvous id (an ncclUniqueId) */
#define RT64_GATHER_DIRECT_HANDLE_BYTES 64   /* sizeof(hipIpcMemHandle_t): RT64_GetGatherDirectHandle / RT64_SetGatherDirect */
#define RT64_HALO_ROWS 62              /* rows of filter input around a row that its SVGF result depends on: 2 x (1 + 2 + 4 + 8 + 16) */
#define RT64_HALO_BYTES_PER_PIXEL 24   /* filter input of a pixel: RGBA16F colour + variance (8) and the guide record (16) */
typedef struct { int peer; int send; int y0, y1; void *host; size_t bytes; } RT64_HALO_REGION;     /* rows [y0, y1) of the frame; host buffer = the rows' colour + variance, then their guide records */
typedef void (*RT64_HALO_EXCHANGE)(void *user, const RT64_HALO_REGION *regions, int count);

#define RT64_X(member, symbol, ret, args) typedef ret (*member##Ptr) args;
RT64_EXT_API_LIST(RT64_X)
#undef RT64_X

typedef struct {
#define RT64_X(member, symbol, ret, args) member##Ptr member;
    RT64_EXT_API_LIST(RT64_X)
#undef RT64_X
} RT64_LIBRARY_EXT;

RT64_INLINE RT64_LIBRARY_EXT RT64_LoadLibraryExt(RT64_LIBRARY lib) {
    RT64_LIBRARY_EXT ext;
#define RT64_X(member, symbol, ret, args) ext.member = lib.handle ? (member##Ptr)(RT64_DLSYM(lib.handle, #symbol)) : 0;
    RT64_EXT_API_LIST(RT64_X)
#undef RT64_X
    return ext;
}

/* ---- layout checks (x86-64 SysV == MSVC x64 for these PODs; probed values from SURVEY.md 8b) -- */
#if defined(__cplusplus)
#   define RT64_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#   define RT64_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
RT64_STATIC_ASSERT(sizeof(RT64_MATRIX4) == 64, "RT64_MATRIX4");
RT64_STATIC_ASSERT(sizeof(RT64_RECT) == 16, "RT64_RECT");
RT64_STATIC_ASSERT(sizeof(RT64_MATERIAL) == 132, "RT64_MATERIAL");
RT64_STATIC_ASSERT(offsetof(RT64_MATERIAL, specularColor) == 36, "RT64_MATERIAL.specularColor");
RT64_STATIC_ASSERT(offsetof(RT64_MATERIAL, selfLight) == 68, "RT64_MATERIAL.selfLight");
RT64_STATIC_ASSERT(offsetof(RT64_MATERIAL, fogColor) == 84, "RT64_MATERIAL.fogColor");
RT64_STATIC_ASSERT(offsetof(RT64_MATERIAL, diffuseColorMix) == 96, "RT64_MATERIAL.diffuseColorMix");
RT64_STATIC_ASSERT(offsetof(RT64_MATERIAL, lockMask) == 124, "RT64_MATERIAL.lockMask");
RT64_STATIC_ASSERT(offsetof(RT64_MATERIAL, enabledAttributes) == 128, "RT64_MATERIAL.enabledAttributes");
RT64_STATIC_ASSERT(sizeof(RT64_LIGHT) == 60, "RT64_LIGHT");
RT64_STATIC_ASSERT(sizeof(RT64_SCENE_DESC) == 84, "RT64_SCENE_DESC");
RT64_STATIC_ASSERT(sizeof(RT64_VIEW_DESC) == 32, "RT64_VIEW_DESC");
RT64_STATIC_ASSERT(offsetof(RT64_VIEW_DESC, upscaler) == 20, "RT64_VIEW_DESC.upscaler");
RT64_STATIC_ASSERT(offsetof(RT64_VIEW_DESC, upscalerMode) == 21, "RT64_VIEW_DESC.upscalerMode");
RT64_STATIC_ASSERT(offsetof(RT64_VIEW_DESC, upscalerSharpness) == 24, "RT64_VIEW_DESC.upscalerSharpness");
RT64_STATIC_ASSERT(offsetof(RT64_VIEW_DESC, denoiserEnabled) == 28, "RT64_VIEW_DESC.denoiserEnabled");
RT64_STATIC_ASSERT(sizeof(RT64_INSTANCE_DESC) == 336, "RT64_INSTANCE_DESC");
RT64_STATIC_ASSERT(offsetof(RT64_INSTANCE_DESC, transform) == 8, "RT64_INSTANCE_DESC.transform");
RT64_STATIC_ASSERT(offsetof(RT64_INSTANCE_DESC, material) == 168, "RT64_INSTANCE_DESC.material");
RT64_STATIC_ASSERT(offsetof(RT64_INSTANCE_DESC, scissorRect) == 300, "RT64_INSTANCE_DESC.scissorRect");
RT64_STATIC_ASSERT(offsetof(RT64_INSTANCE_DESC, flags) == 332, "RT64_INSTANCE_DESC.flags");
RT64_STATIC_ASSERT(sizeof(RT64_TEXTURE_DESC) == 32, "RT64_TEXTURE_DESC");

#endif /* RT64_H_INCLUDED */
