/*
 * sample_host.c -- a plain C host of librt64.so: the reference's sample application without its Win32 window.
 *
 * It does what /root/reference/src/sample/main.cpp does with rt64lib.dll: RT64_LoadLibrary() fills the function-pointer
 * table (main.cpp:419-432 -> include/rt64.h), setupRT64Scene() (main.cpp:201-412) creates the scene description, the
 * shader 0x01200a00, the light, the view, seven textures, the sky plane, the sphere mesh from sphere.obj (unrolled,
 * uv = acos(n.xy)), the two raster-only HUD triangles and the floor quad scaled x10, and every frame repeats the calls of
 * the WM_PAINT handler (main.cpp:97-134): SetViewPerspective, SetInstanceDescription of the sphere, SetSceneLights,
 * DrawDevice.  Where the reference presents to a swap chain this host reads the back buffer back
 * (RT64_ReadbackDevice, an additive export) and prints a checksum; tests/test_gpu_c_host.py compares it with the frame
 * the Python/ctypes harness renders.
 *
 * `--ranks N` runs the multi-GPU path from C as well: N processes (one per GPU, forked BEFORE anything touches a GPU), rank r on HIP
 * device r, the rendezvous id of RT64_GetGatherUniqueId handed from rank 0 to the others through a pipe; every rank creates the same
 * scene, RT64_CreateGather partitions the frame's rows, and each frame is RT64_DrawDevice + RT64_SubmitGather; rank 0 prints the
 * checksum of the gathered frame (`--ranks 1` runs the same calls with a world of one).
 *
 * Only include/rt64.h is needed to build it -- the library is bound at run time with dlopen/dlsym like the game does:
 *     gcc -O2 -Iinclude tools/sample_host.c -o tools/sample_host -ldl -lz -lm
 *     RT64_LIBRARY_PATH=sm64rt-legacy-renderer_amd/librt64.so tools/sample_host --width 640 --height 360 --frames 3
 * PNG files are decoded by a small zlib-based reader below (the reference uses stb_image: 8-bit RGBA output, 16-bit
 * samples keep their high byte); grass_dif.dds is handed over as a file image (RT64_TEXTURE_FORMAT_DDS).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include <zlib.h>

#include "rt64.h"

typedef struct { RT64_VECTOR4 position; RT64_VECTOR3 normal; RT64_VECTOR2 uv; RT64_VECTOR4 input1; } VERTEX;   /* main.cpp:36-41, 52 bytes */

static struct {
    RT64_LIBRARY lib; RT64_LIBRARY_EXT ext;
    RT64_DEVICE *device; RT64_SCENE *scene; RT64_VIEW *view; RT64_SHADER *shader;
    RT64_TEXTURE *tex[7];
    RT64_MESH *sphereMesh, *hudMesh, *hudAltMesh, *floorMesh;
    RT64_INSTANCE *hudAltInstance, *sphereInstance, *hudInstance, *floorInstance;
    RT64_MATRIX4 viewMatrix, sphereTransform;
    RT64_MATERIAL baseMaterial;
    RT64_LIGHT lights[1]; int lightCount;
    RT64_INSTANCE_DESC sphereDesc;
} RT64;

static void die(const char *what) { fprintf(stderr, "sample_host: %s\n", what); exit(2); }

static unsigned char *read_file(const char *dir, const char *name, size_t *size) {
    char path[1024]; snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "sample_host: cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *p = (unsigned char *)malloc((size_t)n + 1);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) die("short read");
    fclose(f); p[n] = 0; *size = (size_t)n;
    return p;
}

/* ---- PNG (non-interlaced, 8-bit RGB / RGBA / grey, 16-bit grey) -> RGBA8 ------------------------------------------ */
static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static int paeth(int a, int b, int c) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

static unsigned char *load_png_rgba8(const char *dir, const char *name, int *w, int *h) {
    size_t size; unsigned char *file = read_file(dir, name, &size);
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (size < 33 || memcmp(file, sig, 8) != 0) die("not a PNG file");
    unsigned char *idat = (unsigned char *)malloc(size); size_t idatLen = 0;
    int width = 0, height = 0, depth = 0, ctype = 0, interlace = 0;
    for (size_t off = 8; off + 12 <= size;) {
        uint32_t len = be32(file + off); const unsigned char *type = file + off + 4, *body = file + off + 8;
        if (off + 12 + len > size) die("truncated PNG chunk");
        if (!memcmp(type, "IHDR", 4)) { width = (int)be32(body); height = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!memcmp(type, "IDAT", 4)) { memcpy(idat + idatLen, body, len); idatLen += len; }
        else if (!memcmp(type, "IEND", 4)) break;
        off += 12 + len;
    }
    int channels = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 6 ? 4 : (ctype == 4 ? 2 : 0)));
    if (!channels || interlace || (depth != 8 && depth != 16)) die("unsupported PNG layout");
    const int bpp = channels * depth / 8; const size_t stride = (size_t)width * bpp;
    uLongf rawLen = (uLongf)((stride + 1) * height);
    unsigned char *raw = (unsigned char *)malloc(rawLen);
    if (uncompress(raw, &rawLen, idat, (uLong)idatLen) != Z_OK || rawLen != (stride + 1) * (size_t)height) die("PNG inflate failed");
    unsigned char *out = (unsigned char *)malloc((size_t)width * height * 4), *prev = (unsigned char *)calloc(stride, 1);
    for (int y = 0; y < height; y++) {
        unsigned char *line = raw + (size_t)y * (stride + 1) + 1; const int filter = line[-1];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)bpp ? line[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v = line[i];
            switch (filter) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break; default: break; }
            line[i] = (unsigned char)v;
        }
        memcpy(prev, line, stride);
        for (int x = 0; x < width; x++) {
            const unsigned char *s = line + (size_t)x * bpp; unsigned char *d = out + ((size_t)y * width + x) * 4;
            const int step = depth / 8;                       /* 16-bit samples are big-endian: the first byte is the high one */
            if (channels >= 3) { d[0] = s[0]; d[1] = s[step]; d[2] = s[2 * step]; d[3] = channels == 4 ? s[3 * step] : 255; }
            else { d[0] = d[1] = d[2] = s[0]; d[3] = channels == 2 ? s[step] : 255; }
        }
    }
    free(prev); free(raw); free(idat); free(file);
    *w = width; *h = height;
    return out;
}

static RT64_TEXTURE *loadTexturePNG(const char *dir, const char *name) {         /* main.cpp:156-171 */
    int w, h; unsigned char *rgba = load_png_rgba8(dir, name, &w, &h);
    RT64_TEXTURE_DESC d; d.bytes = rgba; d.byteCount = w * h * 4; d.format = RT64_TEXTURE_FORMAT_RGBA8; d.width = w; d.height = h; d.rowPitch = w * 4;
    RT64_TEXTURE *t = RT64.lib.CreateTexture(RT64.device, d);
    free(rgba);                                                                  /* the library copied it during the call */
    if (!t) { fprintf(stderr, "sample_host: CreateTexture(%s): %s\n", name, RT64.lib.GetLastError()); exit(2); }
    return t;
}
static RT64_TEXTURE *loadTextureDDS(const char *dir, const char *name) {         /* main.cpp:173-193 */
    size_t size; unsigned char *file = read_file(dir, name, &size);
    RT64_TEXTURE_DESC d; d.bytes = file; d.byteCount = (int)size; d.format = RT64_TEXTURE_FORMAT_DDS; d.width = d.height = d.rowPitch = -1;
    RT64_TEXTURE *t = RT64.lib.CreateTexture(RT64.device, d);
    free(file);
    if (!t) { fprintf(stderr, "sample_host: CreateTexture(%s): %s\n", name, RT64.lib.GetLastError()); exit(2); }
    return t;
}

static RT64_MATRIX4 identity(void) { RT64_MATRIX4 m; memset(&m, 0, sizeof(m)); m.m[0][0] = m.m[1][1] = m.m[2][2] = m.m[3][3] = 1.0f; return m; }

/* tinyobj::LoadObj(triangulate) + the unrolling loop of main.cpp:262-287: positions / normals / faces "v/vt/vn". */
static VERTEX *load_obj_unrolled(const char *dir, const char *name, int *count) {
    size_t size; char *text = (char *)read_file(dir, name, &size);
    size_t np = 0, nn = 0, nf = 0, cp = 1024, cn = 1024, cf = 1024;
    float *pos = (float *)malloc(cp * 12), *nrm = (float *)malloc(cn * 12); int *faces = (int *)malloc(cf * 6 * sizeof(int));
    for (char *line = strtok(text, "\n"); line; line = strtok(NULL, "\n")) {
        if (line[0] == 'v' && (line[1] == ' ' || line[1] == 'n')) {
            const int isN = line[1] == 'n'; char *p = line + 2; float v[3];
            for (int k = 0; k < 3; k++) v[k] = (float)strtod(p, &p);          /* double, then one rounding to float (as the Python harness parses) */
            if (isN) { if (nn == cn) { cn *= 2; nrm = (float *)realloc(nrm, cn * 12); } memcpy(nrm + 3 * nn++, v, 12); }
            else { if (np == cp) { cp *= 2; pos = (float *)realloc(pos, cp * 12); } memcpy(pos + 3 * np++, v, 12); }
        }
        else if (line[0] == 'f' && line[1] == ' ') {
            int corner[16][2], k = 0; char *p = line + 2;
            while (*p && k < 16) {
                while (*p == ' ') p++;
                if (!*p || *p == '\r') break;
                int vi = (int)strtol(p, &p, 10), ti = 0, ni = 0;
                if (*p == '/') { p++; if (*p != '/') ti = (int)strtol(p, &p, 10); if (*p == '/') { p++; ni = (int)strtol(p, &p, 10); } }
                (void)ti; corner[k][0] = vi; corner[k][1] = ni; k++;
            }
            for (int t = 1; t + 1 < k; t++) {                                  /* fan triangulation */
                if (nf == cf) { cf *= 2; faces = (int *)realloc(faces, cf * 6 * sizeof(int)); }
                int *f = faces + 6 * nf++;
                f[0] = corner[0][0]; f[1] = corner[0][1]; f[2] = corner[t][0]; f[3] = corner[t][1]; f[4] = corner[t + 1][0]; f[5] = corner[t + 1][1];
            }
        }
    }
    VERTEX *verts = (VERTEX *)calloc(nf * 3, sizeof(VERTEX));
    for (size_t f = 0; f < nf; f++)
        for (int c = 0; c < 3; c++) {
            const float *p = pos + 3 * (faces[6 * f + 2 * c] - 1), *n = nrm + 3 * (faces[6 * f + 2 * c + 1] - 1);
            VERTEX *v = verts + 3 * f + c;
            v->position.x = p[0]; v->position.y = p[1]; v->position.z = p[2]; v->position.w = 1.0f;
            v->normal.x = n[0]; v->normal.y = n[1]; v->normal.z = n[2];
            v->uv.x = (float)acos((double)n[0]); v->uv.y = (float)acos((double)n[1]);   /* main.cpp:278; double, rounded once: see sample_scene.py */
            v->input1.x = v->input1.y = v->input1.z = v->input1.w = 1.0f;
        }
    free(pos); free(nrm); free(faces); free(text);
    *count = (int)(nf * 3);
    return verts;
}

static void setupRT64Scene(const char *assets) {                                /* main.cpp:201-412 */
    RT64.scene = RT64.lib.CreateScene(RT64.device);
    RT64_SCENE_DESC sceneDesc; memset(&sceneDesc, 0, sizeof(sceneDesc));
    sceneDesc.ambientBaseColor.x = sceneDesc.ambientBaseColor.y = sceneDesc.ambientBaseColor.z = 0.1f;
    sceneDesc.ambientNoGIColor.x = sceneDesc.ambientNoGIColor.y = sceneDesc.ambientNoGIColor.z = 0.2f;
    sceneDesc.eyeLightDiffuseColor.x = sceneDesc.eyeLightDiffuseColor.y = sceneDesc.eyeLightDiffuseColor.z = 0.08f;
    sceneDesc.eyeLightSpecularColor.x = sceneDesc.eyeLightSpecularColor.y = sceneDesc.eyeLightSpecularColor.z = 0.04f;
    sceneDesc.skyDiffuseMultiplier.x = sceneDesc.skyDiffuseMultiplier.y = sceneDesc.skyDiffuseMultiplier.z = 1.0f;
    sceneDesc.skyYawOffset = 0.0f; sceneDesc.giDiffuseStrength = 0.7f; sceneDesc.giSkyStrength = 0.35f;
    RT64.lib.SetSceneDescription(RT64.scene, sceneDesc);

    const int shaderFlags = RT64_SHADER_RASTER_ENABLED | RT64_SHADER_RAYTRACE_ENABLED | RT64_SHADER_NORMAL_MAP_ENABLED | RT64_SHADER_SPECULAR_MAP_ENABLED;
    RT64.shader = RT64.lib.CreateShader(RT64.device, 0x01200a00, RT64_SHADER_FILTER_LINEAR, RT64_SHADER_ADDRESSING_WRAP, RT64_SHADER_ADDRESSING_WRAP, shaderFlags);
    if (!RT64.shader) die(RT64.lib.GetLastError());

    memset(RT64.lights, 0, sizeof(RT64.lights));                                 /* static storage in the reference: unset members are 0 */
    RT64.lights[0].position.x = 15000.0f; RT64.lights[0].position.y = 30000.0f; RT64.lights[0].position.z = 15000.0f;
    RT64.lights[0].attenuationRadius = 1e9f; RT64.lights[0].pointRadius = 5000.0f;
    RT64.lights[0].diffuseColor.x = 0.8f; RT64.lights[0].diffuseColor.y = 0.75f; RT64.lights[0].diffuseColor.z = 0.65f;
    RT64.lights[0].specularColor = RT64.lights[0].diffuseColor;
    RT64.lights[0].shadowOffset = 0.0f; RT64.lights[0].attenuationExponent = 1.0f; RT64.lights[0].groupBits = RT64_LIGHT_GROUP_DEFAULT;
    RT64.lightCount = 1;

    RT64.view = RT64.lib.CreateView(RT64.scene);

    /* creation order of the harness (sample_scene.py): sphere textures, sky, floor textures */
    RT64.tex[0] = loadTextureDDS(assets, "grass_dif.dds"); RT64.tex[1] = loadTexturePNG(assets, "grass_nrm.png"); RT64.tex[2] = loadTexturePNG(assets, "grass_spc.png");
    RT64.tex[3] = loadTexturePNG(assets, "clouds.png");
    RT64.tex[4] = loadTexturePNG(assets, "tiles_dif.png"); RT64.tex[5] = loadTexturePNG(assets, "tiles_nrm.png"); RT64.tex[6] = loadTexturePNG(assets, "tiles_spc.png");
    RT64.lib.SetViewSkyPlane(RT64.view, RT64.tex[3]);

    RT64.viewMatrix = identity();                                                 /* main.cpp:250-258: camera at (0, 2, 10) looking down -z */
    RT64.viewMatrix.m[3][1] = -2.0f; RT64.viewMatrix.m[3][2] = -10.0f;

    int sphereCount = 0; VERTEX *sphere = load_obj_unrolled(assets, "sphere.obj", &sphereCount);
    unsigned int *sphereIdx = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)sphereCount);
    for (int i = 0; i < sphereCount; i++) sphereIdx[i] = (unsigned int)i;
    RT64.sphereMesh = RT64.lib.CreateMesh(RT64.device, RT64_MESH_RAYTRACE_ENABLED | RT64_MESH_RAYTRACE_FAST_TRACE | RT64_MESH_RAYTRACE_COMPACT);
    RT64.lib.SetMesh(RT64.sphereMesh, sphere, sphereCount, (int)sizeof(VERTEX), sphereIdx, sphereCount);
    free(sphere); free(sphereIdx);                                                /* copied during the call (main.cpp:169,191 free right away too) */

    memset(&RT64.baseMaterial, 0, sizeof(RT64.baseMaterial));                     /* main.cpp:292-310 */
    RT64.baseMaterial.uvDetailScale = 1.0f; RT64.baseMaterial.reflectionFresnelFactor = 1.0f;
    RT64.baseMaterial.specularColor.x = RT64.baseMaterial.specularColor.y = RT64.baseMaterial.specularColor.z = 1.0f;
    RT64.baseMaterial.specularExponent = 1.0f; RT64.baseMaterial.solidAlphaMultiplier = 1.0f; RT64.baseMaterial.shadowAlphaMultiplier = 1.0f;
    RT64.baseMaterial.lightGroupMaskBits = RT64_LIGHT_GROUP_MASK_ALL;
    RT64.baseMaterial.fogColor.x = 0.3f; RT64.baseMaterial.fogColor.y = 0.5f; RT64.baseMaterial.fogColor.z = 0.7f; RT64.baseMaterial.fogMul = 1.0f;

    VERTEX hud[3]; memset(hud, 0, sizeof(hud));                                   /* main.cpp:312-338 */
    const float hx[3] = { -1.0f, -0.5f, -0.75f }, hy[3] = { 0.1f, 0.1f, 0.3f }, hu[3] = { 0.0f, 1.0f, 0.0f }, hv[3] = { 0.0f, 0.0f, 1.0f };
    for (int k = 0; k < 3; k++) {
        hud[k].position.x = hx[k]; hud[k].position.y = hy[k]; hud[k].position.w = 1.0f; hud[k].normal.y = 1.0f; hud[k].uv.x = hu[k]; hud[k].uv.y = hv[k];
        hud[k].input1.x = hud[k].input1.y = hud[k].input1.z = hud[k].input1.w = 1.0f;
    }
    unsigned int hudIdx[3] = { 0, 1, 2 };
    RT64.hudMesh = RT64.lib.CreateMesh(RT64.device, 0);
    RT64.lib.SetMesh(RT64.hudMesh, hud, 3, (int)sizeof(VERTEX), hudIdx, 3);
    for (int k = 0; k < 3; k++) hud[k].position.y += 0.15f;
    RT64.hudAltMesh = RT64.lib.CreateMesh(RT64.device, 0);
    RT64.lib.SetMesh(RT64.hudAltMesh, hud, 3, (int)sizeof(VERTEX), hudIdx, 3);

    VERTEX floorV[4]; memset(floorV, 0, sizeof(floorV));                          /* main.cpp:377-392 */
    const float fx[4] = { -1.5f, 1.0f, -1.5f, 1.0f }, fz[4] = { -1.0f, -1.0f, 1.0f, 1.0f }, fu[4] = { 0.0f, 1.0f, 0.0f, 1.0f }, fv[4] = { 0.0f, 0.0f, 1.0f, 1.0f };
    for (int k = 0; k < 4; k++) {
        floorV[k].position.x = fx[k]; floorV[k].position.z = fz[k]; floorV[k].position.w = 1.0f; floorV[k].normal.y = 1.0f; floorV[k].uv.x = fu[k]; floorV[k].uv.y = fv[k];
        floorV[k].input1.x = floorV[k].input1.y = floorV[k].input1.z = floorV[k].input1.w = 1.0f;
    }
    unsigned int floorIdx[6] = { 2, 1, 0, 1, 2, 3 };
    RT64.floorMesh = RT64.lib.CreateMesh(RT64.device, RT64_MESH_RAYTRACE_ENABLED);
    RT64.lib.SetMesh(RT64.floorMesh, floorV, 4, (int)sizeof(VERTEX), floorIdx, 6);

    /* instances in the creation order of main.cpp:356-411 */
    RT64_INSTANCE_DESC d; memset(&d, 0, sizeof(d));
    d.transform = identity(); d.previousTransform = identity(); d.shader = RT64.shader; d.material = RT64.baseMaterial;
    RT64.hudAltInstance = RT64.lib.CreateInstance(RT64.scene);
    d.mesh = RT64.hudAltMesh; d.diffuseTexture = RT64.tex[4]; d.normalTexture = NULL; d.specularTexture = NULL; d.flags = 0;
    RT64.lib.SetInstanceDescription(RT64.hudAltInstance, d);

    RT64.sphereInstance = RT64.lib.CreateInstance(RT64.scene);
    RT64.sphereTransform = identity();
    d.mesh = RT64.sphereMesh; d.diffuseTexture = RT64.tex[0]; d.normalTexture = RT64.tex[1]; d.specularTexture = RT64.tex[2];
    RT64.sphereDesc = d;
    RT64.lib.SetInstanceDescription(RT64.sphereInstance, d);

    RT64.hudInstance = RT64.lib.CreateInstance(RT64.scene);
    d.mesh = RT64.hudMesh; d.diffuseTexture = RT64.tex[0]; d.normalTexture = NULL; d.specularTexture = NULL; d.flags = RT64_INSTANCE_RASTER_BACKGROUND;
    RT64.lib.SetInstanceDescription(RT64.hudInstance, d);

    RT64.floorInstance = RT64.lib.CreateInstance(RT64.scene);
    d.transform = identity(); d.transform.m[0][0] = d.transform.m[1][1] = d.transform.m[2][2] = 10.0f; d.previousTransform = d.transform;
    d.mesh = RT64.floorMesh; d.diffuseTexture = RT64.tex[4]; d.normalTexture = RT64.tex[5]; d.specularTexture = RT64.tex[6]; d.flags = 0;
    RT64.lib.SetInstanceDescription(RT64.floorInstance, d);
}

static void drawFrame(void) {                                                    /* WM_PAINT, main.cpp:97-134 */
    RT64.lib.SetViewPerspective(RT64.view, RT64.viewMatrix, (45.0f * 3.14159265358979323846f) / 180.0f, 0.1f, 1000.0f, true);
    RT64.sphereDesc.transform = RT64.sphereTransform; RT64.sphereDesc.previousTransform = RT64.sphereTransform;
    RT64.lib.SetInstanceDescription(RT64.sphereInstance, RT64.sphereDesc);
    RT64.lib.SetSceneLights(RT64.scene, RT64.lights, RT64.lightCount);
    RT64.lib.DrawDevice(RT64.device, 1, 1000.0f / 60.0f);
}

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

int main(int argc, char **argv) {
    int width = 1280, height = 720, frames = 3, warmup = 0;                      /* main.cpp:435-436: 1280 x 720 window */
    const char *assets = "assets/sample", *dump = NULL;
    int selftest = 0, ranks = 0, bands = 0, direct = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--width") && i + 1 < argc) width = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--height") && i + 1 < argc) height = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);             /* untimed frames before the `frames` timed ones */
        else if (!strcmp(argv[i], "--assets") && i + 1 < argc) assets = argv[++i];
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "--selftest")) selftest = 1;
        else if (!strcmp(argv[i], "--ranks") && i + 1 < argc) ranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--bands")) bands = 1;
        else if (!strcmp(argv[i], "--direct")) direct = 1;                       /* direct gather: rows stored into rank 0's frame slots through an IPC mapping (RT64_SetGatherDirect) */
        else { fprintf(stderr, "usage: %s [--width W] [--height H] [--frames N] [--warmup W] [--assets DIR] [--dump frame.rgba] [--selftest] [--ranks N [--bands] [--direct]]\n", argv[0]); return 2; }
    }
    if (selftest) {       /* the asset readers alone (no library, no GPU): byte sums the CPU test suite compares with the Python harness's loaders */
        static const char *pngs[6] = { "grass_nrm.png", "grass_spc.png", "clouds.png", "tiles_dif.png", "tiles_nrm.png", "tiles_spc.png" };
        printf("{");
        for (int k = 0; k < 6; k++) {
            int w, h; unsigned char *rgba = load_png_rgba8(assets, pngs[k], &w, &h);
            unsigned long long sum = 0, th = 1469598103934665603ull;      /* weighted byte sum + FNV-1a 64 of the texels (the latter is what oracle/ref_inputs_dump.cpp prints for stb_image's bytes) */
            for (size_t i = 0; i < (size_t)w * h * 4; i++) { sum += (unsigned long long)rgba[i] * (1 + (i & 3)); th = (th ^ rgba[i]) * 1099511628211ull; }
            printf("\"%s\": [%d, %d, %llu, \"%016llx\"], ", pngs[k], w, h, sum, th); free(rgba);
        }
        int n = 0; VERTEX *v = load_obj_unrolled(assets, "sphere.obj", &n);
        unsigned long long fnv = 1469598103934665603ull, pn = 1469598103934665603ull; const unsigned char *b = (const unsigned char *)v;
        for (size_t i = 0; i < (size_t)n * sizeof(VERTEX); i++) fnv = (fnv ^ b[i]) * 1099511628211ull;
        for (int i = 0; i < n; i++) {          /* position.xyz + normal.xyz per vertex: the part tiny_obj_loader decides (oracle/ref_inputs_dump.cpp: sphere.posnrm) */
            const unsigned char *q = (const unsigned char *)&v[i].position; for (int k = 0; k < 12; k++) pn = (pn ^ q[k]) * 1099511628211ull;
            q = (const unsigned char *)&v[i].normal; for (int k = 0; k < 12; k++) pn = (pn ^ q[k]) * 1099511628211ull;
        }
        printf("\"sphere.obj\": [%d, \"%016llx\", \"%016llx\"]}\n", n, fnv, pn); free(v);
        return 0;
    }
    /* multi-GPU: fork the other ranks now, before the library is loaded or any GPU call is made; ids travel over one pipe per rank */
    int rank = 0, idPipe[64][2];
    if (ranks > 64) ranks = 64;
    if (ranks > 1) {
        for (int r = 1; r < ranks; r++) if (pipe(idPipe[r]) != 0) die("pipe");
        for (int r = 1; r < ranks; r++) {
            pid_t pid = fork();
            if (pid < 0) die("fork");
            if (pid == 0) { rank = r; break; }
        }
    }
    RT64.lib = RT64_LoadLibrary();                                               /* main.cpp:419-432 */
    if (RT64.lib.handle == 0) die("failed to load the library (set RT64_LIBRARY_PATH)");
    RT64.ext = RT64_LoadLibraryExt(RT64.lib);
    if (!RT64.ext.CreateDeviceHeadless || !RT64.ext.ReadbackDevice) die("librt64.so lacks the headless extensions");
    RT64.device = RT64.ext.CreateDeviceHeadless(width, height, ranks > 0 ? rank : -1);   /* stands in for CreateDevice(hwnd) of a width x height window */
    if (!RT64.device) { fprintf(stderr, "sample_host: CreateDevice: %s\n", RT64.lib.GetLastError()); return 3; }
    setupRT64Scene(assets);
    const size_t bytes = (size_t)width * height * 4;
    unsigned char *frame = (unsigned char *)malloc(bytes);
    double msPerFrame = 0.0;
    if (ranks > 0) {                                                             /* the frame's rows over `ranks` GPUs, gathered on rank 0 */
        unsigned char id[RT64_GATHER_ID_BYTES];
        if (!RT64.ext.CreateGather || !RT64.ext.SubmitGather || !RT64.ext.ReadbackGather) die("librt64.so lacks the gather exports");
        if (rank == 0) {
            if (!RT64.ext.GetGatherUniqueId(id, sizeof(id))) die(RT64.lib.GetLastError());
            for (int r = 1; r < ranks; r++) if (write(idPipe[r][1], id, sizeof(id)) != (ssize_t)sizeof(id)) die("id pipe");
        }
        else if (read(idPipe[rank][0], id, sizeof(id)) != (ssize_t)sizeof(id)) die("id pipe");
        RT64_GATHER *gather = RT64.ext.CreateGather(RT64.device, id, sizeof(id), rank, ranks, bands);
        if (!gather) { fprintf(stderr, "sample_host: CreateGather: %s\n", RT64.lib.GetLastError()); return 5; }
        if (direct) {                                                            /* rank 0 exports its frame slots, the handle travels like the id, every rank switches over */
            unsigned char handle[RT64_GATHER_DIRECT_HANDLE_BYTES];
            if (!RT64.ext.GetGatherDirectHandle || !RT64.ext.SetGatherDirect) die("librt64.so lacks the direct-gather exports");
            if (rank == 0) {
                if (RT64.ext.GetGatherDirectHandle(gather, handle, sizeof(handle)) != sizeof(handle)) die(RT64.lib.GetLastError());
                for (int r = 1; r < ranks; r++) if (write(idPipe[r][1], handle, sizeof(handle)) != (ssize_t)sizeof(handle)) die("handle pipe");
            }
            else if (read(idPipe[rank][0], handle, sizeof(handle)) != (ssize_t)sizeof(handle)) die("handle pipe");
            if (!RT64.ext.SetGatherDirect(gather, handle, sizeof(handle), 1)) die(RT64.lib.GetLastError());
        }
        if (RT64.ext.SetDeviceOption) RT64.ext.SetDeviceOption(RT64.device, "sync_present", 0);        /* frames are enqueued: the exchange of frame k runs beside the rendering of frame k + 1 (and pixel-local frames beside each other) */
        int slot = -1;
        for (int f = 0; f < warmup; f++) { drawFrame(); slot = RT64.ext.SubmitGather(gather); if (slot < 0) die(RT64.lib.GetLastError()); }
        if (warmup) RT64.ext.ReadbackGather(gather, slot, frame, bytes, 0);
        const double t0 = now_ms();
        for (int f = 0; f < frames; f++) { drawFrame(); slot = RT64.ext.SubmitGather(gather); if (slot < 0) die(RT64.lib.GetLastError()); }
        const size_t got = RT64.ext.ReadbackGather(gather, slot, frame, bytes, 0);          /* every rank waits for its part; rank 0 gets the frame */
        msPerFrame = (now_ms() - t0) / (frames > 0 ? frames : 1);
        RT64.ext.DestroyGather(gather);
        if (rank != 0) { RT64.lib.DestroyDevice(RT64.device); return 0; }
        if (got != bytes) { fprintf(stderr, "sample_host: ReadbackGather: %s\n", RT64.lib.GetLastError()); return 4; }
        int failed = 0;
        for (int r = 1; r < ranks; r++) { int st = 0; if (wait(&st) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) failed = 1; }
        if (failed) die("a rank failed");
    }
    else {
        for (int f = 0; f < warmup; f++) drawFrame();
        const double t0 = now_ms();
        for (int f = 0; f < frames; f++) drawFrame();                          /* RT64_DrawDevice returns when the frame is in the back buffer */
        msPerFrame = (now_ms() - t0) / (frames > 0 ? frames : 1);
        if (RT64.ext.ReadbackDevice(RT64.device, RT64_IMAGE_FINAL_RGBA8, frame, bytes) != bytes) { fprintf(stderr, "sample_host: readback: %s\n", RT64.lib.GetLastError()); return 4; }
    }
    unsigned long long sum = 0, fnv = 1469598103934665603ull;
    for (size_t i = 0; i < bytes; i++) { sum += frame[i]; fnv = (fnv ^ frame[i]) * 1099511628211ull; }
    RT64_INSTANCE *picked = ranks > 1 ? NULL : RT64.lib.GetViewRaytracedInstanceAt(RT64.view, width / 2, height / 2);          /* right click of main.cpp:76-83 */
    const char *pickedName = picked == RT64.sphereInstance ? "sphere" : (picked == RT64.floorInstance ? "floor" : (picked ? "other" : "none"));
    RT64_FRAME_STATS st; memset(&st, 0, sizeof(st)); st.structSize = (unsigned int)sizeof(st);
    float gpuMs = 0.0f;
    if (RT64.ext.GetDeviceStats && RT64.ext.GetDeviceStats(RT64.device, &st)) gpuMs = st.msTotal;
    printf("{\"host\": \"C (tools/sample_host.c)\", \"width\": %d, \"height\": %d, \"frames\": %d, \"ranks\": %d, \"checksum\": %llu, \"fnv1a\": \"%016llx\", \"picked_center\": \"%s\", \"gpu_ms_last_frame\": %.4f, \"ms_per_frame\": %.5f}\n",
           width, height, frames, ranks, sum, fnv, pickedName, gpuMs, msPerFrame);
    if (dump) { FILE *f = fopen(dump, "wb"); if (f) { fwrite(frame, 1, bytes, f); fclose(f); } }
    free(frame);

    RT64.lib.DestroyMesh(RT64.sphereMesh); RT64.lib.DestroyMesh(RT64.hudMesh); RT64.lib.DestroyMesh(RT64.hudAltMesh); RT64.lib.DestroyMesh(RT64.floorMesh);
    for (int k = 0; k < 7; k++) RT64.lib.DestroyTexture(RT64.tex[k]);
    RT64.lib.DestroyShader(RT64.shader);
    RT64.lib.DestroyDevice(RT64.device);                                          /* deletes scenes -> views + instances (rt64_device.cpp:97-100) */
    RT64_UnloadLibrary(RT64.lib);
    return 0;
}
