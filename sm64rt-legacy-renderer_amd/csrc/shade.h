// shade.h -- device-side shading library: software texture sampler, colour-combiner any-hit programs, lights, sky.
//
// MI355X-side implementation of the reference's HLSL (file:line relative to /root/reference/src/rt64lib):
//   runtime-generated any-hit programs  private/rt64_shader.cpp:156-226 (vertex fetch), :228-310 (combiner formulas),
//                                       :444-581 (surface any-hit), :594-663 (shadow any-hit)
//   shaders/Ray.hlsli:37-94 (ray differentials), Random.hlsli:14-64, BlueNoise.hlsli:7-13, Lights.hlsli:27-168,
//   BgSky.hlsli:14-93, Fog.hlsli:5-27, Color.hlsli:9-43
// The combiner is evaluated by a small interpreter over GpuCombiner instead of generating one program per shaderId.
// Textures are sampled in software (uchar4 loads + fp32 bilinear/trilinear, Texture spec T1-T3 in DESIGN.md): CDNA has
// no sampler hardware to lean on and it keeps texel arithmetic reproducible.
#pragma once
#include "trace.h"

// ---- texel store -------------------------------------------------------------------------------------------------

// Texel address under the sampler's addressing mode.  POW2 (both sizes of the texture are powers of two -- GpuTexture::pow2, the
// usual case): wrap / mirror need a mask, not the ~20-instruction integer remainder (same result for every i, negative ones included).
template <bool POW2> DEV int tex_address(int i, int n, uint32_t mode) {
    if (mode == 2) return i < 0 ? 0 : (i >= n ? n - 1 : i);                 // CLAMP
    if (mode == 1) {                                                         // MIRROR
        const int p = 2 * n;
        int j;
        if (POW2) j = i & (p - 1); else { j = i % p; if (j < 0) j += p; }
        return j < n ? j : p - 1 - j;
    }
    if (POW2) return i & (n - 1);                                            // WRAP
    int j = i % n; if (j < 0) j += n;
    return j;
}

// ---- wave-uniform table reads ---------------------------------------------------------------------------------------
// The per-frame tables (GpuInstance, GpuTexture) are uploaded before the kernel starts and never written by it, so they are read
// through the constant address space: with a wave-uniform index the loads are scalar (s_load into SGPRs, no VGPRs, no exec-masked
// branches on the values); with a divergent index they are ordinary vector loads.  `waterfall` makes an index uniform: it runs
// its body once per distinct key among the active lanes, with the key in a scalar register.
template <class T> DEV T load_const(const T *p) {
    typedef const T __attribute__((address_space(4))) *CP;
    T r; __builtin_memcpy(&r, reinterpret_cast<CP>(reinterpret_cast<uintptr_t>(p)), sizeof(T)); return r;
}
template <class F> DEV void waterfall(uint32_t key, F &&f) {
    // The loop condition is a wave vote, not `for (;;) ... break`: in the compiler's single-thread view the first-lane value is loop
    // invariant, so a loop that can only leave when key == first-lane(key) is "infinite or one trip" and the comparison gets deleted.
    bool todo = true;
    while (__ballot(todo) != 0ull) {
        if (todo) {
            const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
            if (key == k) {
                f((uint32_t)__builtin_amdgcn_readfirstlane((int)key));       // read again inside the branch: keeps the vector `key` from being substituted back in
                todo = false;
            }
        }
    }
}

// The fields of a GpuTexture the sampler needs; the mip offset is indexed by the (per-lane) level, so it stays in memory.
struct TexView { const uint8_t *texels; uint32_t width, height, mips, pow2; const uint32_t *mipOffset; };
struct TexHead { const uint8_t *texels; uint32_t width, height, mips, pow2; };
static_assert(offsetof(GpuTexture, mipOffset) == sizeof(TexHead), "GpuTexture head");
DEV TexView tex_view(const GpuTexture *t) {             // table entry (constant address space)
    const TexHead h = load_const(reinterpret_cast<const TexHead *>(t));
    TexView v; v.texels = h.texels; v.width = h.width; v.height = h.height; v.mips = h.mips; v.pow2 = h.pow2; v.mipOffset = t->mipOffset;
    return v;
}
DEV TexView tex_view_arg(const GpuTexture __attribute__((address_space(4))) &t) {         // a GpuTexture inside the kernel arguments (FrameParams::background)
    TexView v; v.texels = t.texels; v.width = t.width; v.height = t.height; v.mips = t.mips; v.pow2 = t.pow2; v.mipOffset = nullptr;          // a one-level texture: tex_mip_offset never reads the table
    return v;
}

// Offset of a mip level in texels (per-lane level: a vector load from the texture's table; a scalar fetch per distinct level of the
// wave was measured 5x slower for the whole shading kernel -- the vote loop serialises the sample around it).
DEV uint32_t tex_mip_offset(const TexView &t, uint32_t level) {
    if (t.mips == 1) return 0;                               // mipOffset[0] == 0 (Texture::set*, rt64_host.cpp)
    return t.mipOffset[level];
}
DEV f4 tex_texel(const TexView &t, uint32_t mipOffset, int x, int y, int w) {
    const uint32_t *base = reinterpret_cast<const uint32_t *>(t.texels) + mipOffset;
    uint32_t v = base[(size_t)y * (size_t)w + (size_t)x];
    const float k = 1.0f / 255.0f;
    return mk4((float)(v & 0xFF) * k, (float)((v >> 8) & 0xFF) * k, (float)((v >> 16) & 0xFF) * k, (float)(v >> 24) * k);
}

template <bool POW2>
DEV f4 tex_sample_level_impl(const TexView &t, float u, float v, uint32_t level, uint32_t filter, uint32_t hAddr, uint32_t vAddr) {
    int w = max((int)(t.width >> level), 1), h = max((int)(t.height >> level), 1);
    const uint32_t mo = tex_mip_offset(t, level);
    if (filter == 0) {
        int x = tex_address<POW2>((int)floorf(u * (float)w), w, hAddr), y = tex_address<POW2>((int)floorf(v * (float)h), h, vAddr);
        return tex_texel(t, mo, x, y, w);
    }
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = tex_address<POW2>((int)x0f, w, hAddr), x1 = tex_address<POW2>((int)x0f + 1, w, hAddr);
    int y0 = tex_address<POW2>((int)y0f, h, vAddr), y1 = tex_address<POW2>((int)y0f + 1, h, vAddr);
    f4 c00 = tex_texel(t, mo, x0, y0, w), c10 = tex_texel(t, mo, x1, y0, w), c01 = tex_texel(t, mo, x0, y1, w), c11 = tex_texel(t, mo, x1, y1, w);
    f4 r;
    { float top = c00.x + fx * (c10.x - c00.x), bot = c01.x + fx * (c11.x - c01.x); r.x = top + fy * (bot - top); }
    { float top = c00.y + fx * (c10.y - c00.y), bot = c01.y + fx * (c11.y - c01.y); r.y = top + fy * (bot - top); }
    { float top = c00.z + fx * (c10.z - c00.z), bot = c01.z + fx * (c11.z - c01.z); r.z = top + fy * (bot - top); }
    { float top = c00.w + fx * (c10.w - c00.w), bot = c01.w + fx * (c11.w - c01.w); r.w = top + fy * (bot - top); }
    return r;
}
// gBackground has the screen's size, whatever that is: its sampler keeps both addressing forms in every build
DEV f4 tex_sample_level_any(const TexView &t, float u, float v, uint32_t level, uint32_t filter, uint32_t hAddr, uint32_t vAddr) {
    return t.pow2 ? tex_sample_level_impl<true>(t, u, v, level, filter, hAddr, vAddr) : tex_sample_level_impl<false>(t, u, v, level, filter, hAddr, vAddr);
}
DEV f4 tex_sample_level(const TexView &t, float u, float v, uint32_t level, uint32_t filter, uint32_t hAddr, uint32_t vAddr) {
#ifdef RT_ASSUME_SIMPLE
    return tex_sample_level_impl<true>(t, u, v, level, filter, hAddr, vAddr);       // the host checked: every texture of the frame is a power of two in both sizes
#endif
    return t.pow2 ? tex_sample_level_impl<true>(t, u, v, level, filter, hAddr, vAddr) : tex_sample_level_impl<false>(t, u, v, level, filter, hAddr, vAddr);
}

DEV f4 tex_sample_grad(const TexView &t, float u, float v, f2 ddx, f2 ddy, uint32_t filter, uint32_t hAddr, uint32_t vAddr) {
    if (t.mips == 1) return tex_sample_level(t, u, v, 0, filter, hAddr, vAddr);
    float w0 = (float)t.width, h0 = (float)t.height;
    float ax = ddx.x * w0, ay = ddx.y * h0, bx = ddy.x * w0, by = ddy.y * h0;
    float rho = fmaxf(s_sqrt(ax * ax + ay * ay), s_sqrt(bx * bx + by * by));
    float lod = rho > 0.0f ? log2f(rho) : 0.0f;
    float maxLod = (float)(t.mips - 1);
    if (!(lod > 0.0f)) lod = 0.0f;
    if (lod > maxLod) lod = maxLod;
    if (filter == 0) return tex_sample_level(t, u, v, (uint32_t)(int)(lod + 0.5f), filter, hAddr, vAddr);
    int l0 = (int)floorf(lod), l1 = l0 + 1 < (int)t.mips ? l0 + 1 : (int)t.mips - 1;
    float f = lod - (float)l0;
    f4 a = tex_sample_level(t, u, v, (uint32_t)l0, filter, hAddr, vAddr), b = tex_sample_level(t, u, v, (uint32_t)l1, filter, hAddr, vAddr);
    return mk4(a.x + f * (b.x - a.x), a.y + f * (b.y - a.y), a.z + f * (b.z - a.z), a.w + f * (b.w - a.w));
}

// ---- RNG, colour, sky, fog ------------------------------------------------------------------------------------------

DEV uint32_t init_rand(uint32_t val0, uint32_t val1, uint32_t backoff) {      // Random.hlsli:14-26
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (uint32_t n = 0; n < backoff; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
DEV float next_rand(uint32_t &s) { s = 1664525u * s + 1013904223u; return (float)(s & 0x00FFFFFFu) / (float)0x01000000; }   // :28-37

DEV f3 hue_to_rgb(float hue) {                                                 // Color.hlsli:9-14
    f3 rgb = mk3(fabsf(hue * 6.0f - 3.0f) * 1.0f + -1.0f, fabsf(hue * 6.0f - 2.0f) * -1.0f + 2.0f, fabsf(hue * 6.0f - 4.0f) * -1.0f + 2.0f);
    return mk3(saturatef(rgb.x), saturatef(rgb.y), saturatef(rgb.z));
}
DEV f3 mod_rgb_with_hsl(f3 rgb, f3 mod) {                                       // Color.hlsli:16-43
    const float EPS = 1e-10f;
    float p0, p1, p2, p3, q0, q1, q2, q3;
    if (rgb.y < rgb.z) { p0 = rgb.z; p1 = rgb.y; p2 = -1.0f; p3 = 2.0f / 3.0f; } else { p0 = rgb.y; p1 = rgb.z; p2 = 0.0f; p3 = -1.0f / 3.0f; }
    if (rgb.x < p0) { q0 = p0; q1 = p1; q2 = p3; q3 = rgb.x; } else { q0 = rgb.x; q1 = p1; q2 = p2; q3 = p0; }
    float c = q0 - fminf(q3, q1);
    float h = fabsf((q3 - q1) / (6.0f * c + EPS) + q2);
    float z = q0 - c * 0.5f;
    float s = c / (1.0f - fabsf(z * 2.0f - 1.0f) + EPS);
    float H = h + mod.x, S = s + mod.y, L = z + mod.z;
    f3 base = hue_to_rgb(H);
    float cc = (1.0f - fabsf(2.0f * L - 1.0f)) * S;
    return mk3(saturatef((base.x - 0.5f) * cc + L), saturatef((base.y - 0.5f) * cc + L), saturatef((base.z - 0.5f) * cc + L));
}

DEV f2 fake_envmap_uv(f3 d, float yawOffset) {                                  // BgSky.hlsli:14-18
    float yaw = hlsl_fmod(yawOffset + atan2f(d.x, -d.z) + RT_PI, RT_TWO_PI);
    float pitch = hlsl_fmod(atan2f(-d.y, sqrtf(d.x * d.x + d.z * d.z)) + RT_PI, RT_TWO_PI);
    f2 r; r.x = yaw / RT_TWO_PI; r.y = pitch / RT_TWO_PI; return r;
}

// ComputeSkyPlaneUV, BgSky.hlsli:20-52.  Everything that depends only on the view (yaw, pitch, aspect) is evaluated once
// per frame on the host (sky_plane_base in rt64_host.cpp: P.skyBase = {baseU, baseV, 0.25 * ratioDivision, 0.25}).
DEV f2 sky_plane_uv(PRef P, f2 uv) {
    f2 r;
    r.x = P.skyBase[0] + uv.x * P.skyBase[2];
    r.y = P.skyBase[1] + uv.y * P.skyBase[3];
    return r;
}

DEV f4 sky_finish(PRef P, f4 tex) {
    f4 sky = mk4(tex.x * P.skyDiffuseMultiplier[0], tex.y * P.skyDiffuseMultiplier[1], tex.z * P.skyDiffuseMultiplier[2], tex.w);
    if (P.skyHSLModifier[0] != 0.0f || P.skyHSLModifier[1] != 0.0f || P.skyHSLModifier[2] != 0.0f) {
        f3 r = mod_rgb_with_hsl(xyz(sky), mk3(P.skyHSLModifier[0], P.skyHSLModifier[1], P.skyHSLModifier[2]));
        sky.x = r.x; sky.y = r.y; sky.z = r.z;
    }
    return sky;
}
DEV f4 sample_sky_2d(PRef P, f2 screenUV) {                       // SampleSky2D :54-70
    if (P.skyPlaneTexIndex < 0) return mk4(0, 0, 0, 0);
    f2 uv = sky_plane_uv(P, screenUV);
    return sky_finish(P, tex_sample_level(tex_view(P.textures + P.skyPlaneTexIndex), uv.x, uv.y, 0, 1, 0, 0));
}
// LINEAR / WRAP sample of level 0 from the tiled copy of the sky plane: the arithmetic of tex_sample_level_impl<true> (same values, bit for
// bit), only the texel address differs.
DEV f4 sky_tiled_sample(PRef P, float u, float v) {
    const uint32_t lw = P.skyTiledLog2W, lh = P.skyTiledLog2H;
    const int w = 1 << lw, h = 1 << lh;
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    const int x0 = (int)x0f & (w - 1), x1 = ((int)x0f + 1) & (w - 1), y0 = (int)y0f & (h - 1), y1 = ((int)y0f + 1) & (h - 1);
    auto texel = [&](int tx, int ty) -> f4 {
        const uint32_t idx = ((((uint32_t)ty >> 2) << (lw - 2)) + ((uint32_t)tx >> 2)) * 16u + (((uint32_t)ty & 3u) << 2) + ((uint32_t)tx & 3u);
        const uint32_t t = P.skyTiled[idx];
        const float k = 1.0f / 255.0f;
        return mk4((float)(t & 0xFF) * k, (float)((t >> 8) & 0xFF) * k, (float)((t >> 16) & 0xFF) * k, (float)(t >> 24) * k);
    };
    f4 c00 = texel(x0, y0), c10 = texel(x1, y0), c01 = texel(x0, y1), c11 = texel(x1, y1);
    f4 r;
    { float top = c00.x + fx * (c10.x - c00.x), bot = c01.x + fx * (c11.x - c01.x); r.x = top + fy * (bot - top); }
    { float top = c00.y + fx * (c10.y - c00.y), bot = c01.y + fx * (c11.y - c01.y); r.y = top + fy * (bot - top); }
    { float top = c00.z + fx * (c10.z - c00.z), bot = c01.z + fx * (c11.z - c01.z); r.z = top + fy * (bot - top); }
    { float top = c00.w + fx * (c10.w - c00.w), bot = c01.w + fx * (c11.w - c01.w); r.w = top + fy * (bot - top); }
    return r;
}
DEV f4 sample_sky_plane(PRef P, f3 rayDirection) {                // SampleSkyPlane :72-87
    if (P.skyPlaneTexIndex < 0) return mk4(0, 0, 0, 0);
    f2 uv = fake_envmap_uv(rayDirection, P.skyYawOffset);
    if (P.skyTiled) return sky_finish(P, sky_tiled_sample(P, uv.x, uv.y));
    return sky_finish(P, tex_sample_level(tex_view(P.textures + P.skyPlaneTexIndex), uv.x, uv.y, 0, 1, 0, 0));
}
// gBackground: the raster background instances drawn into a screen-size RGBA8 target (rt64_view.cpp:1296-1319, raster.hip),
// sampled with the static LINEAR / WRAP sampler (BgSky.hlsli:89-95).  No background instance => transparent black.
DEV f3 sample_background_2d(PRef P, f2 screenUV) {
    if (!P.background.texels) return mk3s(0.0f);
    return xyz(tex_sample_level_any(tex_view_arg(P.background), screenUV.x, screenUV.y, 0, 1, 0, 0));
}
DEV f3 sample_background_envmap(PRef P, f3 rayDirection) {
    if (!P.background.texels) return mk3s(0.0f);
    const f2 uv = fake_envmap_uv(rayDirection, 0.0f);
    return xyz(tex_sample_level_any(tex_view_arg(P.background), uv.x, uv.y, 0, 1, 0, 0));
}

// lerp(background, sky, sky.a) as the ray-gen shaders write it (BgSky.hlsli:89-95 with SampleSkyPlane / SampleSky2D).  Under an opaque sky
// texel (a >= 1) that is the sky itself up to one rounding of (sky - bg) + bg, so the gBackground fetch is skipped there: for a bounce
// ray it is a random access into a screen-size image (C5: 4.2 GB of bounce_miss_kernel's fetches), for nothing.
DEV f3 sky_over_background_envmap(PRef P, f3 rayDirection) {
    const f4 sky = sample_sky_plane(P, rayDirection);
    if (sky.w >= 1.0f) return xyz(sky);
    return lerp3(sample_background_envmap(P, rayDirection), xyz(sky), sky.w);
}
DEV f3 sky_over_background_2d(PRef P, f2 screenUV) {
    const f4 sky = sample_sky_2d(P, screenUV);
    if (sky.w >= 1.0f) return xyz(sky);
    return lerp3(sample_background_2d(P, screenUV), xyz(sky), sky.w);
}

DEV f4 fog_from_camera(PRef P, const RT64_MATERIAL &m, f3 position) {    // Fog.hlsli:5-18
    f4 clip = mul4(cmat(P.viewProj).m, mk4(position.x, position.y, position.z, 1.0f));
    clip.z = clip.z * 2.0f - clip.w;
    float winv = 1.0f / fmaxf(clip.w, 0.001f);
    return mk4(m.fogColor.x, m.fogColor.y, m.fogColor.z, clampf((clip.z * winv * m.fogMul + m.fogOffset) / 255.0f, 0.0f, 1.0f));
}
DEV f4 fog_from_origin(const RT64_MATERIAL &m, f3 position, f3 origin) {               // Fog.hlsli:20-27
    float distance = len3(position - origin);
    return mk4(m.fogColor.x, m.fogColor.y, m.fogColor.z, clampf(((distance + m.fogOffset) / m.fogMul) * 0.5f, 0.0f, 1.0f));
}

DEV f3 blue_noise(PRef P, uint32_t px, uint32_t py, uint32_t frame) {     // BlueNoise.hlsli:7-13
    uint32_t f = frame % 64u;
    uint32_t bx = (f % 8u) * 64u + px % 64u, by = (f / 8u) * 64u + py % 64u;
    uint32_t v = reinterpret_cast<const uint32_t *>(P.blueNoise)[(size_t)by * 512u + bx];
    const float k = 1.0f / 255.0f;
    return mk3((float)(v & 0xFF) * k, (float)((v >> 8) & 0xFF) * k, (float)((v >> 16) & 0xFF) * k);
}

// ---- ray differentials, Ray.hlsli:37-94 ----------------------------------------------------------------------------

struct RayDiff { f3 dOdx, dOdy, dDdx, dDdy; };

DEV void compute_ray_diffs(f3 nonNormDir, f3 right, f3 up, float vw, float vh, f3 &dDdx, f3 &dDdy) {
    float dd = dot3(nonNormDir, nonNormDir);
    float divd = s_div(2.0f, dd * s_sqrt(dd));
    float dr = dot3(nonNormDir, right), du = dot3(nonNormDir, up);
    dDdx = ((right * dd - nonNormDir * dr) * divd) * s_rcp(vw);
    dDdy = -(((up * dd - nonNormDir * du) * divd) * s_rcp(vh));
}

// ---- vertex fetch + combiner ----------------------------------------------------------------------------------------

struct VertexData {
    f3 pos[3], posW[3];
    f2 uv[3];
    f4 input[4];
    f3 vertexPosition, vertexNormal, triangleNormal, vertexTangent, vertexBinormal;
    f2 vertexUV;
};

DEV f3 ld_f3(const uint8_t *p) { const float *f = reinterpret_cast<const float *>(p); return mk3(f[0], f[1], f[2]); }

// Copy of the GpuInstance fields the any-hit programs read (scalar registers when the instance index is wave-uniform; only the
// fields a program touches are actually loaded).
struct Mat16 { float m[16]; };
struct InstView {
    GpuCombiner cc; RT64_MATERIAL material; Mat16 objectToWorld, objectToWorldNormal, objectToWorldPrevious;
    const uint8_t *vertices; const uint32_t *indices; int32_t texDiffuse, texNormal, texSpecular; uint32_t filter, hAddr, vAddr, flags;
};
struct CombinerWords { uint32_t w[sizeof(GpuCombiner) / 4]; };          // the combiner's int8 / int16 fields come out of whole-dword scalar loads
static_assert(sizeof(GpuCombiner) % 4 == 0 && offsetof(GpuInstance, cc) % 4 == 0, "GpuCombiner is read as dwords");
struct InstTail { int32_t texDiffuse, texNormal, texSpecular; uint32_t filter, hAddr, vAddr, flags; };
static_assert(offsetof(GpuInstance, flags) - offsetof(GpuInstance, texDiffuse) == offsetof(InstTail, flags), "GpuInstance tail");
DEV InstView inst_view(PRef P, uint32_t instance) {
    const GpuInstance *g = P.instances + instance;
    InstView v;
    const CombinerWords cw = load_const(reinterpret_cast<const CombinerWords *>(&g->cc));
    __builtin_memcpy(&v.cc, &cw, sizeof(v.cc));
    v.material = load_const(&g->material);
    v.objectToWorld = load_const(reinterpret_cast<const Mat16 *>(g->objectToWorld));
    v.objectToWorldNormal = load_const(reinterpret_cast<const Mat16 *>(g->objectToWorldNormal));
    v.objectToWorldPrevious = load_const(reinterpret_cast<const Mat16 *>(g->objectToWorldPrevious));
    v.vertices = load_const(&g->vertices); v.indices = load_const(&g->indices);
    const InstTail tl = load_const(reinterpret_cast<const InstTail *>(&g->texDiffuse));
    v.texDiffuse = tl.texDiffuse; v.texNormal = tl.texNormal; v.texSpecular = tl.texSpecular; v.filter = tl.filter; v.hAddr = tl.hAddr; v.vAddr = tl.vAddr; v.flags = tl.flags;
    return v;
}

DEV void get_vertex_data(const InstView &in, uint32_t prim, const float b[3], bool wantTangent, VertexData &vd) {   // rt64_shader.cpp:156-226
    const GpuCombiner &cc = in.cc;
    const uint8_t *vp[3];
    f3 norm[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        vp[k] = in.vertices + (size_t)in.indices[3 * prim + k] * (size_t)cc.vertexSize;
        vd.pos[k] = ld_f3(vp[k]);
        vd.posW[k] = mul_point(in.objectToWorld.m, vd.pos[k]);
        norm[k] = ld_f3(vp[k] + cc.normalOffset);
    }
    vd.vertexPosition = (vd.pos[0] * b[0] + vd.pos[1] * b[1]) + vd.pos[2] * b[2];
    f3 vn = (norm[0] * b[0] + norm[1] * b[1]) + norm[2] * b[2];
    f3 tn = -cross3(vd.pos[2] - vd.pos[0], vd.pos[1] - vd.pos[0]);
    vd.vertexNormal = (vn.x != 0.0f || vn.y != 0.0f || vn.z != 0.0f) ? normalize3(vn) : tn;
    vd.triangleNormal = normalize3(mul_vector(in.objectToWorldNormal.m, tn));
    if (cc.vertexUV) {
#pragma unroll
        for (int k = 0; k < 3; k++) { const float *f = reinterpret_cast<const float *>(vp[k] + cc.uvOffset); vd.uv[k].x = f[0]; vd.uv[k].y = f[1]; }
        vd.vertexUV.x = vd.uv[0].x * b[0] + vd.uv[1].x * b[1] + vd.uv[2].x * b[2];
        vd.vertexUV.y = vd.uv[0].y * b[0] + vd.uv[1].y * b[1] + vd.uv[2].y * b[2];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {          // fully unrolled + predicated: keeps VertexData in registers (no dynamic indexing)
        if (i < cc.inputCount) {
            float r[4];
            const float *f0 = reinterpret_cast<const float *>(vp[0] + cc.inputOffset[i]);
            const float *f1 = reinterpret_cast<const float *>(vp[1] + cc.inputOffset[i]);
            const float *f2p = reinterpret_cast<const float *>(vp[2] + cc.inputOffset[i]);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) r[ch] = f0[ch] * b[0] + f1[ch] * b[1] + f2p[ch] * b[2];
            r[3] = cc.optAlpha ? (f0[3] * b[0] + f1[3] * b[1] + f2p[3] * b[2]) : 1.0f;
            vd.input[i] = mk4(r[0], r[1], r[2], r[3]);
        }
        else vd.input[i] = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (wantTangent) {                                                          // :201-225
        float uva = vd.uv[1].x - vd.uv[0].x, uvb = vd.uv[2].x - vd.uv[0].x;
        float uvc = vd.uv[1].y - vd.uv[0].y, uvd = vd.uv[2].y - vd.uv[0].y;
        float uvk = uvb * uvc - uva * uvd;
        f3 dpos1 = vd.pos[1] - vd.pos[0], dpos2 = vd.pos[2] - vd.pos[0];
        f3 tangent;
        if (uvk != 0.0f) { f3 n = dpos2 * uvc - dpos1 * uvd; tangent = normalize3(n * s_rcp(uvk)); }
        else if (uva != 0.0f) tangent = normalize3(dpos1 * s_rcp(uva));
        else if (uvb != 0.0f) tangent = normalize3(dpos2 * s_rcp(uvb));
        else tangent = mk3s(0.0f);
        float d1x = vd.uv[1].x - vd.uv[0].x, d1y = -(vd.uv[1].y - vd.uv[0].y);
        float d2x = vd.uv[2].x - vd.uv[1].x, d2y = -(vd.uv[2].y - vd.uv[1].y);
        float crz = d1x * d2y - d1y * d2x;
        float binormalMult = (crz < 0.0f) ? -1.0f : 1.0f;
        vd.vertexTangent = tangent;
        vd.vertexBinormal = cross3(tangent, vd.vertexNormal) * binormalMult;
    }
}

DEV f4 pick_input(const VertexData &vd, int item) {      // vd.input[item - 1] without a dynamically indexed array
    // component-wise value selects (a ?: over the struct lvalues would become a pointer select and pin VertexData in scratch)
    const f4 a = vd.input[0], b = vd.input[1], c = vd.input[2], d = vd.input[3];
    f4 r;
    r.x = item == 1 ? a.x : (item == 2 ? b.x : (item == 3 ? c.x : d.x));
    r.y = item == 1 ? a.y : (item == 2 ? b.y : (item == 3 ? c.y : d.y));
    r.z = item == 1 ? a.z : (item == 2 ? b.z : (item == 3 ? c.z : d.z));
    r.w = item == 1 ? a.w : (item == 2 ? b.w : (item == 3 ? c.w : d.w));
    return r;
}

DEV f4 color_input(int item, bool with_alpha, bool inputs_have_alpha, bool hint_single, const VertexData &vd, f4 t0, f4 t1) {   // :228-258
    f4 r;
    switch (item) {
    default: case 0: return mk4(0.0f, 0.0f, 0.0f, with_alpha ? 0.0f : 1.0f);
    case 1: case 2: case 3: case 4:
        r = pick_input(vd, item);
        if (!(with_alpha || !inputs_have_alpha)) r.w = 1.0f;
        return r;
    case 5: r = t0; if (!with_alpha) r.w = 1.0f; return r;
    case 6: return mk4(t0.w, t0.w, t0.w, (hint_single || with_alpha) ? t0.w : 1.0f);
    case 7: r = t1; if (!with_alpha) r.w = 1.0f; return r;
    }
}

DEV f4 color_formula(const GpuCombiner &cc, bool with_alpha, bool opt_alpha, const VertexData &vd, f4 t0, f4 t1) {   // :260-273
    const int8_t *c = cc.c[0];
    if (cc.doSingle[0]) return color_input(c[3], with_alpha, opt_alpha, false, vd, t0, t1);
    if (cc.doMultiply[0]) {
        f4 a = color_input(c[0], with_alpha, opt_alpha, false, vd, t0, t1), b = color_input(c[2], with_alpha, opt_alpha, true, vd, t0, t1);
        return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
    }
    if (cc.doMix[0]) {
        f4 x = color_input(c[1], with_alpha, opt_alpha, false, vd, t0, t1), y = color_input(c[0], with_alpha, opt_alpha, false, vd, t0, t1);
        f4 s = color_input(c[2], with_alpha, opt_alpha, true, vd, t0, t1);
        return mk4(lerpf(x.x, y.x, s.x), lerpf(x.y, y.y, s.y), lerpf(x.z, y.z, s.z), lerpf(x.w, y.w, s.w));
    }
    f4 a = color_input(c[0], with_alpha, opt_alpha, false, vd, t0, t1), b = color_input(c[1], with_alpha, opt_alpha, false, vd, t0, t1);
    f4 s = color_input(c[2], with_alpha, opt_alpha, true, vd, t0, t1), d = color_input(c[3], with_alpha, opt_alpha, false, vd, t0, t1);
    return mk4((a.x - b.x) * s.x + d.x, (a.y - b.y) * s.x + d.y, (a.z - b.z) * s.x + d.z, (a.w - b.w) * s.x + d.w);
}

DEV float alpha_input(int item, const VertexData &vd, f4 t0, f4 t1) {            // :275-295
    switch (item) {
    default: case 0: return 0.0f;
    case 1: case 2: case 3: case 4: return pick_input(vd, item).w;
    case 5: case 6: return t0.w;
    case 7: return t1.w;
    }
}
DEV float alpha_formula(const GpuCombiner &cc, const VertexData &vd, f4 t0, f4 t1) {   // :297-310
    const int8_t *c = cc.c[1];
    if (cc.doSingle[1]) return alpha_input(c[3], vd, t0, t1);
    if (cc.doMultiply[1]) return alpha_input(c[0], vd, t0, t1) * alpha_input(c[2], vd, t0, t1);
    if (cc.doMix[1]) return lerpf(alpha_input(c[1], vd, t0, t1), alpha_input(c[0], vd, t0, t1), alpha_input(c[2], vd, t0, t1));
    return (alpha_input(c[0], vd, t0, t1) - alpha_input(c[1], vd, t0, t1)) * alpha_input(c[2], vd, t0, t1) + alpha_input(c[3], vd, t0, t1);
}

// One entry of the per-pixel hit list with the reference's storage precision (rt64_view.cpp:237-241,573-594).
struct HitRecord {
    float dist; f3 flow;                   // gHitDistAndFlow (RGBA32F)
    f4 color;                              // gHitColor    (RGBA8 UNORM, already quantised)
    f3 normal;                             // gHitNormal   (RGBA16 SNORM, already quantised)
    f3 specular;                           // gHitSpecular (RGBA8 UNORM, already quantised)
    uint32_t instanceId;
};

// Surface any-hit, rt64_shader.cpp:444-581.  Returns false when the candidate is ignored before it is stored.
// `in` = inst_view(P, instance); callers that can make the instance wave-uniform (waterfall) get the scalar version of everything below.
DEV bool surface_anyhit_view(PRef P, const InstView &in, uint32_t instance, uint32_t prim, float t, float u, float v, f3 rayDirW,
                             const RayDiff &payloadDiff, uint32_t px, uint32_t py, HitRecord &rec) {
    const GpuCombiner &cc = in.cc;
    const RT64_MATERIAL &mat = in.material;
    const bool normalMap = (in.flags & GPU_INST_NORMAL_MAP) != 0, specularMap = (in.flags & GPU_INST_SPECULAR_MAP) != 0;
    const float b[3] = { 1.0f - u - v, u, v };
    const f4 mix = mk4(mat.diffuseColorMix.x, mat.diffuseColorMix.y, mat.diffuseColorMix.z, mat.diffuseColorMix.w);
    VertexData vd;
    get_vertex_data(in, prim, b, cc.vertexUV && normalMap, vd);

    f2 ddx, ddy; ddx.x = ddx.y = ddy.x = ddy.y = 0.0f;
    f4 t0 = mk4(0, 0, 0, 0), t1 = mk4(1.0f, 0.0f, 1.0f, 1.0f);
    if (cc.useTex0) {
        // propagateRayDiffs + computeBarycentricDifferentials + computeTextureDifferentials, Ray.hlsli:47-94
        f3 N = vd.triangleNormal;
        f3 dodx = payloadDiff.dOdx + payloadDiff.dDdx * t, dody = payloadDiff.dOdy + payloadDiff.dDdy * t;
        float rcpDN = s_rcp(dot3(rayDirW, N));
        float dtdx = -dot3(dodx, N) * rcpDN, dtdy = -dot3(dody, N) * rcpDN;
        dodx = dodx + rayDirW * dtdx; dody = dody + rayDirW * dtdy;
        f3 e01 = vd.posW[1] - vd.posW[0], e02 = vd.posW[2] - vd.posW[0];
        f3 Nu = cross3(e02, N), Nv = cross3(e01, N);
        float du = dot3(Nu, e01), dv = dot3(Nv, e02);
        const float rdu = s_rcp(du), rdv = s_rcp(dv);
        f3 Lu = Nu * rdu, Lv = Nv * rdv;
        float dBdx_x = dot3(Lu, dodx), dBdx_y = dot3(Lv, dodx), dBdy_x = dot3(Lu, dody), dBdy_y = dot3(Lv, dody);
        float uv01x = vd.uv[1].x - vd.uv[0].x, uv01y = vd.uv[1].y - vd.uv[0].y, uv02x = vd.uv[2].x - vd.uv[0].x, uv02y = vd.uv[2].y - vd.uv[0].y;
        ddx.x = dBdx_x * uv01x + dBdx_y * uv02x; ddx.y = dBdx_x * uv01y + dBdx_y * uv02y;
        ddy.x = dBdy_x * uv01x + dBdy_y * uv02x; ddy.y = dBdy_x * uv01y + dBdy_y * uv02y;
        f4 tex = tex_sample_grad(tex_view(P.textures + in.texDiffuse), vd.vertexUV.x, vd.vertexUV.y, ddx, ddy, in.filter, in.hAddr, in.vAddr);
        float k = fmaxf(-mix.w, 0.0f);
        t0 = mk4(lerpf(tex.x, mix.x, k), lerpf(tex.y, mix.y, k), lerpf(tex.z, mix.z, k), tex.w);
    }
    f4 result;
    if (!cc.colorAlphaSame && cc.optAlpha) {
        result = color_formula(cc, false, true, vd, t0, t1);
        result.w = alpha_formula(cc, vd, t0, t1);
    }
    else result = color_formula(cc, cc.optAlpha, cc.optAlpha, vd, t0, t1);
    {
        float k = fmaxf(mix.w, 0.0f);
        result.x = lerpf(result.x, mix.x, k); result.y = lerpf(result.y, mix.y, k); result.z = lerpf(result.z, mix.z, k);
    }
    result.w = clampf(mat.solidAlphaMultiplier * result.w, 0.0f, 1.0f);
    if (cc.optTextureEdge) { if (result.w > 0.3f) result.w = 1.0f; else return false; }     // :502-511
    if (cc.optNoise) {                                                                      // :513-516
        uint32_t seed = init_rand(px + py * (uint32_t)P.width, P.frameCount, 16);
        result.w *= rintf(next_rand(seed));
    }
    f3 vertexNormal = normalize3(mul_vector(in.objectToWorldNormal.m, vd.vertexNormal));
    float normalSign = (dot3(vd.triangleNormal, rayDirW) <= 0.0f) ? 1.0f : -1.0f;
    vertexNormal = vertexNormal * normalSign;
    if (cc.vertexUV && normalMap) {                                                         // :522-533
        f3 tangent = normalize3(mul_vector(in.objectToWorldNormal.m, vd.vertexTangent)) * normalSign;
        f3 binormal = normalize3(mul_vector(in.objectToWorldNormal.m, vd.vertexBinormal)) * normalSign;
        if (in.texNormal >= 0) {
            float s = mat.uvDetailScale;
            f2 gx, gy; gx.x = ddx.x * s; gx.y = ddx.y * s; gy.x = ddy.x * s; gy.y = ddy.y * s;
            f4 tex = tex_sample_grad(tex_view(P.textures + in.texNormal), vd.vertexUV.x * s, vd.vertexUV.y * s, gx, gy, in.filter, in.hAddr, in.vAddr);
            f3 nc = mk3(tex.x * 2.0f - 1.0f, tex.y * 2.0f - 1.0f, tex.z * 2.0f - 1.0f);
            vertexNormal = normalize3((vertexNormal * nc.z + tangent * nc.x) + binormal * nc.y);
        }
    }
    f3 prevWorldPos = mul_point(in.objectToWorldPrevious.m, vd.vertexPosition);
    f3 curWorldPos = mul_point(in.objectToWorld.m, vd.vertexPosition);
    f3 vertexSpecular = mk3s(1.0f);
    if (cc.vertexUV && specularMap && in.texSpecular >= 0) {                               // :539-545
        float s = mat.uvDetailScale;
        f2 gx, gy; gx.x = ddx.x * s; gx.y = ddx.y * s; gy.x = ddy.x * s; gy.y = ddy.y * s;
        f4 tex = tex_sample_grad(tex_view(P.textures + in.texSpecular), vd.vertexUV.x * s, vd.vertexUV.y * s, gx, gy, in.filter, in.hAddr, in.vAddr);
        vertexSpecular = xyz(tex);
    }
    rec.dist = t - mat.depthBias;
    rec.flow = curWorldPos - prevWorldPos;
    rec.color = mk4(q_unorm8(result.x), q_unorm8(result.y), q_unorm8(result.z), q_unorm8(result.w));
    rec.normal = mk3(q_snorm16(vertexNormal.x), q_snorm16(vertexNormal.y), q_snorm16(vertexNormal.z));
    rec.specular = mk3(q_unorm8(vertexSpecular.x), q_unorm8(vertexSpecular.y), q_unorm8(vertexSpecular.z));
    rec.instanceId = instance;
    return true;
}
// Any instance index per lane: one pass per distinct instance of the wave, each with the instance's data in scalar registers.
DEV bool surface_anyhit(PRef P, uint32_t instance, uint32_t prim, float t, float u, float v, f3 rayDirW,
                        const RayDiff &payloadDiff, uint32_t px, uint32_t py, HitRecord &rec) {
    bool ok = false;
    waterfall(instance, [&](uint32_t k) { ok = surface_anyhit_view(P, inst_view(P, k), k, prim, t, u, v, rayDirW, payloadDiff, px, py, rec); });
    return ok;
}

// Shadow any-hit alpha, rt64_shader.cpp:611-659.  Negative = candidate ignored (texture edge).
DEV float shadow_anyhit_alpha_view(PRef P, const InstView &in, uint32_t prim, float u, float v, uint32_t px, uint32_t py) {
    const GpuCombiner &cc = in.cc;
    const float b[3] = { 1.0f - u - v, u, v };
    VertexData vd;
    get_vertex_data(in, prim, b, false, vd);
    f4 t0 = mk4(0, 0, 0, 0), t1 = mk4(1.0f, 0.0f, 1.0f, 1.0f);
    if (cc.useTex0) t0 = tex_sample_level(tex_view(P.textures + in.texDiffuse), vd.vertexUV.x, vd.vertexUV.y, 0, in.filter, in.hAddr, in.vAddr);
    float a;
    if (!cc.colorAlphaSame && cc.optAlpha) a = alpha_formula(cc, vd, t0, t1);
    else a = color_formula(cc, cc.optAlpha, cc.optAlpha, vd, t0, t1).w;
    a = clampf(a * in.material.shadowAlphaMultiplier, 0.0f, 1.0f);
    if (cc.optTextureEdge) { if (a > 0.3f) a = 1.0f; else return -1.0f; }
    if (cc.optNoise) {
        uint32_t seed = init_rand(px + py * (uint32_t)P.width, P.frameCount, 16);
        a *= rintf(next_rand(seed));
    }
    return a;
}
DEV float shadow_anyhit_alpha(PRef P, uint32_t instance, uint32_t prim, float u, float v, uint32_t px, uint32_t py) {
    float a = 0.0f;
    waterfall(instance, [&](uint32_t k) { a = shadow_anyhit_alpha_view(P, inst_view(P, k), prim, u, v, px, py); });
    return a;
}

// ---- shadow rays + lights, Lights.hlsli -------------------------------------------------------------------------------

struct ShadeEnv {                // per-lane traversal resources handed down to the light loop
    TraceStack stk;
    TraceCounts cnt;
    uint32_t shadowRays;
    float *lightIntensity;       // LDS, [wave][slots][RT_LANES] (this lane's column): candidate intensities of ComputeLightsRandom
    uint8_t *lightIndex;         // LDS, [wave][slots][RT_LANES]
};

template <bool CACHED = false>
DEV float trace_shadow(PRef P, ShadeEnv &env, f3 origin, f3 dir, float tmin, float tmax, uint32_t px, uint32_t py) {   // :27-52
    float o[3] = { origin.x, origin.y, origin.z }, d[3] = { dir.x, dir.y, dir.z };
    float shadowHit = 1.0f;
    env.shadowRays++;
    trace_ray<CACHED>(P, o, d, tmin, tmax, false /* SKIP_BACKFACE_SHADOWS undefined */, env.stk,
              [&](float, float u, float v, uint32_t instance, uint32_t prim, float &, uint32_t instFlags, float) -> bool {
                  if (instFlags & GPU_INST_SHADOW_OPAQUE) { shadowHit = 0.0f; return true; }   // payload.shadowHit = 0 (:661)
#ifdef RT_ASSUME_SIMPLE
                  shadowHit = 0.0f; return true;          // the host checked: every instance of the frame is shadow-opaque, the any-hit program is never needed
#endif
                  float a = shadow_anyhit_alpha(P, instance, prim, u, v, px, py);
                  if (a < 0.0f) return false;
                  shadowHit = fmaxf(shadowHit - a, 0.0f);
                  return !(shadowHit > 0.0f);           // accepted + ACCEPT_FIRST_HIT_AND_END_SEARCH
              }, env.cnt);
    return shadowHit;
}

DEV f3 ld_v3(const RT64_VECTOR3 &v) { return mk3(v.x, v.y, v.z); }

DEV float light_intensity_simple(const RT64_LIGHT &L, f3 position, f3 normal, float ignoreNormalFactor) {   // :54-65
    f3 lp = ld_v3(L.position);
    float lightDistance = len3(position - lp);
    f3 lightDirection = normalize3(lp - position);
    float NdotL = dot3(normal, lightDirection);
    float surfaceBias = fmaxf(lerpf(NdotL, 1.0f, ignoreNormalFactor) + 0.707106f, 0.0f);
    float f = s_pow(fmaxf(1.0f - s_div(lightDistance, L.attenuationRadius), 0.0f), L.attenuationExponent) * surfaceBias;
    return f * (L.diffuseColor.x + L.diffuseColor.y + L.diffuseColor.z);
}

template <bool CACHED = false>
DEV f3 compute_light(PRef P, ShadeEnv &env, uint32_t px, uint32_t py, uint32_t lightIndex, f3 rayDirection,
                     const RT64_MATERIAL &m, f3 position, f3 normal, f3 specular, bool checkShadows) {   // :67-113
    RT64_LIGHT L;
    waterfall(lightIndex, [&](uint32_t k) { L = load_const(P.lights + k); });      // one scalar fetch per distinct light chosen in the wave
    f3 lp = ld_v3(L.position);
    f3 lightDirection = normalize3(lp - position);
    float lightPointRadius = (P.diSamples > 0) ? L.pointRadius : 0.0f;
    f3 perpX = cross3(-lightDirection, mk3(0.0f, 1.0f, 0.0f));
    if (perpX.x == 0.0f && perpX.y == 0.0f && perpX.z == 0.0f) perpX.x = 1.0f;
    f3 perpY = cross3(perpX, -lightDirection);
    uint32_t maxSamples = P.diSamples > 1 ? P.diSamples : 1, samples = maxSamples;
    float lLambert = 0.0f, lShadow = 0.0f; f3 lSpec = mk3s(0.0f);
    while (samples > 0) {
        // Point light (radius 0): the disc offsets are finite numbers times 0 (a blue-noise byte / 255 is never 0.5, so the disc
        // vector is never 0 / 0), i.e. the sample position is lp exactly -- no blue-noise fetch needed for the same value.
        f3 samplePosition = lp;
        if (lightPointRadius != 0.0f || !(lightDirection.x == lightDirection.x)) {      // (a NaN light direction -- shading point exactly at the light -- keeps the reference's NaN)
            f3 bn = blue_noise(P, px, py, P.frameCount + samples);
            float scx = bn.x * 2.0f - 1.0f, scy = bn.y * 2.0f - 1.0f;
            float len = s_sqrt(scx * scx + scy * scy), sat = saturatef(len), rlen = s_rcp(len);
            scx = scx * rlen * sat; scy = scy * rlen * sat;
            samplePosition = (lp + (perpX * scx) * lightPointRadius) + (perpY * scy) * lightPointRadius;
        }
        float sampleDistance = len3_exact(position - samplePosition);          // ray tmax and ...
        f3 sampleDirection = normalize3_exact(samplePosition - position);      // ... ray direction: exact
        float sampleIntensityFactor = s_pow(fmaxf(1.0f - s_div(sampleDistance, L.attenuationRadius), 0.0f), L.attenuationExponent);
        f3 reflectedLight = reflect3(-sampleDirection, normal);
        float NdotL = fmaxf(dot3(normal, sampleDirection), 0.0f);
        float sampleLambert = lerpf(NdotL, 1.0f, m.ignoreNormalFactor) * sampleIntensityFactor;
        float sampleShadow = 1.0f;
        if (checkShadows)
            sampleShadow = trace_shadow<CACHED>(P, env, position, sampleDirection, RT_RAY_MIN_DISTANCE + m.shadowRayBias, sampleDistance - L.shadowOffset, px, py);
        float sp = s_pow(fmaxf(saturatef(dot3(reflectedLight, -rayDirection) * sampleIntensityFactor), 0.0f), m.specularExponent);
        const float rs = s_rcp((float)maxSamples);
        lLambert += sampleLambert * rs;
        lSpec = lSpec + (specular * sp) * rs;
        lShadow += sampleShadow * rs;
        samples--;
    }
    f3 r = ld_v3(L.diffuseColor) * lLambert + ld_v3(L.specularColor) * lSpec;
    return r * lShadow;
}

template <bool CACHED = false>
DEV f3 compute_lights_random(PRef P, ShadeEnv &env, uint32_t px, uint32_t py, f3 rayDirection, uint32_t instanceId,
                             f3 position, f3 normal, f3 specular, uint32_t maxLightCount, bool checkShadows) {   // :115-168
    f3 result = mk3s(0.0f);
    // the material fields the light loop reads, fetched once per distinct instance of the wave through scalar loads
    RT64_MATERIAL m;
    waterfall(instanceId, [&](uint32_t k) {
        const RT64_MATERIAL mk = load_const(&P.instances[k].material);
        m.lightGroupMaskBits = mk.lightGroupMaskBits; m.ignoreNormalFactor = mk.ignoreNormalFactor; m.shadowRayBias = mk.shadowRayBias; m.specularExponent = mk.specularExponent;
    });
    if (m.lightGroupMaskBits == 0) return result;
    // sLightIntensities / sLightIndices (Lights.hlsli:121-122) live in LDS, one column per lane: dynamically indexed
    // per-lane arrays would otherwise go to scratch memory.
    float *sInt = env.lightIntensity; uint8_t *sIdx = env.lightIndex;
    uint32_t sCount = 0; float total = 0.0f;
    for (uint32_t l = 0; l < P.lightCount && sCount < RT64_MAX_LIGHTS; l++) {
        const RT64_LIGHT Ll = load_const(P.lights + l);           // uniform index: scalar loads
        if (m.lightGroupMaskBits & Ll.groupBits) {
            float li = light_intensity_simple(Ll, position, normal, m.ignoreNormalFactor);
            if (li > RT_EPSILON) { sInt[sCount * RT_LANES] = li; sIdx[sCount * RT_LANES] = (uint8_t)l; total += li; sCount++; }
        }
    }
    if (sCount == 1 && maxLightCount >= 1)       // one candidate: it is chosen whatever the random number is, with probability 1 (randomRange / cInt = total / total)
        return compute_light<CACHED>(P, env, px, py, sIdx[0], rayDirection, m, position, normal, specular, checkShadows);
    float randomRange = total;
    uint32_t lCount = sCount < maxLightCount ? sCount : maxLightCount;
    bool useProbability = lCount == 1;
    for (uint32_t s = 0; s < lCount; s++) {
        float r = blue_noise(P, px, py, P.frameCount + s).x * randomRange;
        uint32_t chosen = 0; float rInt = sInt[0];
        while (chosen < sCount - 1 && r >= rInt) { chosen++; rInt += sInt[chosen * RT_LANES]; }
        float cInt = sInt[chosen * RT_LANES]; uint32_t cIdx = sIdx[chosen * RT_LANES];
        float invProbability = useProbability ? s_div(randomRange, cInt) : 1.0f;
        sInt[chosen * RT_LANES] = 0.0f; randomRange -= cInt;
        result = result + compute_light<CACHED>(P, env, px, py, cIdx, rayDirection, m, position, normal, specular, checkShadows) * invProbability;
    }
    return result;
}

DEV f3 perpendicular_vector(f3 u) {                                              // Random.hlsli:41-48
    f3 a = mk3(fabsf(u.x), fabsf(u.y), fabsf(u.z));
    uint32_t xm = ((a.x - a.y) < 0.0f && (a.x - a.z) < 0.0f) ? 1u : 0u;
    uint32_t ym = (a.y - a.z) < 0.0f ? (1u ^ xm) : 0u;
    uint32_t zm = 1u ^ (xm | ym);
    return cross3(u, mk3((float)xm, (float)ym, (float)zm));
}
DEV f3 cos_hemisphere_blue_noise(PRef P, uint32_t px, uint32_t py, uint32_t frame, f3 hitNorm) {   // IndirectRayGen.hlsl:18-29
    f3 bn = blue_noise(P, px, py, frame);
    f3 bitangent = perpendicular_vector(hitNorm);
    f3 tangent = cross3(bitangent, hitNorm);
    float r = sqrtf(bn.x);
    float sn, cs;                               // phi = 2 pi bn.y: sine and cosine by direction spec D1 (device_math.h) -- the same bits as the scalar reference tracer's
    sincos_turns(bn.y, sn, cs);
    return (tangent * (r * cs) + bitangent * (r * sn)) + hitNorm * sqrtf(fmaxf(0.0f, 1.0f - bn.x));
}

DEV f2 world_to_screen(const float *viewProj, f3 p) {                            // PrimaryRayGen.hlsl:19-23
    f4 clip = mul4(viewProj, mk4(p.x, p.y, p.z, 1.0f));
    const float rw = s_rcp(clip.w);
    f2 r; r.x = 0.5f + (clip.x * rw) * 0.5f; r.y = 0.5f + (clip.y * rw) * 0.5f; return r;
}
DEV float fresnel_reflect_amount(f3 normal, f3 incident, float reflectivity, float fresnelMultiplier) {   // :25-29
    float ret = s_pow(clampf(1.0f + dot3(normal, incident), RT_EPSILON, 1.0f), 5.0f);
    return reflectivity + ((1.0f - reflectivity) * ret * fresnelMultiplier);
}
DEV void primary_ray(PRef P, uint32_t px, uint32_t py, f3 &origin, f3 &dir, f2 &d) {          // :33-39
    d.x = (((float)px + 0.5f + P.pixelJitter[0]) / (float)P.width) * 2.0f - 1.0f;
    d.y = (((float)py + 0.5f + P.pixelJitter[1]) / (float)P.height) * 2.0f - 1.0f;
    f4 target = mul4(cmat(P.projectionI).m, mk4(d.x, -d.y, 1.0f, 1.0f));
    const Mat16c viewI = cmat(P.viewI);
    origin = mul_point(viewI.m, mk3s(0.0f));
    dir = mul_vector(viewI.m, xyz(target));
}
