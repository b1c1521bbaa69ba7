/* oracle_shade.c -- per-hit and per-light shading math of the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h).
 *
 * Line-by-line restatement of
 *   - the colour-combiner decoder and the runtime-generated any-hit programs, ref:private/rt64_shader.cpp:32-96
 *     (ColorCombinerParams, VertexLayout), :156-226 (getVertexData), :228-310 (colour/alpha formulas),
 *     :444-592 (surface any-hit), :594-674 (shadow any-hit);
 *   - ref:shaders/Ray.hlsli:37-94 (Igehy ray differentials), ref:shaders/Random.hlsli:14-64,
 *     ref:shaders/BlueNoise.hlsli:7-13, ref:shaders/Lights.hlsli:27-168, ref:shaders/BgSky.hlsli:14-93,
 *     ref:shaders/Fog.hlsli:5-27, ref:shaders/Color.hlsli:9-43.
 * The generator is evaluated as an interpreter over the decoded combiner, which is equivalent to expanding it
 * per shaderId (SURVEY appendix A3 expands 0x01200a00 by hand).
 */
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "oracle_shade.h"

/* ---- combiner decode, ref:rt64_shader.cpp:32-96 ---------------------------------------------------------- */

void ocombiner_decode(uint32_t shaderId, OCombiner *cc) {
    memset(cc, 0, sizeof(*cc));
    for (int i = 0; i < 4; i++) {
        cc->c[0][i] = (int)((shaderId >> (i * 3)) & 7);
        cc->c[1][i] = (int)((shaderId >> (12 + i * 3)) & 7);
    }
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 4; j++) {
            int v = cc->c[i][j];
            if (v >= 1 && v <= 4 && v > cc->inputCount) cc->inputCount = v;
            if (v == 5 || v == 6) cc->useTextures[0] = 1;
            if (v == 7) cc->useTextures[1] = 1;
        }
    for (int i = 0; i < 2; i++) {
        cc->do_single[i] = cc->c[i][2] == 0;
        cc->do_multiply[i] = cc->c[i][1] == 0 && cc->c[i][3] == 0;
        cc->do_mix[i] = cc->c[i][1] == cc->c[i][3];
    }
    cc->color_alpha_same = (shaderId & 0xfff) == ((shaderId >> 12) & 0xfff);
    cc->opt_alpha = (shaderId & (1u << 24)) != 0;
    cc->opt_texture_edge = (shaderId & (1u << 26)) != 0;
    cc->opt_noise = (shaderId & (1u << 27)) != 0;
    cc->vertexUV = cc->useTextures[0] || cc->useTextures[1];
    /* VertexLayout(true, true, vertexUV, inputCount, opt_alpha), ref:rt64_shader.cpp:87-95 */
    int sz = 0;
    cc->positionOffset = sz; sz += 16;
    cc->normalOffset = sz; sz += 12;
    cc->uvOffset = sz; if (cc->vertexUV) sz += 8;
    for (int i = 0; i < cc->inputCount; i++) { cc->inputOffset[i] = sz; sz += cc->opt_alpha ? 16 : 12; }
    cc->vertexSize = sz;
}

void oracle_decode_combiner(uint32_t shaderId, int out[28]) {
    OCombiner cc; ocombiner_decode(shaderId, &cc);
    for (int i = 0; i < 4; i++) { out[i] = cc.c[0][i]; out[4 + i] = cc.c[1][i]; }
    out[8] = cc.inputCount; out[9] = cc.useTextures[0]; out[10] = cc.useTextures[1];
    out[11] = cc.do_single[0]; out[12] = cc.do_single[1]; out[13] = cc.do_multiply[0]; out[14] = cc.do_multiply[1];
    out[15] = cc.do_mix[0]; out[16] = cc.do_mix[1]; out[17] = cc.color_alpha_same; out[18] = cc.opt_alpha;
    out[19] = cc.opt_texture_edge; out[20] = cc.opt_noise; out[21] = cc.vertexSize; out[22] = cc.normalOffset;
    out[23] = cc.uvOffset; for (int i = 0; i < 4; i++) out[24 + i] = cc.inputOffset[i];
}

/* ---- RNG, ref:shaders/Random.hlsli:14-37 ------------------------------------------------------------------ */

uint32_t oracle_init_rand(uint32_t val0, uint32_t val1, uint32_t backoff) {
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (uint32_t n = 0; n < backoff; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

float oracle_next_rand(uint32_t *s) {
    *s = 1664525u * (*s) + 1013904223u;
    return (float)(*s & 0x00FFFFFFu) / (float)0x01000000;
}

float oracle_halton(int i, int b) {           /* ref:private/rt64_common.h:347-357 */
    float f = 1.0f, r = 0.0f;
    while (i > 0) { f = f / (float)b; r = r + f * (float)(i % b); i = i / b; }
    return r;
}

/* ---- colour, ref:shaders/Color.hlsli ------------------------------------------------------------------------ */

static of3 hue_to_rgb(float hue) {
    of3 rgb = v3(fabsf(hue * 6.0f - 3.0f) * 1.0f + -1.0f, fabsf(hue * 6.0f - 2.0f) * -1.0f + 2.0f, fabsf(hue * 6.0f - 4.0f) * -1.0f + 2.0f);
    return v3(fsaturate(rgb.x), fsaturate(rgb.y), fsaturate(rgb.z));
}
static of3 rgb_to_hcv(of3 rgb) {
    const float EPS = 1e-10f;
    float p[4], q[4];
    if (rgb.y < rgb.z) { p[0] = rgb.z; p[1] = rgb.y; p[2] = -1.0f; p[3] = 2.0f / 3.0f; }
    else { p[0] = rgb.y; p[1] = rgb.z; p[2] = 0.0f; p[3] = -1.0f / 3.0f; }
    if (rgb.x < p[0]) { q[0] = p[0]; q[1] = p[1]; q[2] = p[3]; q[3] = rgb.x; }
    else { q[0] = rgb.x; q[1] = p[1]; q[2] = p[2]; q[3] = p[0]; }
    float c = q[0] - fminf(q[3], q[1]);
    float h = fabsf((q[3] - q[1]) / (6.0f * c + EPS) + q[2]);
    return v3(h, c, q[0]);
}
void oracle_hsl_to_rgb(const float hsl[3], float out[3]) {
    of3 rgb = hue_to_rgb(hsl[0]);
    float c = (1.0f - fabsf(2.0f * hsl[2] - 1.0f)) * hsl[1];
    out[0] = (rgb.x - 0.5f) * c + hsl[2]; out[1] = (rgb.y - 0.5f) * c + hsl[2]; out[2] = (rgb.z - 0.5f) * c + hsl[2];
}
void oracle_rgb_to_hsl(const float rgb[3], float out[3]) {
    const float EPS = 1e-10f;
    of3 hcv = rgb_to_hcv(v3(rgb[0], rgb[1], rgb[2]));
    float z = hcv.z - hcv.y * 0.5f;
    float s = hcv.y / (1.0f - fabsf(z * 2.0f - 1.0f) + EPS);
    out[0] = hcv.x; out[1] = s; out[2] = z;
}
static of3 mod_rgb_with_hsl(of3 rgb, of3 mod) {   /* ref:Color.hlsli:41-43 */
    float in[3] = { rgb.x, rgb.y, rgb.z }, hsl[3], out[3];
    oracle_rgb_to_hsl(in, hsl);
    hsl[0] += mod.x; hsl[1] += mod.y; hsl[2] += mod.z;
    oracle_hsl_to_rgb(hsl, out);
    return v3(fsaturate(out[0]), fsaturate(out[1]), fsaturate(out[2]));
}

/* ---- sky / background, ref:shaders/BgSky.hlsli -------------------------------------------------------------- */

void oracle_fake_envmap_uv(const float d[3], float yawOffset, float uv[2]) {   /* :14-18 */
    float yaw = hlsl_fmod(yawOffset + atan2f(d[0], -d[2]) + O_PI, O_TWO_PI);
    float pitch = hlsl_fmod(atan2f(-d[1], sqrtf(d[0] * d[0] + d[2] * d[2])) + O_PI, O_TWO_PI);
    uv[0] = yaw / O_TWO_PI; uv[1] = pitch / O_TWO_PI;
}

static of2 sky_plane_uv(of2 uv, const om4 *viewI, of2 viewportSz, float yawOffset) {   /* :20-52 */
    const float SCREEN_WIDTH = 320.0f, SCREEN_HEIGHT = 240.0f;
    const float SKYBOX_WIDTH = 4.0f * SCREEN_WIDTH, SKYBOX_HEIGHT = 4.0f * SCREEN_HEIGHT;
    of2 base = { 0.0f, 0.0f };
    of3 viewDirection = v3normalize(m4_vector(viewI, v3(0.0f, 0.0f, 1.0f)));
    float skyYawRadians = hlsl_fmod(yawOffset + atan2f(viewDirection.x, -viewDirection.z) + O_PI, O_TWO_PI);
    base.x = SCREEN_WIDTH * 360.0f * (skyYawRadians - O_PI) / (90.0f * O_PI * 2.0f);
    float skyPitchRadians = atan2f(-viewDirection.y, sqrtf(viewDirection.x * viewDirection.x + viewDirection.z * viewDirection.z));
    float pitchInDegrees = skyPitchRadians * 360.0f / (O_PI * 2.0f);
    float degreesToScale = 360.0f * pitchInDegrees / 90.0f;
    base.y = degreesToScale + 5.0f * (SCREEN_HEIGHT / 2.0f);
    base.y = fclampf(base.y, SCREEN_HEIGHT, SKYBOX_HEIGHT);
    float aspectRatio = viewportSz.x / viewportSz.y;
    base.x += SCREEN_WIDTH / 2.0f;
    base.x -= (SCREEN_HEIGHT * aspectRatio) / 2.0f;
    base.x /= SKYBOX_WIDTH;
    base.y = (SKYBOX_HEIGHT - base.y) / SKYBOX_HEIGHT;
    float ratioDivision = aspectRatio / (4.0f / 3.0f);
    base.x += uv.x * (0.25f * ratioDivision);     /* HLSL: uv.x * 0.25f * ratioDivision; the view-only factor is folded once per frame */
    base.y += uv.y * 0.25f;
    return base;
}

static of4 sky_finish(const OShadeCtx *c, float tex[4]) {
    of4 sky = { tex[0] * c->desc.skyDiffuseMultiplier.x, tex[1] * c->desc.skyDiffuseMultiplier.y, tex[2] * c->desc.skyDiffuseMultiplier.z, tex[3] };
    of3 m = c->desc.skyHSLModifier;
    if (m.x != 0.0f || m.y != 0.0f || m.z != 0.0f) {
        of3 r = mod_rgb_with_hsl(v3(sky.x, sky.y, sky.z), m);
        sky.x = r.x; sky.y = r.y; sky.z = r.z;
    }
    return sky;
}

of4 oshade_sample_sky_2d(const OShadeCtx *c, of2 screenUV) {       /* SampleSky2D :54-70 */
    of4 zero = { 0, 0, 0, 0 };
    if (!c->sky) return zero;
    of2 vp = { c->viewportW, c->viewportH };
    of2 uv = sky_plane_uv(screenUV, &c->viewI, vp, c->desc.skyYawOffset);
    float tex[4];
    otex_sample_level(c->sky, uv.x, uv.y, 0, 1, 0, 0, tex);
    return sky_finish(c, tex);
}

of4 oshade_sample_sky_plane(const OShadeCtx *c, of3 rayDirection) { /* SampleSkyPlane :72-87 */
    of4 zero = { 0, 0, 0, 0 };
    if (!c->sky) return zero;
    float d[3] = { rayDirection.x, rayDirection.y, rayDirection.z }, uv[2], tex[4];
    oracle_fake_envmap_uv(d, c->desc.skyYawOffset, uv);
    otex_sample_level(c->sky, uv[0], uv[1], 0, 1, 0, 0, tex);
    return sky_finish(c, tex);
}

/* gBackground is the raster-background render target (ref:rt64_view.cpp:1296-1319).  The oracle keeps it as an RGBA8
 * image of the frame size (NULL = cleared to transparent black). */
of3 oshade_sample_background_2d(const OShadeCtx *c, of2 screenUV) {
    if (!c->background) return v3s(0.0f);
    float tex[4];
    otex_sample_level(c->background, screenUV.x, screenUV.y, 0, 1, 0, 0, tex);
    return v3(tex[0], tex[1], tex[2]);
}
of3 oshade_sample_background_envmap(const OShadeCtx *c, of3 rayDirection) {
    if (!c->background) return v3s(0.0f);
    float d[3] = { rayDirection.x, rayDirection.y, rayDirection.z }, uv[2], tex[4];
    oracle_fake_envmap_uv(d, 0.0f, uv);
    otex_sample_level(c->background, uv[0], uv[1], 0, 1, 0, 0, tex);
    return v3(tex[0], tex[1], tex[2]);
}

/* ---- fog, ref:shaders/Fog.hlsli ------------------------------------------------------------------------------ */

of4 oshade_fog_from_camera(const OShadeCtx *c, uint32_t instanceId, of3 position) {     /* :5-18 */
    const OMaterial *m = &c->rt[instanceId].desc.material;
    of4 p = { position.x, position.y, position.z, 1.0f };
    of4 clip = m4_mul_vec(&c->viewProj, p);
    clip.z = clip.z * 2.0f - clip.w;
    float winv = 1.0f / fmaxf(clip.w, 0.001f);
    of4 fog = { m->fogColor.x, m->fogColor.y, m->fogColor.z, 0.0f };
    fog.w = fclampf((clip.z * winv * m->fogMul + m->fogOffset) / 255.0f, 0.0f, 1.0f);
    return fog;
}

of4 oshade_fog_from_origin(const OShadeCtx *c, uint32_t instanceId, of3 position, of3 origin) {   /* :20-27 */
    const OMaterial *m = &c->rt[instanceId].desc.material;
    of4 fog = { m->fogColor.x, m->fogColor.y, m->fogColor.z, 0.0f };
    float distance = v3len(v3sub(position, origin));
    fog.w = fclampf(((distance + m->fogOffset) / m->fogMul) * 0.5f, 0.0f, 1.0f);
    return fog;
}

/* ---- blue noise, ref:shaders/BlueNoise.hlsli:7-13 ------------------------------------------------------------ */

of3 oshade_blue_noise(const OShadeCtx *c, uint32_t px, uint32_t py, uint32_t frame) {
    uint32_t f = frame % 64u;
    uint32_t bx = (f % 8u) * 64u + px % 64u, by = (f / 8u) * 64u + py % 64u;
    const uint8_t *p = c->blueNoise + ((size_t)by * 512u + bx) * 4u;
    return v3((float)p[0] * (1.0f / 255.0f), (float)p[1] * (1.0f / 255.0f), (float)p[2] * (1.0f / 255.0f));
}

/* ---- ray differentials, ref:shaders/Ray.hlsli:37-94 ---------------------------------------------------------- */

void oshade_compute_ray_diffs(of3 nonNormDir, of3 right, of3 up, of2 viewportDims, of3 *dDdx, of3 *dDdy) {
    float dd = v3dot(nonNormDir, nonNormDir);
    float divd = 2.0f / (dd * sqrtf(dd));
    float dr = v3dot(nonNormDir, right), du = v3dot(nonNormDir, up);
    *dDdx = v3scale(v3scale(v3sub(v3scale(right, dd), v3scale(nonNormDir, dr)), divd), 1.0f / viewportDims.x);
    *dDdy = v3neg(v3scale(v3scale(v3sub(v3scale(up, dd), v3scale(nonNormDir, du)), divd), 1.0f / viewportDims.y));
}

static ORayDiff propagate_ray_diffs(ORayDiff rd, of3 D, float t, of3 N) {
    of3 dodx = v3add(rd.dOdx, v3scale(rd.dDdx, t));
    of3 dody = v3add(rd.dOdy, v3scale(rd.dDdy, t));
    float rcpDN = 1.0f / v3dot(D, N);
    float dtdx = -v3dot(dodx, N) * rcpDN, dtdy = -v3dot(dody, N) * rcpDN;
    ORayDiff out = rd;
    out.dOdx = v3add(dodx, v3scale(D, dtdx));
    out.dOdy = v3add(dody, v3scale(D, dtdy));
    return out;
}

static void barycentric_differentials(ORayDiff rd, of3 e01, of3 e02, of3 faceN, of2 *dBdx, of2 *dBdy) {
    of3 Nu = v3cross(e02, faceN), Nv = v3cross(e01, faceN);
    float du = v3dot(Nu, e01), dv = v3dot(Nv, e02);
    of3 Lu = v3(Nu.x / du, Nu.y / du, Nu.z / du), Lv = v3(Nv.x / dv, Nv.y / dv, Nv.z / dv);
    dBdx->x = v3dot(Lu, rd.dOdx); dBdx->y = v3dot(Lv, rd.dOdx);
    dBdy->x = v3dot(Lu, rd.dOdy); dBdy->y = v3dot(Lv, rd.dOdy);
}

/* ---- vertex fetch + combiner evaluation ----------------------------------------------------------------------- */

typedef struct {
    of3 pos[3], posW[3], norm[3];
    of2 uv[3];
    of4 input[4];                          /* interpolated input1..4 */
    of3 vertexPosition, vertexNormal, triangleNormal, vertexTangent, vertexBinormal;
    of2 vertexUV;
} VertexData;

static of3 ld3(const uint8_t *p) { of3 r; memcpy(&r, p, 12); return r; }

static void get_vertex_data(const OInst *in, uint32_t prim, const float b[3], int wantTangent, VertexData *vd) {   /* :156-226 */
    const OMesh *mesh = in->desc.mesh;
    const OCombiner *cc = &in->cc;
    const uint8_t *vp[3];
    for (int k = 0; k < 3; k++) vp[k] = mesh->vertices + (size_t)mesh->indices[3 * prim + k] * (size_t)cc->vertexSize;
    for (int k = 0; k < 3; k++) {
        vd->pos[k] = ld3(vp[k] + cc->positionOffset);
        vd->posW[k] = m4_point(&in->objectToWorld, vd->pos[k]);
        vd->norm[k] = ld3(vp[k] + cc->normalOffset);
    }
    vd->vertexPosition = v3add(v3add(v3scale(vd->pos[0], b[0]), v3scale(vd->pos[1], b[1])), v3scale(vd->pos[2], b[2]));
    of3 vn = v3add(v3add(v3scale(vd->norm[0], b[0]), v3scale(vd->norm[1], b[1])), v3scale(vd->norm[2], b[2]));
    of3 tn = v3neg(v3cross(v3sub(vd->pos[2], vd->pos[0]), v3sub(vd->pos[1], vd->pos[0])));
    vd->vertexNormal = (vn.x != 0.0f || vn.y != 0.0f || vn.z != 0.0f) ? v3normalize(vn) : tn;
    vd->triangleNormal = v3normalize(m4_vector(&in->objectToWorldNormal, tn));
    if (cc->vertexUV) {
        for (int k = 0; k < 3; k++) memcpy(&vd->uv[k], vp[k] + cc->uvOffset, 8);
        vd->vertexUV.x = vd->uv[0].x * b[0] + vd->uv[1].x * b[1] + vd->uv[2].x * b[2];
        vd->vertexUV.y = vd->uv[0].y * b[0] + vd->uv[1].y * b[1] + vd->uv[2].y * b[2];
    }
    for (int i = 0; i < cc->inputCount; i++) {
        float in3[3][4];
        for (int k = 0; k < 3; k++) {
            in3[k][3] = 1.0f;
            memcpy(in3[k], vp[k] + cc->inputOffset[i], cc->opt_alpha ? 16 : 12);
        }
        float r[4];
        for (int ch = 0; ch < 4; ch++) r[ch] = in3[0][ch] * b[0] + in3[1][ch] * b[1] + in3[2][ch] * b[2];
        if (!cc->opt_alpha) r[3] = 1.0f;
        vd->input[i].x = r[0]; vd->input[i].y = r[1]; vd->input[i].z = r[2]; vd->input[i].w = r[3];
    }
    if (wantTangent) {                                                            /* :201-225 */
        float uva = vd->uv[1].x - vd->uv[0].x, uvb = vd->uv[2].x - vd->uv[0].x;
        float uvc = vd->uv[1].y - vd->uv[0].y, uvd = vd->uv[2].y - vd->uv[0].y;
        float uvk = uvb * uvc - uva * uvd;
        of3 dpos1 = v3sub(vd->pos[1], vd->pos[0]), dpos2 = v3sub(vd->pos[2], vd->pos[0]);
        of3 tangent;
        if (uvk != 0.0f) { of3 n = v3sub(v3scale(dpos2, uvc), v3scale(dpos1, uvd)); tangent = v3normalize(v3(n.x / uvk, n.y / uvk, n.z / uvk)); }
        else if (uva != 0.0f) tangent = v3normalize(v3(dpos1.x / uva, dpos1.y / uva, dpos1.z / uva));
        else if (uvb != 0.0f) tangent = v3normalize(v3(dpos2.x / uvb, dpos2.y / uvb, dpos2.z / uvb));
        else tangent = v3s(0.0f);
        of2 duv1 = { vd->uv[1].x - vd->uv[0].x, -(vd->uv[1].y - vd->uv[0].y) };
        of2 duv2 = { vd->uv[2].x - vd->uv[1].x, -(vd->uv[2].y - vd->uv[1].y) };
        float crz = duv1.x * duv2.y - duv1.y * duv2.x;
        float binormalMult = (crz < 0.0f) ? -1.0f : 1.0f;
        vd->vertexTangent = tangent;
        vd->vertexBinormal = v3scale(v3cross(tangent, vd->vertexNormal), binormalMult);
    }
}

static of4 color_input(int item, int with_alpha, int inputs_have_alpha, int hint_single, const VertexData *vd, of4 t0, of4 t1) {   /* :228-258 */
    of4 r;
    switch (item) {
    default: case 0: r.x = r.y = r.z = 0.0f; r.w = with_alpha ? 0.0f : 1.0f; return r;
    case 1: case 2: case 3: case 4:
        r = vd->input[item - 1];
        if (!(with_alpha || !inputs_have_alpha)) r.w = 1.0f;
        return r;
    case 5: r = t0; if (!with_alpha) r.w = 1.0f; return r;
    case 6: r.x = r.y = r.z = t0.w; r.w = (hint_single || with_alpha) ? t0.w : 1.0f; return r;
    case 7: r = t1; if (!with_alpha) r.w = 1.0f; return r;
    }
}

static of4 color_formula(const OCombiner *cc, int with_alpha, int opt_alpha, const VertexData *vd, of4 t0, of4 t1) {   /* :260-273 */
    const int *c = cc->c[0];
    of4 r;
    if (cc->do_single[0]) return color_input(c[3], with_alpha, opt_alpha, 0, vd, t0, t1);
    if (cc->do_multiply[0]) {
        of4 a = color_input(c[0], with_alpha, opt_alpha, 0, vd, t0, t1), b = color_input(c[2], with_alpha, opt_alpha, 1, vd, t0, t1);
        r.x = a.x * b.x; r.y = a.y * b.y; r.z = a.z * b.z; r.w = a.w * b.w; return r;
    }
    if (cc->do_mix[0]) {
        of4 x = color_input(c[1], with_alpha, opt_alpha, 0, vd, t0, t1), y = color_input(c[0], with_alpha, opt_alpha, 0, vd, t0, t1);
        of4 s = color_input(c[2], with_alpha, opt_alpha, 1, vd, t0, t1);
        r.x = flerp(x.x, y.x, s.x); r.y = flerp(x.y, y.y, s.y); r.z = flerp(x.z, y.z, s.z); r.w = flerp(x.w, y.w, s.w); return r;
    }
    of4 a = color_input(c[0], with_alpha, opt_alpha, 0, vd, t0, t1), b = color_input(c[1], with_alpha, opt_alpha, 0, vd, t0, t1);
    of4 s = color_input(c[2], with_alpha, opt_alpha, 1, vd, t0, t1), d = color_input(c[3], with_alpha, opt_alpha, 0, vd, t0, t1);
    r.x = (a.x - b.x) * s.x + d.x; r.y = (a.y - b.y) * s.x + d.y; r.z = (a.z - b.z) * s.x + d.z; r.w = (a.w - b.w) * s.x + d.w;
    return r;
}

static float alpha_input(int item, const VertexData *vd, of4 t0, of4 t1) {   /* :275-295 */
    switch (item) {
    default: case 0: return 0.0f;
    case 1: case 2: case 3: case 4: return vd->input[item - 1].w;
    case 5: case 6: return t0.w;
    case 7: return t1.w;
    }
}

static float alpha_formula(const OCombiner *cc, const VertexData *vd, of4 t0, of4 t1) {   /* :297-310 */
    const int *c = cc->c[1];
    if (cc->do_single[1]) return alpha_input(c[3], vd, t0, t1);
    if (cc->do_multiply[1]) return alpha_input(c[0], vd, t0, t1) * alpha_input(c[2], vd, t0, t1);
    if (cc->do_mix[1]) return flerp(alpha_input(c[1], vd, t0, t1), alpha_input(c[0], vd, t0, t1), alpha_input(c[2], vd, t0, t1));
    return (alpha_input(c[0], vd, t0, t1) - alpha_input(c[1], vd, t0, t1)) * alpha_input(c[2], vd, t0, t1) + alpha_input(c[3], vd, t0, t1);
}

/* Pixel shader of the raster pipeline, ref:private/rt64_shader.cpp:363-387: texVal1 is the constant (1,0,1,1) there. */
void oshade_raster_pixel(const OCombiner *cc, const of4 inputs[4], of4 texVal0, float out[4]) {
    VertexData vd; memset(&vd, 0, sizeof(vd));
    for (int i = 0; i < 4; i++) vd.input[i] = inputs[i];
    of4 t1 = { 1.0f, 0.0f, 1.0f, 1.0f }, result;
    if (!cc->color_alpha_same && cc->opt_alpha) { result = color_formula(cc, 0, 1, &vd, texVal0, t1); result.w = alpha_formula(cc, &vd, texVal0, t1); }
    else result = color_formula(cc, cc->opt_alpha, cc->opt_alpha, &vd, texVal0, t1);
    out[0] = result.x; out[1] = result.y; out[2] = result.z; out[3] = result.w;
}

/* ---- surface any-hit, ref:rt64_shader.cpp:444-581 -------------------------------------------------------------- */

int oshade_surface_anyhit(const OShadeCtx *c, const OHit *hit, of3 rayDirW, ORayDiff payloadDiff, uint32_t px, uint32_t py, OHitRecord *rec) {
    const OInst *in = &c->rt[hit->instance];
    const OCombiner *cc = &in->cc;
    const OMaterial *mat = &in->desc.material;
    int normalMap = (in->desc.shaderFlags & 0x4) != 0, specularMap = (in->desc.shaderFlags & 0x8) != 0;
    float b[3] = { 1.0f - hit->u - hit->v, hit->u, hit->v };
    of4 mix = mat->diffuseColorMix;
    VertexData vd; memset(&vd, 0, sizeof(vd));
    get_vertex_data(in, hit->prim, b, cc->vertexUV && normalMap, &vd);

    of2 ddx = { 0, 0 }, ddy = { 0, 0 };
    of4 t0 = { 0, 0, 0, 0 }, t1 = { 1.0f, 0.0f, 1.0f, 1.0f };
    int filter = (int)in->desc.filter, hA = (int)in->desc.hAddr, vA = (int)in->desc.vAddr;
    if (cc->useTextures[0]) {
        ORayDiff prd = propagate_ray_diffs(payloadDiff, rayDirW, hit->t, vd.triangleNormal);
        of2 dBdx, dBdy;
        barycentric_differentials(prd, v3sub(vd.posW[1], vd.posW[0]), v3sub(vd.posW[2], vd.posW[0]), vd.triangleNormal, &dBdx, &dBdy);
        of2 uv01 = { vd.uv[1].x - vd.uv[0].x, vd.uv[1].y - vd.uv[0].y }, uv02 = { vd.uv[2].x - vd.uv[0].x, vd.uv[2].y - vd.uv[0].y };
        ddx.x = dBdx.x * uv01.x + dBdx.y * uv02.x; ddx.y = dBdx.x * uv01.y + dBdx.y * uv02.y;
        ddy.x = dBdy.x * uv01.x + dBdy.y * uv02.x; ddy.y = dBdy.x * uv01.y + dBdy.y * uv02.y;
        float tex[4];
        otex_sample_grad(in->desc.diffuse, vd.vertexUV.x, vd.vertexUV.y, ddx, ddy, filter, hA, vA, tex);
        float k = fmaxf(-mix.w, 0.0f);
        t0.x = flerp(tex[0], mix.x, k); t0.y = flerp(tex[1], mix.y, k); t0.z = flerp(tex[2], mix.z, k); t0.w = tex[3];
    }
    of4 result;
    if (!cc->color_alpha_same && cc->opt_alpha) {
        result = color_formula(cc, 0, 1, &vd, t0, t1);
        result.w = alpha_formula(cc, &vd, t0, t1);
    }
    else result = color_formula(cc, cc->opt_alpha, cc->opt_alpha, &vd, t0, t1);
    {
        float k = fmaxf(mix.w, 0.0f);
        result.x = flerp(result.x, mix.x, k); result.y = flerp(result.y, mix.y, k); result.z = flerp(result.z, mix.z, k);
    }
    result.w = fclampf(mat->solidAlphaMultiplier * result.w, 0.0f, 1.0f);
    if (cc->opt_texture_edge) {                                     /* TEXTURE_EDGE_ENABLED, :502-511 */
        if (result.w > 0.3f) result.w = 1.0f; else return 0;      /* IgnoreHit() before anything is stored */
    }
    if (cc->opt_noise) {                                            /* :513-516 */
        uint32_t seed = oracle_init_rand(px + py * (uint32_t)c->width, c->frameCount, 16);
        result.w *= nearbyintf(oracle_next_rand(&seed));
    }
    of3 vertexNormal = v3normalize(m4_vector(&in->objectToWorldNormal, vd.vertexNormal));
    float normalSign = (v3dot(vd.triangleNormal, rayDirW) <= 0.0f) ? 1.0f : -1.0f;
    vertexNormal = v3scale(vertexNormal, normalSign);
    if (cc->vertexUV && normalMap) {                                /* :522-533 */
        of3 tangent = v3scale(v3normalize(m4_vector(&in->objectToWorldNormal, vd.vertexTangent)), normalSign);
        of3 binormal = v3scale(v3normalize(m4_vector(&in->objectToWorldNormal, vd.vertexBinormal)), normalSign);
        if (in->desc.normal) {
            float s = mat->uvDetailScale, tex[4];
            of2 gx = { ddx.x * s, ddx.y * s }, gy = { ddy.x * s, ddy.y * s };
            otex_sample_grad(in->desc.normal, vd.vertexUV.x * s, vd.vertexUV.y * s, gx, gy, filter, hA, vA, tex);
            of3 nc = v3(tex[0] * 2.0f - 1.0f, tex[1] * 2.0f - 1.0f, tex[2] * 2.0f - 1.0f);
            vertexNormal = v3normalize(v3add(v3add(v3scale(vertexNormal, nc.z), v3scale(tangent, nc.x)), v3scale(binormal, nc.y)));
        }
    }
    of3 prevWorldPos = m4_point(&in->objectToWorldPrevious, vd.vertexPosition);
    of3 curWorldPos = m4_point(&in->objectToWorld, vd.vertexPosition);
    of3 vertexFlow = v3sub(curWorldPos, prevWorldPos);
    of3 vertexSpecular = v3s(1.0f);
    if (cc->vertexUV && specularMap && in->desc.specular) {        /* :539-545 */
        float s = mat->uvDetailScale, tex[4];
        of2 gx = { ddx.x * s, ddx.y * s }, gy = { ddy.x * s, ddy.y * s };
        otex_sample_grad(in->desc.specular, vd.vertexUV.x * s, vd.vertexUV.y * s, gx, gy, filter, hA, vA, tex);
        vertexSpecular = v3(tex[0], tex[1], tex[2]);
    }
    rec->dist = hit->t - mat->depthBias;                            /* WithDistanceBias, ref:shaders/Instances.hlsli:17-19 */
    rec->flow = vertexFlow;
    rec->color[0] = to_unorm8(result.x); rec->color[1] = to_unorm8(result.y); rec->color[2] = to_unorm8(result.z); rec->color[3] = to_unorm8(result.w);
    rec->normal[0] = to_snorm16(vertexNormal.x); rec->normal[1] = to_snorm16(vertexNormal.y); rec->normal[2] = to_snorm16(vertexNormal.z); rec->normal[3] = to_snorm16(1.0f);
    rec->specular[0] = to_unorm8(vertexSpecular.x); rec->specular[1] = to_unorm8(vertexSpecular.y); rec->specular[2] = to_unorm8(vertexSpecular.z); rec->specular[3] = 255;
    rec->instanceId = (uint16_t)hit->instance;
    rec->geo = *hit;
    return 1;
}

/* ---- shadow any-hit, ref:rt64_shader.cpp:594-663.  Returns the alpha to subtract from payload.shadowHit,
 *      or a negative value when the candidate is ignored (texture edge). ---------------------------------------- */

float oshade_shadow_anyhit_alpha(const OShadeCtx *c, const OHit *hit, uint32_t px, uint32_t py) {
    const OInst *in = &c->rt[hit->instance];
    const OCombiner *cc = &in->cc;
    const OMaterial *mat = &in->desc.material;
    if (in->shadowOpaque) return 2.0f;                              /* payload.shadowHit = 0 (:661) / rule O2: more than enough to saturate */
    float b[3] = { 1.0f - hit->u - hit->v, hit->u, hit->v };
    VertexData vd; memset(&vd, 0, sizeof(vd));
    get_vertex_data(in, hit->prim, b, 0, &vd);
    of4 t0 = { 0, 0, 0, 0 }, t1 = { 1.0f, 0.0f, 1.0f, 1.0f };
    if (cc->useTextures[0]) {
        float tex[4];
        otex_sample_level(in->desc.diffuse, vd.vertexUV.x, vd.vertexUV.y, 0, (int)in->desc.filter, (int)in->desc.hAddr, (int)in->desc.vAddr, tex);
        t0.x = tex[0]; t0.y = tex[1]; t0.z = tex[2]; t0.w = tex[3];
    }
    float a;
    if (!cc->color_alpha_same && cc->opt_alpha) a = alpha_formula(cc, &vd, t0, t1);
    else a = color_formula(cc, cc->opt_alpha, cc->opt_alpha, &vd, t0, t1).w;
    a = fclampf(a * mat->shadowAlphaMultiplier, 0.0f, 1.0f);
    if (cc->opt_texture_edge) { if (a > 0.3f) a = 1.0f; else return -1.0f; }
    if (cc->opt_noise) {
        uint32_t seed = oracle_init_rand(px + py * (uint32_t)c->width, c->frameCount, 16);
        a *= nearbyintf(oracle_next_rand(&seed));
    }
    return a;
}

/* ---- lights, ref:shaders/Lights.hlsli ----------------------------------------------------------------------------- */

typedef struct { const OShadeCtx *c; float shadowHit; uint32_t px, py; } ShadowPayload;

static int shadow_cb(void *user, const OHit *hit, float *tmax, int *terminate) {
    ShadowPayload *p = (ShadowPayload *)user;
    (void)tmax;
    float a = oshade_shadow_anyhit_alpha(p->c, hit, p->px, p->py);
    if (a < 0.0f) return 0;                                         /* IgnoreHit */
    p->shadowHit = fmaxf(p->shadowHit - a, 0.0f);
    if (p->shadowHit > 0.0f) return 0;                              /* IgnoreHit: keep searching */
    *terminate = 1;                                                 /* accepted + RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH */
    return 1;
}

float oshade_trace_shadow(OShadeCtx *c, of3 origin, of3 dir, float tmin, float tmax, uint32_t px, uint32_t py) {   /* :27-52 */
    ORay ray = { { origin.x, origin.y, origin.z }, { dir.x, dir.y, dir.z }, tmin, tmax, 0 };   /* SKIP_BACKFACE_SHADOWS undefined: no cull */
    ShadowPayload p = { c, 1.0f, px, py };
    OTraceCounters ctr = { 0, 0 };
    otrace(c->scene, &ray, c->bruteForce, shadow_cb, &p, &ctr);
    c->shadowRays++; c->nodesShadow += ctr.nodes; c->trisShadow += ctr.tris;
    return p.shadowHit;
}

static float light_intensity_simple(const OShadeCtx *c, uint32_t l, of3 position, of3 normal, float ignoreNormalFactor) {   /* :54-65 */
    const OLight *L = &c->lights[l];
    float lightDistance = v3len(v3sub(position, L->position));
    of3 lightDirection = v3normalize(v3sub(L->position, position));
    float NdotL = v3dot(normal, lightDirection);
    float surfaceBias = fmaxf(flerp(NdotL, 1.0f, ignoreNormalFactor) + 0.707106f, 0.0f);
    float f = powf(fmaxf(1.0f - (lightDistance / L->attenuationRadius), 0.0f), L->attenuationExponent) * surfaceBias;
    return f * (L->diffuseColor.x + L->diffuseColor.y + L->diffuseColor.z);
}

static of3 compute_light(OShadeCtx *c, uint32_t px, uint32_t py, uint32_t lightIndex, of3 rayDirection, uint32_t instanceId,
                         of3 position, of3 normal, of3 specular, int checkShadows) {   /* :67-113 */
    const OMaterial *m = &c->rt[instanceId].desc.material;
    const OLight *L = &c->lights[lightIndex];
    of3 lightDirection = v3normalize(v3sub(L->position, position));
    float lightPointRadius = (c->diSamples > 0) ? L->pointRadius : 0.0f;
    of3 perpX = v3cross(v3neg(lightDirection), v3(0.0f, 1.0f, 0.0f));
    if (perpX.x == 0.0f && perpX.y == 0.0f && perpX.z == 0.0f) perpX.x = 1.0f;
    of3 perpY = v3cross(perpX, v3neg(lightDirection));
    uint32_t maxSamples = c->diSamples > 1 ? c->diSamples : 1, samples = maxSamples;
    float lLambert = 0.0f, lShadow = 0.0f; of3 lSpec = v3s(0.0f);
    while (samples > 0) {
        of3 bn = oshade_blue_noise(c, px, py, c->frameCount + samples);
        of2 sc = { bn.x * 2.0f - 1.0f, bn.y * 2.0f - 1.0f };
        float len = sqrtf(sc.x * sc.x + sc.y * sc.y), sat = fsaturate(len);
        sc.x = sc.x / len * sat; sc.y = sc.y / len * sat;
        of3 samplePosition = v3add(v3add(L->position, v3scale(v3scale(perpX, sc.x), lightPointRadius)), v3scale(v3scale(perpY, sc.y), lightPointRadius));
        float sampleDistance = v3len(v3sub(position, samplePosition));
        of3 sampleDirection = v3normalize(v3sub(samplePosition, position));
        float sampleIntensityFactor = powf(fmaxf(1.0f - (sampleDistance / L->attenuationRadius), 0.0f), L->attenuationExponent);
        of3 reflectedLight = v3reflect(v3neg(sampleDirection), normal);
        float NdotL = fmaxf(v3dot(normal, sampleDirection), 0.0f);
        float sampleLambert = flerp(NdotL, 1.0f, m->ignoreNormalFactor) * sampleIntensityFactor;
        float sampleShadow = 1.0f;
        if (checkShadows)
            sampleShadow = oshade_trace_shadow(c, position, sampleDirection, O_RAY_MIN_DISTANCE + m->shadowRayBias, sampleDistance - L->shadowOffset, px, py);
        float sp = powf(fmaxf(fsaturate(v3dot(reflectedLight, v3neg(rayDirection)) * sampleIntensityFactor), 0.0f), m->specularExponent);
        lLambert += sampleLambert / (float)maxSamples;
        lSpec = v3add(lSpec, v3scale(v3scale(specular, sp), 1.0f / (float)maxSamples));
        lShadow += sampleShadow / (float)maxSamples;
        samples--;
    }
    of3 r = v3add(v3scale(L->diffuseColor, lLambert), v3mul(L->specularColor, lSpec));
    return v3scale(r, lShadow);
}

of3 oshade_lights_random(OShadeCtx *c, uint32_t px, uint32_t py, of3 rayDirection, uint32_t instanceId, of3 position, of3 normal,
                         of3 specular, uint32_t maxLightCount, int checkShadows) {   /* :115-168 */
    of3 result = v3s(0.0f);
    const OMaterial *m = &c->rt[instanceId].desc.material;
    if (m->lightGroupMaskBits == 0) return result;
    uint32_t sCount = 0, sIdx[O_MAX_LIGHTS + 1]; float sInt[O_MAX_LIGHTS + 1], total = 0.0f;
    for (uint32_t l = 0; l < (uint32_t)c->lightCount && sCount < O_MAX_LIGHTS; l++) {
        if (m->lightGroupMaskBits & c->lights[l].groupBits) {
            float li = light_intensity_simple(c, l, position, normal, m->ignoreNormalFactor);
            if (li > O_EPSILON) { sInt[sCount] = li; sIdx[sCount] = l; total += li; sCount++; }
        }
    }
    float randomRange = total;
    uint32_t lCount = sCount < maxLightCount ? sCount : maxLightCount;
    int useProbability = lCount == 1;
    for (uint32_t s = 0; s < lCount; s++) {
        float r = oshade_blue_noise(c, px, py, c->frameCount + s).x * randomRange;
        uint32_t chosen = 0; float rInt = sInt[chosen];
        while (chosen < sCount - 1 && r >= rInt) { chosen++; rInt += sInt[chosen]; }
        float cInt = sInt[chosen]; uint32_t cIdx = sIdx[chosen];
        float invProbability = useProbability ? (randomRange / cInt) : 1.0f;
        sInt[chosen] = 0.0f; randomRange -= cInt;
        result = v3add(result, v3scale(compute_light(c, px, py, cIdx, rayDirection, instanceId, position, normal, specular, checkShadows), invProbability));
    }
    return result;
}

/* ---- Random.hlsli:41-64 / IndirectRayGen.hlsl:18-29 ------------------------------------------------------------- */

static of3 perpendicular_vector(of3 u) {
    of3 a = v3(fabsf(u.x), fabsf(u.y), fabsf(u.z));
    uint32_t xm = ((a.x - a.y) < 0.0f && (a.x - a.z) < 0.0f) ? 1u : 0u;
    uint32_t ym = (a.y - a.z) < 0.0f ? (1u ^ xm) : 0u;
    uint32_t zm = 1u ^ (xm | ym);
    return v3cross(u, v3((float)xm, (float)ym, (float)zm));
}

/* Direction spec D1 (shared with the HIP side, csrc/device_math.h: sincos_turns): sine and cosine of 2 pi u for u in [0, 1] with every operation spelled
 * out, so that the angle of a bounce direction -- a value that defines a RAY -- has the same bits on both sides (libm's and the device library's sinf / cosf
 * are different code; the reference's HLSL sin / cos compile to approximate GPU instructions whose bits nothing specifies).
 *   a = 4 u; k = (int)(a + 0.5f); x = (a - k) * (pi / 2) in [-pi/4, pi/4]; z = x x;
 *   s = fma(x z, fma(z, fma(z, S3, S2), S1), x);  c = fma(z, fma(z, fma(z, fma(z, C4, C3), C2), -0.5), 1)   (Cephes sinf / cosf kernels)
 *   quadrant k & 3 rotates: 0 (s, c), 1 (c, -s), 2 (-s, -c), 3 (-c, s). */
static void sincos_turns(float u, float *sn, float *cs) {
    const float a = u * 4.0f;
    const int k = (int)(a + 0.5f);
    const float x = (a - (float)k) * 1.57079637f, z = x * x;
    const float s = fmaf(x * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), x);
    const float c = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), -0.5f), 1.0f);
    switch (k & 3) {
    case 0: *sn = s; *cs = c; break;
    case 1: *sn = c; *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
    }
}

of3 oshade_cos_hemisphere_blue_noise(const OShadeCtx *c, uint32_t px, uint32_t py, uint32_t frame, of3 hitNorm) {
    of3 bn = oshade_blue_noise(c, px, py, frame);
    of3 bitangent = perpendicular_vector(hitNorm);
    of3 tangent = v3cross(bitangent, hitNorm);
    float r = sqrtf(bn.x);
    float sn, cs;                               /* phi = 2 pi bn.y (IndirectRayGen.hlsl:24): by direction spec D1 */
    sincos_turns(bn.y, &sn, &cs);
    return v3add(v3add(v3scale(tangent, r * cs), v3scale(bitangent, r * sn)), v3scale(hitNorm, sqrtf(fmaxf(0.0f, 1.0f - bn.x))));
}
