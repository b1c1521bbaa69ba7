/* oracle_texture.c -- texel store + software sampler of the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h).
 *
 * Replaces: RT64_CreateTexture (ref:private/rt64_texture.cpp:28-187: RGBA8 upload, 1 mip because mip generation
 * is compiled out at ref:private/rt64_device.cpp:758-762; DDS via DDSTextureLoader keeps the file's own mips) and
 * the GPU sampler hardware behind SampleGrad / SampleLevel (ref:private/rt64_shader.cpp:480,528,543,622 and
 * ref:shaders/BgSky.hlsli:57,74).  Sampler states: MIN_MAG_MIP_{POINT,LINEAR} x {WRAP,MIRROR,CLAMP}^2, MaxAnisotropy 1,
 * LOD bias 0 (ref:private/rt64_view.cpp:699-722); ray-gen static sampler LINEAR/WRAP (ref:rt64_device.cpp:958-973).
 *
 * Texture spec (shared contract with the HIP sampler; everything except log2f is bit-reproducible):
 *   T1 texel value = byte * (1.0f / 255.0f)   (one multiply; 255 maps to exactly 1.0f).
 *   T2 LINEAR: x = u*w - 0.5f, x0 = floorf(x), fx = x - x0 (same for y); the four texels (x0|x0+1, y0|y0+1) addressed
 *      per axis by WRAP i mod w / MIRROR reflect with period 2w / CLAMP to [0, w-1];
 *      c = (c00 + fx*(c10 - c00)) + fy*((c01 + fx*(c11 - c01)) - (c00 + fx*(c10 - c00))).
 *      POINT: texel (floorf(u*w), floorf(v*h)) addressed the same way.
 *   T3 SampleGrad LOD (isotropic, D3D11.3 functional spec 7.18.11 without the allowed approximations):
 *      rho = max(|ddx * (w0,h0)|, |ddy * (w0,h0)|), lod = clamp(log2f(rho), 0, mips-1);
 *      LINEAR: l0 = floor(lod), blend levels l0 and min(l0+1, mips-1) by lod - l0;  POINT: level (int)(lod + 0.5f).
 *   T4 BC7 blocks are decoded once at creation to RGBA8 (public BPTC format; tables in bc7_tables.inc).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_internal.h"
#include "bc7_tables.inc"

/* ---- BC7 ------------------------------------------------------------------------------------------------- */

typedef struct { int ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2; } Bc7Mode;
static const Bc7Mode BC7_MODES[8] = {
    { 3, 4, 0, 0, 4, 0, 1, 0, 3, 0 }, { 2, 6, 0, 0, 6, 0, 0, 1, 3, 0 }, { 3, 6, 0, 0, 5, 0, 0, 0, 2, 0 },
    { 2, 6, 0, 0, 7, 0, 1, 0, 2, 0 }, { 1, 0, 2, 1, 5, 6, 0, 0, 2, 3 }, { 1, 0, 2, 0, 7, 8, 0, 0, 2, 2 },
    { 1, 0, 0, 0, 7, 7, 1, 0, 4, 0 }, { 2, 6, 0, 0, 5, 5, 1, 0, 2, 0 },
};
static const uint8_t BC7_W2[4] = { 0, 21, 43, 64 };
static const uint8_t BC7_W3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 };
static const uint8_t BC7_W4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };

typedef struct { const uint8_t *p; int pos; } BitReader;
static uint32_t bits(BitReader *r, int n) {
    uint32_t v = 0;
    for (int i = 0; i < n; i++, r->pos++)
        v |= (uint32_t)((r->p[r->pos >> 3] >> (r->pos & 7)) & 1u) << i;
    return v;
}
static const uint8_t *bc7_weights(int nbits) { return nbits == 2 ? BC7_W2 : (nbits == 3 ? BC7_W3 : BC7_W4); }
static uint8_t bc7_interp(int e0, int e1, int w) { return (uint8_t)(((64 - w) * e0 + w * e1 + 32) >> 6); }

void oracle_decode_bc7_block(const uint8_t block[16], uint8_t rgba[64]) {
    BitReader r = { block, 0 };
    int mode = 0;
    while (mode < 8 && !bits(&r, 1)) mode++;
    if (mode == 8) { memset(rgba, 0, 64); return; }          /* reserved encoding decodes to transparent black */
    const Bc7Mode *M = &BC7_MODES[mode];
    int partition = (int)bits(&r, M->pb);
    int rotation = (int)bits(&r, M->rb);
    int idxSel = (int)bits(&r, M->isb);
    int ep[6][4];                                            /* [endpoint][channel] */
    int nep = M->ns * 2;
    for (int ch = 0; ch < 3; ch++) for (int e = 0; e < nep; e++) ep[e][ch] = (int)bits(&r, M->cb);
    for (int e = 0; e < nep; e++) ep[e][3] = M->ab ? (int)bits(&r, M->ab) : 255;
    int cbits = M->cb, abits = M->ab;
    if (M->epb) {
        for (int e = 0; e < nep; e++) {
            int p = (int)bits(&r, 1);
            for (int ch = 0; ch < 3; ch++) ep[e][ch] = (ep[e][ch] << 1) | p;
            if (M->ab) ep[e][3] = (ep[e][3] << 1) | p;
        }
        cbits++; if (M->ab) abits++;
    }
    else if (M->spb) {
        for (int s = 0; s < M->ns; s++) {
            int p = (int)bits(&r, 1);
            for (int e = 2 * s; e < 2 * s + 2; e++) {
                for (int ch = 0; ch < 3; ch++) ep[e][ch] = (ep[e][ch] << 1) | p;
                if (M->ab) ep[e][3] = (ep[e][3] << 1) | p;
            }
        }
        cbits++; if (M->ab) abits++;
    }
    for (int e = 0; e < nep; e++) {
        for (int ch = 0; ch < 3; ch++) { int v = ep[e][ch] << (8 - cbits); ep[e][ch] = v | (v >> cbits); }
        if (M->ab) { int v = ep[e][3] << (8 - abits); ep[e][3] = v | (v >> abits); }
    }
    const uint8_t *pt = M->ns == 2 ? &BC7_PARTITION2[partition * 16] : (M->ns == 3 ? &BC7_PARTITION3[partition * 16] : NULL);
    int anchors[3] = { 0, -1, -1 };
    if (M->ns == 2) anchors[1] = BC7_ANCHOR2[partition];
    if (M->ns == 3) { anchors[1] = BC7_ANCHOR3A[partition]; anchors[2] = BC7_ANCHOR3B[partition]; }
    int idx1[16], idx2[16];
    for (int i = 0; i < 16; i++) {
        int s = pt ? pt[i] : 0;
        int nb = M->ib - (i == anchors[s] ? 1 : 0);
        idx1[i] = (int)bits(&r, nb);
    }
    for (int i = 0; i < 16; i++) idx2[i] = M->ib2 ? (int)bits(&r, M->ib2 - (i == 0 ? 1 : 0)) : 0;
    for (int i = 0; i < 16; i++) {
        int s = pt ? pt[i] : 0;
        const int *e0 = ep[2 * s], *e1 = ep[2 * s + 1];
        int ci = idx1[i], ai = idx1[i], cn = M->ib, an = M->ib;
        if (M->ib2) {
            if (idxSel) { ci = idx2[i]; cn = M->ib2; ai = idx1[i]; an = M->ib; }
            else { ci = idx1[i]; cn = M->ib; ai = idx2[i]; an = M->ib2; }
        }
        uint8_t px[4];
        for (int ch = 0; ch < 3; ch++) px[ch] = bc7_interp(e0[ch], e1[ch], bc7_weights(cn)[ci]);
        px[3] = M->ab ? bc7_interp(e0[3], e1[3], bc7_weights(an)[ai]) : 255;
        if (rotation) { uint8_t t = px[3]; px[3] = px[rotation - 1]; px[rotation - 1] = t; }
        memcpy(rgba + 4 * i, px, 4);
    }
}

/* ---- creation ---------------------------------------------------------------------------------------------- */

OTexture *oracle_texture_create_rgba8(const uint8_t *bytes, int width, int height, int rowPitch) {
    OTexture *t = (OTexture *)calloc(1, sizeof(OTexture));
    t->mips = 1; t->w[0] = width; t->h[0] = height;
    t->rgba[0] = (uint8_t *)malloc((size_t)width * height * 4);
    for (int y = 0; y < height; y++) memcpy(t->rgba[0] + (size_t)y * width * 4, bytes + (size_t)y * rowPitch, (size_t)width * 4);
    return t;
}

static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

OTexture *oracle_texture_create_dds(const uint8_t *bytes, size_t byteCount) {
    if (byteCount < 128 || memcmp(bytes, "DDS ", 4) != 0) return NULL;
    uint32_t height = rd32(bytes + 12), width = rd32(bytes + 16), mipCount = rd32(bytes + 28);
    uint32_t pfFlags = rd32(bytes + 80), fourCC = rd32(bytes + 84), rgbBits = rd32(bytes + 88);
    size_t off = 128;
    int bc7 = 0, rgba = 0;
    if ((pfFlags & 0x4) && fourCC == 0x30315844u) {                       /* 'DX10' */
        if (byteCount < 148) return NULL;
        uint32_t fmt = rd32(bytes + 128);
        off = 148;
        if (fmt == 98 || fmt == 99) bc7 = 1;                               /* BC7_UNORM(_SRGB read as UNORM) */
        else if (fmt == 28 || fmt == 29) rgba = 1;                         /* R8G8B8A8_UNORM */
        else return NULL;
    }
    else if ((pfFlags & 0x40) && rgbBits == 32 && rd32(bytes + 92) == 0x000000FFu) rgba = 1;
    else return NULL;
    if (mipCount == 0) mipCount = 1;
    if (mipCount > O_MAX_MIPS) return NULL;
    OTexture *t = (OTexture *)calloc(1, sizeof(OTexture));
    t->mips = (int)mipCount;
    uint32_t w = width, h = height;
    for (uint32_t m = 0; m < mipCount; m++) {
        t->w[m] = (int)w; t->h[m] = (int)h;
        t->rgba[m] = (uint8_t *)malloc((size_t)w * h * 4);
        if (bc7) {
            uint32_t bw = (w + 3) / 4, bh = (h + 3) / 4;
            if (off + (size_t)bw * bh * 16 > byteCount) { oracle_texture_destroy(t); return NULL; }
            for (uint32_t by = 0; by < bh; by++)
                for (uint32_t bx = 0; bx < bw; bx++) {
                    uint8_t px[64];
                    oracle_decode_bc7_block(bytes + off + ((size_t)by * bw + bx) * 16, px);
                    for (uint32_t y = 0; y < 4 && by * 4 + y < h; y++)
                        for (uint32_t x = 0; x < 4 && bx * 4 + x < w; x++)
                            memcpy(t->rgba[m] + (((size_t)(by * 4 + y)) * w + bx * 4 + x) * 4, px + (y * 4 + x) * 4, 4);
                }
            off += (size_t)bw * bh * 16;
        }
        else if (rgba) {
            if (off + (size_t)w * h * 4 > byteCount) { oracle_texture_destroy(t); return NULL; }
            memcpy(t->rgba[m], bytes + off, (size_t)w * h * 4);
            off += (size_t)w * h * 4;
        }
        w = w > 1 ? w / 2 : 1; h = h > 1 ? h / 2 : 1;
    }
    return t;
}

void oracle_texture_destroy(OTexture *t) {
    if (!t) return;
    for (int m = 0; m < O_MAX_MIPS; m++) free(t->rgba[m]);
    free(t);
}

int oracle_texture_info(const OTexture *t, int *width, int *height, int *mips) {
    if (!t) return 0;
    *width = t->w[0]; *height = t->h[0]; *mips = t->mips;
    return 1;
}

const uint8_t *oracle_texture_mip(const OTexture *t, int mip, int *w, int *h) {
    if (!t || mip < 0 || mip >= t->mips) return NULL;
    *w = t->w[mip]; *h = t->h[mip];
    return t->rgba[mip];
}

/* ---- sampling ---------------------------------------------------------------------------------------------- */

static int address(int i, int n, int mode) {
    if (mode == 2) return i < 0 ? 0 : (i >= n ? n - 1 : i);               /* CLAMP */
    if (mode == 1) {                                                      /* MIRROR */
        int p = 2 * n, j = i % p; if (j < 0) j += p;
        return j < n ? j : p - 1 - j;
    }
    int j = i % n; if (j < 0) j += n;                                     /* WRAP */
    return j;
}

static void texel(const OTexture *t, int level, int x, int y, float out[4]) {
    const uint8_t *p = t->rgba[level] + ((size_t)y * t->w[level] + x) * 4;
    for (int c = 0; c < 4; c++) out[c] = (float)p[c] * (1.0f / 255.0f);
}

void otex_sample_level(const OTexture *t, float u, float v, int level, int filter, int hAddr, int vAddr, float out[4]) {
    int w = t->w[level], h = t->h[level];
    if (filter == 0) {
        int x = address((int)floorf(u * (float)w), w, hAddr), y = address((int)floorf(v * (float)h), h, vAddr);
        texel(t, level, x, y, out);
        return;
    }
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = address((int)x0f, w, hAddr), x1 = address((int)x0f + 1, w, hAddr);
    int y0 = address((int)y0f, h, vAddr), y1 = address((int)y0f + 1, h, vAddr);
    float c00[4], c10[4], c01[4], c11[4];
    texel(t, level, x0, y0, c00); texel(t, level, x1, y0, c10); texel(t, level, x0, y1, c01); texel(t, level, x1, y1, c11);
    for (int c = 0; c < 4; c++) {
        float top = c00[c] + fx * (c10[c] - c00[c]);
        float bot = c01[c] + fx * (c11[c] - c01[c]);
        out[c] = top + fy * (bot - top);
    }
}

void otex_sample_grad(const OTexture *t, float u, float v, of2 ddx, of2 ddy, int filter, int hAddr, int vAddr, float out[4]) {
    if (t->mips == 1) { otex_sample_level(t, u, v, 0, filter, hAddr, vAddr, out); return; }
    float w0 = (float)t->w[0], h0 = (float)t->h[0];
    float ax = ddx.x * w0, ay = ddx.y * h0, bx = ddy.x * w0, by = ddy.y * h0;
    float rho = fmaxf(sqrtf(ax * ax + ay * ay), sqrtf(bx * bx + by * by));
    float lod = rho > 0.0f ? log2f(rho) : 0.0f;
    float maxLod = (float)(t->mips - 1);
    if (!(lod > 0.0f)) lod = 0.0f;
    if (lod > maxLod) lod = maxLod;
    if (filter == 0) {
        otex_sample_level(t, u, v, (int)(lod + 0.5f), filter, hAddr, vAddr, out);
        return;
    }
    int l0 = (int)floorf(lod), l1 = l0 + 1 < t->mips ? l0 + 1 : t->mips - 1;
    float f = lod - (float)l0;
    float a[4], b[4];
    otex_sample_level(t, u, v, l0, filter, hAddr, vAddr, a);
    otex_sample_level(t, u, v, l1, filter, hAddr, vAddr, b);
    for (int c = 0; c < 4; c++) out[c] = a[c] + f * (b[c] - a[c]);
}

void oracle_texture_sample(const OTexture *t, float u, float v, float ddxu, float ddxv, float ddyu, float ddyv,
                           int filter, int hAddr, int vAddr, float out[4]) {
    of2 dx = { ddxu, ddxv }, dy = { ddyu, ddyv };
    otex_sample_grad(t, u, v, dx, dy, filter, hAddr, vAddr, out);
}
