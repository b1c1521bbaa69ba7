// bc7.hip -- BC7 (BPTC) block decoder: DDS textures are expanded to RGBA8 once at RT64_CreateTexture time.
//
// The reference hands DDS files to the D3D12 runtime (rt64_texture.cpp:146-187, DDSTextureLoader) and lets the texture
// units decode BC7 on the fly; CDNA has no such unit on the compute path, so the blocks are decoded by this kernel into
// the same RGBA8 texel store every other texture uses (sample scene: grass_dif.dds, DXGI_FORMAT_BC7_UNORM, 10 mips).
// One thread decodes one 4x4 block (128-bit load, 16 x 4-byte stores); 64 MB/s-class work done once per texture.
#include "kernels.h"
#include "bc7_tables.inc"

namespace {

struct Bc7Mode { int8_t ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2; };
__device__ const Bc7Mode MODES[8] = {
    { 3, 4, 0, 0, 4, 0, 1, 0, 3, 0 }, { 2, 6, 0, 0, 6, 0, 0, 1, 3, 0 }, { 3, 6, 0, 0, 5, 0, 0, 0, 2, 0 },
    { 2, 6, 0, 0, 7, 0, 1, 0, 2, 0 }, { 1, 0, 2, 1, 5, 6, 0, 0, 2, 3 }, { 1, 0, 2, 0, 7, 8, 0, 0, 2, 2 },
    { 1, 0, 0, 0, 7, 7, 1, 0, 4, 0 }, { 2, 6, 0, 0, 5, 5, 1, 0, 2, 0 },
};
__device__ const uint8_t W2[4] = { 0, 21, 43, 64 };
__device__ const uint8_t W3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 };
__device__ const uint8_t W4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };

struct Bits {
    unsigned long long lo, hi; int pos;
    __device__ uint32_t get(int n) {
        if (n == 0) return 0;
        unsigned long long v;
        if (pos >= 64) v = hi >> (pos - 64);
        else { v = lo >> pos; if (pos + n > 64) v |= hi << (64 - pos); }
        pos += n;
        return (uint32_t)(v & ((1ull << n) - 1ull));
    }
};

__device__ int weight(int nbits, int idx) { return nbits == 2 ? W2[idx] : (nbits == 3 ? W3[idx] : W4[idx]); }
__device__ int interp(int e0, int e1, int w) { return ((64 - w) * e0 + w * e1 + 32) >> 6; }

__global__ void bc7_decode_kernel(const uint4 *blocks, uint32_t *rgba, uint32_t width, uint32_t height, uint32_t bw, uint32_t bh,
                                  const uint8_t *part2, const uint8_t *part3, const uint8_t *anchor2, const uint8_t *anchor3a, const uint8_t *anchor3b) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bw * bh) return;
    const uint4 raw = blocks[b];
    Bits r; r.lo = (unsigned long long)raw.x | ((unsigned long long)raw.y << 32); r.hi = (unsigned long long)raw.z | ((unsigned long long)raw.w << 32); r.pos = 0;
    int mode = 0;
    while (mode < 8 && !r.get(1)) mode++;
    uint32_t px[16];
    if (mode == 8) { for (int i = 0; i < 16; i++) px[i] = 0; }
    else {
        const Bc7Mode M = MODES[mode];
        const int partition = (int)r.get(M.pb), rotation = (int)r.get(M.rb), idxSel = (int)r.get(M.isb);
        int ep[6][4];
        const int nep = M.ns * 2;
        for (int ch = 0; ch < 3; ch++) for (int e = 0; e < nep; e++) ep[e][ch] = (int)r.get(M.cb);
        for (int e = 0; e < nep; e++) ep[e][3] = M.ab ? (int)r.get(M.ab) : 255;
        int cbits = M.cb, abits = M.ab;
        if (M.epb) {
            for (int e = 0; e < nep; e++) {
                int p = (int)r.get(1);
                for (int ch = 0; ch < 3; ch++) ep[e][ch] = (ep[e][ch] << 1) | p;
                if (M.ab) ep[e][3] = (ep[e][3] << 1) | p;
            }
            cbits++; if (M.ab) abits++;
        }
        else if (M.spb) {
            for (int s = 0; s < M.ns; s++) {
                int p = (int)r.get(1);
                for (int e = 2 * s; e < 2 * s + 2; e++) {
                    for (int ch = 0; ch < 3; ch++) ep[e][ch] = (ep[e][ch] << 1) | p;
                    if (M.ab) ep[e][3] = (ep[e][3] << 1) | p;
                }
            }
            cbits++; if (M.ab) abits++;
        }
        for (int e = 0; e < nep; e++) {
            for (int ch = 0; ch < 3; ch++) { int v = ep[e][ch] << (8 - cbits); ep[e][ch] = v | (v >> cbits); }
            if (M.ab) { int v = ep[e][3] << (8 - abits); ep[e][3] = v | (v >> abits); }
        }
        const uint8_t *pt = M.ns == 2 ? part2 + partition * 16 : (M.ns == 3 ? part3 + partition * 16 : nullptr);
        int anchors[3] = { 0, -1, -1 };
        if (M.ns == 2) anchors[1] = anchor2[partition];
        if (M.ns == 3) { anchors[1] = anchor3a[partition]; anchors[2] = anchor3b[partition]; }
        int idx1[16], idx2[16];
        for (int i = 0; i < 16; i++) { int s = pt ? pt[i] : 0; idx1[i] = (int)r.get(M.ib - (i == anchors[s] ? 1 : 0)); }
        for (int i = 0; i < 16; i++) idx2[i] = M.ib2 ? (int)r.get(M.ib2 - (i == 0 ? 1 : 0)) : 0;
        for (int i = 0; i < 16; i++) {
            const int s = pt ? pt[i] : 0;
            const int *e0 = ep[2 * s], *e1 = ep[2 * s + 1];
            int ci = idx1[i], ai = idx1[i], cn = M.ib, an = M.ib;
            if (M.ib2) {
                if (idxSel) { ci = idx2[i]; cn = M.ib2; ai = idx1[i]; an = M.ib; }
                else { ci = idx1[i]; cn = M.ib; ai = idx2[i]; an = M.ib2; }
            }
            int c[4];
            for (int ch = 0; ch < 3; ch++) c[ch] = interp(e0[ch], e1[ch], weight(cn, ci));
            c[3] = M.ab ? interp(e0[3], e1[3], weight(an, ai)) : 255;
            if (rotation) { int t = c[3]; c[3] = c[rotation - 1]; c[rotation - 1] = t; }
            px[i] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
        }
    }
    const uint32_t bx = b % bw, by = b / bw;
    for (uint32_t y = 0; y < 4 && by * 4 + y < height; y++)
        for (uint32_t x = 0; x < 4 && bx * 4 + x < width; x++)
            rgba[(size_t)(by * 4 + y) * width + bx * 4 + x] = px[y * 4 + x];
}

uint8_t *g_tables_per_device[64] = {};     // device copy: part2[1024] part3[1024] anchor2[64] anchor3a[64] anchor3b[64]

}  // namespace

hipError_t bc7_decode_launch(const uint8_t *blocks, uint8_t *rgba, uint32_t width, uint32_t height, hipStream_t stream) {
    int dev = 0;
    hipGetDevice(&dev);
    uint8_t *&g_tables = g_tables_per_device[dev & 63];
    if (!g_tables) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&g_tables), 1024 + 1024 + 64 * 3);
        if (e != hipSuccess) return e;
        hipMemcpy(g_tables, BC7_PARTITION2, 1024, hipMemcpyHostToDevice);
        hipMemcpy(g_tables + 1024, BC7_PARTITION3, 1024, hipMemcpyHostToDevice);
        hipMemcpy(g_tables + 2048, BC7_ANCHOR2, 64, hipMemcpyHostToDevice);
        hipMemcpy(g_tables + 2112, BC7_ANCHOR3A, 64, hipMemcpyHostToDevice);
        hipMemcpy(g_tables + 2176, BC7_ANCHOR3B, 64, hipMemcpyHostToDevice);
    }
    const uint32_t bw = (width + 3) / 4, bh = (height + 3) / 4, n = bw * bh;
    hipLaunchKernelGGL(bc7_decode_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, reinterpret_cast<const uint4 *>(blocks),
                       reinterpret_cast<uint32_t *>(rgba), width, height, bw, bh, g_tables, g_tables + 1024, g_tables + 2048, g_tables + 2112, g_tables + 2176);
    return hipGetLastError();
}

// ---- 4 x 4-texel tiles ------------------------------------------------------------------------------------------------------------
// dst[((y / 4) * (w / 4) + x / 4) * 16 + (y % 4) * 4 + x % 4] = src[y * w + x]: one 64-byte cache line per tile (FrameParams::skyTiled).
namespace {
__global__ __launch_bounds__(256) void tile_texture_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t w, uint32_t h) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;            // index into dst: consecutive threads write consecutive words
    if (i >= w * h) return;
    const uint32_t tile = i >> 4, in = i & 15u, tilesX = w >> 2;
    const uint32_t x = (tile % tilesX) * 4 + (in & 3u), y = (tile / tilesX) * 4 + (in >> 2);
    dst[i] = src[(size_t)y * w + x];
}
}  // namespace

hipError_t tile_texture_launch(const uint8_t *rgba, uint32_t *tiled, uint32_t width, uint32_t height, hipStream_t stream) {
    if (width < 4 || height < 4 || (width & 3u) || (height & 3u)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tile_texture_kernel, dim3((width * height + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(rgba), tiled, width, height);
    return hipGetLastError();
}
