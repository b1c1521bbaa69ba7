// device_math.h -- small float vector helpers for the HIP kernels (also usable from host code).
//
// The translation units are compiled with -ffp-contract=off: every fused multiply-add is an explicit fmaf(), so
// the geometry pipeline (Morton codes, boxes, ray/box and ray/triangle tests) produces the same bits as the
// scalar reference tracer used by the tests.  Shading code uses plain left-to-right arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <math.h>

#define HD __host__ __device__ __forceinline__
#define DEV __device__ __forceinline__

#define RT_EPSILON 1e-6f                    // Constants.hlsli:5
#define RT_PI 3.14159265f                   // Constants.hlsli:6
#define RT_TWO_PI (RT_PI * 2.0f)
#define RT_APPLY_LIGHTS_MINIMUM_ALPHA 0.5f  // Constants.hlsli:8
#define RT_RAY_MIN_DISTANCE 0.1f            // Ray.hlsli:9
#define RT_RAY_MAX_DISTANCE 100000.0f       // Ray.hlsli:10

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
HD f3 mk3s(float s) { return mk3(s, s, s); }
HD f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
HD f3 xyz(f4 v) { return mk3(v.x, v.y, v.z); }
HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
// Shading-side fast math (device only): v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / v_exp_f32 / v_log_f32 are 1-ulp hardware ops.
// The HLSL this path replaces compiles to the same class of approximate GPU instructions (pow = exp2(y * log2(x)),
// normalize = v * rsqrt(dot)), so IEEE-exact libm results are not the specification here; the tests compare shading
// with a tolerance.  Geometry code (trace.h, lbvh.hip) never uses these.
#if defined(__HIP_DEVICE_COMPILE__)
DEV float s_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEV float s_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
DEV float s_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
DEV float s_pow(float x, float y) { return y == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
#else
HD float s_rcp(float x) { return 1.0f / x; }
HD float s_rsqrt(float x) { return 1.0f / sqrtf(x); }
HD float s_sqrt(float x) { return sqrtf(x); }
HD float s_pow(float x, float y) { return powf(x, y); }
#endif
HD float s_div(float a, float b) { return a * s_rcp(b); }

HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HD float len3(f3 a) { return s_sqrt(dot3(a, a)); }
HD f3 normalize3(f3 a) { float inv = s_rsqrt(dot3(a, a)); return mk3(a.x * inv, a.y * inv, a.z * inv); }   // HLSL normalize = v * rsqrt(dot)
// IEEE-exact variants for values that define RAYS (shadow-ray direction / length): traversal decisions stay bit-identical
// to the scalar reference tracer.
HD float len3_exact(f3 a) { return sqrtf(dot3(a, a)); }
HD f3 normalize3_exact(f3 a) { float inv = 1.0f / len3_exact(a); return mk3(a.x * inv, a.y * inv, a.z * inv); }
// Direction spec D1 (shared with oracle/oracle_shade.c: sincos_turns): sine and cosine of 2 pi u for u in [0, 1], every operation spelled out so that the
// device and the scalar reference tracer produce the same bits -- the angle of a cosine-weighted bounce direction (IndirectRayGen.hlsl:18-29) defines a RAY,
// and the library's sinf / cosf are different code on the two sides.  a = 4 u; k = (int)(a + 0.5f); x = (a - k) * (pi / 2) in [-pi/4, pi/4];
// s = x + x z (S1 + z (S2 + z S3)), c = 1 + z (C1 + z (C2 + z (C3))) with z = x x, as fmaf chains (Cephes' single-precision kernels, |error| < 1e-7);
// the quadrant k rotates (s, c).  (The reference evaluates sin / cos with the GPU's approximate instructions: no bit pattern is specified there.)
HD void sincos_turns(float u, float &sn, float &cs) {
    const float a = u * 4.0f;
    const int k = (int)(a + 0.5f);
    const float x = (a - (float)k) * 1.57079637f, z = x * x;
    const float s = fmaf(x * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), x);
    const float c = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), -0.5f), 1.0f);
    switch (k & 3) {
    case 0: sn = s; cs = c; break;
    case 1: sn = c; cs = -s; break;
    case 2: sn = -s; cs = -c; break;
    default: sn = -c; cs = s; break;
    }
}
HD float lerpf(float a, float b, float t) { return a + t * (b - a); }        // HLSL lerp
HD f3 lerp3(f3 a, f3 b, float t) { return a + (b - a) * t; }
HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
HD float saturatef(float x) { return clampf(x, 0.0f, 1.0f); }
HD f3 reflect3(f3 i, f3 n) { return i - n * (2.0f * dot3(n, i)); }           // HLSL reflect
HD float hlsl_fmod(float x, float y) { return x - y * truncf(x / y); }

// Geometry-side helpers with a fixed fma chain (bit-exact contract, see DESIGN.md "Geometry spec").
HD float g_dot3(const float a[3], const float b[3]) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
HD void g_cross3(const float a[3], const float b[3], float r[3]) {
    r[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    r[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    r[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
HD void g_xform_point(const float *M, const float p[3], float r[3]) {        // p * M, w = 1
    for (int c = 0; c < 3; c++) r[c] = fmaf(p[2], M[8 + c], fmaf(p[1], M[4 + c], fmaf(p[0], M[c], M[12 + c])));
}
HD void g_xform_vector(const float *M, const float p[3], float r[3]) {       // p * M, w = 0
    for (int c = 0; c < 3; c++) r[c] = fmaf(p[2], M[8 + c], fmaf(p[1], M[4 + c], p[0] * M[c]));
}

// Shading-side 4x4 (row-vector convention; HLSL mul(M, v) on the reference's cbuffers == v * M_cpu).
HD f4 mul4(const float *M, f4 v) {
    f4 r;
    r.x = v.x * M[0] + v.y * M[4] + v.z * M[8] + v.w * M[12];
    r.y = v.x * M[1] + v.y * M[5] + v.z * M[9] + v.w * M[13];
    r.z = v.x * M[2] + v.y * M[6] + v.z * M[10] + v.w * M[14];
    r.w = v.x * M[3] + v.y * M[7] + v.z * M[11] + v.w * M[15];
    return r;
}
HD f3 mul_point(const float *M, f3 p) { return xyz(mul4(M, mk4(p.x, p.y, p.z, 1.0f))); }
HD f3 mul_vector(const float *M, f3 p) { return xyz(mul4(M, mk4(p.x, p.y, p.z, 0.0f))); }

// ---- storage formats (D3D conversion rules; the images keep the reference's formats, rt64_view.cpp:152-241) ----
HD uint8_t to_unorm8(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 1.0f) return 255;
    return (uint8_t)floorf(x * 255.0f + 0.5f);
}
HD float from_unorm8(uint8_t v) { return (float)v / 255.0f; }
HD int16_t to_snorm16(float x) {
    if (x != x) return 0;
    x = clampf(x, -1.0f, 1.0f) * 32767.0f;
    return (int16_t)(x >= 0.0f ? x + 0.5f : x - 0.5f);
}
HD float from_snorm16(int16_t v) { return fmaxf((float)v / 32767.0f, -1.0f); }
HD float q_unorm8(float x) { return from_unorm8(to_unorm8(x)); }
HD float q_snorm16(float x) { return from_snorm16(to_snorm16(x)); }

DEV uint16_t f32_to_f16_bits(float x) { return __half_as_ushort(__float2half_rn(x)); }
DEV float f16_bits_to_f32(uint16_t h) { return __half2float(__ushort_as_half(h)); }
DEV float q_f16(float x) { return f16_bits_to_f32(f32_to_f16_bits(x)); }

// RGBA16F image access: one 8-byte load/store per pixel.
DEV void store_rgba16f(uint16_t *img, size_t i, float x, float y, float z, float w) {
    uint2 v;
    v.x = (uint32_t)f32_to_f16_bits(x) | ((uint32_t)f32_to_f16_bits(y) << 16);
    v.y = (uint32_t)f32_to_f16_bits(z) | ((uint32_t)f32_to_f16_bits(w) << 16);
    reinterpret_cast<uint2 *>(img)[i] = v;
}
DEV uint32_t pack_rgba16f_lo(float a, float b) { return (uint32_t)f32_to_f16_bits(a) | ((uint32_t)f32_to_f16_bits(b) << 16); }      // two RGBA16F channels as one dword (store_rgba16f's packing)
DEV f4 unpack_rgba16f_bits(uint32_t xy, uint32_t zw) {
    return mk4(f16_bits_to_f32((uint16_t)(xy & 0xFFFFu)), f16_bits_to_f32((uint16_t)(xy >> 16)), f16_bits_to_f32((uint16_t)(zw & 0xFFFFu)), f16_bits_to_f32((uint16_t)(zw >> 16)));
}
DEV f4 load_rgba16f(const uint16_t *img, size_t i) {
    uint2 v = reinterpret_cast<const uint2 *>(img)[i];
    return mk4(f16_bits_to_f32((uint16_t)(v.x & 0xFFFFu)), f16_bits_to_f32((uint16_t)(v.x >> 16)),
               f16_bits_to_f32((uint16_t)(v.y & 0xFFFFu)), f16_bits_to_f32((uint16_t)(v.y >> 16)));
}
DEV void store_rgba8(uint8_t *img, size_t i, float x, float y, float z, float w) {
    uint32_t v = (uint32_t)to_unorm8(x) | ((uint32_t)to_unorm8(y) << 8) | ((uint32_t)to_unorm8(z) << 16) | ((uint32_t)to_unorm8(w) << 24);
    reinterpret_cast<uint32_t *>(img)[i] = v;
}
DEV f4 load_rgba8(const uint8_t *img, size_t i) {
    uint32_t v = reinterpret_cast<const uint32_t *>(img)[i];
    return mk4(from_unorm8((uint8_t)(v & 0xFF)), from_unorm8((uint8_t)((v >> 8) & 0xFF)), from_unorm8((uint8_t)((v >> 16) & 0xFF)), from_unorm8((uint8_t)(v >> 24)));
}

// SVGF: variance of a pixel with enough history, from its luminance moments -- one definition for the two kernels that may compute it (svgf_variance_kernel and,
// on frames whose GI runs as the wavefront chain, bounce_resolve_kernel), with the multiply-add spelled out so that both round alike whatever their files' contraction mode
HD float svgf_moment_variance(float m1, float m2) { return fmaxf(0.0f, fmaf(-m1, m1, m2)); }
