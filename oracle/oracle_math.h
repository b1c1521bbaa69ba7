/* oracle_math.h -- scalar float helpers for the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h).
 *
 * Build rule: -ffp-contract=off, no -ffast-math.  Every fused multiply-add in the geometry spec is an
 * explicit fmaf(); everything else is an individually rounded IEEE binary32 operation, so the HIP
 * kernels can follow the same operation order and obtain identical bits.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>
#include "rt64_oracle.h"

#define O_EPSILON 1e-6f                    /* ref:shaders/Constants.hlsli:5 */
#define O_PI 3.14159265f                   /* ref:shaders/Constants.hlsli:6 */
#define O_TWO_PI (O_PI * 2.0f)
#define O_APPLY_LIGHTS_MINIMUM_ALPHA 0.5f  /* ref:shaders/Constants.hlsli:8 */
#define O_RAY_MIN_DISTANCE 0.1f            /* ref:shaders/Ray.hlsli:9 */
#define O_RAY_MAX_DISTANCE 100000.0f       /* ref:shaders/Ray.hlsli:10 */

typedef struct { float x, y; } of2;
typedef ov3 of3;
typedef ov4 of4;

static inline of3 v3(float x, float y, float z) { of3 r = { x, y, z }; return r; }
static inline of3 v3s(float s) { return v3(s, s, s); }
static inline of3 v3add(of3 a, of3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline of3 v3sub(of3 a, of3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline of3 v3mul(of3 a, of3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline of3 v3scale(of3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline of3 v3neg(of3 a) { return v3(-a.x, -a.y, -a.z); }
/* Shading-side dot/cross: plain left-to-right sums (HLSL dot has no specified order; tolerance covers it). */
static inline float v3dot(of3 a, of3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline of3 v3cross(of3 a, of3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3len(of3 a) { return sqrtf(v3dot(a, a)); }
static inline of3 v3normalize(of3 a) { float inv = 1.0f / v3len(a); return v3(a.x * inv, a.y * inv, a.z * inv); }   /* HLSL normalize = v * rsqrt(dot); one IEEE divide */
static inline of3 v3lerp(of3 a, of3 b, float t) { return v3add(a, v3scale(v3sub(b, a), t)); }   /* HLSL lerp: a + t*(b-a) */
static inline float flerp(float a, float b, float t) { return a + t * (b - a); }
static inline float fclampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float fsaturate(float x) { return fclampf(x, 0.0f, 1.0f); }
static inline of3 v3reflect(of3 i, of3 n) { return v3sub(i, v3scale(n, 2.0f * v3dot(n, i))); }   /* HLSL reflect */
static inline float hlsl_fmod(float x, float y) { return x - y * truncf(x / y); }

/* Geometry-side dot/cross with a fixed fma chain (bit-exact contract with the HIP kernels). */
static inline float g_dot3(const float a[3], const float b[3]) {
    return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}
static inline void g_cross3(const float a[3], const float b[3], float r[3]) {
    r[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    r[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    r[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
/* p * M for a point (w = 1) / a vector (w = 0), row-vector convention, fixed fma chain. */
static inline void g_xform_point(const om4 *M, const float p[3], float r[3]) {
    for (int c = 0; c < 3; c++)
        r[c] = fmaf(p[2], M->m[2][c], fmaf(p[1], M->m[1][c], fmaf(p[0], M->m[0][c], M->m[3][c])));
}
static inline void g_xform_vector(const om4 *M, const float p[3], float r[3]) {
    for (int c = 0; c < 3; c++)
        r[c] = fmaf(p[2], M->m[2][c], fmaf(p[1], M->m[1][c], p[0] * M->m[0][c]));
}

/* Shading-side 4x4 helpers (row-vector convention; HLSL mul(M, v) on the reference's cbuffers == v * M_cpu,
 * SURVEY appendix A1). */
static inline of4 m4_mul_vec(const om4 *M, of4 v) {
    of4 r;
    r.x = v.x * M->m[0][0] + v.y * M->m[1][0] + v.z * M->m[2][0] + v.w * M->m[3][0];
    r.y = v.x * M->m[0][1] + v.y * M->m[1][1] + v.z * M->m[2][1] + v.w * M->m[3][1];
    r.z = v.x * M->m[0][2] + v.y * M->m[1][2] + v.z * M->m[2][2] + v.w * M->m[3][2];
    r.w = v.x * M->m[0][3] + v.y * M->m[1][3] + v.z * M->m[2][3] + v.w * M->m[3][3];
    return r;
}
static inline of3 m4_point(const om4 *M, of3 p) { of4 v = { p.x, p.y, p.z, 1.0f }; of4 r = m4_mul_vec(M, v); return v3(r.x, r.y, r.z); }
static inline of3 m4_vector(const om4 *M, of3 p) { of4 v = { p.x, p.y, p.z, 0.0f }; of4 r = m4_mul_vec(M, v); return v3(r.x, r.y, r.z); }
static inline void m4_mul(const om4 *A, const om4 *B, om4 *R) {   /* R = A * B, float accumulation left to right */
    om4 t;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            t.m[i][j] = A->m[i][0] * B->m[0][j] + A->m[i][1] * B->m[1][j] + A->m[i][2] * B->m[2][j] + A->m[i][3] * B->m[3][j];
    *R = t;
}
static inline void m4_identity(om4 *M) { memset(M, 0, sizeof(*M)); M->m[0][0] = M->m[1][1] = M->m[2][2] = M->m[3][3] = 1.0f; }
static inline void m4_transpose(const om4 *A, om4 *R) { om4 t; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t.m[i][j] = A->m[j][i]; *R = t; }

/* ---- storage formats (D3D functional spec conversion rules) ---------------------- */

static inline uint8_t to_unorm8(float x) {       /* clamp to [0,1], NaN -> 0, round to nearest */
    if (!(x > 0.0f)) return 0;
    if (x >= 1.0f) return 255;
    return (uint8_t)floorf(x * 255.0f + 0.5f);
}
static inline float from_unorm8(uint8_t v) { return (float)v / 255.0f; }
static inline int16_t to_snorm16(float x) {
    if (x != x) return 0;
    x = fclampf(x, -1.0f, 1.0f) * 32767.0f;
    return (int16_t)(x >= 0.0f ? x + 0.5f : x - 0.5f);
}
static inline float from_snorm16(int16_t v) { return fmaxf((float)v / 32767.0f, -1.0f); }
static inline float q_unorm8(float x) { return from_unorm8(to_unorm8(x)); }
static inline float q_snorm16(float x) { return from_snorm16(to_snorm16(x)); }
static inline float q_f16(float x) { return oracle_f16_to_f32(oracle_f32_to_f16(x)); }
static inline of3 q3_f16(of3 v) { return v3(q_f16(v.x), q_f16(v.y), q_f16(v.z)); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

#endif
