/* oracle_util.c -- storage-format conversions of the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h). */
#include "oracle_internal.h"

/* binary32 -> binary16, round to nearest even (typed UAV stores are specified as RNE here). */
uint16_t oracle_f32_to_f16(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007FFFFFu;
    int32_t exp = (int32_t)((x >> 23) & 0xFFu);
    if (exp == 0xFF) return (uint16_t)(sign | 0x7C00u | (mant ? (0x0200u | (mant >> 13)) : 0u));   /* inf / NaN */
    exp = exp - 127 + 15;
    if (exp >= 0x1F) return (uint16_t)(sign | 0x7C00u);                                              /* overflow -> inf */
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;                                                        /* underflow -> signed zero */
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - exp);
        uint32_t half = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1u))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)exp << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;                                    /* may carry into the exponent: correct */
    return (uint16_t)(sign | half);
}

float oracle_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, mant = h & 0x3FFu;
    if (exp == 0) {
        if (mant == 0) return u2f(sign);
        int e = -1;
        do { e++; mant <<= 1; } while (!(mant & 0x400u));
        return u2f(sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3FFu) << 13));
    }
    if (exp == 0x1F) return u2f(sign | 0x7F800000u | (mant << 13));
    return u2f(sign | ((exp - 15 + 127) << 23) | (mant << 13));
}
