// host_field.h — GF(p) arithmetic on native 64-bit limbs for the few HOST-side pieces of the library:
// the serial MiMC recurrence (air_mimc.hip) and O(n^2) Lagrange interpolation of <= a few hundred points
// (small.hip).  Same modulus and canonical representation as gf128.cuh.
#pragma once
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 hu128;

static inline hu128 hf_p() { return ((hu128)0xFFFFFFFFFFFFFFFFull << 64) | 0xFFFFFFF700000001ull; }
static const hu128 HF_C = (hu128)0x8FFFFFFFFull;  // 2^128 mod p

static inline hu128 hf_reduce(hu128 hi, hu128 lo) {
    hu128 m0 = (hu128)(uint64_t)hi * HF_C, m1 = (hu128)(uint64_t)(hi >> 64) * HF_C;
    hu128 tl = m0 + (m1 << 64);
    hu128 th = (m1 >> 64) + (tl < m0);
    hu128 s = tl + lo;
    unsigned k = s < tl;
    hu128 s2 = s + th * HF_C;
    k += s2 < s;
    while (k) { hu128 s3 = s2 + HF_C; k -= 1; k += s3 < s2; s2 = s3; }
    while (s2 >= hf_p()) s2 -= hf_p();
    return s2;
}
static inline hu128 hf_mul(hu128 a, hu128 b) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    hu128 p00 = (hu128)a0 * b0, p01 = (hu128)a0 * b1, p10 = (hu128)a1 * b0, p11 = (hu128)a1 * b1;
    hu128 mid = p01 + p10, midc = mid < p01;
    hu128 lo = p00 + (mid << 64), c1 = lo < p00;
    hu128 hi = p11 + (mid >> 64) + (midc << 64) + c1;
    return hf_reduce(hi, lo);
}
// Weakly reduced product for latency-bound serial chains (the MiMC recurrence): returns ANY 128-bit representative of
// a*b mod p (inputs may be any 128-bit values).  The fold multiplies by 2^128 mod p = 2^35 + 2^32 - 1 with shifts and
// skips the final compare-and-subtract, which shortens the dependency chain of x -> x^2 -> x^3.
static inline hu128 hf_mul_weak(hu128 a, hu128 b) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    hu128 p00 = (hu128)a0 * b0, p01 = (hu128)a0 * b1, p10 = (hu128)a1 * b0, p11 = (hu128)a1 * b1;
    hu128 mid = p01 + p10, midc = mid < p01;
    hu128 lo = p00 + (mid << 64), c1 = lo < p00;
    hu128 hi = p11 + (mid >> 64) + (midc << 64) + c1;
    // hi * C = ((9*hi) << 32) - hi
    hu128 h8 = hi << 3;
    uint64_t top = (uint64_t)(hi >> 125);
    hu128 h9 = h8 + hi;
    top += (h9 < h8);
    hu128 sh = h9 << 32;
    uint64_t T = (top << 32) | (uint64_t)(h9 >> 96);  // bits of (9*hi << 32) at or above 2^128, < 2^36
    hu128 s = lo + sh;
    int net = (int)(s < lo);
    hu128 d = s - hi;
    net -= (int)(s < hi);
    hu128 tc = (((hu128)T * 9) << 32) - T;             // T * C < 2^73
    hu128 e = d + tc;
    net += (int)(e < d);                               // value = e + net * 2^128, net in {-1, 0, 1, 2}
    if (net > 0) {
        hu128 f = e + (hu128)(unsigned)net * HF_C;
        if (f < e) f += HF_C;
        e = f;
    } else if (net < 0) {
        hu128 f = e - HF_C;
        if (f > e) f -= HF_C;
        e = f;
    }
    return e;
}
static inline hu128 hf_canon(hu128 x) {
    while (x >= hf_p()) x -= hf_p();
    return x;
}
static inline hu128 hf_add(hu128 a, hu128 b) {
    hu128 s = a + b;
    if (s < a || s >= hf_p()) s -= hf_p();
    return s;
}
static inline hu128 hf_sub(hu128 a, hu128 b) { return a >= b ? a - b : a - b + hf_p(); }
static inline hu128 hf_pow(hu128 b, hu128 e) {
    hu128 r = 1;
    while (e) {
        if (e & 1) r = hf_mul(r, b);
        b = hf_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline hu128 hf_inv(hu128 a) { return a ? hf_pow(a, hf_p() - 2) : 0; }
static inline hu128 hf_load(const uint8_t *b) { hu128 v; memcpy(&v, b, 16); return v; }
static inline void hf_store(uint8_t *b, hu128 v) { memcpy(b, &v, 16); }
