// gf128.h — arithmetic in GF(p), p = 2^128 - 9*2^32 + 1, for gfx950 device code (and, for the
// CPU unit tests of this header only, plain host C++).
//
// Elements are kept CANONICAL (value < p) in memory: the same 16 little-endian bytes are hashed into
// Merkle leaves and copied into proofs (lib/Stark.ts:284-296), so no Montgomery-domain conversion
// pass is ever needed.  Reduction uses the shape of the modulus instead: 2^128 == 9*2^32 - 1 (mod p),
// i.e. a 256-bit product folds with shifts and adds only (no multiplications by p or p').
//
// Register form: four 32-bit limbs (gfx950 has no 64x64 multiplier; the widest is
// v_mad_u64_u32 = 32x32+64), loaded/stored as one 128-bit dwordx4 access.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GF_HD __host__ __device__ __forceinline__
#else
#define GF_HD inline
#endif

struct alignas(16) fe {
    uint32_t w0, w1, w2, w3;
};

#define GF_P0 0x00000001u
#define GF_P1 0xFFFFFFF7u
#define GF_P2 0xFFFFFFFFu
#define GF_P3 0xFFFFFFFFu

GF_HD fe fe_make(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { fe r; r.w0 = a; r.w1 = b; r.w2 = c; r.w3 = d; return r; }
GF_HD fe fe_zero() { return fe_make(0, 0, 0, 0); }
GF_HD fe fe_one() { return fe_make(1, 0, 0, 0); }
GF_HD bool fe_is_zero(const fe &a) { return (a.w0 | a.w1 | a.w2 | a.w3) == 0; }
GF_HD bool fe_eq(const fe &a, const fe &b) { return a.w0 == b.w0 && a.w1 == b.w1 && a.w2 == b.w2 && a.w3 == b.w3; }

// a >= p  (p = [1, 0xFFFFFFF7, 0xFFFFFFFF, 0xFFFFFFFF])
GF_HD bool fe_ge_p(const fe &a) {
    return (a.w3 == GF_P3) && (a.w2 == GF_P2) && (a.w1 > GF_P1 || (a.w1 == GF_P1 && a.w0 >= GF_P0));
}

// carry-chain primitives: clang lowers these to v_add_co_u32 / v_addc_co_u32 (v_sub_co / v_subb_co)
GF_HD uint32_t gf_addc(uint32_t x, uint32_t y, uint32_t ci, uint32_t &co) {
#if defined(__clang__)
    unsigned c;
    uint32_t r = __builtin_addc(x, y, ci, &c);
    co = c;
    return r;
#else
    uint64_t t = (uint64_t)x + y + ci;
    co = (uint32_t)(t >> 32);
    return (uint32_t)t;
#endif
}
GF_HD uint32_t gf_subc(uint32_t x, uint32_t y, uint32_t bi, uint32_t &bo) {
#if defined(__clang__)
    unsigned b;
    uint32_t r = __builtin_subc(x, y, bi, &b);
    bo = b;
    return r;
#else
    uint64_t t = (uint64_t)x - y - bi;
    bo = (uint32_t)(t >> 63);
    return (uint32_t)t;
#endif
}

GF_HD fe fe_add(const fe &a, const fe &b) {
    uint32_t c, c2;
    fe r, t;
    r.w0 = gf_addc(a.w0, b.w0, 0u, c);
    r.w1 = gf_addc(a.w1, b.w1, c, c);
    r.w2 = gf_addc(a.w2, b.w2, c, c);
    r.w3 = gf_addc(a.w3, b.w3, c, c);
    // t = r - p (mod 2^128) = r + (9*2^32 - 1); it overflows exactly when r >= p
    t.w0 = gf_addc(r.w0, 0xFFFFFFFFu, 0u, c2);
    t.w1 = gf_addc(r.w1, 8u, c2, c2);
    t.w2 = gf_addc(r.w2, 0u, c2, c2);
    t.w3 = gf_addc(r.w3, 0u, c2, c2);
    bool sel = (c | c2) != 0;
    fe o;
    o.w0 = sel ? t.w0 : r.w0;
    o.w1 = sel ? t.w1 : r.w1;
    o.w2 = sel ? t.w2 : r.w2;
    o.w3 = sel ? t.w3 : r.w3;
    return o;
}

GF_HD fe fe_sub(const fe &a, const fe &b) {
    uint32_t bw, b2;
    fe r, t;
    r.w0 = gf_subc(a.w0, b.w0, 0u, bw);
    r.w1 = gf_subc(a.w1, b.w1, bw, bw);
    r.w2 = gf_subc(a.w2, b.w2, bw, bw);
    r.w3 = gf_subc(a.w3, b.w3, bw, bw);
    // on borrow add p back: r + p == r - (9*2^32 - 1) (mod 2^128)
    t.w0 = gf_subc(r.w0, 0xFFFFFFFFu, 0u, b2);
    t.w1 = gf_subc(r.w1, 8u, b2, b2);
    t.w2 = gf_subc(r.w2, 0u, b2, b2);
    t.w3 = gf_subc(r.w3, 0u, b2, b2);
    fe o;
    o.w0 = bw ? t.w0 : r.w0;
    o.w1 = bw ? t.w1 : r.w1;
    o.w2 = bw ? t.w2 : r.w2;
    o.w3 = bw ? t.w3 : r.w3;
    return o;
}

GF_HD fe fe_neg(const fe &a) { return fe_is_zero(a) ? a : fe_sub(fe_zero(), a); }

// one row of the schoolbook product: o (5 limbs) = ai * b.  Each v_mad_u64_u32 (32x32+64) absorbs the high word of the
// previous one as its addend (ai*bj + hi < 2^64 always), so a row is 4 dependent mads and NO carry-chain ops; the four
// rows of a product are independent of each other and give the scheduler its parallelism.
GF_HD void fe_row_mul(uint32_t ai, const uint32_t b[4], uint32_t o[5]) {
    uint64_t t = (uint64_t)ai * b[0];
    o[0] = (uint32_t)t;
    t = (uint64_t)ai * b[1] + (t >> 32);
    o[1] = (uint32_t)t;
    t = (uint64_t)ai * b[2] + (t >> 32);
    o[2] = (uint32_t)t;
    t = (uint64_t)ai * b[3] + (t >> 32);
    o[3] = (uint32_t)t;
    o[4] = (uint32_t)(t >> 32);
}

// 128x128 -> 256-bit schoolbook product, 32-bit limbs: 16 x v_mad_u64_u32 + 15 carry-chain ops (adding the rows up)
GF_HD void fe_mul_wide(const fe &a, const fe &b, uint32_t r[8]) {
    const uint32_t bv[4] = {b.w0, b.w1, b.w2, b.w3};
    uint32_t r0[5], r1[5], r2[5], r3[5], c;
    fe_row_mul(a.w0, bv, r0);
    fe_row_mul(a.w1, bv, r1);
    fe_row_mul(a.w2, bv, r2);
    fe_row_mul(a.w3, bv, r3);
    r[0] = r0[0];
    r[1] = gf_addc(r0[1], r1[0], 0u, c);
    r[2] = gf_addc(r0[2], r1[1], c, c);
    r[3] = gf_addc(r0[3], r1[2], c, c);
    r[4] = gf_addc(r0[4], r1[3], c, c);
    r[5] = r1[4] + c;
    r[2] = gf_addc(r[2], r2[0], 0u, c);
    r[3] = gf_addc(r[3], r2[1], c, c);
    r[4] = gf_addc(r[4], r2[2], c, c);
    r[5] = gf_addc(r[5], r2[3], c, c);
    r[6] = r2[4] + c;
    r[3] = gf_addc(r[3], r3[0], 0u, c);
    r[4] = gf_addc(r[4], r3[1], c, c);
    r[5] = gf_addc(r[5], r3[2], c, c);
    r[6] = gf_addc(r[6], r3[3], c, c);
    r[7] = r3[4] + c;
}

GF_HD uint32_t gf_fsh(uint32_t hi, uint32_t lo, int s) { return (hi << s) | (lo >> (32 - s)); }  // v_alignbit_b32

// reduce x = hi*2^128 + lo (any 256-bit value) to the canonical residue, ~41 full-rate VALU ops:
//   x == lo + (9*hi << 32) - hi                       (2^128 == 9*2^32 - 1)
//     =: V = (v0..v3) + T*2^128,  T = (v4, v5) < 2^37, V >= 0
//     == (v0..v3) + T*(9*2^32 - 1)                    (second fold, T*C < 2^74)
//     =: w + k*2^128, k in {0,1}; k = 1 leaves w tiny, so a single "+C / conditional -p" finishes.
GF_HD fe fe_reduce_wide(const uint32_t r[8]) {
    uint32_t c, b;
    // A = 9*hi, limbs a0..a4 (a4 < 9): four chained mads instead of shift + carry-add pairs
    uint64_t m = (uint64_t)r[4] * 9u;
    uint32_t a0 = (uint32_t)m;
    m = (uint64_t)r[5] * 9u + (m >> 32);
    uint32_t a1 = (uint32_t)m;
    m = (uint64_t)r[6] * 9u + (m >> 32);
    uint32_t a2 = (uint32_t)m;
    m = (uint64_t)r[7] * 9u + (m >> 32);
    uint32_t a3 = (uint32_t)m, a4 = (uint32_t)(m >> 32);
    // U = lo + (A << 32), six limbs
    uint32_t u1 = gf_addc(r[1], a0, 0u, c);
    uint32_t u2 = gf_addc(r[2], a1, c, c);
    uint32_t u3 = gf_addc(r[3], a2, c, c);
    uint32_t u4 = gf_addc(a3, 0u, c, c);
    uint32_t u5 = a4 + c;
    // V = U - hi  (>= 0 because U >= 9*hi*2^32)
    uint32_t v0 = gf_subc(r[0], r[4], 0u, b);
    uint32_t v1 = gf_subc(u1, r[5], b, b);
    uint32_t v2 = gf_subc(u2, r[6], b, b);
    uint32_t v3 = gf_subc(u3, r[7], b, b);
    uint32_t t0 = gf_subc(u4, 0u, b, b);
    uint32_t t1 = u5 - b;
    // G = 9*T (two limbs), E = T*C = (G << 32) - T (three limbs, >= 0)
    uint32_t g0 = gf_addc(t0 << 3, t0, 0u, c);
    uint32_t g1 = gf_fsh(t1, t0, 3) + t1 + c;
    uint32_t e0 = gf_subc(0u, t0, 0u, b);
    uint32_t e1 = gf_subc(g0, t1, b, b);
    uint32_t e2 = g1 - b;
    // w + k*2^128 = (v0..v3) + E
    fe w;
    uint32_t k;
    w.w0 = gf_addc(v0, e0, 0u, c);
    w.w1 = gf_addc(v1, e1, c, c);
    w.w2 = gf_addc(v2, e2, c, c);
    w.w3 = gf_addc(v3, 0u, c, k);
    // t = w + C (== w - p mod 2^128); take it when k is set or when it overflows (w >= p)
    fe t;
    t.w0 = gf_addc(w.w0, 0xFFFFFFFFu, 0u, c);
    t.w1 = gf_addc(w.w1, 8u, c, c);
    t.w2 = gf_addc(w.w2, 0u, c, c);
    t.w3 = gf_addc(w.w3, 0u, c, c);
    bool sel = (k | c) != 0;
    fe o;
    o.w0 = sel ? t.w0 : w.w0;
    o.w1 = sel ? t.w1 : w.w1;
    o.w2 = sel ? t.w2 : w.w2;
    o.w3 = sel ? t.w3 : w.w3;
    return o;
}

GF_HD fe fe_mul(const fe &a, const fe &b) {
    uint32_t r[8];
    fe_mul_wide(a, b, r);
    return fe_reduce_wide(r);
}
GF_HD fe fe_sqr(const fe &a) { return fe_mul(a, a); }

// b^e, e given as four 32-bit limbs (little endian)
GF_HD fe fe_pow(fe b, const fe &e) {
    fe r = fe_one();
    const uint32_t ev[4] = {e.w0, e.w1, e.w2, e.w3};
    for (int i = 0; i < 4; i++) {
        uint32_t w = ev[i];
        for (int k = 0; k < 32; k++) {
            if (w & 1u) r = fe_mul(r, b);
            b = fe_sqr(b);
            w >>= 1;
        }
    }
    return r;
}
GF_HD fe fe_pow_u64(fe b, uint64_t e) {          // b canonical.  No product by one at the start, no squaring after the top bit:
    if (!e) return fe_one();                      // x^3 is two products, x^5 three (the S-boxes of the example AIRs)
    fe r = b;
    bool have = false;
    for (;;) {
        if (e & 1u) { r = have ? fe_mul(r, b) : b; have = true; }
        e >>= 1;
        if (!e) break;
        b = fe_sqr(b);
    }
    return r;
}

// a^(p-2); p - 2 = 2^128 - 9*2^32 - 1 = [0xFFFFFFFF, 0xFFFFFFF6, 0xFFFFFFFF, 0xFFFFFFFF].  0 -> 0.
GF_HD fe fe_inv(const fe &a) {
    // x_k = a^(2^k - 1)
    fe x1 = a;
    fe x2 = fe_mul(fe_sqr(x1), x1);
    fe x4 = fe_mul(fe_sqr(fe_sqr(x2)), x2);
    fe x8 = x4;
    for (int i = 0; i < 4; i++) x8 = fe_sqr(x8);
    x8 = fe_mul(x8, x4);
    fe x16 = x8;
    for (int i = 0; i < 8; i++) x16 = fe_sqr(x16);
    x16 = fe_mul(x16, x8);
    fe x32 = x16;
    for (int i = 0; i < 16; i++) x32 = fe_sqr(x32);
    x32 = fe_mul(x32, x16);
    fe x64 = x32;
    for (int i = 0; i < 32; i++) x64 = fe_sqr(x64);
    x64 = fe_mul(x64, x32);
    // exponent bits, most significant first: 64 ones | 28 ones, 0110 | 32 ones
    fe r = x64;
    // next 32 bits: 0xFFFFFFF6 = 28 ones then 0,1,1,0
    fe x28 = x16;                      // build a^(2^28 - 1) = x16 * 2^12 ... : x16 -> shift 8 -> *x8 -> shift 4 -> *x4
    for (int i = 0; i < 8; i++) x28 = fe_sqr(x28);
    x28 = fe_mul(x28, x8);
    for (int i = 0; i < 4; i++) x28 = fe_sqr(x28);
    x28 = fe_mul(x28, x4);
    for (int i = 0; i < 28; i++) r = fe_sqr(r);
    r = fe_mul(r, x28);
    r = fe_sqr(r);                     // 0
    r = fe_sqr(r); r = fe_mul(r, a);   // 1
    r = fe_sqr(r); r = fe_mul(r, a);   // 1
    r = fe_sqr(r);                     // 0
    for (int i = 0; i < 32; i++) r = fe_sqr(r);
    r = fe_mul(r, x32);
    return r;
}
