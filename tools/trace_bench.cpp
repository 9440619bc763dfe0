// trace_bench.cpp — host-only latency experiments for the serial MiMC recurrence x <- x^3 + k (gs_mimc_trace).
// g++ -O3 -march=native tools/trace_bench.cpp -o tools/trace_bench && tools/trace_bench
#include <chrono>
#include <cstdio>
#include <vector>

#include "../genstark_amd/csrc/host_field.h"

// ---- experiment: x^3 straight from the limbs' cubes (no x0*x1 product, no normalised square), columns split once (no carry chain),
// every high column folded by a constant of its own weight (one level of products), then the usual small second fold
static inline hfe hf_cube_direct(hfe x) {
    typedef uint64_t u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80, K5 = 80ull * 0x8FFFFFFFFull;   // 2^256 == C21*2^64 + C20, 2^320 == C20*2^64 + K5
    const u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    const hfe p00 = (hfe)x0 * x0, p11 = (hfe)x1 * x1;
    const u64 l0 = (u64)p00, h0 = (u64)(p00 >> 64), l1 = (u64)p11, h1 = (u64)(p11 >> 64);
    const hfe aL = (hfe)l0 * x0, aH = (hfe)h0 * x0, bL = (hfe)l0 * x1, bH = (hfe)h0 * x1;
    const hfe cL = (hfe)l1 * x0, cH = (hfe)h1 * x0, dL = (hfe)l1 * x1, dH = (hfe)h1 * x1;
#define LO(v) ((hfe)(u64)(v))
#define HI(v) ((hfe)(u64)((v) >> 64))
    const hfe col1 = HI(aL) + LO(aH) + 3 * LO(bL);
    const hfe col2 = HI(aH) + 3 * (HI(bL) + LO(bH) + LO(cL));
    const hfe col3 = 3 * (HI(bH) + HI(cL) + LO(cH)) + LO(dL);
    const hfe col4 = 3 * HI(cH) + HI(dL) + LO(dH);
    const u64 e0 = (u64)aL, e1 = (u64)col1;
    u64 e2, e3, e4, e5;
    const bool o2 = __builtin_add_overflow((u64)col2, (u64)(col1 >> 64), &e2);
    const bool o3 = __builtin_add_overflow((u64)col3, (u64)(col2 >> 64), &e3);
    const bool o4 = __builtin_add_overflow((u64)col4, (u64)(col3 >> 64), &e4);
    const bool o5 = __builtin_add_overflow((u64)(dH >> 64), (u64)(col4 >> 64), &e5);
    if (__builtin_expect(o2 | o3 | o4 | o5, 0)) return hf_cube_weak(x);     // a column's low word within 12 of 2^64: probability ~2^-58
    // weight 1: e0 + e2*C + e4*C20 + e5*K5      weight 2^64: e1 + e3*C + e4*C21 + e5*C20
    const hfe m2 = (hfe)e2 * C, m3 = (hfe)e3 * C, m40 = (hfe)e4 * C20, m41 = (hfe)e4 * C21, m50 = (hfe)e5 * K5, m51 = (hfe)e5 * C20;
    // limb sums (each a handful of 64-bit terms: no overflow of the 128-bit accumulators)
    const hfe s0 = (hfe)e0 + LO(m2) + LO(m40) + LO(m50);
    const hfe s1 = (hfe)e1 + HI(m2) + HI(m40) + HI(m50) + LO(m3) + LO(m41) + LO(m51) + HI(s0);
    const hfe T = HI(m3) + HI(m41) + HI(m51) + HI(s1);                      // weight 2^128: < 2^67
    const hfe R = ((hfe)(u64)s1 << 64) | (u64)s0;
    const hfe TC = (hfe)(u64)T * C + (((hfe)(u64)(T >> 64) * C) << 64);
    hfe r = R + TC;
    if (__builtin_expect(r < R, 0)) r += HF_C;
    return r;
#undef LO
#undef HI
}

// ---- experiment: hf_cube_add_weak with the second fold started early: the part of T that does not wait for a1's carry is multiplied by C
// beside the limb sums; the carry (< 16) joins as carry * C
static inline hfe hf_cube_add_early(hfe x, hfe k) {
    typedef uint64_t u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80;
    u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    hfe p00 = (hfe)x0 * x0, p01 = (hfe)x0 * x1, p11 = (hfe)x1 * x1;
    u64 s0 = (u64)p00;
    hfe mid = (p00 >> 64) + ((hfe)(u64)p01 << 1);
    u64 s1 = (u64)mid;
    hfe up = (mid >> 64) + ((p01 >> 64) << 1) + (u64)p11;
    u64 s2 = (u64)up;
    u64 s3 = (u64)(up >> 64) + (u64)(p11 >> 64);
    hfe q00 = (hfe)s0 * x0, q10 = (hfe)s1 * x0, q20 = (hfe)s2 * x0, q30 = (hfe)s3 * x0;
    hfe q01 = (hfe)s0 * x1, q11 = (hfe)s1 * x1, q21 = (hfe)s2 * x1, q31 = (hfe)s3 * x1;
    u64 y0 = (u64)q00;
    hfe c1 = (q00 >> 64) + (u64)q10 + (u64)q01;
    u64 y1 = (u64)c1;
    hfe c2 = (c1 >> 64) + (q10 >> 64) + (q01 >> 64) + (u64)q20 + (u64)q11;
    u64 y2 = (u64)c2;
    hfe c3 = (c2 >> 64) + (q20 >> 64) + (q11 >> 64) + (u64)q30 + (u64)q21;
    u64 y3 = (u64)c3;
    hfe c4 = (c3 >> 64) + (q30 >> 64) + (q21 >> 64) + (u64)q31;
    u64 y4 = (u64)c4;
    u64 y5 = (u64)(c4 >> 64) + (u64)(q31 >> 64);
    hfe A = (hfe)y2 * C, B = (hfe)y3 * C, D = (hfe)y4 * C20, E = (hfe)y4 * C21, G = (hfe)y5 * C20, H = (hfe)y5 * C21;
    hfe T0 = (B >> 64) + (E >> 64) + (G >> 64) + H;                       // < 2^73, ready with the fold products
    hfe TC0 = (hfe)(u64)T0 * C + (((hfe)(u64)(T0 >> 64) * C) << 64);       // beside the limb sums below
    hfe a0 = (hfe)y0 + (u64)k + (u64)A + (u64)D;
    hfe a1 = (hfe)y1 + (u64)(k >> 64) + (u64)(A >> 64) + (u64)(D >> 64) + (u64)B + (u64)E + (u64)G + (u64)(a0 >> 64);
    hfe R = ((hfe)(u64)a1 << 64) | (u64)a0;
    hfe r = R + TC0;
    u64 wrap = r < R;                                                      // (probability 2^-18)
    hfe r2 = r + (hfe)((u64)(a1 >> 64) + 0) * C;                           // the carry of the limb sums: < 16
    const u64 wrap2 = r2 < r;
    if (__builtin_expect(wrap | wrap2, 0)) {                               // every wrap past 2^128 comes back as + C
        const hfe r3 = r2 + (hfe)(wrap + wrap2) * HF_C;
        r2 = r3 < r2 ? r3 + HF_C : r3;
    }
    return r2;
}

// ---- experiment: the last-arriving limb of the cube, y5 = hi(c4) + hi(q31), folded in two pieces: hi(q31) * C^2 starts as soon as the
// product q31 exists, the (tiny: < 16) carry hi(c4) joins y4's high side as carry * 2^64 * C^2 = carry * C^2 << 64 ... kept simple: the
// carry is multiplied too, but off the path of the big product
static inline hfe hf_cube_add_v2(hfe x, hfe k) {
    typedef uint64_t u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80;
    u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    hfe p00 = (hfe)x0 * x0, p01 = (hfe)x0 * x1, p11 = (hfe)x1 * x1;
    u64 s0 = (u64)p00;
    hfe mid = (p00 >> 64) + ((hfe)(u64)p01 << 1);
    u64 s1 = (u64)mid;
    hfe up = (mid >> 64) + ((p01 >> 64) << 1) + (u64)p11;
    u64 s2 = (u64)up;
    u64 s3 = (u64)(up >> 64) + (u64)(p11 >> 64);
    hfe q00 = (hfe)s0 * x0, q10 = (hfe)s1 * x0, q20 = (hfe)s2 * x0, q30 = (hfe)s3 * x0;
    hfe q01 = (hfe)s0 * x1, q11 = (hfe)s1 * x1, q21 = (hfe)s2 * x1, q31 = (hfe)s3 * x1;
    u64 y0 = (u64)q00;
    hfe c1 = (q00 >> 64) + (u64)q10 + (u64)q01;
    u64 y1 = (u64)c1;
    hfe c2 = (c1 >> 64) + (q10 >> 64) + (q01 >> 64) + (u64)q20 + (u64)q11;
    u64 y2 = (u64)c2;
    hfe c3 = (c2 >> 64) + (q20 >> 64) + (q11 >> 64) + (u64)q30 + (u64)q21;
    u64 y3 = (u64)c3;
    hfe c4 = (c3 >> 64) + (q30 >> 64) + (q21 >> 64) + (u64)q31;
    u64 y4 = (u64)c4;
    const u64 y5a = (u64)(q31 >> 64), y5b = (u64)(c4 >> 64);          // y5 = y5a + y5b, y5b < 16
    hfe A = (hfe)y2 * C, B = (hfe)y3 * C, D = (hfe)y4 * C20, E = (hfe)y4 * C21;
    hfe G = (hfe)y5a * C20, H = (hfe)y5a * C21;                          // (start with q31, not with the column sum)
    hfe Gb = (hfe)y5b * C20, Hb = (hfe)y5b * C21;
    hfe a0 = (hfe)y0 + (u64)k + (u64)A + (u64)D;
    hfe a1 = (hfe)y1 + (u64)(k >> 64) + (u64)(A >> 64) + (u64)(D >> 64) + (u64)B + (u64)E + (u64)G + (u64)Gb + (u64)(a0 >> 64);
    hfe T = (B >> 64) + (E >> 64) + (G >> 64) + (Gb >> 64) + H + Hb + (a1 >> 64);
    hfe R = ((hfe)(u64)a1 << 64) | (u64)a0;
    hfe TC = (hfe)(u64)T * C + (((hfe)(u64)(T >> 64) * C) << 64);
    hfe r = R + TC;
    if (__builtin_expect(r < R, 0)) r += HF_C;
    return r;
}
// ---- experiment: the doubled cross product from a doubled limb (2 x1 mod 2^64, the lost top bit paid back as x0 << 64): no 128-bit
// shifts between the square and the cube
static inline hfe hf_cube_add_v3(hfe x, hfe k) {
    typedef uint64_t u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80;
    u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    const u64 x1d = x1 << 1, top = x1 >> 63;
    hfe p00 = (hfe)x0 * x0, p01d = (hfe)x0 * x1d, p11 = (hfe)x1 * x1;   // 2 x0 x1 = p01d + top * x0 * 2^64
    u64 s0 = (u64)p00;
    hfe mid = (p00 >> 64) + (u64)p01d;
    u64 s1 = (u64)mid;
    hfe up = (mid >> 64) + (p01d >> 64) + (u64)p11 + ((0 - top) & x0);
    u64 s2 = (u64)up;
    u64 s3 = (u64)(up >> 64) + (u64)(p11 >> 64);
    hfe q00 = (hfe)s0 * x0, q10 = (hfe)s1 * x0, q20 = (hfe)s2 * x0, q30 = (hfe)s3 * x0;
    hfe q01 = (hfe)s0 * x1, q11 = (hfe)s1 * x1, q21 = (hfe)s2 * x1, q31 = (hfe)s3 * x1;
    u64 y0 = (u64)q00;
    hfe c1 = (q00 >> 64) + (u64)q10 + (u64)q01;
    u64 y1 = (u64)c1;
    hfe c2 = (c1 >> 64) + (q10 >> 64) + (q01 >> 64) + (u64)q20 + (u64)q11;
    u64 y2 = (u64)c2;
    hfe c3 = (c2 >> 64) + (q20 >> 64) + (q11 >> 64) + (u64)q30 + (u64)q21;
    u64 y3 = (u64)c3;
    hfe c4 = (c3 >> 64) + (q30 >> 64) + (q21 >> 64) + (u64)q31;
    u64 y4 = (u64)c4;
    u64 y5 = (u64)(c4 >> 64) + (u64)(q31 >> 64);
    hfe A = (hfe)y2 * C, B = (hfe)y3 * C, D = (hfe)y4 * C20, E = (hfe)y4 * C21, G = (hfe)y5 * C20, H = (hfe)y5 * C21;
    hfe a0 = (hfe)y0 + (u64)k + (u64)A + (u64)D;
    hfe a1 = (hfe)y1 + (u64)(k >> 64) + (u64)(A >> 64) + (u64)(D >> 64) + (u64)B + (u64)E + (u64)G + (u64)(a0 >> 64);
    hfe T = (B >> 64) + (E >> 64) + (G >> 64) + H + (a1 >> 64);
    hfe R = ((hfe)(u64)a1 << 64) | (u64)a0;
    hfe TC = (hfe)(u64)T * C + (((hfe)(u64)(T >> 64) * C) << 64);
    hfe r = R + TC;
    if (__builtin_expect(r < R, 0)) r += HF_C;
    return r;
}

// ---- experiment: the recurrence on AVX-512 IFMA (vpmadd52luq / vpmadd52huq).  Radix 2^44, three limbs per element,
// lane c of a zmm register = column c of a product.  The multiplier side is pre-shifted by 8 bits so that the 52-bit split of the
// instruction falls on the radix: lo52(a * (b << 8)) = ((a b) mod 2^44) << 8, hi52 = (a b) >> 44.  2^132 = 16 * (9 * 2^32 - 1) mod p is a
// 40-bit constant K3, so columns 3.. fold down with the same instruction pair (column c -> limbs c - 3, c - 2).  The multiplicand may be
// lazy (limbs < 2^52); the pre-shifted side must be strict (< 2^44), which is restored once per step through the scalar domain — a
// sequential carry chain across lanes costs more than the round trip.  What this form can NOT avoid, per step: two products, each
// broadcast/shift permutes (3 cycles) + a chain of vpmadd52 (4 cycles each, 3-4 deep unless split) + a fold (permute, vpmadd52 pair,
// permute, adds) — about 75 cycles of dependent latency against the scalar chain's 38: IFMA buys throughput over many independent chains,
// not latency on one.
#if defined(__x86_64__)
#include <immintrin.h>
#define IFMA_TARGET __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq")))
IFMA_TARGET static inline __m512i ifma_bcast(__m512i v, int t) { return _mm512_permutexvar_epi64(_mm512_set1_epi64(t), v); }
template <int S> IFMA_TARGET static inline __m512i ifma_up(__m512i v) { return _mm512_alignr_epi64(v, _mm512_setzero_si512(), 8 - S); }     // lane i <- lane i - S
template <int S> IFMA_TARGET static inline __m512i ifma_down(__m512i v) { return _mm512_alignr_epi64(_mm512_setzero_si512(), v, S); }       // lane i <- lane i + S
// columns of a (NA lazy limbs) times the strict element whose limbs << 8 are x8
template <int NA> IFMA_TARGET static inline __m512i ifma_mul(__m512i a, __m512i x8) {
    const __m512i z = _mm512_setzero_si512();
    __m512i L0 = _mm512_madd52lo_epu64(z, ifma_bcast(a, 0), x8), H0 = _mm512_madd52hi_epu64(z, ifma_bcast(a, 0), ifma_up<1>(x8));
    __m512i L1 = _mm512_madd52lo_epu64(z, ifma_bcast(a, 1), ifma_up<1>(x8)), H1 = _mm512_madd52hi_epu64(z, ifma_bcast(a, 1), ifma_up<2>(x8));
    L0 = _mm512_madd52lo_epu64(L0, ifma_bcast(a, 2), ifma_up<2>(x8));
    H0 = _mm512_madd52hi_epu64(H0, ifma_bcast(a, 2), ifma_up<3>(x8));
    if (NA > 3) {
        L1 = _mm512_madd52lo_epu64(L1, ifma_bcast(a, 3), ifma_up<3>(x8));
        H1 = _mm512_madd52hi_epu64(H1, ifma_bcast(a, 3), ifma_up<4>(x8));
    }
    return _mm512_add_epi64(_mm512_srli_epi64(_mm512_add_epi64(L0, L1), 8), _mm512_add_epi64(H0, H1));
}
// columns 3.. of P come down as column * K3: low 44 bits to limb c - 3, the rest to limb c - 2
IFMA_TARGET static inline __m512i ifma_fold(__m512i P) {
    const __m512i z = _mm512_setzero_si512();
    const __m512i K8 = _mm512_set1_epi64((long long)(((9ull << 32) - 1) * 16) << 8);
    const __m512i F = ifma_down<3>(P);
    const __m512i lo = _mm512_srli_epi64(_mm512_madd52lo_epu64(z, F, K8), 8), hi = ifma_up<1>(_mm512_madd52hi_epu64(z, F, K8));
    return _mm512_add_epi64(_mm512_add_epi64(_mm512_maskz_mov_epi64(0x07, P), lo), hi);
}
IFMA_TARGET static double run_ifma(uint64_t steps, const std::vector<hfe> &rc, hfe seed, std::vector<hfe> &t) {
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t M44 = (1ull << 44) - 1;
    hfe x = seed;
    uint32_t ri = 0, nrc = (uint32_t)rc.size();
    for (uint64_t i = 0; i < steps; i++) {
        t[i] = hf_mimc_out(x);
        const uint64_t x0 = (uint64_t)x & M44, x1 = (uint64_t)(x >> 44) & M44, x2 = (uint64_t)(x >> 88);          // strict limbs (x2 < 2^40)
        const __m512i X = _mm512_set_epi64(0, 0, 0, 0, 0, (long long)x2, (long long)x1, (long long)x0), X8 = _mm512_slli_epi64(X, 8);
        __m512i S = ifma_fold(ifma_mul<3>(X, X8));             // x^2: limbs 0..3 lazy (< 2^48), limb 3 = what column 5 left above 2^132
        __m512i Y = ifma_fold(ifma_mul<4>(S, X8));             // x^3: limbs 0..4
        Y = ifma_fold(Y);                                      // limbs 3, 4 once more: limbs 0..2, each < 2^52
        alignas(64) uint64_t y[8];
        _mm512_store_si512((__m512i *)y, Y);
        // back to a weak 128-bit value: y0 + y1 2^44 + y2 2^88, the part of y2 above 2^40 is a multiple of 2^128 == C
        hfe v = (hfe)y[0] + ((hfe)y[1] << 44);
        v = hf_add_weak(v, (hfe)(y[2] & ((1ull << 40) - 1)) << 88);
        v = hf_add_weak(v, (hfe)(y[2] >> 40) * HF_C);
        x = hf_add_weak(v, rc[ri]);
        if (++ri == nrc) ri = 0;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count();
}
#endif

// ---- round 5: the chain's step with the carry chains written out by rows is hf_cube_add_rows of host_field.h (adopted for BMI2 cores);
// the variants below start from it
#if defined(__x86_64__)
// the same, and the fold's sums written so that what arrives LAST is added last: the terms that wait for y4 after the ones that do not, the
// terms that wait for y5 (the end of the carry chain) after those; FOLD = 1: additionally the part of T that does not wait for y5 is
// multiplied by C beside the rest (T = Te + Tl, T C = Te C + Tl C: the second product starts from y5's terms alone)
template <int FOLD>
__attribute__((target("bmi2"))) static inline hfe hf_cube_add_rows2(hfe x, hfe k) {
    typedef unsigned long long u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80;
    const u64 x0 = (u64)x, x1 = (u64)(x >> 64), k0 = (u64)k, k1 = (u64)(k >> 64);
    u64 a0, a1, b0, b1, c0, c1;
    a0 = _mulx_u64(x0, x0, &a1);
    b0 = _mulx_u64(x0, x1, &b1);
    c0 = _mulx_u64(x1, x1, &c1);
    u64 d0, d1, d2, s1, s2, s3;
    unsigned char cy;
    cy = _addcarry_u64(0, b0, b0, &d0);
    cy = _addcarry_u64(cy, b1, b1, &d1);
    d2 = cy;
    cy = _addcarry_u64(0, a1, d0, &s1);
    cy = _addcarry_u64(cy, c0, d1, &s2);
    (void)_addcarry_u64(cy, c1, d2, &s3);
    const u64 s0 = a0;
    u64 p0l, p0h, p1l, p1h, p2l, p2h, p3l, p3h, q0l, q0h, q1l, q1h, q2l, q2h, q3l, q3h;
    p0l = _mulx_u64(s0, x0, &p0h); q0l = _mulx_u64(s0, x1, &q0h);
    p1l = _mulx_u64(s1, x0, &p1h); q1l = _mulx_u64(s1, x1, &q1h);
    p2l = _mulx_u64(s2, x0, &p2h); q2l = _mulx_u64(s2, x1, &q2h);
    p3l = _mulx_u64(s3, x0, &p3h); q3l = _mulx_u64(s3, x1, &q3h);
    u64 r1, r2, r3, r4, t2, t3, t4, t5;
    cy = _addcarry_u64(0, p0h, p1l, &r1);
    cy = _addcarry_u64(cy, p1h, p2l, &r2);
    cy = _addcarry_u64(cy, p2h, p3l, &r3);
    (void)_addcarry_u64(cy, p3h, 0, &r4);
    cy = _addcarry_u64(0, q0h, q1l, &t2);
    cy = _addcarry_u64(cy, q1h, q2l, &t3);
    cy = _addcarry_u64(cy, q2h, q3l, &t4);
    (void)_addcarry_u64(cy, q3h, 0, &t5);
    u64 y1, y2, y3, y4, y5;
    const u64 y0 = p0l;
    cy = _addcarry_u64(0, r1, q0l, &y1);
    cy = _addcarry_u64(cy, r2, t2, &y2);
    cy = _addcarry_u64(cy, r3, t3, &y3);
    cy = _addcarry_u64(cy, r4, t4, &y4);
    (void)_addcarry_u64(cy, t5, 0, &y5);
    u64 Al, Ah, Bl, Bh, Dl, Dh, El, Eh, Gl, Gh, Hl, Hh;
    Al = _mulx_u64(y2, C, &Ah); Bl = _mulx_u64(y3, C, &Bh);
    Dl = _mulx_u64(y4, C20, &Dh); El = _mulx_u64(y4, C21, &Eh);
    Gl = _mulx_u64(y5, C20, &Gh); Hl = _mulx_u64(y5, C21, &Hh);
    // column 0: y0 + k0 + Al | + Dl           column 1: y1 + k1 + Ah + Bl | + Dh + El | + Gl          column 2..: Bh | + Eh | + Gh + H
    const hfe e0 = (hfe)y0 + k0 + Al;
    const hfe e1 = (hfe)y1 + k1 + Ah + Bl;
    const hfe a0s = e0 + Dl;
    const hfe m1 = e1 + Dh + El + (u64)(a0s >> 64);
    const hfe a1s = m1 + Gl;
    const hfe R = ((hfe)(u64)a1s << 64) | (u64)a0s;
    hfe r;
    if (FOLD == 0) {
        const hfe T = ((hfe)Bh + Eh) + (Gh + (((hfe)Hh << 64) | Hl)) + (u64)(a1s >> 64);
        u64 Tl, Th;
        Tl = _mulx_u64((u64)T, C, &Th);
        const hfe TC = (((hfe)Th << 64) | Tl) + (((hfe)((u64)(T >> 64) * C)) << 64);
        r = R + TC;
    } else {
        const hfe Te = (hfe)Bh + Eh + (u64)(m1 >> 64);                              // < 2^66: everything that does not wait for y5
        const hfe Tl_ = (hfe)Gh + (((hfe)Hh << 64) | Hl) + (u64)(a1s >> 64) - (u64)(m1 >> 64);   // y5's terms + the carry Gl caused (0 or 1 more than m1's)
        u64 el, eh, ll, lh;
        el = _mulx_u64((u64)Te, C, &eh);
        ll = _mulx_u64((u64)Tl_, C, &lh);
        const hfe TeC = (((hfe)eh << 64) | el) + (((hfe)((u64)(Te >> 64) * C)) << 64);
        const hfe TlC = (((hfe)lh << 64) | ll) + (((hfe)((u64)(Tl_ >> 64) * C)) << 64);
        const hfe r1_ = R + TeC;                                                       // Te C < 2^102, Tl C < 2^110: at most one wrap in total (checked below)
        r = r1_ + TlC;
        if (__builtin_expect(r1_ < R, 0)) r += HF_C;
        if (__builtin_expect(r < TlC, 0)) r += HF_C;
        return r;
    }
    if (__builtin_expect(r < R, 0)) r += HF_C;
    return r;
}
// the multiplier port is on the critical path (19 mulx + 1 imul per step, one issued per cycle): SH80 = the two products by 80 (the high
// limb of C^2) as shifts and adds; FINAL = the small product T1 * C joins the limb it lands on before the last product arrives; SHC = the
// two products by C = 2^35 + 2^32 - 1 of the first fold as shifts and adds too
template <int SH80, int FINAL, int SHC>
__attribute__((target("bmi2"))) static inline hfe hf_cube_add_rows3(hfe x, hfe k) {
    typedef unsigned long long u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80;
    const u64 x0 = (u64)x, x1 = (u64)(x >> 64), k0 = (u64)k, k1 = (u64)(k >> 64);
    u64 a0, a1, b0, b1, c0, c1;
    a0 = _mulx_u64(x0, x0, &a1);
    b0 = _mulx_u64(x0, x1, &b1);
    c0 = _mulx_u64(x1, x1, &c1);
    u64 d0, d1, d2, s1, s2, s3;
    unsigned char cy;
    cy = _addcarry_u64(0, b0, b0, &d0);
    cy = _addcarry_u64(cy, b1, b1, &d1);
    d2 = cy;
    cy = _addcarry_u64(0, a1, d0, &s1);
    cy = _addcarry_u64(cy, c0, d1, &s2);
    (void)_addcarry_u64(cy, c1, d2, &s3);
    const u64 s0 = a0;
    u64 p0l, p0h, p1l, p1h, p2l, p2h, p3l, p3h, q0l, q0h, q1l, q1h, q2l, q2h, q3l, q3h;
    p0l = _mulx_u64(s0, x0, &p0h); q0l = _mulx_u64(s0, x1, &q0h);
    p1l = _mulx_u64(s1, x0, &p1h); q1l = _mulx_u64(s1, x1, &q1h);
    p2l = _mulx_u64(s2, x0, &p2h); q2l = _mulx_u64(s2, x1, &q2h);
    p3l = _mulx_u64(s3, x0, &p3h); q3l = _mulx_u64(s3, x1, &q3h);
    u64 r1, r2, r3, r4, t2, t3, t4, t5;
    cy = _addcarry_u64(0, p0h, p1l, &r1);
    cy = _addcarry_u64(cy, p1h, p2l, &r2);
    cy = _addcarry_u64(cy, p2h, p3l, &r3);
    (void)_addcarry_u64(cy, p3h, 0, &r4);
    cy = _addcarry_u64(0, q0h, q1l, &t2);
    cy = _addcarry_u64(cy, q1h, q2l, &t3);
    cy = _addcarry_u64(cy, q2h, q3l, &t4);
    (void)_addcarry_u64(cy, q3h, 0, &t5);
    u64 y1, y2, y3, y4, y5;
    const u64 y0 = p0l;
    cy = _addcarry_u64(0, r1, q0l, &y1);
    cy = _addcarry_u64(cy, r2, t2, &y2);
    cy = _addcarry_u64(cy, r3, t3, &y3);
    cy = _addcarry_u64(cy, r4, t4, &y4);
    (void)_addcarry_u64(cy, t5, 0, &y5);
    u64 Al, Ah, Bl, Bh, Dl, Dh, El, Eh, Gl, Gh, Hl, Hh;
    if (SHC) {
        const hfe A = ((hfe)y2 << 35) + ((hfe)y2 << 32) - y2, B = ((hfe)y3 << 35) + ((hfe)y3 << 32) - y3;
        Al = (u64)A; Ah = (u64)(A >> 64); Bl = (u64)B; Bh = (u64)(B >> 64);
    } else {
        Al = _mulx_u64(y2, C, &Ah); Bl = _mulx_u64(y3, C, &Bh);
    }
    Dl = _mulx_u64(y4, C20, &Dh);
    Gl = _mulx_u64(y5, C20, &Gh);
    if (SH80) {
        const hfe E = ((hfe)y4 << 6) + ((hfe)y4 << 4), H = ((hfe)y5 << 6) + ((hfe)y5 << 4);
        El = (u64)E; Eh = (u64)(E >> 64); Hl = (u64)H; Hh = (u64)(H >> 64);
    } else {
        El = _mulx_u64(y4, C21, &Eh); Hl = _mulx_u64(y5, C21, &Hh);
    }
    const hfe e0 = (hfe)y0 + k0 + Al, e1 = (hfe)y1 + k1 + Ah + Bl;
    const hfe a0s = e0 + Dl;
    const hfe a1s = e1 + Dh + El + (u64)(a0s >> 64) + Gl;
    const hfe T = ((hfe)Bh + Eh) + (Gh + (((hfe)Hh << 64) | Hl)) + (u64)(a1s >> 64);
    u64 Tl, Th;
    Tl = _mulx_u64((u64)T, C, &Th);
    if (FINAL) {
        // r = (a1s_lo + T1 C + Th) 2^64 + (a0s_lo + Tl): the parts that do not wait for the last product are summed first
        const u64 T1C = (u64)(T >> 64) * C;                        // T1 < 2^9: the product fits
        u64 hi_early, lo, hi;
        const unsigned char w1 = _addcarry_u64(0, (u64)a1s, T1C, &hi_early);
        unsigned char c0_ = _addcarry_u64(0, (u64)a0s, Tl, &lo);
        const unsigned char w2 = _addcarry_u64(c0_, hi_early, Th, &hi);
        hfe r = ((hfe)hi << 64) | lo;
        if (__builtin_expect(w1 | w2, 0)) r += HF_C;               // one wrap past 2^128 at most (R + T C < 2^128 + 2^110)
        return r;
    }
    const hfe R = ((hfe)(u64)a1s << 64) | (u64)a0s;
    const hfe TC = (((hfe)Th << 64) | Tl) + (((hfe)((u64)(T >> 64) * C)) << 64);
    hfe r = R + TC;
    if (__builtin_expect(r < R, 0)) r += HF_C;
    return r;
}
template <int SH80, int FINAL, int SHC>
__attribute__((target("bmi2"))) static double run_rows3(uint64_t steps, const std::vector<hfe> &rc, hfe seed, std::vector<hfe> &t) {
    auto t0 = std::chrono::steady_clock::now();
    hfe x = seed;
    uint32_t ri = 0, nrc = (uint32_t)rc.size();
    for (uint64_t i = 0; i < steps; i++) {
        t[i] = hf_mimc_out(x);
        x = hf_cube_add_rows3<SH80, FINAL, SHC>(x, rc[ri]);
        if (++ri == nrc) ri = 0;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count();
}
template <int FOLD>
__attribute__((target("bmi2"))) static double run_rows2(uint64_t steps, const std::vector<hfe> &rc, hfe seed, std::vector<hfe> &t) {
    auto t0 = std::chrono::steady_clock::now();
    hfe x = seed;
    uint32_t ri = 0, nrc = (uint32_t)rc.size();
    for (uint64_t i = 0; i < steps; i++) {
        t[i] = hf_mimc_out(x);
        x = hf_cube_add_rows2<FOLD>(x, rc[ri]);
        if (++ri == nrc) ri = 0;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count();
}
__attribute__((target("bmi2,adx"))) static double run_rows(uint64_t steps, const std::vector<hfe> &rc, hfe seed, std::vector<hfe> &t) {
    auto t0 = std::chrono::steady_clock::now();
    hfe x = seed;
    uint32_t ri = 0, nrc = (uint32_t)rc.size();
    for (uint64_t i = 0; i < steps; i++) {
        t[i] = hf_mimc_out(x);
        x = hf_cube_add_rows(x, rc[ri]);
        if (++ri == nrc) ri = 0;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count();
}
#endif

template <int V>
static double run(uint64_t steps, const std::vector<hfe> &rc, hfe seed, std::vector<hfe> &t) {
    auto t0 = std::chrono::steady_clock::now();
    hfe x = seed;
    uint32_t ri = 0, nrc = (uint32_t)rc.size();
    for (uint64_t i = 0; i < steps; i++) {
        if (V & 8) {          // the product's chain since round 4: k joins the cube's first fold, weak chain, canonical value beside it
            t[i] = hf_mimc_out(x);
            x = (V & 64) ? hf_cube_add_v3(x, rc[ri]) : (V & 32) ? hf_cube_add_v2(x, rc[ri]) : (V & 16) ? hf_cube_add_early(x, rc[ri]) : hf_mimc_step_weak(x, rc[ri]);
            if (++ri == nrc) ri = 0;
            continue;
        }
        hfe y = (V & 4) ? hf_cube_direct(x) : (V & 2) ? hf_cube_weak(x) : hf_mul_weak(hf_mul_weak(x, x), x);
        hfe sum = y + rc[ri];
        if (V & 1) {          // weak chain, canonicalisation off the critical path
            t[i] = hf_canon(x);
            if (__builtin_expect(sum < y, 0)) sum += HF_C;
            x = sum;
        } else {              // canonical chain
            t[i] = x;
            if (sum < y) sum += HF_C;
            x = hf_canon(sum);
        }
        if (++ri == nrc) ri = 0;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count();
}

int main() {
    const uint64_t steps = 1 << 20;
    std::vector<hfe> rc(64), t0(steps), t1(steps), t2(steps), t3(steps);
    for (int i = 0; i < 64; i++) rc[i] = hf_pow(3, 1000 + i);
    hfe seed = 3;
    for (int rep = 0; rep < 4; rep++) {
        double a = run<0>(steps, rc, seed, t0), b = run<1>(steps, rc, seed, t1), c = run<2>(steps, rc, seed, t2), d = run<3>(steps, rc, seed, t3);
        bool ok = true;
        for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t1[i] && t0[i] == t2[i] && t0[i] == t3[i];
        double e = run<4>(steps, rc, seed, t1), f = run<5>(steps, rc, seed, t3);
        for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t1[i] && t0[i] == t3[i];
        double g8 = run<8>(steps, rc, seed, t2);
        for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
        double g24 = run<24>(steps, rc, seed, t2);
        for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
        double g40 = run<40>(steps, rc, seed, t2);
        for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
        double g72 = run<72>(steps, rc, seed, t2);
        for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
        printf("2^20 steps: cube + k in one fold, weak chain %.2f ms | + second fold started early %.2f ms | y5 in two pieces %.2f ms | doubled limb %.2f ms\n", g8, g24, g40, g72);
#if defined(__x86_64__)
        if (__builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx")) {
            double gr = run_rows(steps, rc, seed, t2);
            for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
            double g2 = run_rows2<0>(steps, rc, seed, t2);
            for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
            double g3 = run_rows2<1>(steps, rc, seed, t2);
            for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
            printf("2^20 steps: carry chains by rows (mulx + adc) %.2f ms | + fold sums ordered by arrival %.2f ms | + second fold split early / late %.2f ms\n", gr, g2, g3);
            double h1 = run_rows3<1, 0, 0>(steps, rc, seed, t2);
            for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
            double h2 = run_rows3<0, 1, 0>(steps, rc, seed, t2);
            for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
            double h3 = run_rows3<1, 1, 0>(steps, rc, seed, t2);
            for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
            double h4 = run_rows3<1, 1, 1>(steps, rc, seed, t2);
            for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
            printf("2^20 steps: rows, x80 by shifts %.2f ms | rows, last sum reordered %.2f ms | both %.2f ms | both + xC by shifts %.2f ms\n", h1, h2, h3, h4);
        }
        if (__builtin_cpu_supports("avx512ifma")) {
            double gi = run_ifma(steps, rc, seed, t2);
            for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t2[i];
            printf("2^20 steps: AVX-512 IFMA form (radix 2^44, vpmadd52, one chain) %.2f ms\n", gi);
        } else {
            printf("2^20 steps: AVX-512 IFMA form: this CPU has no avx512ifma\n");
        }
#endif
        printf("2^20 steps: A/canon-chain %.2f ms | A/weak-chain %.2f ms | cube/canon-chain %.2f ms | cube/weak-chain %.2f ms | direct/canon %.2f ms | direct/weak %.2f ms  (%s)\n", a, b, c, d, e, f,
               ok ? "all equal" : "DIFF");
    }
    return 0;
}
