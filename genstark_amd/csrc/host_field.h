// host_field.h — GF(p) arithmetic on native 64-bit limbs for the few HOST-side pieces of the library:
// the serial MiMC recurrence (air_mimc.hip) and O(n^2) Lagrange interpolation of <= a few hundred points
// (small.hip).  Same modulus and canonical representation as gf128.h.
#pragma once
#if defined(GS_SMALL_Q)
#include "host_field_small.h"
#elif defined(GS_WIDE_BITS)
#include "host_field_wide.h"
#else
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 hfe;

static inline hfe hf_p() { return ((hfe)0xFFFFFFFFFFFFFFFFull << 64) | 0xFFFFFFF700000001ull; }
static const hfe HF_C = (hfe)0x8FFFFFFFFull;  // 2^128 mod p

static inline hfe hf_reduce(hfe hi, hfe lo) {
    hfe m0 = (hfe)(uint64_t)hi * HF_C, m1 = (hfe)(uint64_t)(hi >> 64) * HF_C;
    hfe tl = m0 + (m1 << 64);
    hfe th = (m1 >> 64) + (tl < m0);
    hfe s = tl + lo;
    unsigned k = s < tl;
    hfe s2 = s + th * HF_C;
    k += s2 < s;
    while (k) { hfe s3 = s2 + HF_C; k -= 1; k += s3 < s2; s2 = s3; }
    while (s2 >= hf_p()) s2 -= hf_p();
    return s2;
}
static inline hfe hf_mul(hfe a, hfe b) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    hfe p00 = (hfe)a0 * b0, p01 = (hfe)a0 * b1, p10 = (hfe)a1 * b0, p11 = (hfe)a1 * b1;
    hfe mid = p01 + p10, midc = mid < p01;
    hfe lo = p00 + (mid << 64), c1 = lo < p00;
    hfe hi = p11 + (mid >> 64) + (midc << 64) + c1;
    return hf_reduce(hi, lo);
}
// Weakly reduced product for latency-bound serial chains (the MiMC recurrence): returns ANY 128-bit representative of
// a*b mod p (inputs may be any 128-bit values).  All-additive fold, no data-dependent branches (the sign fix-ups of a
// shift/subtract fold mispredict every other product): with C = 2^128 mod p and hi = h1*2^64 + h0,
//   a*b = lo + h0*C + ((h1*C mod 2^64) << 64) + ((h1*C >> 64) + carries) * C        (mod p)
// and the final compare-and-subtract is skipped, which shortens the dependency chain of x -> x^2 -> x^3.
static inline hfe hf_mul_weak(hfe a, hfe b) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    hfe p00 = (hfe)a0 * b0, p01 = (hfe)a0 * b1, p10 = (hfe)a1 * b0, p11 = (hfe)a1 * b1;
    hfe mid = p01 + p10;
    uint64_t midc = mid < p01;
    hfe lo = p00 + (mid << 64);
    uint64_t c1 = lo < p00;
    hfe hi = p11 + (mid >> 64) + ((hfe)midc << 64) + c1;
    const uint64_t cc = (uint64_t)HF_C;
    hfe m0 = (hfe)(uint64_t)hi * cc, m1 = (hfe)(uint64_t)(hi >> 64) * cc;   // each < 2^100
    hfe t = m0 + (m1 << 64);
    uint64_t k = t < m0;
    hfe s = lo + t;
    k += s < lo;
    hfe top = (hfe)((uint64_t)(m1 >> 64) + k) * cc;                              // (< 2^37) * C < 2^73
    hfe r = s + top;
    if (__builtin_expect(r < s, 0)) r += HF_C;                                         // probability ~2^-55
    return r;
}
// Weak x^3 for the MiMC recurrence in one go: 256-bit square, 384-bit product, ONE fold of the upper 256 bits
// (2^128 == C, 2^256 == C^2 mod p).  10% shorter dependency chain than two hf_mul_weak on Zen 5 (tools/trace_bench.cpp).
static inline hfe hf_cube_weak(hfe x) {
    typedef uint64_t u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80;   // C^2 = 81*2^64 - 18*2^32 + 1 = C21*2^64 + C20
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
    hfe a0 = (hfe)y0 + (u64)A + (u64)D;
    hfe a1 = (hfe)y1 + (u64)(A >> 64) + (u64)(D >> 64) + (u64)B + (u64)E + (u64)G + (u64)(a0 >> 64);
    hfe T = (B >> 64) + (E >> 64) + (G >> 64) + H + (a1 >> 64);          // < 2^73
    hfe R = ((hfe)(u64)a1 << 64) | (u64)a0;
    hfe TC = (hfe)(u64)T * C + (((hfe)(u64)(T >> 64) * C) << 64);
    hfe r = R + TC;
    if (__builtin_expect(r < R, 0)) r += HF_C;
    return r;
}

static inline hfe hf_canon(hfe x) {
    while (x >= hf_p()) x -= hf_p();
    return x;
}
// products inside a long chain whose end is canonicalised once (host_pow in air_vm.hip)
#define HF_CHAIN_MUL hf_mul_weak
#define HF_CHAIN_END hf_canon
// sum of two weak values (any 128-bit representatives), weak again: a wrap past 2^128 comes back as + C (2^128 == C mod p), branch-free.
// The wrapped sum can be as large as 2^128 - 2, so adding C may wrap ONCE more (when both operands are within ~2^36 of 2^128); that
// second wrap leaves a value below C and is paid back the same way — a third is impossible.
static inline hfe hf_add_weak(hfe a, hfe b) {
    const hfe s = a + b;
    const hfe t = s + (((hfe)0 - (hfe)(s < a)) & HF_C);
    return t + (((hfe)0 - (hfe)(t < s)) & HF_C);
}
#define HF_CHAIN_ADD hf_add_weak
static inline hfe hf_add(hfe a, hfe b) {
    hfe s = a + b;
    if (s < a || s >= hf_p()) s -= hf_p();
    return s;
}
static inline hfe hf_sub(hfe a, hfe b) { return a >= b ? a - b : a - b + hf_p(); }
static inline hfe hf_pow(hfe b, hfe e) {
    hfe r = 1;
    while (e) {
        if (e & 1) r = hf_mul(r, b);
        b = hf_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline hfe hf_inv(hfe a) { return a ? hf_pow(a, hf_p() - 2) : 0; }
static inline bool hf_is_zero(hfe a) { return a == 0; }
static inline hfe hf_load(const uint8_t *b) { hfe v; memcpy(&v, b, 16); return v; }
static inline void hf_store(uint8_t *b, hfe v) { memcpy(b, &v, 16); }
// x^3 + k in one go, weak in and weak out (any 128-bit representatives; k < 2^128): hf_cube_weak with k's two limbs joining the limb sums
// of the first fold.  A separate "+ k" after the cube wraps past 2^128 every other step (k is a uniform residue) — a select or a
// mispredicted branch on the chain — and a canonical chain adds its compare-and-subtract to every step; here the constant costs no
// step of the dependency chain at all (the sums it joins have other terms arriving later), the only fix-up left is the one-in-2^55
// wrap of R + T*C, and the canonical value the trace stores is computed beside the chain (hf_mimc_out), not on it.
// (always_inline: gs_mimc_trace compiles its loop a second time for BMI2 cores — a function with other target attributes inlines this only when told to)
__attribute__((always_inline)) static inline hfe hf_cube_add_weak(hfe x, hfe k) {
    typedef uint64_t u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80;   // C^2 = 81*2^64 - 18*2^32 + 1 = C21*2^64 + C20
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
    hfe a0 = (hfe)y0 + (u64)k + (u64)A + (u64)D;                                                       // (four 64-bit terms: < 2^66)
    hfe a1 = (hfe)y1 + (u64)(k >> 64) + (u64)(A >> 64) + (u64)(D >> 64) + (u64)B + (u64)E + (u64)G + (u64)(a0 >> 64);
    hfe T = (B >> 64) + (E >> 64) + (G >> 64) + H + (a1 >> 64);          // < 2^73
    hfe R = ((hfe)(u64)a1 << 64) | (u64)a0;
    hfe TC = (hfe)(u64)T * C + (((hfe)(u64)(T >> 64) * C) << 64);
    hfe r = R + TC;
    if (__builtin_expect(r < R, 0)) r += HF_C;
    return r;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
// The same step with the carry chains written out (round 5): mulx products, the 256-bit square and the 384-bit cube summed ROW by row as
// adc chains instead of the column sums over 128-bit temporaries above — the same algorithm and the same values (any 128-bit
// representative in, one out), but the compiler turns the 128-bit form into setc / movzx / add-adc pairs on the dependency chain, and the
// chain is all there is: 7.3 ms instead of 8.0 ms per 2^20 steps on the GPU box's EPYC 9575F (tools/trace_bench.cpp,
// profiles/r05_c_*; no difference on the build container's Xeon).  Needs BMI2: used by the chain's build for such cores only.
__attribute__((always_inline, target("bmi2"))) static inline hfe hf_cube_add_rows(hfe x, hfe k) {
    typedef unsigned long long u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80;   // C^2 = 81*2^64 - 18*2^32 + 1 = C21*2^64 + C20
    const u64 x0 = (u64)x, x1 = (u64)(x >> 64), k0 = (u64)k, k1 = (u64)(k >> 64);
    u64 a0, a1, b0, b1, c0, c1;
    a0 = _mulx_u64(x0, x0, &a1);
    b0 = _mulx_u64(x0, x1, &b1);
    c0 = _mulx_u64(x1, x1, &c1);
    u64 d0, d1, d2, s1, s2, s3;
    unsigned char cy;
    cy = _addcarry_u64(0, b0, b0, &d0);                   // d = 2 b (129 bits)
    cy = _addcarry_u64(cy, b1, b1, &d1);
    d2 = cy;
    cy = _addcarry_u64(0, a1, d0, &s1);                   // s = x^2 = a + d 2^64 + c 2^128
    cy = _addcarry_u64(cy, c0, d1, &s2);
    (void)_addcarry_u64(cy, c1, d2, &s3);                 // (x^2 < 2^256: no carry out)
    const u64 s0 = a0;
    u64 p0l, p0h, p1l, p1h, p2l, p2h, p3l, p3h, q0l, q0h, q1l, q1h, q2l, q2h, q3l, q3h;
    p0l = _mulx_u64(s0, x0, &p0h); q0l = _mulx_u64(s0, x1, &q0h);
    p1l = _mulx_u64(s1, x0, &p1h); q1l = _mulx_u64(s1, x1, &q1h);
    p2l = _mulx_u64(s2, x0, &p2h); q2l = _mulx_u64(s2, x1, &q2h);
    p3l = _mulx_u64(s3, x0, &p3h); q3l = _mulx_u64(s3, x1, &q3h);
    u64 r1, r2, r3, r4, t2, t3, t4, t5;
    cy = _addcarry_u64(0, p0h, p1l, &r1);                 // row 0: s * x0 = (r4 r3 r2 r1 p0l)
    cy = _addcarry_u64(cy, p1h, p2l, &r2);
    cy = _addcarry_u64(cy, p2h, p3l, &r3);
    (void)_addcarry_u64(cy, p3h, 0, &r4);
    cy = _addcarry_u64(0, q0h, q1l, &t2);                 // row 1: s * x1 = (t5 t4 t3 t2 q0l), one limb up
    cy = _addcarry_u64(cy, q1h, q2l, &t3);
    cy = _addcarry_u64(cy, q2h, q3l, &t4);
    (void)_addcarry_u64(cy, q3h, 0, &t5);
    u64 y1, y2, y3, y4, y5;
    const u64 y0 = p0l;
    cy = _addcarry_u64(0, r1, q0l, &y1);                  // x^3 = (y5 .. y0)
    cy = _addcarry_u64(cy, r2, t2, &y2);
    cy = _addcarry_u64(cy, r3, t3, &y3);
    cy = _addcarry_u64(cy, r4, t4, &y4);
    (void)_addcarry_u64(cy, t5, 0, &y5);
    u64 Al, Ah, Bl, Bh, Dl, Dh, El, Eh, Gl, Gh, Hl, Hh;
    Al = _mulx_u64(y2, C, &Ah); Bl = _mulx_u64(y3, C, &Bh);
    Dl = _mulx_u64(y4, C20, &Dh); El = _mulx_u64(y4, C21, &Eh);
    Gl = _mulx_u64(y5, C20, &Gh); Hl = _mulx_u64(y5, C21, &Hh);
    // the fold of hf_cube_add_weak, the terms that arrive last added last
    const hfe e0 = (hfe)y0 + k0 + Al, e1 = (hfe)y1 + k1 + Ah + Bl;
    const hfe a0s = e0 + Dl;
    const hfe a1s = e1 + Dh + El + (u64)(a0s >> 64) + Gl;
    const hfe T = ((hfe)Bh + Eh) + (Gh + (((hfe)Hh << 64) | Hl)) + (u64)(a1s >> 64);          // < 2^73
    const hfe R = ((hfe)(u64)a1s << 64) | (u64)a0s;
    u64 Tl, Th;
    Tl = _mulx_u64((u64)T, C, &Th);
    const hfe TC = (((hfe)Th << 64) | Tl) + (((hfe)((u64)(T >> 64) * C)) << 64);
    hfe r = R + TC;
    if (__builtin_expect(r < R, 0)) r += HF_C;
    return r;
}
#define HF_HAVE_CUBE_ADD_ROWS 1
#endif
__attribute__((always_inline)) static inline hfe hf_mimc_step_weak(hfe xw, hfe k) { return hf_cube_add_weak(xw, k); }
__attribute__((always_inline)) static inline hfe hf_mimc_out(hfe xw) { return xw >= hf_p() ? xw - hf_p() : xw; }      // a weak value is below 2^128 < 2p
// one step of the MiMC recurrence x <- x^3 + k (examples/mimc/utils.ts:7-15) on the weak cube
static inline hfe hf_mimc_step(hfe x, hfe k) {
    hfe y = hf_cube_weak(x);                           // any representative of x^3
    hfe sum = y + k;
    if (sum < y) sum += HF_C;                          // wrapped past 2^128: +2^128 == +C (the wrapped value is small)
    return hf_canon(sum);
}
#endif  // GS_SMALL_Q
