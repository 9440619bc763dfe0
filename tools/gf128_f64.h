// gf128_f64.h — EXPERIMENT (not part of the library): a per-lane product in GF(p), p = 2^128 - 9*2^32 + 1, on the
// fp64 FMA pipe of gfx950 instead of v_mad_u64_u32.  An element is three limbs in radix 2^43 held as doubles (integers, signed),
//        value = l0 + l1 * 2^43 + l2 * 2^86   (any integer congruent to the element mod p).
// A partial product a * b (|a b| < 2^94) is split EXACTLY at the radix with two FMAs and one subtraction:
//        t = fma(a, b, M)  with M = 1.5 * 2^95: the unit in the last place of t is 2^43, so t - M is a * b rounded to a multiple of 2^43,
//        hi = t - M,  lo = fma(a, b, -hi)   (|lo| <= 2^42, exact: the product and its rounding differ by less than one ulp)
// 53-bit multiplicands instead of 32-bit ones: 9 partial products instead of 25 (radix 2^26) — but every one of them is 3 operations on a
// pipe whose instructions cost what a v_mad_u64_u32 costs (1.83 vs 2.0 ns per wave-instruction, profiles/r02_a_instruction_costs.txt),
// and the reduction 2^129 == 2 * (9 * 2^32 - 1) needs the same split three (+ one) times more.  73 fp64 operations per product.
// tools/microbench5 times it beside lz_mul_vm; tests/test_lazy_field.py checks it on the host (IEEE fma is the same function there).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../genstark_amd/csrc/gf128.h"

struct fz {
    double l[3];
};
#define FZ_R 8796093022208.0                                 /* 2^43 */
#define FZ_RI (1.0 / 8796093022208.0)
#define FZ_M (1.5 * 4503599627370496.0 * 8796093022208.0)    /* 1.5 * 2^52 * 2^43 */
#define FZ_K 77309411326.0                                   /* 2^129 mod p = 2 * (9 * 2^32 - 1) */

#if defined(__HIP_DEVICE_COMPILE__)
#define FZ_FMA(a, b, c) __fma_rn(a, b, c)
#define FZ_ADD(a, b) __dadd_rn(a, b)
#define FZ_SUB(a, b) __dsub_rn(a, b)
#define FZ_MUL(a, b) __dmul_rn(a, b)
#else
// (the host harness is compiled with -ffp-contract=off: a contraction of `t - M` into an FMA would change nothing here, one of
// hi * RI + x would not either — every operation below is exact by construction — but the check should not depend on that)
#define FZ_FMA(a, b, c) fma(a, b, c)
#define FZ_ADD(a, b) ((a) + (b))
#define FZ_SUB(a, b) ((a) - (b))
#define FZ_MUL(a, b) ((a) * (b))
#endif

GF_HD void fz_split(double a, double b, double &hi, double &lo) {
    const double t = FZ_FMA(a, b, FZ_M);
    hi = FZ_SUB(t, FZ_M);
    lo = FZ_FMA(a, b, -hi);
}
// canonical 16-byte element -> three limbs in [0, 2^43) (l2 < 2^42)
GF_HD fz fz_unpack(const fe &a) {
    const uint64_t lo = (uint64_t)a.w0 | ((uint64_t)a.w1 << 32), hi = (uint64_t)a.w2 | ((uint64_t)a.w3 << 32);
    fz r;
    r.l[0] = (double)(int64_t)(lo & ((1ull << 43) - 1));
    r.l[1] = (double)(int64_t)(((lo >> 43) | (hi << 21)) & ((1ull << 43) - 1));
    r.l[2] = (double)(int64_t)(hi >> 22);
    return r;
}
// x * w: x lazy (|limb| < 2^47: a radix-16 network's worth of sums of normalised values), w normalised (|limb| <= 2^43).
// Result near-normalised: |l0| < 2^44 (the last fold lands there after the carries), |l1| <= 2^42, |l2| <= 2^42 — still a valid multiplier.
GF_HD fz fz_mul(const fz &x, const fz &w) {
    double LO[5], HI[5];
#pragma unroll
    for (int c = 0; c < 5; c++) LO[c] = HI[c] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double h, l;
            fz_split(x.l[i], w.l[j], h, l);
            LO[i + j] = FZ_ADD(LO[i + j], l);
            HI[i + j] = FZ_ADD(HI[i + j], h);
        }
    // columns of radix 2^43: D_c = LO_c + HI_(c-1) / 2^43
    double D[6];
    D[0] = LO[0];
#pragma unroll
    for (int c = 1; c < 5; c++) D[c] = FZ_FMA(HI[c - 1], FZ_RI, LO[c]);
    D[5] = FZ_MUL(HI[4], FZ_RI);
    // columns 3, 4, 5 come down as D * K (2^129 == K): low part to column c - 3, high part to column c - 2
    double h3, l3, h4, l4, h5, l5, h6, l6;
    fz_split(D[3], FZ_K, h3, l3);
    fz_split(D[4], FZ_K, h4, l4);
    fz_split(D[5], FZ_K, h5, l5);
    double r0 = FZ_ADD(D[0], l3);
    double r1 = FZ_ADD(FZ_FMA(h3, FZ_RI, D[1]), l4);
    double r2 = FZ_ADD(FZ_FMA(h4, FZ_RI, D[2]), l5);
    const double r3 = FZ_MUL(h5, FZ_RI);                       // what column 5 left at 2^129 again (< 2^43)
    fz_split(r3, FZ_K, h6, l6);
    r0 = FZ_ADD(r0, l6);
    r1 = FZ_FMA(h6, FZ_RI, r1);
    // carries (round to nearest multiple of the radix: signed limbs)
    double c0 = FZ_SUB(FZ_ADD(r0, FZ_M), FZ_M);
    r0 = FZ_SUB(r0, c0);
    r1 = FZ_FMA(c0, FZ_RI, r1);
    double c1 = FZ_SUB(FZ_ADD(r1, FZ_M), FZ_M);
    r1 = FZ_SUB(r1, c1);
    r2 = FZ_FMA(c1, FZ_RI, r2);
    double c2 = FZ_SUB(FZ_ADD(r2, FZ_M), FZ_M);              // multiples of 2^129 still in limb 2 (a handful): back as c2 / 2^43 * K at limb 0
    r2 = FZ_SUB(r2, c2);
    r0 = FZ_FMA(FZ_MUL(c2, FZ_RI), FZ_K, r0);
    fz y;
    y.l[0] = r0; y.l[1] = r1; y.l[2] = r2;
    return y;
}
GF_HD fz fz_add(const fz &a, const fz &b) { fz r; for (int i = 0; i < 3; i++) r.l[i] = FZ_ADD(a.l[i], b.l[i]); return r; }
GF_HD fz fz_sub(const fz &a, const fz &b) { fz r; for (int i = 0; i < 3; i++) r.l[i] = FZ_SUB(a.l[i], b.l[i]); return r; }
