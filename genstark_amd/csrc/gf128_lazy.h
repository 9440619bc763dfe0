// gf128_lazy.h — "lazy" arithmetic in GF(p), p = 2^128 - 9*2^32 + 1, for the butterfly networks of ntt.hip (gfx950).
//
// Why a second representation beside gf128.h's canonical four 32-bit limbs: on gfx950 a carry-chain instruction
// (v_add_co / v_addc_co) and every three-operand instruction cost about twice a plain two-operand 32-bit operation, and a
// canonical add/sub needs 8 carry operations + 4 selects (tools/microbench4.hip, profiles/r02_a_instruction_costs.md).
// Here an element is five SIGNED 32-bit limbs in radix B = 2^26,
//        value = l0 + l1*B + l2*B^2 + l3*B^3 + l4*B^4   (any integer congruent to the element mod p),
// so that
//   * add / sub are five independent v_add_u32 / v_sub_u32 — no carries, no selects, no reduction: limbs simply grow by
//     one bit per butterfly level (six bits of headroom = a whole radix-16 network between two multiplications);
//   * a product is 25 x v_mad_i64_i32 into 64-bit column accumulators (no carry chains either) followed by ONE carry
//     propagation done with v_alignbit / v_and / v_mad, and the fold 2^130 == 2304*B - 4 (mod p)  [2^128 == 9*2^32 - 1];
//   * for a multiplier known in advance (the radix-16 twiddles, the same for every lane) the five shifted copies
//     W_i = w * B^i mod p are tabulated, which removes the four high columns and their fold entirely ("W-form").
// Canonical 16-byte elements exist only in memory: lz_unpack after the load, lz_pack before the store.
//
// Bounds (checked by tests/test_lazy_field.py over extreme and random inputs, and by the interval notes below):
//   NN ("near-normalised", what lz_unpack, lz_mul_* and lz_norm return):
//        l0 in (-2^8, B + 2^8), l1 in (-2^17, B + 2^17), l2, l3 in [0, B), l4 in [0, 2^24)
//   a multiplier input x may be any sum/difference of up to 6 NN values (sum of |limbs| < 2^30.6 is what the column
//   accumulators need: |column| < 2^57 so that a carry fits 32 bits); the radix-16 DIF network only ever multiplies
//   differences, which are at most 4 NN values apart.
#pragma once
#include <stdint.h>

#include "gf128.h"

#define LZ_B 26
#define LZ_M 0x3ffffff

// Products by the small constants of the folds (2304, -4, 147456, -256, 4).  Written in C the compiler either strength-reduces them
// into 64-bit shift/subtract sequences (carry chains: exactly what this file avoids) or, when the constant lives in another basic
// block, widens them to full 64 x 64-bit products (four instructions); one explicit v_mad each keeps them what they are.
GF_HD int64_t lz_mad_ik(int32_t a, int32_t k, int64_t c) {      // c + a * k, signed
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d;
    uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(k), "v"(c));
    return d;
#else
    return c + (int64_t)a * k;
#endif
}
GF_HD int64_t lz_mad_uk(uint32_t a, uint32_t k, int64_t c) {    // c + a * k, unsigned factors
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d;
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(k), "v"(c));
    return d;
#else
    return c + (int64_t)((uint64_t)a * k);
#endif
}
struct lzk {   // kept so that call sites read the same on host and device; carries nothing any more
    int unused;
};
GF_HD lzk lzk_make() { lzk K; K.unused = 0; return K; }

struct lz {
    int32_t l[5];
};

// W-form of a multiplier: row i holds the NN limbs of (w * B^i mod p), canonical (limbs 0..3 < 2^26, limb 4 < 2^24)
struct lzw {
    int32_t w[5][5];
};

GF_HD lz lz_unpack(const fe &a) {   // 8 operations; the result is exactly normalised (l4 < 2^24)
    lz r;
    r.l[0] = (int32_t)(a.w0 & LZ_M);
    r.l[1] = (int32_t)(((a.w0 >> 26) | (a.w1 << 6)) & LZ_M);
    r.l[2] = (int32_t)(((a.w1 >> 20) | (a.w2 << 12)) & LZ_M);
    r.l[3] = (int32_t)(((a.w2 >> 14) | (a.w3 << 18)) & LZ_M);
    r.l[4] = (int32_t)(a.w3 >> 8);
    return r;
}

GF_HD lz lz_add(const lz &a, const lz &b) {
    lz r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
GF_HD lz lz_sub(const lz &a, const lz &b) {
    lz r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}

// carry of a 64-bit column into the next limb: bits 26..57 as a signed 32-bit value (exact for |c| < 2^57); one v_alignbit_b32
GF_HD int32_t lz_carry(int64_t c) { return (int32_t)(c >> LZ_B); }
GF_HD int32_t lz_low(int64_t c) { return (int32_t)((uint32_t)c & LZ_M); }

// the common tail of both products: five 64-bit columns c[0..4] (value = sum c[j] B^j, |c[j]| < 2^57 - 2^44) -> NN limbs
GF_HD lz lz_fold_columns(int64_t c[5], const lzk &K) {
    lz y;
    // top column first: everything above 2^130 comes down as T0 * (2304*B - 4)
    const int32_t t0 = lz_carry(c[4]);
    int32_t y4 = lz_low(c[4]);
    c[1] = lz_mad_ik(t0, 2304, c[1]);
    c[0] = lz_mad_ik(t0, -4, c[0]);
    int32_t k = lz_carry(c[0]);
    y.l[0] = lz_low(c[0]);
    c[1] += k;
    k = lz_carry(c[1]);
    y.l[1] = lz_low(c[1]);
    c[2] += k;
    k = lz_carry(c[2]);
    y.l[2] = lz_low(c[2]);
    c[3] += k;
    k = lz_carry(c[3]);
    y.l[3] = lz_low(c[3]);
    // k (|k| < 2^31) lands on limb 4: its low 26 bits stay, what is left of it and bits 24.. of limb 4 are folded once more,
    // now at 2^128 == 576*B - 1  (t1 is a handful of bits)
    y4 += k & LZ_M;
    const int32_t t1 = ((k >> LZ_B) << 2) + (y4 >> 24);
    y.l[4] = y4 & 0xffffff;
    y.l[1] += t1 * 576;
    y.l[0] -= t1;
    return y;
}

// x * w, the multiplier given by its W-form (lane-uniform on the GPU: the 25 words sit in SGPRs)
GF_HD lz lz_mul_u(const lz &x, const lzw &W, const lzk &K) {
    int64_t c[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        int64_t s = (int64_t)x.l[0] * W.w[0][j];
#pragma unroll
        for (int i = 1; i < 5; i++) s += (int64_t)x.l[i] * W.w[i][j];
        c[j] = s;
    }
    return lz_fold_columns(c, K);
}

// x * w, the multiplier an ordinary NN element (per-lane twiddles read from a table): nine columns, the four high ones
// (weight 2^130 * B^k) are brought down first.  A 64-bit column is split as c = h * 2^32 + l (h signed, l unsigned: its two registers):
//     c * 2^130 == 2304*B*c - 4*c,   2304*c = 2304*l + 147456*B*h,   4*c = 4*l + 256*B*h        (2^32 = 64*B)
// nine columns (value = sum c[k] B^k, |c[k]| < 2^57) -> NN limbs: the four high ones first, then the common tail
GF_HD lz lz_fold9(int64_t c[9], const lzk &K);
GF_HD lz lz_mul_v(const lz &x, const lz &w, const lzk &K) {
    int64_t c[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int64_t s = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = k - i;
            if (j >= 0 && j < 5) s += (int64_t)x.l[i] * w.l[j];
        }
        c[k] = s;
    }
    return lz_fold9(c, K);
}
// x * x for an NN x: 15 products instead of 25 (the cross terms once, with a doubled factor: limbs stay below 2^27.1, a column is at
// most three terms below 2^53.2).  What the long exponentiations of AIR programs are made of (csrc/air_jit.hip: emit_pow).
GF_HD lz lz_sqr(const lz &x, const lzk &K) {
    const int32_t d0 = 2 * x.l[0], d1 = 2 * x.l[1], d2 = 2 * x.l[2], d3 = 2 * x.l[3];
    int64_t c[9];
    c[0] = (int64_t)x.l[0] * x.l[0];
    c[1] = (int64_t)d0 * x.l[1];
    c[2] = (int64_t)d0 * x.l[2] + (int64_t)x.l[1] * x.l[1];
    c[3] = (int64_t)d0 * x.l[3] + (int64_t)d1 * x.l[2];
    c[4] = (int64_t)d0 * x.l[4] + (int64_t)d1 * x.l[3] + (int64_t)x.l[2] * x.l[2];
    c[5] = (int64_t)d1 * x.l[4] + (int64_t)d2 * x.l[3];
    c[6] = (int64_t)d2 * x.l[4] + (int64_t)x.l[3] * x.l[3];
    c[7] = (int64_t)d3 * x.l[4];
    c[8] = (int64_t)x.l[4] * x.l[4];
    return lz_fold9(c, K);
}
GF_HD lz lz_fold9(int64_t c[9], const lzk &K) {
#pragma unroll
    for (int k = 8; k >= 5; k--) {   // top down: column 8 spills into column 5, which is folded last
        const int32_t h = (int32_t)(c[k] >> 32);
        const uint32_t lo = (uint32_t)c[k];
        // -4*lo as an unsigned product: 4 * ~lo = 4 * (2^32 - 1) - 4 * lo; the constant is taken back right away (the compiler
        // folds the four of them into the first addend of columns 0..3)
        c[k - 3] = lz_mad_ik(h, 147456, c[k - 3]);
        c[k - 4] = lz_mad_uk(lo, 2304u, c[k - 4]);
        c[k - 4] = lz_mad_ik(h, -256, c[k - 4]);
        c[k - 5] = lz_mad_uk(~lo, 4u, c[k - 5]) - (((int64_t)1 << 34) - 4);
    }
    return lz_fold_columns(c, K);
}

// x * w * B^-5 (mod p) — Montgomery's REDC on the nine columns of the product, from the BOTTOM.  In this radix
//     p = 1 - 576*B + 2^24*B^4   (9*2^32 = 576*B, 2^128 = 2^24*B^4),   p == 1 (mod B),
// so the multiple of p that clears the low limb of column i is q = -(c_i mod B) — no multiplication to find it — and adding q*p*B^i
// leaves c_i - lo = carry*B, c_{i+1} += 576*lo, c_{i+4} -= 2^24*lo: two v_mad per step instead of the four a HIGH column costs in
// lz_fold9 (35 + 5 carry v_mad per product instead of 46).  After five steps columns 5..8 hold the result; the common carry tail
// brings it to NN limbs.  The multiplier is stored premultiplied by R = B^5 = 2^130 (twiddle tables: x * (w R) / R = x * w), so the
// DATA never leaves the ordinary domain and every other routine of this file applies to it unchanged.
// Bounds: |c_k| < 2^57 - 2^51 on entry (every input class of the pass kernels: tests/test_lazy_field.py) — a step adds less than
// 2^50 + 2^36 + 2^31 to a column; the result is within (x w / R - p, x w / R], a few units of 2^128: t below is a handful of bits.
GF_HD lz lz_mul_vm(const lz &x, const lz &wR, const lzk &K) {
    int64_t c[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int64_t s = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = k - i;
            if (j >= 0 && j < 5) s += (int64_t)x.l[i] * wR.l[j];
        }
        c[k] = s;
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int32_t lo = lz_low(c[i]);
        const int32_t k = lz_carry(c[i]);
        c[i + 1] = lz_mad_ik(lo, 576, c[i + 1]);
        c[i + 1] = lz_mad_ik(k, 1, c[i + 1]);
        c[i + 4] = lz_mad_ik(lo, -(1 << 24), c[i + 4]);
    }
    lz y;
    int32_t k = lz_carry(c[5]);
    y.l[0] = lz_low(c[5]);
    c[6] = lz_mad_ik(k, 1, c[6]);
    k = lz_carry(c[6]);
    y.l[1] = lz_low(c[6]);
    c[7] = lz_mad_ik(k, 1, c[7]);
    k = lz_carry(c[7]);
    y.l[2] = lz_low(c[7]);
    c[8] = lz_mad_ik(k, 1, c[8]);
    k = lz_carry(c[8]);               // limb 4 of the result, signed, a few bits above 2^24 at most
    y.l[3] = lz_low(c[8]);
    const int32_t t = k >> 24;        // the part at 2^128 == 576*B - 1
    y.l[4] = k & 0xffffff;
    y.l[1] += t * 576;
    y.l[0] -= t;
    return y;
}
// R^2 mod p and R mod p (R = 2^130) as canonical elements: x * R = lz_mul_vm(x, R^2)
//   R  = 2^130 mod p = 4 * (9 * 2^32 - 1) = 36 * 2^32 - 4
//   R2 = R^2 mod p   = 1296 * 2^64 - 288 * 2^32 + 16
GF_HD fe lz_mont_r() { return fe_make(0xFFFFFFFCu, 35u, 0u, 0u); }
GF_HD fe lz_mont_r2() { return fe_make(16u, 0xFFFFFEE0u, 1295u, 0u); }

// carry propagation without a product (elements that skip a twiddle): any limbs |l_i| < 2^31 - 2^6 -> NN
GF_HD lz lz_norm(const lz &x) {
    lz y;
    int32_t k = x.l[0] >> LZ_B;
    y.l[0] = x.l[0] & LZ_M;
    int32_t v = x.l[1] + k;
    k = v >> LZ_B;
    y.l[1] = v & LZ_M;
    v = x.l[2] + k;
    k = v >> LZ_B;
    y.l[2] = v & LZ_M;
    v = x.l[3] + k;
    k = v >> LZ_B;
    y.l[3] = v & LZ_M;
    v = x.l[4] + k;
    const int32_t t = v >> 24;     // |t| < 2^7
    y.l[4] = v & 0xffffff;
    y.l[1] += t * 576;
    y.l[0] -= t;
    return y;
}

// any lazy value (|l_i| <= 2^30 + 2^22, |l4| <= 2^28 + 2^12) -> a 16-byte element: the canonical residue (CANONICAL = true, what
// leaves the library) or any representative below 2^128 (false: intermediate passes, read back by lz_unpack).
// Almost everything happens in the limb domain with plain 32-bit operations: a multiple of p makes the value positive, carries are
// propagated exactly, the part above 2^128 comes back as t * (576*B - 1) and the carries are propagated once more.  What is left for
// saturated 128-bit arithmetic is the one-in-2^86 case that this sum reaches 2^128 again (one conditional +C, cannot repeat: the
// wrapped value is tiny) and, for the canonical form, the final conditional -p.
template <bool CANONICAL>
GF_HD fe lz_pack_t(const lz &x) {
    // 32*p in radix 2^26 = 2^133 - 576*32*B + 32: limbs {32, -18432, 0, 0, 2^29} (value-equal; the limbs need not be normalised)
    int32_t v = x.l[0] + 32;
    int32_t k = v >> LZ_B;
    int32_t l0 = v & LZ_M;
    v = x.l[1] - 18432 + k;
    k = v >> LZ_B;
    int32_t l1 = v & LZ_M;
    v = x.l[2] + k;
    k = v >> LZ_B;
    int32_t l2 = v & LZ_M;
    v = x.l[3] + k;
    k = v >> LZ_B;
    int32_t l3 = v & LZ_M;
    int32_t l4 = x.l[4] + (1 << 29) + k;      // >= 0: the whole value is positive; < 2^30 + 2^28
    const int32_t t = l4 >> 24;                 // quotient by 2^128, < 2^7
    l4 &= 0xffffff;
    // + t * (2^128 mod p) = t * (576*B - 1), then exact carries again (l0 may have gone slightly negative)
    v = l0 - t;
    k = v >> LZ_B;
    const uint32_t m0 = (uint32_t)(v & LZ_M);
    v = l1 + 576 * t + k;
    k = v >> LZ_B;
    const uint32_t m1 = (uint32_t)(v & LZ_M);
    v = l2 + k;
    k = v >> LZ_B;
    const uint32_t m2 = (uint32_t)(v & LZ_M);
    v = l3 + k;
    k = v >> LZ_B;
    const uint32_t m3 = (uint32_t)(v & LZ_M);
    const uint32_t m4 = (uint32_t)(l4 + k);     // <= 2^24: bit 24 set means the sum reached 2^128 (then everything below is tiny)
    fe r;
    r.w0 = m0 | (m1 << 26);
    r.w1 = (m1 >> 6) | (m2 << 20);
    r.w2 = (m2 >> 12) | (m3 << 14);
    r.w3 = (m3 >> 18) | (m4 << 8);              // bit 24 of m4 falls off the top: r = value mod 2^128
    const uint32_t m = 0u - (m4 >> 24);
    uint32_t c;
    r.w0 = gf_addc(r.w0, m, 0u, c);             // + C = 9*2^32 - 1 when it did
    r.w1 = gf_addc(r.w1, m & 8u, c, c);
    r.w2 = gf_addc(r.w2, 0u, c, c);
    r.w3 = gf_addc(r.w3, 0u, c, c);
    if (!CANONICAL) return r;
    // canonical: subtract p when r >= p, i.e. when r + C overflows
    fe s;
    s.w0 = gf_addc(r.w0, 0xFFFFFFFFu, 0u, c);
    s.w1 = gf_addc(r.w1, 8u, c, c);
    s.w2 = gf_addc(r.w2, 0u, c, c);
    s.w3 = gf_addc(r.w3, 0u, c, c);
    fe o;
    o.w0 = c ? s.w0 : r.w0;
    o.w1 = c ? s.w1 : r.w1;
    o.w2 = c ? s.w2 : r.w2;
    o.w3 = c ? s.w3 : r.w3;
    return o;
}
GF_HD fe lz_pack(const lz &x) { return lz_pack_t<true>(x); }
GF_HD fe lz_pack_weak(const lz &x) { return lz_pack_t<false>(x); }

// multiply an NN value by B modulo p (one limb up; the limb that falls off the top comes back as 2304*B - 4): the building
// block of the W-form.  Input NN, output NN.
GF_HD lz lz_shift_limb(const lz &x, const lzk &K) {
    int64_t c[5];
    c[0] = lz_mad_ik(x.l[4], -4, 0);
    c[1] = lz_mad_ik(x.l[4], 2304, (int64_t)x.l[0]);
    c[2] = x.l[1];
    c[3] = x.l[2];
    c[4] = x.l[3];
    return lz_fold_columns(c, K);
}

// W-form of a canonical element (host: plan tables; device: per-lane running products)
GF_HD void lz_wform(const fe &w, lzw &W) {
    fe cur = w;
    const fe Bm = fe_make(1u << 26, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 5; i++) {
        lz u = lz_unpack(cur);
#pragma unroll
        for (int j = 0; j < 5; j++) W.w[i][j] = u.l[j];
        if (i < 4) cur = fe_mul(cur, Bm);
    }
}
