// host_field_wide.h — the host-side GF(p) helpers of host_field.h for the 256- / 224-bit build flavours (see gf_wide.h):
// same names; an element wraps the device header's `fe` (its functions are host + device) and converts from small integers.
#pragma once
#include <stdint.h>
#include <string.h>

struct hfe {
    fe v;
    hfe() : v(fe_zero()) {}
    hfe(uint64_t x) : v(fe_make((uint32_t)x, (uint32_t)(x >> 32), 0, 0)) {}
};

static inline hfe hf_wrap(const fe &v) { hfe r; r.v = v; return r; }
static inline hfe hf_add(hfe a, hfe b) { return hf_wrap(fe_add(a.v, b.v)); }
static inline hfe hf_sub(hfe a, hfe b) { return hf_wrap(fe_sub(a.v, b.v)); }

// The host product runs on four 64-bit limbs (the device header's 32-bit limbs cost a host core ~170 ns per product, this ~30):
// 4 x 4 schoolbook on unsigned __int128, then folds with 2^B == C (mod p), B = 256 / 224, and one conditional subtraction.
// Same canonical value as fe_mul.
typedef unsigned __int128 hw_u128;
static inline void hw_limbs(const fe &a, uint64_t x[4]) {
    for (int i = 0; i < 4; i++) x[i] = (uint64_t)a.w[2 * i] | ((uint64_t)a.w[2 * i + 1] << 32);
}
static inline hfe hw_element(const uint64_t x[4]) {          // x < 2p -> canonical element
    uint64_t pl[4], d[4];
    fe pf;
    for (int i = 0; i < GF_LIMBS; i++) pf.w[i] = gf_p_limb(i);
    hw_limbs(pf, pl);
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        hw_u128 t = (hw_u128)x[i] - pl[i] - borrow;
        d[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1u;
    }
    const uint64_t *r = borrow ? x : d;
    hfe o;
    for (int i = 0; i < 4; i++) { o.v.w[2 * i] = (uint32_t)r[i]; o.v.w[2 * i + 1] = (uint32_t)(r[i] >> 32); }
    return o;
}
#if GS_WIDE_BITS == 0
// runtime modulus: the device header's product (two word-serial Montgomery reductions on 32-bit limbs), as it compiles for the host
static inline hfe hf_mul(hfe a, hfe b) { return hf_wrap(fe_mul(a.v, b.v)); }
#else
#if GS_WIDE_BITS == 256
static inline void hw_reduce(const uint64_t t[8], uint64_t r[4]) {
    const uint64_t C = 351ull * 4294967296ull - 1ull;          // 2^256 mod p, 41 bits
    uint64_t carry = 0;
    for (int i = 0; i < 4; i++) {                               // lo + hi * C: five limbs, the fifth below 2^42
        hw_u128 m = (hw_u128)t[i + 4] * C + t[i] + carry;
        r[i] = (uint64_t)m;
        carry = (uint64_t)(m >> 64);
    }
    hw_u128 m = (hw_u128)carry * C;                             // < 2^83
    hw_u128 a = (hw_u128)r[0] + (uint64_t)m;
    r[0] = (uint64_t)a;
    a = (hw_u128)r[1] + (uint64_t)(m >> 64) + (uint64_t)(a >> 64);
    r[1] = (uint64_t)a;
    a = (hw_u128)r[2] + (uint64_t)(a >> 64);
    r[2] = (uint64_t)a;
    a = (hw_u128)r[3] + (uint64_t)(a >> 64);
    r[3] = (uint64_t)a;
    if ((uint64_t)(a >> 64)) {                                  // wrapped past 2^256: the wrapped value is small, + C cannot wrap again
        a = (hw_u128)r[0] + C;
        r[0] = (uint64_t)a;
        for (int i = 1; i < 4; i++) { a = (hw_u128)r[i] + (uint64_t)(a >> 64); r[i] = (uint64_t)a; }
    }
}
#else
// p = 2^224 - 2^96 + 1: the word c_k (k >= 7) of the product weighs 2^(32k) == 2^(32(k-4)) - 2^(32(k-7)), once more for k - 4 >= 7.
// Collected per 32-bit word of the result (signed carries), then the few units that spill past 2^224 go round again.
static inline void hw_reduce(const uint64_t t[8], uint64_t r[4]) {
    int64_t c[14];
    for (int i = 0; i < 7; i++) { c[2 * i] = (int64_t)(t[i] & 0xFFFFFFFFull); c[2 * i + 1] = (int64_t)(t[i] >> 32); }
    int64_t w[7] = {c[0] - c[7] - c[11], c[1] - c[8] - c[12], c[2] - c[9] - c[13], c[3] + c[7] - c[10] + c[11],
                    c[4] + c[8] - c[11] + c[12], c[5] + c[9] - c[12] + c[13], c[6] + c[10] - c[13]};
    int64_t top = 0;
    for (;;) {
        int64_t acc = 0;
        w[0] -= top;
        w[3] += top;
        for (int i = 0; i < 7; i++) {
            acc += w[i];
            w[i] = acc & 0xFFFFFFFFll;
            acc >>= 32;                                       // arithmetic: a borrow travels as -1
        }
        top = acc;
        if (!top) break;
    }
    for (int i = 0; i < 3; i++) r[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    r[3] = (uint64_t)w[6];
}
#endif
static inline hfe hf_mul(hfe a, hfe b) {
    uint64_t x[4], y[4], t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r[4];
    hw_limbs(a.v, x);
    hw_limbs(b.v, y);
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            hw_u128 m = (hw_u128)x[i] * y[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)m;
            carry = (uint64_t)(m >> 64);
        }
        t[i + 4] = carry;
    }
    hw_reduce(t, r);
    return hw_element(r);
}
#endif
static inline hfe hf_pow(hfe b, hfe e) {
    hfe r(1);
    int top = GF_LIMBS - 1;
    while (top > 0 && !e.v.w[top]) top--;
    for (int i = 0; i <= top; i++) {
        uint32_t w = e.v.w[i];
        for (int k = 0; k < 32 && (w || i < top); k++) {
            if (w & 1u) r = hf_mul(r, b);
            b = hf_mul(b, b);
            w >>= 1;
        }
    }
    return r;
}
static inline hfe hf_inv(hfe a) {                      // Fermat, 0 -> 0 (as fe_inv)
    hfe e;
    for (int i = 0; i < GF_LIMBS; i++) e.v.w[i] = gf_p_limb(i);
    uint64_t borrow = 2;
    for (int i = 0; i < GF_LIMBS; i++) {
        uint64_t t = (uint64_t)e.v.w[i] - borrow;
        e.v.w[i] = (uint32_t)t;
        borrow = (t >> 32) & 1u;
    }
    return hf_pow(a, e);
}
static inline hfe hf_mimc_step(hfe x, hfe k) { return hf_add(hf_mul(hf_mul(x, x), x), k); }   // examples/mimc/utils.ts:7-15
// (the 128-bit flavour keeps the chain weak and canonicalises beside it: host_field.h)
static inline hfe hf_mimc_step_weak(hfe x, hfe k) { return hf_mimc_step(x, k); }
static inline hfe hf_mimc_out(hfe x) { return x; }
static inline bool hf_is_zero(hfe a) { return fe_is_zero(a.v); }
static inline hfe hf_load(const uint8_t *b) { hfe r; memcpy(&r.v, b, sizeof(fe)); return r; }
static inline void hf_store(uint8_t *b, hfe x) { memcpy(b, &x.v, sizeof(fe)); }
