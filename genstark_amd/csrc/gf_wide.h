// gf_wide.h — GF(p) for the two multi-limb primes of the reference's examples, fixed at build time (-DGS_WIDE_BITS=256 |
// 224), with the names of gf128.h so that every kernel of the library compiles for them unchanged (build flavours
// libgstark_hip_p256.so / libgstark_hip_p224.so):
//     p256 = 2^256 - 351*2^32 + 1     examples/mimc/mimc256.ts:13
//     p224 = 2^224 - 2^96 + 1         assembly/lib224.aa:3, examples/elliptic/pointmul.aa
// An element is 32 bytes little-endian (8 x u32; the top limb of a 224-bit element is zero), canonical (< p).
//
// The reference runs these fields on galois' generic BigInt code (its wasm path is 128-bit only) and nothing here is tuned
// either: a product is a schoolbook NL x NL limb multiply on 64-bit accumulators, folded three times with
// 2^(32*NL) == C (C = 2^(32*NL) - p: 351*2^32 - 1, resp. 2^96 - 1) and corrected with one conditional subtraction.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GF_HD __host__ __device__ __forceinline__
#else
#define GF_HD inline
#endif

#ifndef GS_WIDE_BITS
#error "gf_wide.h needs -DGS_WIDE_BITS=256, 224 or 0 (0: the modulus is set at run time, gs_set_modulus)"
#endif

#define GF_LIMBS 8                       // storage limbs of an element
struct alignas(16) fe {
    uint32_t w[GF_LIMBS];
};

#if GS_WIDE_BITS == 256
#define GF_NL 8                          // limbs the arithmetic runs on
#define GF_CW 2                          // limbs of C = 2^256 - p = 350*2^32 + (2^32 - 1)
#define GF_P_LIMBS {0x00000001u, 0xFFFFFEA1u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}
#define GF_C_LIMBS {0xFFFFFFFFu, 350u, 0u}
#elif GS_WIDE_BITS == 224
#define GF_NL 7
#define GF_CW 3                          // C = 2^224 - p = 2^96 - 1
#define GF_P_LIMBS {0x00000001u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u}
#define GF_C_LIMBS {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}
#elif GS_WIDE_BITS == 0
// ---- the runtime-modulus flavour (libgstark_hip_rt.so; index.ts:14 re-exports createPrimeField(modulus) for ANY prime): p is any odd
// modulus below 2^256, fixed once per process by gs_set_modulus (include/gstark.h).  Elements stay CANONICAL in memory like every other
// flavour's (hashes, the wire format and the host all see plain little-endian integers), so a product is two Montgomery reductions:
// a * b * R^-1, then * R^2 * R^-1 (R = 2^256) — word-serial REDC needs nothing of p but its limbs and -p^-1 mod 2^32.  Four times the
// multiplies of a tuned field and none of its tricks: this flavour exists so that every modulus works, not to be fast.
#define GF_NL 8
#define GF_RUNTIME_MODULUS 1
struct GfRuntime {
    uint32_t p[GF_LIMBS];        // the modulus
    uint32_t r2[GF_LIMBS];       // R^2 mod p
    uint32_t n0inv;              // -p^-1 mod 2^32
    uint32_t set;                // 1 once gs_set_modulus has run
};
#if defined(__HIPCC__)
// no relocatable device code: every translation unit has its own copy of the constants and pushes them itself (common.h: gs_rt_unit)
static __constant__ GfRuntime gf_rt_device;
#endif
inline GfRuntime &gf_rt_host() { static GfRuntime v = {}; return v; }       // one per shared object (vague linkage)
GF_HD const GfRuntime &gf_rt() {
#if defined(__HIP_DEVICE_COMPILE__)
    return gf_rt_device;
#else
    return gf_rt_host();
#endif
}
GF_HD uint32_t gf_p_limb(int i) { return gf_rt().p[i]; }
#else
#error "GS_WIDE_BITS must be 256, 224 or 0"
#endif

#if GS_WIDE_BITS != 0
GF_HD uint32_t gf_p_limb(int i) { const uint32_t p[GF_LIMBS] = GF_P_LIMBS; return p[i]; }
GF_HD uint32_t gf_c_limb(int i) { const uint32_t c[3] = GF_C_LIMBS; return c[i]; }
#endif

GF_HD fe fe_make(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    fe r;
    r.w[0] = a; r.w[1] = b; r.w[2] = c; r.w[3] = d;
#pragma unroll
    for (int i = 4; i < GF_LIMBS; i++) r.w[i] = 0;
    return r;
}
GF_HD fe fe_zero() { return fe_make(0, 0, 0, 0); }
GF_HD fe fe_one() { return fe_make(1, 0, 0, 0); }
GF_HD bool fe_is_zero(const fe &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) o |= a.w[i];
    return o == 0;
}
GF_HD bool fe_eq(const fe &a, const fe &b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) o |= a.w[i] ^ b.w[i];
    return o == 0;
}
GF_HD bool fe_ge_p(const fe &a) {
    // a >= p  <=>  a - p does not borrow
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) {
        uint64_t d = (uint64_t)a.w[i] - gf_p_limb(i) - borrow;
        borrow = (d >> 32) & 1u;
    }
    return borrow == 0;
}

// r = a - p if (carry || a >= p) else a, for a value carry*2^256 + a < 2p
GF_HD fe gf_cond_sub_p(const fe &a, uint32_t carry) {
    fe d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) {
        uint64_t t = (uint64_t)a.w[i] - gf_p_limb(i) - borrow;
        d.w[i] = (uint32_t)t;
        borrow = (t >> 32) & 1u;
    }
    const bool take = carry != 0 || borrow == 0;
    fe r;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) r.w[i] = take ? d.w[i] : a.w[i];
    return r;
}

GF_HD fe fe_add(const fe &a, const fe &b) {
    fe s;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) {
        uint64_t t = (uint64_t)a.w[i] + b.w[i] + carry;
        s.w[i] = (uint32_t)t;
        carry = t >> 32;
    }
    return gf_cond_sub_p(s, (uint32_t)carry);
}
GF_HD fe fe_sub(const fe &a, const fe &b) {
    fe d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) {
        uint64_t t = (uint64_t)a.w[i] - b.w[i] - borrow;
        d.w[i] = (uint32_t)t;
        borrow = (t >> 32) & 1u;
    }
    // a < b: add p back (the wrapped difference plus p is the canonical result modulo 2^256)
    fe r;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) {
        uint64_t t = (uint64_t)d.w[i] + (borrow ? gf_p_limb(i) : 0u) + carry;
        r.w[i] = (uint32_t)t;
        carry = t >> 32;
    }
    return r;
}
GF_HD fe fe_neg(const fe &a) { return fe_is_zero(a) ? a : fe_sub(fe_zero(), a); }

#if GS_WIDE_BITS != 0
// out[0 .. GF_NL + NH) (+ one carry limb) = lo[0 .. GF_NL) + hi[0 .. NH) * C
template <int NH>
GF_HD void gf_fold(const uint32_t *lo, const uint32_t *hi, uint32_t *out /* GF_NL + GF_CW + 1 limbs, zero-extended */) {
    constexpr int NO = GF_NL + GF_CW + 1;
    uint32_t acc[NO];
#pragma unroll
    for (int i = 0; i < NO; i++) acc[i] = i < GF_NL ? lo[i] : 0u;
#pragma unroll
    for (int j = 0; j < GF_CW; j++) {
        const uint32_t cj = gf_c_limb(j);
        uint64_t carry = 0;
#pragma unroll
        for (int i = 0; i < NH; i++) {
            if (i + j < NO) {
                uint64_t t = (uint64_t)hi[i] * cj + acc[i + j] + carry;
                acc[i + j] = (uint32_t)t;
                carry = t >> 32;
            }
        }
#pragma unroll
        for (int k = NH + j; k < NO; k++) {
            uint64_t t = (uint64_t)acc[k] + carry;
            acc[k] = (uint32_t)t;
            carry = t >> 32;
        }
    }
#pragma unroll
    for (int i = 0; i < NO; i++) out[i] = acc[i];
}

GF_HD fe fe_mul(const fe &a, const fe &b) {
    // schoolbook product, row by row
    uint32_t t[2 * GF_NL];
#pragma unroll
    for (int i = 0; i < 2 * GF_NL; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < GF_NL; i++) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < GF_NL; j++) {
            uint64_t m = (uint64_t)a.w[i] * b.w[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)m;
            carry = m >> 32;
        }
        t[i + GF_NL] = (uint32_t)carry;
    }
    // fold 1: NL high limbs -> value < 2^(32*(NL+CW)) * (1 + eps): CW + 1 high limbs
    uint32_t u[GF_NL + GF_CW + 1], v[GF_NL + GF_CW + 1], z[GF_NL + GF_CW + 1];
    gf_fold<GF_NL>(t, t + GF_NL, u);
    // fold 2: (CW + 1 limbs) * C fits NL limbs -> a carry of 0 or 1
    gf_fold<GF_CW + 1>(u, u + GF_NL, v);
    // fold 3: the carry; cannot carry again (a carry out of fold 2 leaves a small low part)
    gf_fold<1>(v, v + GF_NL, z);
    fe r;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) r.w[i] = i < GF_NL ? z[i] : 0u;
    return gf_cond_sub_p(r, 0);
}
#else
// t (2 * NL limbs, < p * 2^256) -> t * 2^-256 mod p, canonical: word-serial Montgomery reduction
GF_HD fe gf_redc(uint32_t *t /* 2 * GF_NL + 1 limbs, the last one 0 on entry */) {
    const GfRuntime &rt = gf_rt();
#pragma unroll
    for (int i = 0; i < GF_NL; i++) {
        const uint32_t m = t[i] * rt.n0inv;
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < GF_NL; j++) {
            uint64_t x = (uint64_t)m * rt.p[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)x;
            carry = x >> 32;
        }
#pragma unroll
        for (int k = i + GF_NL; k <= 2 * GF_NL; k++) {
            uint64_t x = (uint64_t)t[k] + carry;
            t[k] = (uint32_t)x;
            carry = x >> 32;
        }
    }
    fe r;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) r.w[i] = t[GF_NL + i];
    return gf_cond_sub_p(r, t[2 * GF_NL]);
}
GF_HD void gf_schoolbook(const fe &a, const uint32_t *b, uint32_t *t /* 2 * GF_NL + 1 limbs */) {
#pragma unroll
    for (int i = 0; i <= 2 * GF_NL; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < GF_NL; i++) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < GF_NL; j++) {
            uint64_t m = (uint64_t)a.w[i] * b[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)m;
            carry = m >> 32;
        }
        t[i + GF_NL] = (uint32_t)carry;
    }
}
// NOT inlined: a product is ~600 instructions here and the NTT butterflies alone hold hundreds of them — inlined, ntt.hip of this
// flavour takes more than a quarter of an hour to compile; as a call it builds like the other flavours
#if defined(__HIPCC__)
static __host__ __device__ __noinline__
#else
static inline
#endif
fe fe_mul(const fe &a, const fe &b) {
    uint32_t t[2 * GF_NL + 1];
    gf_schoolbook(a, b.w, t);
    const fe ab = gf_redc(t);                     // a b R^-1
    gf_schoolbook(ab, gf_rt().r2, t);
    return gf_redc(t);                            // a b R^-1 R^2 R^-1 = a b
}
#endif
GF_HD fe fe_sqr(const fe &a) { return fe_mul(a, a); }

// b^e, e given as an element's worth of limbs (little endian)
GF_HD fe fe_pow(fe b, const fe &e) {
    fe r = fe_one();
    for (int i = 0; i < GF_LIMBS; i++) {
        uint32_t w = e.w[i];
        for (int k = 0; k < 32; k++) {
            if (w & 1u) r = fe_mul(r, b);
            b = fe_mul(b, b);
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
        b = fe_mul(b, b);
    }
    return r;
}
GF_HD fe fe_inv(const fe &a) {                 // Fermat: a^(p-2); 0 -> 0
    fe e;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) e.w[i] = gf_p_limb(i);
    // p - 2: p's low limb is 1, so borrow through the zero limbs above it
    uint64_t borrow = 2;
#pragma unroll
    for (int i = 0; i < GF_LIMBS; i++) {
        uint64_t t = (uint64_t)e.w[i] - borrow;
        e.w[i] = (uint32_t)t;
        borrow = (t >> 32) & 1u;
    }
    return fe_pow(a, e);
}

#if GS_WIDE_BITS == 0
// host: fix this shared object's copy of the field constants (gs_set_modulus of the library; the native driver when it is bound to a
// runtime-modulus library).  0 = done (or already set to the same modulus), -1 = not an odd modulus >= 3, -2 = another modulus is set.
inline int gf_rt_configure(const uint32_t p[GF_LIMBS]) {
    GfRuntime &rt = gf_rt_host();
    if (rt.set) {
        for (int i = 0; i < GF_LIMBS; i++) if (rt.p[i] != p[i]) return -2;
        return 0;
    }
    bool small = true;
    for (int i = 1; i < GF_LIMBS; i++) small = small && !p[i];
    if (!(p[0] & 1u) || (small && p[0] < 3)) return -1;         // Montgomery reduction needs an odd modulus
    GfRuntime v = {};
    for (int i = 0; i < GF_LIMBS; i++) v.p[i] = p[i];
    uint32_t inv = 1;                                           // p^-1 mod 2^32 by Newton's iteration, then negated
    for (int i = 0; i < 5; i++) inv *= 2u - p[0] * inv;
    v.n0inv = 0u - inv;
    rt = v;                                                     // fe_add below reads the modulus through gf_rt_host()
    fe r = fe_one();
    for (int i = 0; i < 2 * 32 * GF_NL; i++) r = fe_add(r, r); // 2^512 mod p
    for (int i = 0; i < GF_LIMBS; i++) rt.r2[i] = r.w[i];
    rt.set = 1;
    return 0;
}
#endif
