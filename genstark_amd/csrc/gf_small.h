// gf_small.h — GF(q) for a prime q < 2^64 fixed at build time (-DGS_SMALL_Q=<q>ull), with the names and the 16-byte element
// layout of gf128.h, so that every kernel of the library compiles for it unchanged (build flavours libgstark_hip_q64.so /
// libgstark_hip_q32.so: the fields of examples/rescue/hash2x64.ts:10 and examples/demo/fibonacci.ts:14).
//
// The reference itself accelerates only the 128-bit field (galois' wasm path; every other modulus runs its generic BigInt
// code), and nothing here is tuned either: an element is kept canonical in the two low limbs, the two high limbs are zero, one
// product is a 64x64 -> 128-bit multiply folded with 2^64 == 2^64 - q (Solinas-style, q = 2^64 - c with c < 2^36) or, for
// q < 2^32, a 64-bit remainder by a constant.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GF_HD __host__ __device__ __forceinline__
#else
#define GF_HD inline
#endif

#ifndef GS_SMALL_Q
#error "gf_small.h needs -DGS_SMALL_Q=<prime below 2^64>"
#endif

struct alignas(16) fe {
    uint32_t w0, w1, w2, w3;
};

#define GF_Q ((uint64_t)GS_SMALL_Q)
#define GF_P0 ((uint32_t)GF_Q)
#define GF_P1 ((uint32_t)(GF_Q >> 32))
#define GF_P2 0u
#define GF_P3 0u

GF_HD fe fe_make(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { fe r; r.w0 = a; r.w1 = b; r.w2 = c; r.w3 = d; return r; }
GF_HD fe fe_zero() { return fe_make(0, 0, 0, 0); }
GF_HD fe fe_one() { return fe_make(1, 0, 0, 0); }
GF_HD bool fe_is_zero(const fe &a) { return (a.w0 | a.w1 | a.w2 | a.w3) == 0; }
GF_HD bool fe_eq(const fe &a, const fe &b) { return a.w0 == b.w0 && a.w1 == b.w1 && a.w2 == b.w2 && a.w3 == b.w3; }
GF_HD uint64_t fe_u64(const fe &a) { return (uint64_t)a.w0 | ((uint64_t)a.w1 << 32); }
GF_HD fe fe_from(uint64_t v) { return fe_make((uint32_t)v, (uint32_t)(v >> 32), 0, 0); }
GF_HD bool fe_ge_p(const fe &a) { return (a.w2 | a.w3) != 0 || fe_u64(a) >= GF_Q; }

GF_HD uint64_t gfs_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    return (s < a || s >= GF_Q) ? s - GF_Q : s;          // a + b < 2q < 2^65: one subtraction, exact modulo 2^64
}
GF_HD uint64_t gfs_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a - b + GF_Q; }

GF_HD uint64_t gfs_mulhi(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

GF_HD uint64_t gfs_mul(uint64_t a, uint64_t b) {
#if GS_SMALL_Q < 0x100000000ull
    return (a * b) % GF_Q;                               // a, b < 2^32: the product fits 64 bits
#else
    // q = 2^64 - c: x = hi*2^64 + lo == hi*c + lo; three folds bring the high part from 64 to 0 bits (c < 2^36)
    const uint64_t c = 0ull - GF_Q;
    uint64_t hi = gfs_mulhi(a, b), lo = a * b;
    uint64_t h1 = gfs_mulhi(hi, c), l1 = hi * c;
    uint64_t s1 = l1 + lo;
    h1 += s1 < l1;                                       // < 2^36 + 1
    uint64_t h2 = gfs_mulhi(h1, c), l2 = h1 * c;         // h1*c < 2^73
    uint64_t s2 = l2 + s1;
    h2 += s2 < l2;                                       // < 2^10
    uint64_t t = h2 * c;                                 // < 2^46
    uint64_t s3 = s2 + t;
    if (s3 < t) s3 += c;                                 // wrapped past 2^64 == c; the wrapped value is tiny
    return s3 >= GF_Q ? s3 - GF_Q : s3;
#endif
}

GF_HD fe fe_add(const fe &a, const fe &b) { return fe_from(gfs_add(fe_u64(a), fe_u64(b))); }
GF_HD fe fe_sub(const fe &a, const fe &b) { return fe_from(gfs_sub(fe_u64(a), fe_u64(b))); }
GF_HD fe fe_neg(const fe &a) { return fe_is_zero(a) ? a : fe_from(GF_Q - fe_u64(a)); }
GF_HD fe fe_mul(const fe &a, const fe &b) { return fe_from(gfs_mul(fe_u64(a), fe_u64(b))); }
GF_HD fe fe_sqr(const fe &a) { return fe_mul(a, a); }

// b^e, e given as four 32-bit limbs (little endian; exponents up to 128 bits as in the 128-bit build)
GF_HD fe fe_pow(fe b, const fe &e) {
    uint64_t r = 1, x = fe_u64(b);
    const uint32_t ev[4] = {e.w0, e.w1, e.w2, e.w3};
    for (int i = 0; i < 4; i++) {
        uint32_t w = ev[i];
        for (int k = 0; k < 32; k++) {
            if (w & 1u) r = gfs_mul(r, x);
            x = gfs_mul(x, x);
            w >>= 1;
        }
    }
    return fe_from(r);
}
GF_HD fe fe_pow_u64(fe b, uint64_t e) {
    uint64_t r = 1, x = fe_u64(b);
    while (e) {
        if (e & 1u) r = gfs_mul(r, x);
        x = gfs_mul(x, x);
        e >>= 1;
    }
    return fe_from(r);
}
GF_HD fe fe_inv(const fe &a) { return fe_pow_u64(a, GF_Q - 2); }      // 0 -> 0
