/*
 * oracle/gf128.h — TEST INFRASTRUCTURE (CPU oracle).  Not part of the product; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * Scalar arithmetic in GF(p), p = 2^128 - 9*2^32 + 1 (examples/mimc/mimc128.ts:13,
 * examples/rescue/hash4x128.ts:10).  The reference performs these operations inside
 * @guildofweavers/galois@0.4.22 (absent from /root/reference; package-lock.json:12-20); the
 * results of add/sub/mul/inv/exp modulo a prime are uniquely determined, so this is a restatement
 * of the mathematical definition, checked against Python big-int arithmetic in tests/.
 *
 * parity unpinned: no reference-produced vectors exist for this path.
 */
#ifndef ORACLE_GF128_H
#define ORACLE_GF128_H

#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef u128 fe; /* canonical representative in [0, p) */

#define FE_P_LO 0xFFFFFFF700000001ULL
#define FE_P_HI 0xFFFFFFFFFFFFFFFFULL
#define FE_C ((u128)0x8FFFFFFFFULL) /* 2^128 mod p = 9*2^32 - 1 */

static inline fe fe_p(void) { return ((u128)FE_P_HI << 64) | FE_P_LO; }

static inline fe fe_load(const uint8_t *b) {
    uint64_t lo, hi;
    memcpy(&lo, b, 8);
    memcpy(&hi, b + 8, 8);
    return ((u128)hi << 64) | lo;
}
static inline void fe_store(uint8_t *b, fe a) {
    uint64_t lo = (uint64_t)a, hi = (uint64_t)(a >> 64);
    memcpy(b, &lo, 8);
    memcpy(b + 8, &hi, 8);
}

static inline fe fe_add(fe a, fe b) {
    fe s = a + b;
    if (s < a || s >= fe_p()) s -= fe_p(); /* wrap-around subtraction is exact: a+b < 2p */
    return s;
}
static inline fe fe_sub(fe a, fe b) { return a >= b ? a - b : a - b + fe_p(); }
static inline fe fe_neg(fe a) { return a ? fe_p() - a : 0; }

/* reduce hi*2^128 + lo */
static inline fe fe_reduce256(u128 hi, u128 lo) {
    /* hi*2^128 == hi*C; hi*C is < 2^164: split as th*2^128 + tl */
    u128 m0 = (u128)(uint64_t)hi * FE_C;
    u128 m1 = (u128)(uint64_t)(hi >> 64) * FE_C;
    u128 tl = m0 + (m1 << 64);
    u128 th = (m1 >> 64) + (tl < m0);
    u128 s = tl + lo;
    unsigned k = s < tl;
    u128 r = th * FE_C; /* th < 2^37 -> r < 2^73 */
    u128 s2 = s + r;
    k += s2 < s;
    while (k) { /* each pending carry is worth 2^128 == C */
        u128 s3 = s2 + FE_C;
        k -= 1;
        k += s3 < s2;
        s2 = s3;
    }
    while (s2 >= fe_p()) s2 -= fe_p();
    return s2;
}

static inline fe fe_mul(fe a, fe b) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u128 mid = p01 + p10;
    u128 midc = mid < p01;
    u128 lo = p00 + (mid << 64);
    u128 c1 = lo < p00;
    u128 hi = p11 + (mid >> 64) + (midc << 64) + c1;
    return fe_reduce256(hi, lo);
}

static inline fe fe_exp(fe b, u128 e) {
    fe r = 1;
    while (e) {
        if (e & 1) r = fe_mul(r, b);
        b = fe_mul(b, b);
        e >>= 1;
    }
    return r;
}
/* a^-1 = a^(p-2); 0 -> 0 */
static inline fe fe_inv(fe a) { return a ? fe_exp(a, fe_p() - 2) : 0; }
static inline fe fe_div(fe a, fe b) { return fe_mul(a, fe_inv(b)); }

#endif
