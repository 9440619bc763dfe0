/*
 * oracle/gf_small.h — TEST INFRASTRUCTURE (CPU oracle).  The arithmetic of oracle/gf128.h for a prime q < 2^64 fixed at build time
 * (-DGS_SMALL_Q=<q>ull): the checker of the small-field build flavours of the HIP library (genstark_amd/csrc/gf_small.h).
 * Plain remainders of 128-bit integers: the mathematical definition, nothing shared with the device code.
 * parity unpinned (see gf128.h).
 */
#ifndef ORACLE_GF_SMALL_H
#define ORACLE_GF_SMALL_H

#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef u128 fe; /* canonical representative in [0, q), stored in 16 bytes like the 128-bit field */

static inline fe fe_p(void) { return (u128)(uint64_t)GS_SMALL_Q; }
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
static inline fe fe_add(fe a, fe b) { return (a + b) % fe_p(); }
static inline fe fe_sub(fe a, fe b) { return (a + fe_p() - b % fe_p()) % fe_p(); }
static inline fe fe_neg(fe a) { return a % fe_p() ? fe_p() - a % fe_p() : 0; }
static inline fe fe_mul(fe a, fe b) { return (a % fe_p()) * (b % fe_p()) % fe_p(); }
static inline fe fe_exp(fe b, u128 e) {
    fe r = 1;
    b %= fe_p();
    while (e) {
        if (e & 1) r = fe_mul(r, b);
        b = fe_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline fe fe_inv(fe a) { return a % fe_p() ? fe_exp(a, fe_p() - 2) : 0; }
static inline fe fe_div(fe a, fe b) { return fe_mul(a, fe_inv(b)); }

#endif
