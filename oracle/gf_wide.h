/*
 * oracle/gf_wide.h — TEST INFRASTRUCTURE (CPU oracle).  The arithmetic of oracle/gf128.h for the two multi-limb primes of the
 * reference's examples, fixed at build time (-DGS_WIDE_BITS=256: 2^256 - 351*2^32 + 1, examples/mimc/mimc256.ts:13;
 * -DGS_WIDE_BITS=224: 2^224 - 2^96 + 1, assembly/lib224.aa:3): the checker of the wide build flavours of the HIP library
 * (genstark_amd/csrc/gf_wide.h).  Elements are 32 bytes little-endian.
 *
 * The mathematical definition on C23 bit-precise integers — the 512-bit product a * b reduced with 2^BITS == 2^BITS - p until it
 * fits, then by subtraction (a generic 512-bit `%` gives the same values ten times slower: -DGS_ORACLE_PLAIN_MOD selects it) —
 * nothing shared with the device code.
 * gcc 11 has no _BitInt: this flavour is compiled with the ROCm clang (oracle/Makefile).
 * parity unpinned (see gf128.h).
 */
#ifndef ORACLE_GF_WIDE_H
#define ORACLE_GF_WIDE_H

#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef unsigned _BitInt(256) fe;      /* canonical representative in [0, p) */
typedef unsigned _BitInt(512) fe2;
typedef fe fexp;                       /* an exponent as wide as an element */
#define FE_BYTES 32

#if GS_WIDE_BITS == 0
/* the runtime-modulus flavour (liboracle_rt.so, checker of libgstark_hip_rt.so): the modulus is whatever gs_set_modulus stored; every
 * reduction is the language's own `%` on 512-bit integers — the definition, nothing else */
#define GS_ORACLE_PLAIN_MOD 1
extern fe gs_oracle_runtime_modulus;
#endif
static inline fe fe_p(void) {
#if GS_WIDE_BITS == 0
    return gs_oracle_runtime_modulus;
#elif GS_WIDE_BITS == 256
    return (fe)0 - ((fe)351 << 32) + 1;                 /* 2^256 wraps to 0 */
#elif GS_WIDE_BITS == 224
    return ((fe)1 << 224) - ((fe)1 << 96) + 1;
#else
#error "GS_WIDE_BITS must be 256, 224 or 0"
#endif
}
static inline fe fe_load(const uint8_t *b) {
    fe v = 0;
    for (int i = 31; i >= 0; i--) v = (v << 8) | b[i];
    return v;
}
static inline void fe_store(uint8_t *b, fe a) {
    for (int i = 0; i < 32; i++) { b[i] = (uint8_t)(a & 0xFF); a >>= 8; }
}
static inline fe fe_mod(fe2 t) {
#ifdef GS_ORACLE_PLAIN_MOD
    return (fe)(t % (fe2)fe_p());
#else
    const fe2 one = 1, c = (one << GS_WIDE_BITS) - (fe2)fe_p(), mask = (one << GS_WIDE_BITS) - 1;
    while (t >> GS_WIDE_BITS) t = (t >> GS_WIDE_BITS) * c + (t & mask);      /* 2^BITS == c (mod p) */
    while (t >= (fe2)fe_p()) t -= (fe2)fe_p();
    return (fe)t;
#endif
}
static inline fe fe_add(fe a, fe b) { return fe_mod((fe2)a + (fe2)b); }
static inline fe fe_sub(fe a, fe b) { return fe_mod((fe2)a + (fe2)fe_p() - (fe2)fe_mod(b)); }
static inline fe fe_neg(fe a) { return fe_mod(a) ? fe_p() - fe_mod(a) : 0; }
static inline fe fe_mul(fe a, fe b) { return fe_mod((fe2)a * (fe2)b); }
static inline fe fe_exp(fe b, fexp e) {
    fe r = 1;
    b = fe_mod(b);
    while (e) {
        if (e & 1) r = fe_mul(r, b);
        b = fe_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline fe fe_inv(fe a) { return fe_mod(a) ? fe_exp(a, fe_p() - 2) : 0; }
static inline fe fe_div(fe a, fe b) { return fe_mul(a, fe_inv(b)); }

#endif
