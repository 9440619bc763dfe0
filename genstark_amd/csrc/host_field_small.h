// host_field_small.h — the host-side GF(q) helpers of host_field.h for the small-field build flavours (see gf_small.h):
// same names, an element is an unsigned __int128 whose value is below q < 2^64.
#pragma once
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 hfe;

#define HF_Q ((hfe)(uint64_t)GS_SMALL_Q)
static const hfe HF_C = 0;     // "2^128 mod p" correction of the MiMC recurrence: never taken, sums stay far below 2^128

static inline hfe hf_p() { return HF_Q; }
static inline hfe hf_canon(hfe x) { return x % HF_Q; }
static inline hfe hf_reduce(hfe hi, hfe lo) {           // (hi * 2^128 + lo) mod q
    hfe r = hi % HF_Q;
    for (int i = 0; i < 2; i++) r = (r << 64) % HF_Q;         // r * 2^128 in two steps: r < 2^64 keeps r << 64 inside 128 bits
    return (r + lo % HF_Q) % HF_Q;
}
static inline hfe hf_add(hfe a, hfe b) { return (a + b) % HF_Q; }
static inline hfe hf_sub(hfe a, hfe b) { return (a + HF_Q - b) % HF_Q; }
static inline hfe hf_mul(hfe a, hfe b) { return (a % HF_Q) * (b % HF_Q) % HF_Q; }
static inline hfe hf_mul_weak(hfe a, hfe b) { return hf_mul(a, b); }
static inline hfe hf_cube_weak(hfe x) { return hf_mul(hf_mul(x, x), x); }
static inline hfe hf_pow(hfe b, hfe e) {
    hfe r = 1;
    b %= HF_Q;
    while (e) {
        if (e & 1) r = hf_mul(r, b);
        b = hf_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline hfe hf_inv(hfe a) { return a % HF_Q ? hf_pow(a, HF_Q - 2) : 0; }
static inline hfe hf_mimc_step(hfe x, hfe k) { return hf_add(hf_cube_weak(x), k % HF_Q); }
// (the 128-bit flavour keeps the chain weak and canonicalises beside it: host_field.h)
static inline hfe hf_mimc_step_weak(hfe x, hfe k) { return hf_mimc_step(x, k); }
static inline hfe hf_mimc_out(hfe x) { return x; }
static inline bool hf_is_zero(hfe a) { return a % HF_Q == 0; }
static inline hfe hf_load(const uint8_t *b) { hfe v; memcpy(&v, b, 16); return v; }
static inline void hf_store(uint8_t *b, hfe v) { memcpy(b, &v, 16); }
