// host_field_small.h — the host-side GF(q) helpers of host_field.h for the small-field build flavours (see gf_small.cuh):
// same names, an element is an unsigned __int128 whose value is below q < 2^64.
#pragma once
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 hu128;

#define HF_Q ((hu128)(uint64_t)GS_SMALL_Q)
static const hu128 HF_C = 0;     // "2^128 mod p" correction of the MiMC recurrence: never taken, sums stay far below 2^128

static inline hu128 hf_p() { return HF_Q; }
static inline hu128 hf_canon(hu128 x) { return x % HF_Q; }
static inline hu128 hf_reduce(hu128 hi, hu128 lo) {           // (hi * 2^128 + lo) mod q
    hu128 r = hi % HF_Q;
    for (int i = 0; i < 2; i++) r = (r << 64) % HF_Q;         // r * 2^128 in two steps: r < 2^64 keeps r << 64 inside 128 bits
    return (r + lo % HF_Q) % HF_Q;
}
static inline hu128 hf_add(hu128 a, hu128 b) { return (a + b) % HF_Q; }
static inline hu128 hf_sub(hu128 a, hu128 b) { return (a + HF_Q - b) % HF_Q; }
static inline hu128 hf_mul(hu128 a, hu128 b) { return (a % HF_Q) * (b % HF_Q) % HF_Q; }
static inline hu128 hf_mul_weak(hu128 a, hu128 b) { return hf_mul(a, b); }
static inline hu128 hf_cube_weak(hu128 x) { return hf_mul(hf_mul(x, x), x); }
static inline hu128 hf_pow(hu128 b, hu128 e) {
    hu128 r = 1;
    b %= HF_Q;
    while (e) {
        if (e & 1) r = hf_mul(r, b);
        b = hf_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline hu128 hf_inv(hu128 a) { return a % HF_Q ? hf_pow(a, HF_Q - 2) : 0; }
static inline hu128 hf_load(const uint8_t *b) { hu128 v; memcpy(&v, b, 16); return v; }
static inline void hf_store(uint8_t *b, hu128 v) { memcpy(b, &v, 16); }
