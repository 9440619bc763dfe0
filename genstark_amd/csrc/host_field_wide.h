// host_field_wide.h — the host-side GF(p) helpers of host_field.h for the 256- / 224-bit build flavours (see gf_wide.cuh):
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
static inline hfe hf_mul(hfe a, hfe b) { return hf_wrap(fe_mul(a.v, b.v)); }
static inline hfe hf_pow(hfe b, hfe e) { return hf_wrap(fe_pow(b.v, e.v)); }
static inline hfe hf_inv(hfe a) { return hf_wrap(fe_inv(a.v)); }
static inline hfe hf_mimc_step(hfe x, hfe k) { return hf_add(hf_mul(hf_mul(x, x), x), k); }   // examples/mimc/utils.ts:7-15
static inline bool hf_is_zero(hfe a) { return fe_is_zero(a.v); }
static inline hfe hf_load(const uint8_t *b) { hfe r; memcpy(&r.v, b, sizeof(fe)); return r; }
static inline void hf_store(uint8_t *b, hfe x) { memcpy(b, &x.v, sizeof(fe)); }
