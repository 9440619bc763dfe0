/*
 * oracle/oracle_abi.c — TEST INFRASTRUCTURE (CPU oracle).
 *
 * A host-memory restatement of every entry point of include/gstark.h, used only as the checker:
 * tests/ run the same call through the HIP library and through this file and compare bytes, and
 * the "not gpu" tests run the Python host mirror on top of it.  The product never loads it
 * (genstark_amd/_abi.py refuses any backend whose gs_backend_name() is not "hip-gfx950" unless a
 * test injects it).  "Device pointers" here are plain malloc'd host pointers.
 *
 * Each function follows the mathematical definition of the galois/merkle member named in
 * gstark.h (the packages themselves are absent from /root/reference), using
 * the simplest algorithm available (radix-2 NTT, serial Montgomery-trick inversion, one hash at a
 * time) so that it shares no structure with the HIP kernels it checks.
 *
 * parity unpinned (see gf128.h).
 */
#include "../include/gstark.h"
#if defined(GS_SMALL_Q)
#include "gf_small.h"   /* checker flavour for a prime below 2^64 */
#elif defined(GS_WIDE_BITS)
#include "gf_wide.h"    /* checker flavour for the 256- / 224-bit primes (32-byte elements; needs a C23 _BitInt compiler) */
#else
#include "gf128.h"
#endif
#ifndef FE_BYTES
#define FE_BYTES 16     /* bytes of one element in memory */
typedef u128 fexp;      /* an exponent as wide as an element */
#endif
#include "hashes.h"
#include <stdio.h>
#include <stdlib.h>

struct gs_ctx {
    char err[256];
    uint8_t rb[64][256];      /* posted read-backs: the copy is taken at once (host memory), the ticket discipline is the library's */
    uint32_t rb_bytes[64];
    uint64_t rb_next;
};

static int fail(gs_ctx *c, int code, const char *msg) {
    if (c) snprintf(c->err, sizeof c->err, "%s", msg);
    return code;
}
static int is_pow2(uint64_t x) { return x && !(x & (x - 1)); }

int gs_abi_version(void) { return GS_ABI_VERSION; }
const char *gs_backend_name(void) { return "oracle-cpu"; }
int gs_ctx_create(int device, void *stream, gs_ctx **out) {
    (void)device; (void)stream;
#if defined(GS_WIDE_BITS) && GS_WIDE_BITS == 0
    if (!fe_p()) return GS_ERR_UNSUPPORTED;          /* gs_set_modulus first */
#endif
    gs_ctx *c = (gs_ctx *)calloc(1, sizeof *c);
    if (!c) return GS_ERR_OOM;
    *out = c;
    return GS_OK;
}
void gs_ctx_destroy(gs_ctx *c) { free(c); }
const char *gs_last_error(const gs_ctx *c) { return c ? c->err : "null context"; }
int gs_sync(gs_ctx *c) { (void)c; return GS_OK; }
void *gs_stream(gs_ctx *c) { (void)c; return NULL; }
int gs_element_size(void) { return FE_BYTES; }
int gs_field_modulus(gs_elt *out) { fe_store(out, fe_p()); return GS_OK; }
#if defined(GS_WIDE_BITS) && GS_WIDE_BITS == 0
fe gs_oracle_runtime_modulus = 0;
#endif
/* include/gstark.h: the runtime-modulus flavour takes its (odd) modulus here, once per process; a fixed flavour accepts only its own */
int gs_set_modulus(const uint8_t *modulus_le, uint32_t bytes) {
    uint8_t buf[FE_BYTES] = {0};
    if (!modulus_le || !bytes || bytes > FE_BYTES) return GS_ERR_ARG;
    memcpy(buf, modulus_le, bytes);
#if defined(GS_WIDE_BITS) && GS_WIDE_BITS == 0
    const fe want = fe_load(buf);
    if (gs_oracle_runtime_modulus) return gs_oracle_runtime_modulus == want ? GS_OK : GS_ERR_UNSUPPORTED;
    if (!(want & 1) || want < 3) return GS_ERR_ARG;
    gs_oracle_runtime_modulus = want;
    return GS_OK;
#else
    uint8_t mine[FE_BYTES];
    fe_store(mine, fe_p());
    return memcmp(mine, buf, FE_BYTES) ? GS_ERR_UNSUPPORTED : GS_OK;
#endif
}

int gs_alloc(gs_ctx *c, uint64_t bytes, void **p) {
    *p = malloc(bytes ? bytes : 1);
    return *p ? GS_OK : fail(c, GS_ERR_OOM, "malloc failed");
}
int gs_free(gs_ctx *c, void *p) { (void)c; free(p); return GS_OK; }
int gs_cache_trim(gs_ctx *c) { (void)c; return GS_OK; }
int gs_upload(gs_ctx *c, void *d, const void *s, uint64_t n) { (void)c; memcpy(d, s, n); return GS_OK; }
int gs_download(gs_ctx *c, void *d, const void *s, uint64_t n) { (void)c; memcpy(d, s, n); return GS_OK; }
int gs_copy(gs_ctx *c, void *d, const void *s, uint64_t n) { (void)c; memmove(d, s, n); return GS_OK; }
int gs_gather(gs_ctx *c, const void *src, uint64_t rec, const uint64_t *idx, uint64_t count, void *out) {
    (void)c;
    for (uint64_t i = 0; i < count; i++) memcpy((uint8_t *)out + i * rec, (const uint8_t *)src + idx[i] * rec, rec);
    return GS_OK;
}

int gs_air_jit(gs_ctx *c, int enable) { (void)c; (void)enable; return GS_OK; }   /* the oracle interprets */
uint64_t gs_air_jit_launches(const gs_ctx *c) { (void)c; return 0; }
/* (a measurement aid of the HIP library: the oracle launches no kernels and tallies nothing) */
int gs_traffic_enable(gs_ctx *c, int on) { (void)c; (void)on; return GS_OK; }
int gs_traffic_read(gs_ctx *c, struct gs_traffic_entry *out, uint32_t cap, uint32_t *count) { (void)c; (void)out; (void)cap; if (count) *count = 0; return GS_OK; }
int gs_air_jit_check(int kind, const uint32_t *code, uint32_t ninstr, const uint32_t *icode, uint32_t init_ninstr, const uint8_t *consts,
                     uint32_t nconsts, uint32_t vm_regs, uint32_t registers, const uint64_t *lens, uint32_t nstatic, char *log_out,
                     uint64_t log_cap) {
    (void)consts; (void)nconsts; (void)kind; (void)code; (void)ninstr; (void)icode; (void)init_ninstr; (void)vm_regs; (void)registers; (void)lens; (void)nstatic;
    if (log_out && log_cap) snprintf(log_out, log_cap, "the oracle has no code generator");
    return GS_ERR_UNSUPPORTED;
}
int gs_defer_begin(gs_ctx *c) { (void)c; return GS_OK; }   /* host memory: every read-back is immediate */
int gs_defer_end(gs_ctx *c) { (void)c; return GS_OK; }
int gs_readback_post(gs_ctx *c, const void *src, uint32_t bytes, uint64_t *ticket) {
    if (!c || !src || !ticket) return GS_ERR_ARG;
    if (!bytes || bytes > 256 || (bytes & 15)) return fail(c, GS_ERR_ARG, "readback_post: 16..256 bytes, a multiple of 16");
    memcpy(c->rb[c->rb_next % 64], src, bytes);
    c->rb_bytes[c->rb_next % 64] = bytes;
    *ticket = c->rb_next++;
    return GS_OK;
}
int gs_readback_wait(gs_ctx *c, uint64_t ticket, void *host_dst) {
    if (!c || !host_dst) return GS_ERR_ARG;
    if (ticket >= c->rb_next || c->rb_next - ticket > 64) return fail(c, GS_ERR_ARG, "readback_wait: ticket is not outstanding");
    memcpy(host_dst, c->rb[ticket % 64], c->rb_bytes[ticket % 64]);
    return GS_OK;
}

/* The loops below are independent per index; built with -fopenmp (liboracle_omp.so: the all-cores CPU baseline of bench.py) they
 * run on every host core, built without it (liboracle.so, the checker) the pragmas vanish.  Loops that carry a running product
 * are cut into fixed blocks, each block starting from its own power (fe_exp): same values, block by block. */
#if defined(_OPENMP)
#include <omp.h>
#define PAR_FOR _Pragma("omp parallel for schedule(static)")
#else
#define PAR_FOR
#endif
#define BLOCK 4096   /* elements per block of a running product */

#define EL(p, i) fe_load((const uint8_t *)(p) + FE_BYTES * (uint64_t)(i))
#define ST(p, i, v) fe_store((uint8_t *)(p) + FE_BYTES * (uint64_t)(i), (v))

int gs_power_series(gs_ctx *c, const gs_elt *base, uint64_t n, void *out) {
    (void)c;
    fe b = fe_load(base);
    PAR_FOR
    for (uint64_t s0 = 0; s0 < n; s0 += BLOCK) {
        fe x = fe_exp(b, (fexp)s0);
        for (uint64_t i = s0; i < n && i < s0 + BLOCK; i++) { ST(out, i, x); x = fe_mul(x, b); }
    }
    return GS_OK;
}
int gs_vec_add(gs_ctx *c, const void *a, const void *b, uint64_t n, void *o) {
    (void)c;
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) ST(o, i, fe_add(EL(a, i), EL(b, i)));
    return GS_OK;
}
int gs_vec_sub(gs_ctx *c, const void *a, const void *b, uint64_t n, void *o) {
    (void)c;
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) ST(o, i, fe_sub(EL(a, i), EL(b, i)));
    return GS_OK;
}
int gs_vec_mul(gs_ctx *c, const void *a, const void *b, uint64_t n, void *o) {
    (void)c;
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) ST(o, i, fe_mul(EL(a, i), EL(b, i)));
    return GS_OK;
}
int gs_vec_add_scalar(gs_ctx *c, const void *a, const gs_elt *s, uint64_t n, void *o) {
    (void)c; fe k = fe_load(s);
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) ST(o, i, fe_add(EL(a, i), k));
    return GS_OK;
}
int gs_vec_sub_scalar(gs_ctx *c, const void *a, const gs_elt *s, uint64_t n, void *o) {
    (void)c; fe k = fe_load(s);
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) ST(o, i, fe_sub(EL(a, i), k));
    return GS_OK;
}
int gs_vec_mul_scalar(gs_ctx *c, const void *a, const gs_elt *s, uint64_t n, void *o) {
    (void)c; fe k = fe_load(s);
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) ST(o, i, fe_mul(EL(a, i), k));
    return GS_OK;
}

/* Montgomery-trick inversion, zeros map to zero: serial inside a block of BLOCK elements (one field inversion per block) */
static void batch_inv_block(const void *a, uint64_t lo, uint64_t hi, fe *out) {
    fe acc = 1;
    for (uint64_t i = lo; i < hi; i++) {
        out[i] = acc;
        fe v = EL(a, i);
        if (v) acc = fe_mul(acc, v);
    }
    fe inv = fe_inv(acc);
    for (uint64_t i = hi; i-- > lo;) {
        fe v = EL(a, i);
        if (v) {
            fe r = fe_mul(out[i], inv);
            inv = fe_mul(inv, v);
            out[i] = r;
        } else out[i] = 0;
    }
}
static int batch_inv(const void *a, uint64_t n, fe *out) {
    PAR_FOR
    for (uint64_t s0 = 0; s0 < n; s0 += BLOCK) batch_inv_block(a, s0, s0 + BLOCK < n ? s0 + BLOCK : n, out);
    return 0;
}
int gs_vec_inv(gs_ctx *c, const void *a, uint64_t n, void *o) {
    fe *t = (fe *)malloc((n ? n : 1) * sizeof(fe));
    if (!t) return fail(c, GS_ERR_OOM, "malloc failed");
    batch_inv(a, n, t);
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) ST(o, i, t[i]);
    free(t);
    return GS_OK;
}
int gs_vec_div(gs_ctx *c, const void *a, const void *b, uint64_t n, void *o) {
    fe *t = (fe *)malloc((n ? n : 1) * sizeof(fe));
    if (!t) return fail(c, GS_ERR_OOM, "malloc failed");
    batch_inv(b, n, t);
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) ST(o, i, fe_mul(EL(a, i), t[i]));
    free(t);
    return GS_OK;
}
int gs_vec_exp(gs_ctx *c, const void *a, const gs_elt *e, uint64_t n, void *o) {
    (void)c; fexp ee = fe_load(e);
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) ST(o, i, fe_exp(EL(a, i), ee));
    return GS_OK;
}
int gs_combine_many(gs_ctx *c, const void *const *vecs, const uint8_t *coeffs, uint32_t count, uint64_t n, void *o) {
    if (count == 0) return fail(c, GS_ERR_ARG, "combine_many: bad count");
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) {
        fe s = 0;
        for (uint32_t j = 0; j < count; j++) s = fe_add(s, fe_mul(EL(vecs[j], i), fe_load(coeffs + FE_BYTES * j)));
        ST(o, i, s);
    }
    return GS_OK;
}
/* term by term as the reference builds it: every adjusted vector v_j o powers, then the merge of all 2*count terms, then + plus */
int gs_combine_adjusted(gs_ctx *c, const void *const *vecs, const uint8_t *coeffs, const uint8_t *adj, uint32_t count, const void *powers,
                        const void *plus, uint64_t n, void *o) {
    if (!vecs || !o || (!coeffs && !adj) || (adj && !powers)) return GS_ERR_ARG;
    if (count == 0) return fail(c, GS_ERR_ARG, "combine_adjusted: no vectors");
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) {
        fe s = 0;
        if (coeffs) for (uint32_t j = 0; j < count; j++) s = fe_add(s, fe_mul(EL(vecs[j], i), fe_load(coeffs + FE_BYTES * j)));
        if (adj) for (uint32_t j = 0; j < count; j++) s = fe_add(s, fe_mul(fe_mul(EL(vecs[j], i), EL(powers, i)), fe_load(adj + FE_BYTES * j)));
        if (plus) s = fe_add(s, EL(plus, i));
        ST(o, i, s);
    }
    return GS_OK;
}
int gs_combine(gs_ctx *c, const void *a, const void *b, uint64_t n, gs_elt *out) {
    (void)c; fe s = 0;
    for (uint64_t i = 0; i < n; i++) s = fe_add(s, fe_mul(EL(a, i), EL(b, i)));
    fe_store(out, s);
    return GS_OK;
}
int gs_pluck(gs_ctx *c, const void *v, uint64_t vlen, uint64_t skip, uint64_t times, void *o) {
    if (!vlen) return fail(c, GS_ERR_ARG, "pluck: empty vector");
    PAR_FOR
    for (uint64_t i = 0; i < times; i++) ST(o, i, EL(v, (i * skip) % vlen));
    return GS_OK;
}
/* the definitions, term by term: denominators materialised, inverted with the serial Montgomery trick, multiplied */
/* the point list of a coset: x_i = shift * w^i.  The plain entries are the coset forms at shift = 1. */
int gs_zero_poly_inverses(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t steps, const gs_elt *x_last, void *o) {
    uint8_t one[64];
    memset(one, 0, sizeof one);
    one[0] = 1;
    return gs_zero_poly_inverses_coset(c, omega, n, (const gs_elt *)one, steps, x_last, o);
}
int gs_zero_poly_inverses_coset(gs_ctx *c, const gs_elt *omega, uint64_t n, const gs_elt *shift, uint64_t steps, const gs_elt *x_last, void *o) {
    if (!is_pow2(n) || !is_pow2(steps) || steps > n) return fail(c, GS_ERR_ARG, "zero_poly_inverses: bad sizes");
    if (n / steps > 32) return fail(c, GS_ERR_UNSUPPORTED, "zero_poly_inverses: n / steps above 32");
    const fe sh = fe_load(shift), shs = fe_exp(sh, (fexp)steps);
    fe w = fe_load(omega), xl = fe_load(x_last), ws = fe_exp(w, (fexp)steps);
    uint8_t *den = (uint8_t *)malloc((n ? n : 1) * FE_BYTES);
    fe *inv = (fe *)malloc((n ? n : 1) * sizeof(fe));
    if (!den || !inv) { free(den); free(inv); return fail(c, GS_ERR_OOM, "malloc failed"); }
    PAR_FOR
    for (uint64_t s0 = 0; s0 < n; s0 += BLOCK) {
        fe d = fe_mul(shs, fe_exp(ws, (fexp)s0));
        for (uint64_t i = s0; i < n && i < s0 + BLOCK; i++) { ST(den, i, fe_sub(d, 1)); d = fe_mul(d, ws); }      /* x_i^steps - 1 */
    }
    batch_inv(den, n, inv);
    PAR_FOR
    for (uint64_t s0 = 0; s0 < n; s0 += BLOCK) {
        fe x = fe_mul(sh, fe_exp(w, (fexp)s0));
        for (uint64_t i = s0; i < n && i < s0 + BLOCK; i++) { ST(o, i, fe_mul(fe_sub(x, xl), inv[i])); x = fe_mul(x, w); }
    }
    free(den); free(inv);
    return GS_OK;
}
int gs_div_by_domain_roots(gs_ctx *c, const void *num, uint32_t rows, uint64_t n, const gs_elt *omega, const uint64_t *root_index,
                           const uint32_t *roots_per_row, uint32_t max_roots, void *o) {
    uint8_t one[64];
    memset(one, 0, sizeof one);
    one[0] = 1;
    return gs_div_by_domain_roots_coset(c, num, rows, n, omega, (const gs_elt *)one, root_index, roots_per_row, max_roots, o);
}
int gs_div_by_domain_roots_coset(gs_ctx *c, const void *num, uint32_t rows, uint64_t n, const gs_elt *omega, const gs_elt *shift, const uint64_t *root_index,
                                 const uint32_t *roots_per_row, uint32_t max_roots, void *o) {
    if (!is_pow2(n)) return fail(c, GS_ERR_ARG, "div_by_domain_roots: n must be a power of two");
    fe w = fe_load(omega);
    const fe sh = fe_load(shift);
    uint8_t *den = (uint8_t *)malloc((n ? n : 1) * FE_BYTES);
    fe *inv = (fe *)malloc((n ? n : 1) * sizeof(fe));
    if (!den || !inv) { free(den); free(inv); return fail(c, GS_ERR_OOM, "malloc failed"); }
    for (uint32_t r = 0; r < rows; r++) {
        if (roots_per_row[r] > max_roots || roots_per_row[r] > 4) { free(den); free(inv); return fail(c, GS_ERR_UNSUPPORTED, "div_by_domain_roots: at most 4 roots per row"); }
        fe root[4];
        for (uint32_t a = 0; a < roots_per_row[r]; a++) root[a] = fe_exp(w, (fexp)(root_index[(uint64_t)r * max_roots + a] % n));
        PAR_FOR
        for (uint64_t s0 = 0; s0 < n; s0 += BLOCK) {
            fe x = fe_mul(sh, fe_exp(w, (fexp)s0));
            for (uint64_t i = s0; i < n && i < s0 + BLOCK; i++) {
                fe z = 1;
                for (uint32_t a = 0; a < roots_per_row[r]; a++) z = fe_mul(z, fe_sub(x, root[a]));
                ST(den, i, z);
                x = fe_mul(x, w);
            }
        }
        batch_inv(den, n, inv);
        PAR_FOR
        for (uint64_t i = 0; i < n; i++) ST(o, (uint64_t)r * n + i, fe_mul(EL(num, (uint64_t)r * n + i), inv[i]));
    }
    free(den); free(inv);
    return GS_OK;
}
/* The tail of the composition polynomial + the linear combination (include/gstark.h: gs_composition_tail), restated as the sequence of
 * steps it replaces — each one this file's own entry: D = Q / Z (CompositionPolynomial.ts:113-121), I_b over the domain, P_b - I_b,
 * / Z_b (BoundaryConstraints.ts:71-95), the degree-adjusted merge with D (CompositionPolynomial.ts:124-146), then
 * LinearCombination.computeMany with C added (LinearCombination.ts:36-64). */
int gs_composition_tail_coset(gs_ctx *c, uint64_t n, const gs_elt *omega, const gs_elt *shift, const void *q, const void *z_inv_in, uint64_t z_steps,
                              const gs_elt *x_last,
                        const void *const *b_vecs, uint32_t bcount, const uint8_t *ipolys, uint32_t ilen, const uint64_t *root_index,
                        const uint32_t *roots_per_row, uint32_t max_roots, const uint8_t *b_coeffs, const uint8_t *b_adj, const void *const *l_vecs,
                        uint32_t lcount, const uint8_t *l_coeffs, const uint8_t *l_adj, const void *powers_in, uint64_t powers_exponent, void *c_out,
                        void *l_out) {
    if (!c || !omega || !shift || !q || (!z_inv_in && !x_last) || !l_out) return GS_ERR_ARG;
    if (bcount && (!b_vecs || !ipolys || !roots_per_row || !b_coeffs || (max_roots && !root_index))) return GS_ERR_ARG;
    if (lcount && (!l_vecs || !l_coeffs)) return GS_ERR_ARG;
    if ((b_adj || l_adj) && !powers_in && !powers_exponent) return GS_ERR_ARG;
    /* the two domain-only vectors, when the caller did not make them: ZeroPolynomial.ts:36-44 + CompositionPolynomial.ts:117, and the
     * power series of the degree adjustment (CompositionPolynomial.ts:124-138) */
    const void *z_inv = z_inv_in, *powers = powers_in;
    uint8_t *zmade = NULL, *pmade = NULL;
    if (!z_inv) {
        if (!is_pow2(n) || !is_pow2(z_steps) || z_steps > n) return fail(c, GS_ERR_ARG, "composition_tail: steps must be a power of two <= n");
        if (n / z_steps > 32) return fail(c, GS_ERR_UNSUPPORTED, "composition_tail: n / steps above 32 (pass 1/Z as a vector)");
        zmade = (uint8_t *)malloc((n ? n : 1) * FE_BYTES);
        if (!zmade) return fail(c, GS_ERR_OOM, "malloc failed");
        int zr = gs_zero_poly_inverses_coset(c, omega, n, shift, z_steps, x_last, zmade);
        if (zr) { free(zmade); return zr; }
        z_inv = zmade;
    }
    if ((b_adj || l_adj) && !powers) {
        pmade = (uint8_t *)malloc((n ? n : 1) * FE_BYTES);
        if (!pmade) { free(zmade); return fail(c, GS_ERR_OOM, "malloc failed"); }
        /* x_i^e for x_i = shift * omega^i: the power series of omega^e, times shift^e (the strided share of the domain-wide series) */
        uint8_t base[FE_BYTES], se[FE_BYTES];
        fe_store(base, fe_exp(fe_load(omega), (fexp)(powers_exponent % n)));
        fe_store(se, fe_exp(fe_load(shift), (fexp)powers_exponent));
        int pr = gs_power_series(c, (const gs_elt *)base, n, pmade);
        if (!pr) pr = gs_vec_mul_scalar(c, pmade, (const gs_elt *)se, n, pmade);
        if (pr) { free(zmade); free(pmade); return pr; }
        powers = pmade;
    }
    if (!is_pow2(n)) { free(zmade); free(pmade); return fail(c, GS_ERR_ARG, "composition_tail: n must be a power of two"); }
    if (bcount > 64) { free(zmade); free(pmade); return fail(c, GS_ERR_UNSUPPORTED, "composition_tail: at most 64 boundary rows"); }
    if (bcount && (ilen == 0 || ilen > 4)) { free(zmade); free(pmade); return fail(c, GS_ERR_UNSUPPORTED, "composition_tail: 1..4 interpolant coefficients per row"); }
    for (uint32_t r = 0; r < bcount; r++)
        if (roots_per_row[r] > max_roots || roots_per_row[r] > 4) { free(zmade); free(pmade); return fail(c, GS_ERR_UNSUPPORTED, "composition_tail: at most 4 roots per row"); }
    int rc = GS_OK;
    uint8_t *cbuf = (uint8_t *)malloc((n ? n : 1) * FE_BYTES), *mat = NULL;
    if (!cbuf) { free(zmade); free(pmade); return fail(c, GS_ERR_OOM, "malloc failed"); }
    rc = gs_vec_mul(c, q, z_inv, n, cbuf);
    if (!rc && bcount) {
        mat = (uint8_t *)malloc((size_t)3 * bcount * n * FE_BYTES);
        if (!mat) { free(cbuf); free(zmade); free(pmade); return fail(c, GS_ERR_OOM, "malloc failed"); }
        uint8_t *iv = mat, *pi = mat + (size_t)bcount * n * FE_BYTES, *bq = pi + (size_t)bcount * n * FE_BYTES;
        /* I_b over the coset: its coefficients scaled by shift^k, then the values at the powers of omega */
        uint8_t *isc = (uint8_t *)malloc((size_t)bcount * ilen * FE_BYTES);
        if (!isc) { free(mat); free(cbuf); free(zmade); free(pmade); return fail(c, GS_ERR_OOM, "malloc failed"); }
        for (uint32_t r = 0; r < bcount; r++) {
            fe sk = 1;
            for (uint32_t k = 0; k < ilen; k++) { ST(isc, (size_t)r * ilen + k, fe_mul(EL(ipolys, (size_t)r * ilen + k), sk)); sk = fe_mul(sk, fe_load(shift)); }
        }
        rc = gs_eval_polys_at_roots(c, isc, bcount, ilen, omega, n, iv);
        free(isc);
        if (!rc) rc = gs_sub_matrix_from_vectors(c, b_vecs, iv, bcount, n, pi);
        if (!rc) rc = gs_div_by_domain_roots_coset(c, pi, bcount, n, omega, shift, root_index, roots_per_row, max_roots, bq);
        if (!rc) {
            const void *rows[64];
            if (bcount > 64) rc = fail(c, GS_ERR_UNSUPPORTED, "composition_tail: at most 64 boundary rows");
            for (uint32_t r = 0; !rc && r < bcount; r++) rows[r] = bq + (size_t)r * n * FE_BYTES;
            if (!rc) rc = gs_combine_adjusted(c, rows, b_coeffs, b_adj, bcount, b_adj ? powers : NULL, cbuf, n, cbuf);
        }
    }
    if (!rc && c_out) memcpy(c_out, cbuf, n * FE_BYTES);
    if (!rc) {
        if (lcount) rc = gs_combine_adjusted(c, l_vecs, l_coeffs, l_adj, lcount, l_adj ? powers : NULL, cbuf, n, l_out);
        else memcpy(l_out, cbuf, n * FE_BYTES);
    }
    free(mat); free(cbuf); free(zmade); free(pmade);
    return rc;
}
int gs_composition_tail(gs_ctx *c, uint64_t n, const gs_elt *omega, const void *q, const void *z_inv, uint64_t z_steps, const gs_elt *x_last,
                        const void *const *b_vecs, uint32_t bcount, const uint8_t *ipolys, uint32_t ilen, const uint64_t *root_index,
                        const uint32_t *roots_per_row, uint32_t max_roots, const uint8_t *b_coeffs, const uint8_t *b_adj, const void *const *l_vecs,
                        uint32_t lcount, const uint8_t *l_coeffs, const uint8_t *l_adj, const void *powers, uint64_t powers_exponent, void *c_out,
                        void *l_out) {
    uint8_t one[64];
    memset(one, 0, sizeof one);
    one[0] = 1;
    return gs_composition_tail_coset(c, n, omega, (const gs_elt *)one, q, z_inv, z_steps, x_last, b_vecs, bcount, ipolys, ilen, root_index, roots_per_row, max_roots,
                                     b_coeffs, b_adj, l_vecs, lcount, l_coeffs, l_adj, powers, powers_exponent, c_out, l_out);
}
int gs_transpose_vector(gs_ctx *c, const void *v, uint64_t n, uint32_t cols, uint64_t step, void *o) {
    if (!cols || !step || n % ((uint64_t)cols * step)) return fail(c, GS_ERR_ARG, "transpose_vector: n %% (cols*step) != 0");
    uint64_t rows = n / ((uint64_t)cols * step);
    PAR_FOR
    for (uint64_t r = 0; r < rows; r++)
        for (uint32_t k = 0; k < cols; k++) ST(o, r * cols + k, EL(v, (r + (uint64_t)k * rows) * step));
    return GS_OK;
}
int gs_gather_words(gs_ctx *c, const void *addrs, uint64_t count, void *dst) {
    (void)c;
    const uint64_t *a = (const uint64_t *)addrs;
    for (uint64_t t = 0; t < count; t++) memcpy((uint8_t *)dst + 16 * t, (const void *)(uintptr_t)a[t], 16);
    return GS_OK;
}
int gs_transpose_records(gs_ctx *c, const void *src, uint64_t rows, uint64_t cols, uint64_t rec, void *dst) {
    if (!rec || rec % 16) return fail(c, GS_ERR_ARG, "transpose_records: record size must be a multiple of 16");
    for (uint64_t r = 0; r < rows; r++)
        for (uint64_t k = 0; k < cols; k++) memcpy((uint8_t *)dst + (k * rows + r) * rec, (const uint8_t *)src + (r * cols + k) * rec, rec);
    return GS_OK;
}
int gs_transpose_matrix(gs_ctx *c, const void *m, uint64_t rows, uint64_t cols, void *o) {
    (void)c;
    PAR_FOR
    for (uint64_t r = 0; r < rows; r++)
        for (uint64_t k = 0; k < cols; k++) ST(o, k * rows + r, EL(m, r * cols + k));
    return GS_OK;
}
int gs_sub_matrix_from_vectors(gs_ctx *c, const void *const *vecs, const void *m, uint32_t rows, uint64_t cols, void *o) {
    (void)c;
    for (uint32_t r = 0; r < rows; r++)
        for (uint64_t i = 0; i < cols; i++) ST(o, (uint64_t)r * cols + i, fe_sub(EL(vecs[r], i), EL(m, (uint64_t)r * cols + i)));
    return GS_OK;
}

/* in-place iterative radix-2 decimation-in-time NTT over natural-order input; tw[k] = omega^k for k < n/2 */
static void ntt_inplace(fe *a, uint64_t n, fe omega) {
    if (n < 2) return;
    fe *tw = (fe *)malloc((n / 2) * sizeof(fe));
    if (!tw) abort();
    PAR_FOR
    for (uint64_t s0 = 0; s0 < n / 2; s0 += BLOCK) {
        fe x = fe_exp(omega, (fexp)s0);
        for (uint64_t k = s0; k < n / 2 && k < s0 + BLOCK; k++) { tw[k] = x; x = fe_mul(x, omega); }
    }
    int bits = 0;
    while (((uint64_t)1 << bits) < n) bits++;
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) { /* bit reversal */
        uint64_t j = 0;
        for (int b = 0; b < bits; b++) j |= ((i >> b) & 1) << (bits - 1 - b);
        if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (uint64_t len = 2; len <= n; len <<= 1) {
        const uint64_t half = len / 2, stride = n / len;
        PAR_FOR
        for (uint64_t b = 0; b < n / 2; b++) {       /* butterfly b: block b / half, position b % half */
            const uint64_t j = b % half, i = (b / half) * len;
            fe u = a[i + j], v = fe_mul(a[i + j + half], tw[j * stride]);
            a[i + j] = fe_add(u, v);
            a[i + j + half] = fe_sub(u, v);
        }
    }
    free(tw);
}
static int check_root(gs_ctx *c, fe w, uint64_t n) { /* omega must generate the n-th roots of unity */
    if (n == 1) return w == 1 ? GS_OK : fail(c, GS_ERR_ARG, "ntt: omega must be 1 for n = 1");
    return fe_exp(w, n / 2) == fe_p() - 1 ? GS_OK : fail(c, GS_ERR_ARG, "ntt: omega is not a primitive n-th root of unity");
}
int gs_eval_polys_at_roots(gs_ctx *c, const void *polys, uint32_t rows, uint64_t plen, const gs_elt *omega,
                           uint64_t n, void *out) {
    if (!is_pow2(n) || plen > n) return fail(c, GS_ERR_ARG, "eval_polys_at_roots: n must be a power of two >= poly_len");
    if (check_root(c, fe_load(omega), n)) return GS_ERR_ARG;
    fe *t = (fe *)malloc(n * sizeof(fe));
    if (!t) return fail(c, GS_ERR_OOM, "malloc failed");
    fe w = fe_load(omega);
    for (uint32_t r = 0; r < rows; r++) {
        PAR_FOR
        for (uint64_t i = 0; i < n; i++) t[i] = i < plen ? EL(polys, (uint64_t)r * plen + i) : 0;
        ntt_inplace(t, n, w);
        PAR_FOR
        for (uint64_t i = 0; i < n; i++) ST(out, (uint64_t)r * n + i, t[i]);
    }
    free(t);
    return GS_OK;
}
int gs_interpolate_roots(gs_ctx *c, const void *ys, uint32_t rows, const gs_elt *omega, uint64_t n, void *out) {
    if (!is_pow2(n)) return fail(c, GS_ERR_ARG, "interpolate_roots: n must be a power of two");
    if (check_root(c, fe_load(omega), n)) return GS_ERR_ARG;
    fe *t = (fe *)malloc(n * sizeof(fe));
    if (!t) return fail(c, GS_ERR_OOM, "malloc failed");
    fe winv = fe_inv(fe_load(omega)), ninv = fe_inv((fe)n);
    for (uint32_t r = 0; r < rows; r++) {
        PAR_FOR
        for (uint64_t i = 0; i < n; i++) t[i] = EL(ys, (uint64_t)r * n + i);
        ntt_inplace(t, n, winv);
        PAR_FOR
        for (uint64_t i = 0; i < n; i++) ST(out, (uint64_t)r * n + i, fe_mul(t[i], ninv));
    }
    free(t);
    return GS_OK;
}
int gs_eval_poly_at(gs_ctx *c, const void *poly, uint64_t len, const gs_elt *x, gs_elt *out) {
    (void)c; fe xx = fe_load(x), s = 0;
    for (uint64_t i = len; i-- > 0;) s = fe_add(fe_mul(s, xx), EL(poly, i));
    fe_store(out, s);
    return GS_OK;
}

/* cubic through 4 points by Lagrange basis expansion */
static void lagrange4(const fe x[4], const fe y[4], fe cof[4]) {
    cof[0] = cof[1] = cof[2] = cof[3] = 0;
    /* the four denominators prod_{m != j} (x_j - x_m), inverted together (one field inversion per row; a zero denominator -
     * repeated x - inverts to zero like fe_inv does) */
    fe den[4], pre[4], acc = 1;
    for (int j = 0; j < 4; j++) {
        den[j] = 1;
        for (int m = 0; m < 4; m++) if (m != j) den[j] = fe_mul(den[j], fe_sub(x[j], x[m]));
        pre[j] = acc;
        if (den[j]) acc = fe_mul(acc, den[j]);
    }
    fe inv = fe_inv(acc);
    for (int j = 3; j >= 0; j--) {
        if (den[j]) { fe r = fe_mul(pre[j], inv); inv = fe_mul(inv, den[j]); den[j] = r; }
    }
    for (int j = 0; j < 4; j++) {
        fe num[4] = {1, 0, 0, 0}; /* prod_{m != j} (X - x_m), ascending coefficients */
        int deg = 0;
        for (int m = 0; m < 4; m++) {
            if (m == j) continue;
            fe nx = fe_neg(x[m]), nw[4] = {0, 0, 0, 0};
            for (int d = 0; d <= deg; d++) { /* num *= (X - x_m) */
                nw[d] = fe_add(nw[d], fe_mul(num[d], nx));
                nw[d + 1] = fe_add(nw[d + 1], num[d]);
            }
            deg++;
            for (int d = 0; d <= deg; d++) num[d] = nw[d];
        }
        fe s = fe_mul(y[j], den[j]);
        for (int d = 0; d < 4; d++) cof[d] = fe_add(cof[d], fe_mul(num[d], s));
    }
}
int gs_interpolate_quartic_batch(gs_ctx *c, const void *xs, const void *ys, uint64_t rows, void *out) {
    (void)c;
    for (uint64_t r = 0; r < rows; r++) {
        fe x[4], y[4], k[4];
        for (int j = 0; j < 4; j++) { x[j] = EL(xs, r * 4 + j); y[j] = EL(ys, r * 4 + j); }
        lagrange4(x, y, k);
        for (int j = 0; j < 4; j++) ST(out, r * 4 + j, k[j]);
    }
    return GS_OK;
}
int gs_interpolate_quartic_domain(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *ys,
                                  uint64_t rows, void *out) {
    if (rows * 4 * step != n) return fail(c, GS_ERR_ARG, "interpolate_quartic_domain: rows*4*step != n");
    fe w = fe_load(omega);
    for (uint64_t r = 0; r < rows; r++) {
        fe x[4], y[4], k[4];
        for (int j = 0; j < 4; j++) { x[j] = fe_exp(w, (fexp)((r + (uint64_t)j * rows) * step)); y[j] = EL(ys, r * 4 + j); }
        lagrange4(x, y, k);
        for (int j = 0; j < 4; j++) ST(out, r * 4 + j, k[j]);
    }
    return GS_OK;
}
int gs_fri_fold(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const gs_elt *xp, void *out) {
    if (m < 4 || m * step != n) return fail(c, GS_ERR_ARG, "fri_fold: column length * step != n");
    fe w = fe_load(omega), X = fe_load(xp);
    uint64_t rows = m / 4;
    /* row r sits at x_r = omega^(r*step), its three companions at x_r * zeta^j with zeta = omega^(rows*step) (a 4th root of unity) */
    const fe g = fe_exp(w, (fexp)step), zeta = fe_exp(w, (fexp)(rows * step));
    PAR_FOR
    for (uint64_t s0 = 0; s0 < rows; s0 += BLOCK) {
        fe xr = fe_exp(g, (fexp)s0);
        for (uint64_t r = s0; r < rows && r < s0 + BLOCK; r++, xr = fe_mul(xr, g)) {
            fe x[4], y[4], k[4], s = 0;
            x[0] = xr;
            for (int j = 1; j < 4; j++) x[j] = fe_mul(x[j - 1], zeta);
            for (int j = 0; j < 4; j++) y[j] = EL(column, r + (uint64_t)j * rows);
            lagrange4(x, y, k);
            for (int j = 3; j >= 0; j--) s = fe_add(fe_mul(s, X), k[j]);
            ST(out, r, s);
        }
    }
    return GS_OK;
}
int gs_fri_fold_seeded_scaled(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const void *seed32,
                              const gs_elt *scale, void *out) {
    /* x = prng(seed) * scale, prng(seed) = sha256(seed) as a big-endian integer, mod p (LowDegreeProver.ts:194), then the plain folding step */
    uint8_t d[32], xb[64];
    orc_sha256((const uint8_t *)seed32, 32, d);
    fe x = 0;
    for (int i = 0; i < 32; i++) x = fe_add(fe_mul(x, (fe)256), (fe)d[i]);
    x = fe_mul(x, fe_load(scale));
    memset(xb, 0, sizeof xb);
    fe_store(xb, x);
    return gs_fri_fold(c, omega, n, step, column, m, (const gs_elt *)xb, out);
}
int gs_fri_fold_seeded(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const void *seed32, void *out) {
    uint8_t one[FE_BYTES];
    memset(one, 0, sizeof one);
    one[0] = 1;
    return gs_fri_fold_seeded_scaled(c, omega, n, step, column, m, seed32, one, out);
}
int gs_eval_quartic_batch(gs_ctx *c, const void *polys, uint64_t rows, const gs_elt *x, void *out) {
    (void)c; fe xx = fe_load(x);
    PAR_FOR
    for (uint64_t r = 0; r < rows; r++) {
        fe s = EL(polys, r * 4 + 3);
        for (int j = 2; j >= 0; j--) s = fe_add(fe_mul(s, xx), EL(polys, r * 4 + j));
        ST(out, r, s);
    }
    return GS_OK;
}

int gs_hash_digest(gs_ctx *c, gs_hash_alg alg, const uint8_t *msg, uint64_t len, uint8_t out[32]) {
    (void)c; orc_hash((int)alg, msg, (size_t)len, out); return GS_OK;
}
int gs_hash_merge_rows(gs_ctx *c, gs_hash_alg alg, const void *const *vecs, uint32_t count, uint64_t n, void *out) {
    if (count == 0) return fail(c, GS_ERR_ARG, "hash_merge_rows: bad count");
    PAR_FOR
    for (uint64_t i = 0; i < n; i++) {
        uint8_t small[FE_BYTES * GS_MAX_COMBINE], *buf = count <= GS_MAX_COMBINE ? small : (uint8_t *)malloc(FE_BYTES * (size_t)count);
        if (!buf) abort();
        for (uint32_t j = 0; j < count; j++) memcpy(buf + FE_BYTES * j, (const uint8_t *)vecs[j] + FE_BYTES * i, FE_BYTES);
        orc_hash((int)alg, buf, FE_BYTES * (size_t)count, (uint8_t *)out + 32 * i);
        if (buf != small) free(buf);
    }
    return GS_OK;
}
int gs_hash_digest_values(gs_ctx *c, gs_hash_alg alg, const void *buf, uint64_t vs, uint64_t count, void *out) {
    (void)c;
    PAR_FOR
    for (uint64_t i = 0; i < count; i++) orc_hash((int)alg, (const uint8_t *)buf + vs * i, (size_t)vs, (uint8_t *)out + 32 * i);
    return GS_OK;
}
int gs_merkle_build(gs_ctx *c, gs_hash_alg alg, const void *leaves, uint64_t n, void *nodes) {
    if (!is_pow2(n) || n < 2) return fail(c, GS_ERR_ARG, "merkle_build: n must be a power of two >= 2");
    uint8_t *nd = (uint8_t *)nodes;
    memset(nd, 0, 32);
    PAR_FOR
    for (uint64_t i = 0; i < n / 2; i++) orc_hash((int)alg, (const uint8_t *)leaves + 64 * i, 64, nd + 32 * (n / 2 + i));
    for (uint64_t lvl = n / 4; lvl >= 1; lvl >>= 1) {      /* nodes [lvl, 2*lvl): one level, children in [2*lvl, 4*lvl) */
        PAR_FOR
        for (uint64_t i = lvl; i < 2 * lvl; i++) orc_hash((int)alg, nd + 64 * i, 64, nd + 32 * i);
    }
    return GS_OK;
}
int gs_merkle_commit_rows(gs_ctx *c, gs_hash_alg alg, const void *const *vecs, uint32_t count, uint64_t n, void *leaves, void *nodes) {
    /* Hash.mergeVectorRows then MerkleTree.create (lib/Stark.ts:115-118), literally */
    int rc = gs_hash_merge_rows(c, alg, vecs, count, n, leaves);
    if (rc) return rc;
    return gs_merkle_build(c, alg, leaves, n, nodes);
}

/* the members one after the other: the tree, its root posted, prng(root) */
int gs_merkle_commit_rows_seed(gs_ctx *c, gs_hash_alg alg, const void *const *vecs, uint32_t count, uint64_t n, void *leaves, void *nodes,
                               void *point_out, uint64_t *ticket) {
    if (!c || (!point_out && !ticket)) return GS_ERR_ARG;
    int rc = gs_merkle_commit_rows(c, alg, vecs, count, n, leaves, nodes);
    if (rc) return rc;
    const uint8_t *root = (const uint8_t *)nodes + 32;
    if (ticket && (rc = gs_readback_post(c, root, 32, ticket))) return rc;
    if (point_out) {
        uint8_t d[32];
        orc_sha256(root, 32, d);
        fe x = 0;
        for (int i = 0; i < 32; i++) x = fe_add(fe_mul(x, (fe)256), (fe)d[i]);
        fe_store((uint8_t *)point_out, x);
    }
    return GS_OK;
}
int gs_fri_fold_at(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const void *x_dev, void *out) {
    if (!x_dev) return GS_ERR_ARG;
    uint8_t xb[64];
    memset(xb, 0, sizeof xb);
    memcpy(xb, x_dev, FE_BYTES);
    return gs_fri_fold(c, omega, n, step, column, m, (const gs_elt *)xb, out);
}

/* gs_fri_layers restated as what its contract says it equals: per layer, gs_fri_fold_at then gs_merkle_commit_rows_seed over the four
 * quarters of the folded column (LowDegreeProver.ts:189-202), the next point taken from the tree just built. */
int gs_fri_layers(gs_ctx *c, gs_hash_alg alg, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t len, const void *x_dev,
                  uint32_t nlayers, struct gs_fri_layer *layers) {
    if (!c || !omega || !column || !x_dev || !layers || !nlayers) return GS_ERR_ARG;
    if (len < 32 || (len & (len - 1)) || len * step != n) return fail(c, GS_ERR_ARG, "fri_layers: len must be a power of two >= 32 with len * step == n");
    if (2 * (uint64_t)nlayers > 62 || (len >> (2 * nlayers)) < 8) return fail(c, GS_ERR_ARG, "fri_layers: too many layers for this column");
    uint8_t point[64];
    memset(point, 0, sizeof point);
    memcpy(point, x_dev, FE_BYTES);
    for (uint32_t i = 0; i < nlayers; i++) {
        struct gs_fri_layer *L = &layers[i];
        if (!L->next || !L->leaves || !L->nodes) return GS_ERR_ARG;
        int rc = gs_fri_fold_at(c, omega, n, step, column, len, point, L->next);
        if (rc) return rc;
        const uint64_t rows = len / 4, q = rows / 4;
        const void *quarters[4];
        for (int k = 0; k < 4; k++) quarters[k] = (const uint8_t *)L->next + (uint64_t)k * q * FE_BYTES;
        uint8_t next_point[64];
        memset(next_point, 0, sizeof next_point);
        if ((rc = gs_merkle_commit_rows_seed(c, alg, quarters, 4, q, L->leaves, L->nodes, next_point, &L->ticket))) return rc;
        if (L->point_out) memcpy(L->point_out, next_point, FE_BYTES);
        memcpy(point, next_point, FE_BYTES);
        column = L->next;
        len = rows;
        step *= 4;
    }
    return GS_OK;
}

int gs_mimc_trace(gs_ctx *c, const gs_elt *seed, const uint8_t *rc, uint32_t nrc, uint64_t steps, void *out) {
    if (!nrc || !steps) return fail(c, GS_ERR_ARG, "mimc_trace: empty");
    fe x = fe_load(seed);
    for (uint64_t i = 0; i < steps; i++) {
        ST(out, i, x);
        x = fe_add(fe_mul(fe_mul(x, x), x), fe_load(rc + FE_BYTES * (i % nrc)));
    }
    return GS_OK;
}
int gs_mimc_constraints(gs_ctx *c, const void *p, uint64_t nc, uint64_t shift, const void *k, uint64_t klen, void *out) {
    if (!klen || !nc) return fail(c, GS_ERR_ARG, "mimc_constraints: empty");
    PAR_FOR
    for (uint64_t j = 0; j < nc; j++) {
        fe x = EL(p, j), nx = EL(p, (j + shift) % nc);
        fe t = fe_add(fe_mul(fe_mul(x, x), x), EL(k, j % klen));
        ST(out, j, fe_sub(nx, t));
    }
    return GS_OK;
}
/* the definition, line by line of the header comment: Q, the two divisors inverted with the serial Montgomery trick, the sum */
int gs_mimc_composition(gs_ctx *c, const void *p, uint64_t n, uint64_t steps, const gs_elt *omega, const void *k, uint64_t klen,
                        const uint8_t *coeffs, uint64_t q_inc, uint64_t b_inc, const uint8_t *ipoly, const uint64_t *root_index, uint32_t nroots,
                        const uint8_t *lc_coeffs, void *out) {
    if (!is_pow2(n) || !is_pow2(steps) || steps > n || !klen) return fail(c, GS_ERR_ARG, "mimc_composition: bad sizes");
    if (n / steps > 32 || !nroots || nroots > 4) return fail(c, GS_ERR_UNSUPPORTED, "mimc_composition: n / steps <= 32 and 1..4 assertions");
    if (q_inc % steps || b_inc % steps) return fail(c, GS_ERR_ARG, "mimc_composition: degree increments must be multiples of the trace length");
    fe w = fe_load(omega), d0 = fe_load(coeffs), d1 = fe_load(coeffs + FE_BYTES), b0 = fe_load(coeffs + 2 * FE_BYTES), b1 = fe_load(coeffs + 3 * FE_BYTES);
    fe x_last = fe_exp(w, (fexp)((steps - 1) * (n / steps))), root[4];
    for (uint32_t a = 0; a < nroots; a++) root[a] = fe_exp(w, (fexp)(root_index[a] % n));
    uint8_t *den = (uint8_t *)malloc(2 * n * FE_BYTES);
    fe *inv = (fe *)malloc(2 * n * sizeof(fe));
    if (!den || !inv) { free(den); free(inv); return fail(c, GS_ERR_OOM, "malloc failed"); }
    /* x^steps, x^q_inc, x^b_inc along the domain: running products of omega^steps, omega^q_inc, omega^b_inc */
    const fe ws = fe_exp(w, (fexp)steps), wq = fe_exp(w, (fexp)q_inc), wb = fe_exp(w, (fexp)b_inc);
    PAR_FOR
    for (uint64_t s0 = 0; s0 < n; s0 += BLOCK) {
        fe x = fe_exp(w, (fexp)s0), xs = fe_exp(ws, (fexp)s0);
        for (uint64_t i = s0; i < n && i < s0 + BLOCK; i++) {
            fe zb = 1;
            for (uint32_t a = 0; a < nroots; a++) zb = fe_mul(zb, fe_sub(x, root[a]));
            ST(den, i, fe_sub(xs, 1));
            ST(den, n + i, zb);
            x = fe_mul(x, w);
            xs = fe_mul(xs, ws);
        }
    }
    batch_inv(den, 2 * n, inv);
    PAR_FOR
    for (uint64_t s0 = 0; s0 < n; s0 += BLOCK) {
      fe x = fe_exp(w, (fexp)s0), xq = fe_exp(wq, (fexp)s0), xb = fe_exp(wb, (fexp)s0);
      for (uint64_t i = s0; i < n && i < s0 + BLOCK; i++, xq = fe_mul(xq, wq), xb = fe_mul(xb, wb)) {
        fe pi = EL(p, i), pn = EL(p, (i + n / steps) % n);
        fe q = fe_sub(pn, fe_add(fe_mul(fe_mul(pi, pi), pi), EL(k, i % klen)));
        fe d = fe_mul(fe_mul(fe_mul(q, fe_add(d0, fe_mul(d1, xq))), fe_sub(x, x_last)), inv[i]);
        fe iv = 0;
        for (uint32_t cidx = nroots; cidx-- > 0;) iv = fe_add(fe_mul(iv, x), fe_load(ipoly + FE_BYTES * cidx));
        fe bq = fe_mul(fe_sub(pi, iv), inv[n + i]);
        fe r = fe_add(d, fe_mul(bq, fe_add(b0, fe_mul(b1, xb))));
        if (lc_coeffs) r = fe_add(r, fe_mul(pi, fe_add(fe_load(lc_coeffs), fe_mul(fe_load(lc_coeffs + FE_BYTES), xb))));
        ST(out, i, r);
        x = fe_mul(x, w);
      }
    }
    free(den); free(inv);
    return GS_OK;
}

/* ---- MerkleTree.proveBatch restated (same layout as oracle/pyref.py MerkleTree.prove_batch) ---- */
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
int gs_merkle_prove_batch(gs_ctx *c, const void *leaves, const void *nodes, uint64_t n, const uint64_t *idx, uint32_t count,
                          uint8_t *values_out, uint32_t *ncols_out, uint32_t *col_lens, uint8_t *nodes_out, uint64_t cap) {
    if (!is_pow2(n) || n < 2 || !count) return fail(c, GS_ERR_ARG, "merkle_prove_batch: bad arguments");
    const uint8_t *lv = (const uint8_t *)leaves, *nd = (const uint8_t *)nodes;
    uint64_t *sorted = (uint64_t *)malloc(count * sizeof(uint64_t) * 4);
    if (!sorted) return fail(c, GS_ERR_OOM, "malloc failed");
    uint64_t *norm = sorted + count, *cur = norm + count, *nxt = cur + count;
    for (uint32_t i = 0; i < count; i++) {
        if (idx[i] >= n) { free(sorted); return fail(c, GS_ERR_ARG, "merkle_prove_batch: index out of range"); }
        sorted[i] = idx[i];
        memcpy(values_out + 32 * (uint64_t)i, lv + 32 * idx[i], 32);
    }
    qsort(sorted, count, sizeof(uint64_t), cmp_u64);
    uint32_t ncols = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (i && sorted[i] == sorted[i - 1]) { free(sorted); return fail(c, GS_ERR_ARG, "merkle_prove_batch: repeating indexes"); }
        uint64_t e = sorted[i] & ~1ull;
        if (!ncols || norm[ncols - 1] != e) norm[ncols++] = e;
    }
    /* column storage: per column up to depth digests */
    int depth = 0;
    while ((1ull << depth) < n) depth++;
    uint8_t *cols = (uint8_t *)malloc((size_t)ncols * depth * 32);
    if (!cols) { free(sorted); return fail(c, GS_ERR_OOM, "malloc failed"); }
    for (uint32_t i = 0; i < ncols; i++) col_lens[i] = 0;
#define PUSH(col, src) do { memcpy(cols + ((size_t)(col) * depth + col_lens[col]) * 32, (src), 32); col_lens[col]++; } while (0)
    uint32_t pos = 0; /* walk sorted[] alongside norm[] to know which of the pair was requested */
    for (uint32_t i = 0; i < ncols; i++) {
        uint64_t e = norm[i];
        int has0 = 0, has1 = 0;
        while (pos < count && (sorted[pos] & ~1ull) == e) { if (sorted[pos] & 1) has1 = 1; else has0 = 1; pos++; }
        if (has0 && !has1) PUSH(i, lv + 32 * (e + 1));
        else if (!has0 && has1) PUSH(i, lv + 32 * e);
        cur[i] = (e + n) >> 1;
    }
    uint32_t len = ncols;
    for (int d = depth - 1; d > 0; d--) {
        uint32_t nl = 0;
        for (uint32_t i = 0; i < len; i++) {
            uint64_t sib = cur[i] ^ 1;
            if (i + 1 < len && cur[i + 1] == sib) i++;
            else PUSH(i, nd + 32 * sib);
            nxt[nl++] = sib >> 1;
        }
        uint64_t *t = cur; cur = nxt; nxt = t;
        len = nl;
    }
#undef PUSH
    uint64_t total = 0;
    for (uint32_t i = 0; i < ncols; i++) total += col_lens[i];
    int rc = GS_OK;
    if (total > cap) rc = fail(c, GS_ERR_ARG, "merkle_prove_batch: nodes_out too small");
    else {
        uint64_t o = 0;
        for (uint32_t i = 0; i < ncols; i++) { memcpy(nodes_out + 32 * o, cols + (size_t)i * depth * 32, 32 * (size_t)col_lens[i]); o += col_lens[i]; }
        *ncols_out = ncols;
    }
    free(cols);
    free(sorted);
    return rc;
}

int gs_small_interpolate(const uint8_t *xs, const uint8_t *ys, uint32_t n, uint8_t *out) {
    if (n == 0 || n > 4096) return GS_ERR_ARG;
    fe *x = (fe *)malloc(sizeof(fe) * n * 4);
    if (!x) return GS_ERR_OOM;
    fe *y = x + n, *num = y + n, *acc = num + n;
    for (uint32_t i = 0; i < n; i++) { x[i] = fe_load(xs + FE_BYTES * i); y[i] = fe_load(ys + FE_BYTES * i); acc[i] = 0; }
    for (uint32_t j = 0; j < n; j++) { /* plain O(n^2)-per-basis Lagrange: rebuild each numerator from scratch */
        uint32_t deg = 0;
        fe den = 1;
        num[0] = 1;
        for (uint32_t m = 0; m < n; m++) {
            if (m == j) continue;
            fe nx = fe_neg(x[m]);
            num[deg + 1] = num[deg];
            for (uint32_t d = deg; d >= 1; d--) num[d] = fe_add(num[d - 1], fe_mul(num[d], nx));
            num[0] = fe_mul(num[0], nx);
            deg++;
            den = fe_mul(den, fe_sub(x[j], x[m]));
        }
        fe s = fe_mul(y[j], fe_inv(den));
        for (uint32_t d = 0; d < n; d++) acc[d] = fe_add(acc[d], fe_mul(num[d], s));
    }
    for (uint32_t d = 0; d < n; d++) fe_store(out + FE_BYTES * d, acc[d]);
    free(x);
    return GS_OK;
}
int gs_small_eval_poly(const uint8_t *poly, uint32_t len, const uint8_t *xs, uint32_t m, uint8_t *out) {
    for (uint32_t i = 0; i < m; i++) {
        fe x = fe_load(xs + FE_BYTES * i), s = 0;
        for (uint32_t k = len; k-- > 0;) s = fe_add(fe_mul(s, x), fe_load(poly + FE_BYTES * k));
        fe_store(out + FE_BYTES * i, s);
    }
    return GS_OK;
}

/* ---- generic AIR programs (include/gstark.h): the simplest possible interpreter, both for the trace and per point ---- */
static fe vm_pow_u32(fe b, uint32_t e) { return fe_exp(b, (fexp)e); }
static int air_check(gs_ctx *c, const uint32_t *code, uint32_t n, uint32_t nconsts, uint32_t vm, uint32_t regs, uint32_t nstatic, uint32_t nout, int allow_next) {
    if (!code || !n || !vm || vm > GS_AIR_MAX_VM_REGS || !regs || regs > GS_AIR_MAX_REGISTERS || nstatic > GS_AIR_MAX_REGISTERS) return fail(c, GS_ERR_ARG, "air program: bad shape");
    for (uint32_t pc = 0; pc < n; pc++) {
        uint32_t op = code[4 * pc], d = code[4 * pc + 1], a = code[4 * pc + 2], b = code[4 * pc + 3];
        int ok;
        switch (op) {
            case 0: ok = d < vm && a < nconsts; break;
            case 1: ok = d < vm && a < regs; break;
            case 2: ok = allow_next && d < vm && a < regs; break;
            case 3: ok = d < vm && a < nstatic; break;
            case 4: case 5: case 6: ok = d < vm && a < vm && b < vm; break;
            case 7: ok = d < vm && a < vm; break;
            case 8: ok = d < vm && a < vm && b < nconsts; break;
            case 9: ok = d < nout && a < vm; break;
            default: ok = 0;
        }
        if (!ok) return fail(c, GS_ERR_ARG, "air program: invalid instruction");
    }
    return GS_OK;
}
int gs_air_trace(gs_ctx *c, const uint32_t *code, uint32_t n, const uint8_t *consts, uint32_t nconsts, uint32_t vmn, uint32_t regs,
                 const uint8_t *svals, const uint32_t *speriods, uint32_t nstatic, const uint8_t *row0, uint64_t steps, void *out) {
    if (air_check(c, code, n, nconsts, vmn, regs, nstatic, regs, 0)) return GS_ERR_ARG;
    if (!steps) return fail(c, GS_ERR_ARG, "air_trace: empty");
    fe vm[GS_AIR_MAX_VM_REGS], row[GS_AIR_MAX_REGISTERS], next[GS_AIR_MAX_REGISTERS];
    uint64_t soff[GS_AIR_MAX_REGISTERS];
    uint64_t o = 0;
    for (uint32_t s = 0; s < nstatic; s++) { if (!speriods[s]) return fail(c, GS_ERR_ARG, "air_trace: empty static register"); soff[s] = o; o += speriods[s]; }
    for (uint32_t r = 0; r < regs; r++) row[r] = fe_load(row0 + FE_BYTES * r);
    for (uint64_t i = 0; i < steps; i++) {
        for (uint32_t r = 0; r < regs; r++) ST(out, (uint64_t)r * steps + i, row[r]);
        if (i + 1 == steps) break;
        for (uint32_t r = 0; r < regs; r++) next[r] = row[r];
        for (uint32_t pc = 0; pc < n; pc++) {
            uint32_t op = code[4 * pc], d = code[4 * pc + 1], a = code[4 * pc + 2], b = code[4 * pc + 3];
            switch (op) {
                case 0: vm[d] = fe_load(consts + FE_BYTES * a); break;
                case 1: vm[d] = row[a]; break;
                case 3: vm[d] = fe_load(svals + FE_BYTES * (soff[a] + i % speriods[a])); break;
                case 4: vm[d] = fe_add(vm[a], vm[b]); break;
                case 5: vm[d] = fe_sub(vm[a], vm[b]); break;
                case 6: vm[d] = fe_mul(vm[a], vm[b]); break;
                case 7: vm[d] = vm_pow_u32(vm[a], b); break;
                case 8: vm[d] = fe_exp(vm[a], fe_load(consts + FE_BYTES * b)); break;
                default: next[d] = vm[a]; break;
            }
        }
        for (uint32_t r = 0; r < regs; r++) row[r] = next[r];
    }
    return GS_OK;
}
/* include/gstark.h gs_air_trace_segments: independent runs, each the serial loop above started from its own first row */
int gs_air_trace_segments(gs_ctx *c, const uint32_t *code, uint32_t n, const uint32_t *icode, uint32_t in, const uint8_t *consts, uint32_t nconsts,
                          uint32_t vmn, uint32_t regs, const uint8_t *svals, const uint32_t *speriods, uint32_t nstatic, const uint8_t *rows0,
                          uint64_t segments, uint64_t seglen, void *out) {
    if (air_check(c, code, n, nconsts, vmn, regs, nstatic, regs, 0)) return GS_ERR_ARG;
    if (in && air_check(c, icode, in, nconsts, vmn, regs, 0, regs, 0)) return GS_ERR_ARG;
    if (!segments || !seglen) return fail(c, GS_ERR_ARG, "air_trace_segments: empty");
    fe vm[GS_AIR_MAX_VM_REGS], row[GS_AIR_MAX_REGISTERS], next[GS_AIR_MAX_REGISTERS];
    uint64_t soff[GS_AIR_MAX_REGISTERS], o = 0, steps = segments * seglen;
    for (uint32_t s = 0; s < nstatic; s++) { if (!speriods[s]) return fail(c, GS_ERR_ARG, "air_trace_segments: empty static register"); soff[s] = o; o += speriods[s]; }
    for (uint64_t g = 0; g < segments; g++) {
        for (uint32_t r = 0; r < regs; r++) row[r] = fe_load(rows0 + FE_BYTES * (g * regs + r));
        if (in) {   /* init block: inputs -> first row */
            for (uint32_t r = 0; r < regs; r++) next[r] = row[r];
            for (uint32_t pc = 0; pc < in; pc++) {
                uint32_t op = icode[4 * pc], d = icode[4 * pc + 1], a = icode[4 * pc + 2], b = icode[4 * pc + 3];
                switch (op) {
                    case 0: vm[d] = fe_load(consts + FE_BYTES * a); break;
                    case 1: vm[d] = row[a]; break;
                    case 4: vm[d] = fe_add(vm[a], vm[b]); break;
                    case 5: vm[d] = fe_sub(vm[a], vm[b]); break;
                    case 6: vm[d] = fe_mul(vm[a], vm[b]); break;
                    case 7: vm[d] = vm_pow_u32(vm[a], b); break;
                    case 8: vm[d] = fe_exp(vm[a], fe_load(consts + FE_BYTES * b)); break;
                    default: next[d] = vm[a]; break;
                }
            }
            for (uint32_t r = 0; r < regs; r++) row[r] = next[r];
        }
        for (uint64_t k = 0; k < seglen; k++) {
            const uint64_t i = g * seglen + k;
            for (uint32_t r = 0; r < regs; r++) ST(out, (uint64_t)r * steps + i, row[r]);
            if (k + 1 == seglen) break;
            for (uint32_t r = 0; r < regs; r++) next[r] = row[r];
            for (uint32_t pc = 0; pc < n; pc++) {
                uint32_t op = code[4 * pc], d = code[4 * pc + 1], a = code[4 * pc + 2], b = code[4 * pc + 3];
                switch (op) {
                    case 0: vm[d] = fe_load(consts + FE_BYTES * a); break;
                    case 1: vm[d] = row[a]; break;
                    case 3: vm[d] = fe_load(svals + FE_BYTES * (soff[a] + i % speriods[a])); break;
                    case 4: vm[d] = fe_add(vm[a], vm[b]); break;
                    case 5: vm[d] = fe_sub(vm[a], vm[b]); break;
                    case 6: vm[d] = fe_mul(vm[a], vm[b]); break;
                    case 7: vm[d] = vm_pow_u32(vm[a], b); break;
                    case 8: vm[d] = fe_exp(vm[a], fe_load(consts + FE_BYTES * b)); break;
                    default: next[d] = vm[a]; break;
                }
            }
            for (uint32_t r = 0; r < regs; r++) row[r] = next[r];
        }
    }
    return GS_OK;
}
/* registers read in place from the columns of a larger domain (include/gstark.h): register a at point j is p[a * prow + j * pstride] */
int gs_air_constraints_strided(gs_ctx *c, const uint32_t *code, uint32_t n, const uint8_t *consts, uint32_t nconsts, uint32_t vmn, uint32_t regs,
                               uint32_t ncons, const void *p, uint64_t prow, uint64_t pstride, uint64_t nc, uint64_t shift, const void *stab,
                               const uint64_t *slens, uint32_t nstatic, void *out) {
    if (air_check(c, code, n, nconsts, vmn, regs, nstatic, ncons, 1)) return GS_ERR_ARG;
    if (!nc) return fail(c, GS_ERR_ARG, "air_constraints: empty domain");
    if (!pstride || (nc - 1) > UINT64_MAX / pstride || (nc - 1) * pstride >= prow) return fail(c, GS_ERR_ARG, "air_constraints: points at this stride do not fit the rows");
    uint64_t soff[GS_AIR_MAX_REGISTERS], o = 0;
    for (uint32_t s = 0; s < nstatic; s++) { if (!slens[s]) return fail(c, GS_ERR_ARG, "air_constraints: empty static table"); soff[s] = o; o += slens[s]; }
    PAR_FOR
    for (uint64_t j = 0; j < nc; j++) {
        fe vm[GS_AIR_MAX_VM_REGS];       /* (the points are independent: a scratch file per iteration) */
        uint64_t jn = (j + shift) % nc;
        for (uint32_t pc = 0; pc < n; pc++) {
            uint32_t op = code[4 * pc], d = code[4 * pc + 1], a = code[4 * pc + 2], b = code[4 * pc + 3];
            switch (op) {
                case 0: vm[d] = fe_load(consts + FE_BYTES * a); break;
                case 1: vm[d] = EL(p, (uint64_t)a * prow + j * pstride); break;
                case 2: vm[d] = EL(p, (uint64_t)a * prow + jn * pstride); break;
                case 3: vm[d] = EL(stab, soff[a] + j % slens[a]); break;
                case 4: vm[d] = fe_add(vm[a], vm[b]); break;
                case 5: vm[d] = fe_sub(vm[a], vm[b]); break;
                case 6: vm[d] = fe_mul(vm[a], vm[b]); break;
                case 7: vm[d] = vm_pow_u32(vm[a], b); break;
                case 8: vm[d] = fe_exp(vm[a], fe_load(consts + FE_BYTES * b)); break;
                default: ST(out, (uint64_t)d * nc + j, vm[a]); break;
            }
        }
    }
    return GS_OK;
}
int gs_air_constraints(gs_ctx *c, const uint32_t *code, uint32_t n, const uint8_t *consts, uint32_t nconsts, uint32_t vmn, uint32_t regs,
                       uint32_t ncons, const void *p, uint64_t nc, uint64_t shift, const void *stab, const uint64_t *slens, uint32_t nstatic, void *out) {
    return gs_air_constraints_strided(c, code, n, consts, nconsts, vmn, regs, ncons, p, nc, 1, nc, shift, stab, slens, nstatic, out);
}

/* include/gstark.h gs_pseudorandom_indexes — QueryIndexGenerator.ts:39-67 on the host */
#define SHA256(m, l, o) orc_sha256((m), (l), (o))
int gs_pseudorandom_indexes(const uint8_t *seed, uint32_t seed_len, uint32_t count, uint64_t max_, uint32_t exclude, uint64_t *out) {
    if (!seed || !out || !max_ || (count && !out)) return GS_ERR_ARG;
    uint64_t max_count = exclude ? max_ - max_ / exclude : max_;
    if (max_count < count) return GS_ERR_ARG;
    uint8_t st[32], msg[33], dg[32];
    SHA256(seed, seed_len, st);                       /* state: 256-bit big-endian integer */
    uint32_t found = 0;
    for (uint64_t i = 0; i < (uint64_t)count * 1000 && found < count; i++) {
        /* v = state + i, 33 bytes big-endian */
        uint64_t carry = i;
        msg[0] = 0;
        for (int k = 31; k >= 0; k--) { uint64_t t = (uint64_t)st[k] + (carry & 0xFF); msg[k + 1] = (uint8_t)t; carry = (carry >> 8) + (t >> 8); }
        msg[0] = (uint8_t)carry;
        /* hex digits without leading zeros; an odd count drops the last nibble: bytes = (v >> 4) then */
        int lead = 0;
        while (lead < 33 && msg[lead] == 0) lead++;
        int nhex = lead == 33 ? 0 : (33 - lead) * 2 - ((msg[lead] >> 4) == 0 ? 1 : 0);
        uint8_t buf[33];
        int nbytes = nhex / 2;
        if (nhex & 1) {                               /* shift right by one nibble */
            for (int k = 0; k < nbytes; k++) {
                /* byte k of the result = nibbles 2k, 2k+1 of the hex string */
                int hi_n = 2 * k, lo_n = 2 * k + 1;   /* nibble index from the most significant nibble of the string */
                int first = lead * 2 + 1;             /* string starts at the low nibble of msg[lead] */
                int a = first + hi_n, b = first + lo_n;
                uint8_t na = (a & 1) ? (msg[a >> 1] & 15) : (msg[a >> 1] >> 4);
                uint8_t nb = (b & 1) ? (msg[b >> 1] & 15) : (msg[b >> 1] >> 4);
                buf[k] = (uint8_t)((na << 4) | nb);
            }
        } else {
            for (int k = 0; k < nbytes; k++) buf[k] = msg[lead + k];
        }
        SHA256(buf, (size_t)nbytes, dg);
        /* index = dg (big-endian 256-bit) mod max */
        unsigned __int128 r = 0;
        for (int k = 0; k < 32; k++) r = ((r << 8) | dg[k]) % max_;
        uint64_t index = (uint64_t)r;
        if (exclude && index % exclude == 0) continue;
        int dup = 0;
        for (uint32_t k = 0; k < found; k++) if (out[k] == index) { dup = 1; break; }
        if (dup) continue;
        out[found++] = index;
    }
    return found == count ? GS_OK : GS_ERR_ARG;
}
#undef SHA256
