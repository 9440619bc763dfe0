// air_mimc.hip — the air-assembly ProvingContext pieces on the prove() path for the MiMC AIR
// (examples/mimc/mimc128Assembly.ts:28-51): execution trace generation and transition-constraint
// evaluation over the composition domain (lib/Stark.ts:97; lib/components/CompositionPolynomial.ts:76).
#include "common.h"
#include "host_field.h"

// q[j] = p[(j + shift) mod nc] - (p[j]^3 + k[j mod klen])
__global__ void k_mimc_constraints(const fe *__restrict__ p, uint64_t nc, uint64_t shift, const fe *__restrict__ k, uint64_t klen,
                                   fe *__restrict__ out) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < nc; j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t jn = j + shift;
        if (jn >= nc) jn -= nc;
        fe x = p[j], nx = p[jn];
        fe x3 = fe_mul(fe_mul(x, x), x);
        out[j] = fe_sub(nx, fe_add(x3, k[j % klen]));
    }
}

// CompositionPolynomial.evaluateAll for this AIR in ONE pass over the evaluation domain (see include/gstark.h): per point nine field
// products, reads p[i], p[i + E], k, two entries of the domain's table u = 1/(omega^j - 1); everything that depends on x^T only
// (the degree-adjustment factors and 1/(x^T - 1): E distinct values) comes from two E-entry tables built on the host.
#define GS_MIMC_COMP_MAX_PERIOD 32
#define GS_MIMC_COMP_MAX_ROOTS 4
struct MimcCompArgs {
    fe qz[GS_MIMC_COMP_MAX_PERIOD];   // (d0 + d1 * g^(j*qm)) / (g^j - 1),  g = omega^steps
    fe bt[GS_MIMC_COMP_MAX_PERIOD];   // (b0 + b1 * g^(j*bm)) * prod_a omega^-k_a
    fe lt[GS_MIMC_COMP_MAX_PERIOD];   // l0 + l1 * g^(j*bm)   (linear combination with P; zeros when not asked for)
    fe ipoly[GS_MIMC_COMP_MAX_ROOTS];
    uint64_t root[GS_MIMC_COMP_MAX_ROOTS];
    fe x_last;
};
__global__ void k_mimc_composition(const fe *__restrict__ p, uint64_t n, uint64_t shift, const fe *__restrict__ k, uint64_t klen,
                                   const fe *__restrict__ tw_lo, const fe *__restrict__ tw_hi, int log_lo, int logn, const fe *__restrict__ u,
                                   MimcCompArgs a, uint32_t period, uint32_t nroots, int with_lc, fe *__restrict__ out) {
    __shared__ fe qz[GS_MIMC_COMP_MAX_PERIOD], bt[GS_MIMC_COMP_MAX_PERIOD], lt[GS_MIMC_COMP_MAX_PERIOD];
    if (threadIdx.x < period) { qz[threadIdx.x] = a.qz[threadIdx.x]; bt[threadIdx.x] = a.bt[threadIdx.x]; lt[threadIdx.x] = a.lt[threadIdx.x]; }
    __syncthreads();
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        fe x = tw_lo[i & ((1ull << log_lo) - 1)];
        if (logn > log_lo) x = fe_mul(x, tw_hi[i >> log_lo]);
        const fe pi = p[i], pn = p[(i + shift) & (n - 1)];
        const fe q = fe_sub(pn, fe_add(fe_mul(fe_mul(pi, pi), pi), k[i % klen]));                    // transition constraint
        const fe d = fe_mul(fe_mul(q, fe_sub(x, a.x_last)), qz[i & (period - 1)]);                     // Q * adjustment / Z
        fe iv = a.ipoly[nroots - 1];                                                                   // I(x), Horner
        for (int c = (int)nroots - 2; c >= 0; c--) iv = fe_add(fe_mul(iv, x), a.ipoly[c]);
        fe b = fe_mul(fe_sub(pi, iv), bt[i & (period - 1)]);
        for (uint32_t r = 0; r < nroots; r++) b = fe_mul(b, u[(i + n - a.root[r]) & (n - 1)]);         // / prod (x - x_a)
        fe r = fe_add(d, b);
        if (with_lc) r = fe_add(r, fe_mul(pi, lt[i & (period - 1)]));                                 // wave-uniform
        out[i] = r;
    }
}

// The MiMC recurrence x <- x^3 + k is a serial dependency chain (examples/mimc/utils.ts:7-15): like
// the reference (generated JS over one input) it runs on one host core, here on native 64-bit limbs
// (host_field.h), written straight into a pinned staging buffer that is then copied to the device.
// One chunk of the chain.  Two forms of the same step: the portable one (128-bit temporaries) for any x86-64 / any host, and for cores with
// BMI2 — what MI355X hosts are: EPYC 9005 — the one with mulx products and the carry chains written out by rows, scheduled for Zen 5: 9 % on
// the chain there (tools/trace_bench.cpp, profiles/r05_c_*); picked at run time, never assumed.
#define GS_MIMC_CHUNK_BODY(STEP)                                                                                \
    for (uint64_t i = base; i < end; i++) {                                                                    \
        t[i] = hf_mimc_out(x);                /* the canonical value, beside the chain */                      \
        x = STEP(x, rc[ri]);                  /* the chain itself stays weak (host_field.h) */                 \
        if (++ri == nrc) ri = 0;                                                                               \
    }                                                                                                          \
    *ri_io = ri;                                                                                               \
    return x;
static hfe mimc_chunk_generic(hfe *t, hfe x, const hfe *rc, uint32_t nrc, uint32_t *ri_io, uint64_t base, uint64_t end) {
    uint32_t ri = *ri_io;
    GS_MIMC_CHUNK_BODY(hf_mimc_step_weak)
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(GS_SMALL_Q) && !defined(GS_WIDE_BITS)
#define GS_MIMC_CHUNK_V3 1
__attribute__((target("arch=x86-64-v3,tune=znver5"))) static hfe mimc_chunk_v3(hfe *t, hfe x, const hfe *rc, uint32_t nrc, uint32_t *ri_io, uint64_t base,
                                                                               uint64_t end) {
    uint32_t ri = *ri_io;
    GS_MIMC_CHUNK_BODY(hf_cube_add_rows)      // mulx + adc chains by rows (host_field.h): 7.3 instead of 8.0 ms per 2^20 steps on the EPYC 9575F
}
#endif
extern "C" {

int gs_mimc_trace(gs_ctx *c, const gs_elt *seed, const uint8_t *rc_host, uint32_t nrc, uint64_t steps, void *out) {
    if (!c || !seed || !rc_host || !out) return GS_ERR_ARG;
    if (!nrc || !steps) return gs_fail(c, GS_ERR_ARG, "mimc_trace: empty");
    std::vector<hfe> rc(nrc);
    for (uint32_t i = 0; i < nrc; i++) rc[i] = hf_load(rc_host + GS_ELT * i);
    // pinned, grow-only; waits only for the previous trace's upload, so device work the caller queued before this call
    // (domains, Z(x) inverses, ...) runs while this core grinds through the recurrence
    int rcode = gs_trace_begin(c, steps * GS_ELT);
    if (rcode) return rcode;
    hfe *t = (hfe *)c->h_trace;
    hfe x = hf_load(seed);
    uint32_t ri = 0;
    const uint64_t CHUNK = 1ull << 16;            // copy finished chunks while the next one is being generated
#ifdef GS_MIMC_CHUNK_V3
    static const bool v3 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("fma");
#else
    const bool v3 = false;
#endif
    for (uint64_t base = 0; base < steps; base += CHUNK) {
        const uint64_t end = base + CHUNK < steps ? base + CHUNK : steps;
#ifdef GS_MIMC_CHUNK_V3
        if (v3) x = mimc_chunk_v3(t, x, rc.data(), nrc, &ri, base, end);
        else
#endif
            x = mimc_chunk_generic(t, x, rc.data(), nrc, &ri, base, end);
        (void)v3;
        GS_HIP(c, hipMemcpyAsync((uint8_t *)out + base * GS_ELT, t + base, (end - base) * GS_ELT, hipMemcpyHostToDevice, c->stream));
    }
    return gs_trace_end(c);   // no synchronisation: consumers are ordered on the stream
}

int gs_mimc_constraints(gs_ctx *c, const void *p_comp, uint64_t nc, uint64_t shift, const void *k_table, uint64_t klen, void *out) {
    if (!c || !p_comp || !k_table || !out) return GS_ERR_ARG;
    if (!nc || !klen) return gs_fail(c, GS_ERR_ARG, "mimc_constraints: empty");
    hipLaunchKernelGGL(k_mimc_constraints, dim3(gs_grid(nc)), dim3(256), 0, c->stream, (const fe *)p_comp, nc, shift % nc,
                       (const fe *)k_table, klen, (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_mimc_composition(gs_ctx *c, const void *p_eval, uint64_t n, uint64_t steps, const gs_elt *omega, const void *k_table, uint64_t klen,
                        const uint8_t *coeffs_host, uint64_t q_inc, uint64_t b_inc, const uint8_t *ipoly_host, const uint64_t *root_index_host,
                        uint32_t nroots, const uint8_t *lc_coeffs_host, void *out) {
    if (!c || !p_eval || !omega || !k_table || !coeffs_host || !ipoly_host || !root_index_host || !out) return GS_ERR_ARG;
    if (!gs_is_pow2(n) || !gs_is_pow2(steps) || steps > n || !klen) return gs_fail(c, GS_ERR_ARG, "mimc_composition: bad sizes");
    const uint64_t period = n / steps;
    if (period > GS_MIMC_COMP_MAX_PERIOD || nroots == 0 || nroots > GS_MIMC_COMP_MAX_ROOTS)
        return gs_fail(c, GS_ERR_UNSUPPORTED, "mimc_composition: n / steps <= %d and 1..%d assertions", GS_MIMC_COMP_MAX_PERIOD, GS_MIMC_COMP_MAX_ROOTS);
    if (q_inc % steps || b_inc % steps) return gs_fail(c, GS_ERR_ARG, "mimc_composition: degree increments must be multiples of the trace length");
    const fe w = fe_from_bytes(omega);
    const fe *lo, *hi, *u;
    int log_lo;
    int rc = gs_plan_pow_tables(c, w, n, &lo, &hi, &log_lo);
    if (!rc) rc = gs_plan_inverse_table(c, w, n, &u);
    if (rc) return rc;
    const fe d0 = fe_from_bytes(coeffs_host), d1 = fe_from_bytes(coeffs_host + GS_ELT), b0 = fe_from_bytes(coeffs_host + 2 * GS_ELT),
             b1 = fe_from_bytes(coeffs_host + 3 * GS_ELT);
    const fe l0 = lc_coeffs_host ? fe_from_bytes(lc_coeffs_host) : fe_zero(), l1 = lc_coeffs_host ? fe_from_bytes(lc_coeffs_host + GS_ELT) : fe_zero();
    MimcCompArgs a;
    uint64_t ksum = 0;
    for (uint32_t r = 0; r < GS_MIMC_COMP_MAX_ROOTS; r++) {
        a.root[r] = r < nroots ? root_index_host[r] & (n - 1) : 0;
        a.ipoly[r] = r < nroots ? fe_from_bytes(ipoly_host + GS_ELT * r) : fe_zero();
        if (r < nroots) ksum = (ksum + a.root[r]) & (n - 1);
    }
    const fe scale = gs_memo_pow(c, w, (n - ksum) & (n - 1));                // prod omega^-k_a
    // what depends on the domain and the degrees alone, remembered per context: [3j] 1/(g^j - 1) (j = 0: 0^-1 = 0), [3j+1] g^(j*qm),
    // [3j+2] g^(j*bm) for g = omega^steps (x^steps = g^(i mod period)), then x_last — sixteen inversions of 4 us each otherwise, on the
    // host between the arrival of the evaluation root and this launch
    const std::vector<fe> &tab = gs_memo(c, gs_memo_key("mimc_comp").add(w).add(n).add(steps).add(q_inc).add(b_inc), [&](std::vector<fe> &t) {
        const fe g = fe_pow_u64(w, steps);
        const fe gq = fe_pow_u64(g, (q_inc / steps) % period), gb = fe_pow_u64(g, (b_inc / steps) % period);
        fe gj = fe_one(), gqj = fe_one(), gbj = fe_one();
        for (uint64_t j = 0; j < period; j++) {
            t.push_back(fe_inv(fe_sub(gj, fe_one())));
            t.push_back(gqj);
            t.push_back(gbj);
            gj = fe_mul(gj, g); gqj = fe_mul(gqj, gq); gbj = fe_mul(gbj, gb);
        }
        t.push_back(fe_pow_u64(w, (steps - 1) * period));
    });
    for (uint64_t j = 0; j < GS_MIMC_COMP_MAX_PERIOD; j++) {
        if (j < period) {
            a.qz[j] = fe_mul(fe_add(d0, fe_mul(d1, tab[3 * j + 1])), tab[3 * j]);
            a.bt[j] = fe_mul(fe_add(b0, fe_mul(b1, tab[3 * j + 2])), scale);
            a.lt[j] = fe_add(l0, fe_mul(l1, tab[3 * j + 2]));
        } else {
            a.qz[j] = a.bt[j] = a.lt[j] = fe_zero();
        }
    }
    a.x_last = tab[3 * period];
    gs_traffic(c, 2 * n * GS_ELT, n, "k_mimc_composition");      // P over the evaluation domain in (its neighbour P(x g) is the same vector), L out
    hipLaunchKernelGGL(k_mimc_composition, dim3(gs_grid(n)), dim3(256), 0, c->stream, (const fe *)p_eval, n, period, (const fe *)k_table, klen, lo, hi,
                       log_lo, gs_log2(n), u, a, (uint32_t)period, nroots, lc_coeffs_host ? 1 : 0, (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

}  // extern "C"
