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

// The MiMC recurrence x <- x^3 + k is a serial dependency chain (examples/mimc/utils.ts:7-15): like
// the reference (generated JS over one input) it runs on one host core, here on native 64-bit limbs
// (host_field.h), written straight into a pinned staging buffer that is then copied to the device.
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
    for (uint64_t base = 0; base < steps; base += CHUNK) {
        const uint64_t end = base + CHUNK < steps ? base + CHUNK : steps;
        for (uint64_t i = base; i < end; i++) {
            t[i] = x;
            x = hf_mimc_step(x, rc[ri]);
            if (++ri == nrc) ri = 0;
        }
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

}  // extern "C"
