// air_mimc.hip — the air-assembly ProvingContext pieces on the prove() path for the MiMC AIR
// (examples/mimc/mimc128Assembly.ts:28-51): execution trace generation and transition-constraint
// evaluation over the composition domain (lib/Stark.ts:97; lib/components/CompositionPolynomial.ts:76).
#include "common.h"

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
// the reference (generated JS over one input) it runs on one host core, here on native 64-bit limbs.
typedef unsigned __int128 u128;
static inline u128 host_reduce(u128 hi, u128 lo) {
    const u128 C = (u128)0x8FFFFFFFFull;  // 2^128 mod p
    const u128 P = ((u128)0xFFFFFFFFFFFFFFFFull << 64) | 0xFFFFFFF700000001ull;
    u128 m0 = (u128)(uint64_t)hi * C, m1 = (u128)(uint64_t)(hi >> 64) * C;
    u128 tl = m0 + (m1 << 64);
    u128 th = (m1 >> 64) + (tl < m0);
    u128 s = tl + lo;
    unsigned k = s < tl;
    u128 s2 = s + th * C;
    k += s2 < s;
    while (k) { u128 s3 = s2 + C; k -= 1; k += s3 < s2; s2 = s3; }
    while (s2 >= P) s2 -= P;
    return s2;
}
static inline u128 host_mul(u128 a, u128 b) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u128 mid = p01 + p10, midc = mid < p01;
    u128 lo = p00 + (mid << 64), c1 = lo < p00;
    u128 hi = p11 + (mid >> 64) + (midc << 64) + c1;
    return host_reduce(hi, lo);
}
static inline u128 host_add(u128 a, u128 b) {
    const u128 P = ((u128)0xFFFFFFFFFFFFFFFFull << 64) | 0xFFFFFFF700000001ull;
    u128 s = a + b;
    if (s < a || s >= P) s -= P;
    return s;
}

extern "C" {

int gs_mimc_trace(gs_ctx *c, const uint8_t seed[16], const uint8_t *rc_host, uint32_t nrc, uint64_t steps, void *out) {
    if (!c || !seed || !rc_host || !out) return GS_ERR_ARG;
    if (!nrc || !steps) return gs_fail(c, GS_ERR_ARG, "mimc_trace: empty");
    std::vector<u128> rc(nrc);
    for (uint32_t i = 0; i < nrc; i++) memcpy(&rc[i], rc_host + 16 * i, 16);
    void *h = nullptr;
    GS_HIP(c, hipHostMalloc(&h, steps * 16, hipHostMallocDefault));
    u128 *t = (u128 *)h;
    u128 x;
    memcpy(&x, seed, 16);
    uint32_t ri = 0;
    for (uint64_t i = 0; i < steps; i++) {
        t[i] = x;
        x = host_add(host_mul(host_mul(x, x), x), rc[ri]);
        if (++ri == nrc) ri = 0;
    }
    hipError_t e = hipMemcpyAsync(out, h, steps * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipHostFree(h);
    if (e != hipSuccess) return gs_fail(c, GS_ERR_DEVICE, "mimc_trace copy: %s", hipGetErrorString(e));
    return GS_OK;
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
