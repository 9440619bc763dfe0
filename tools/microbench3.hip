// tools/microbench3.hip — does hand-interleaving two independent modular multiplications at source level remove the
// carry-hazard s_nops the compiler leaves in fe_mul?  (experiment; build like microbench.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "../genstark_amd/csrc/gf128.h"

// two independent products, statement-interleaved
#define ROW2(A, B, ai_a, ai_b, OA, OB)                                                                             \
    {                                                                                                              \
        uint64_t pa0 = (uint64_t)(ai_a) * A##b[0], pb0 = (uint64_t)(ai_b) * B##b[0];                               \
        uint64_t pa1 = (uint64_t)(ai_a) * A##b[1], pb1 = (uint64_t)(ai_b) * B##b[1];                               \
        uint64_t pa2 = (uint64_t)(ai_a) * A##b[2], pb2 = (uint64_t)(ai_b) * B##b[2];                               \
        uint64_t pa3 = (uint64_t)(ai_a) * A##b[3], pb3 = (uint64_t)(ai_b) * B##b[3];                               \
        uint32_t ca, cb;                                                                                           \
        OA[0] = (uint32_t)pa0; OB[0] = (uint32_t)pb0;                                                              \
        OA[1] = gf_addc((uint32_t)pa1, (uint32_t)(pa0 >> 32), 0u, ca); OB[1] = gf_addc((uint32_t)pb1, (uint32_t)(pb0 >> 32), 0u, cb); \
        OA[2] = gf_addc((uint32_t)pa2, (uint32_t)(pa1 >> 32), ca, ca); OB[2] = gf_addc((uint32_t)pb2, (uint32_t)(pb1 >> 32), cb, cb); \
        OA[3] = gf_addc((uint32_t)pa3, (uint32_t)(pa2 >> 32), ca, ca); OB[3] = gf_addc((uint32_t)pb3, (uint32_t)(pb2 >> 32), cb, cb); \
        OA[4] = (uint32_t)(pa3 >> 32) + ca; OB[4] = (uint32_t)(pb3 >> 32) + cb;                                    \
    }

__device__ __forceinline__ void fe_mul2(const fe &a0, const fe &b0, const fe &a1, const fe &b1, fe &o0, fe &o1) {
    const uint32_t Ab[4] = {b0.w0, b0.w1, b0.w2, b0.w3}, Bb[4] = {b1.w0, b1.w1, b1.w2, b1.w3};
    uint32_t ra0[5], ra1[5], ra2[5], ra3[5], rb0[5], rb1[5], rb2[5], rb3[5];
    ROW2(A, B, a0.w0, a1.w0, ra0, rb0)
    ROW2(A, B, a0.w1, a1.w1, ra1, rb1)
    ROW2(A, B, a0.w2, a1.w2, ra2, rb2)
    ROW2(A, B, a0.w3, a1.w3, ra3, rb3)
    uint32_t x[8], y[8], ca, cb;
#define BOTH(EA, EB) EA; EB;
    BOTH(x[0] = ra0[0], y[0] = rb0[0])
    BOTH(x[1] = gf_addc(ra0[1], ra1[0], 0u, ca), y[1] = gf_addc(rb0[1], rb1[0], 0u, cb))
    BOTH(x[2] = gf_addc(ra0[2], ra1[1], ca, ca), y[2] = gf_addc(rb0[2], rb1[1], cb, cb))
    BOTH(x[3] = gf_addc(ra0[3], ra1[2], ca, ca), y[3] = gf_addc(rb0[3], rb1[2], cb, cb))
    BOTH(x[4] = gf_addc(ra0[4], ra1[3], ca, ca), y[4] = gf_addc(rb0[4], rb1[3], cb, cb))
    BOTH(x[5] = ra1[4] + ca, y[5] = rb1[4] + cb)
    BOTH(x[2] = gf_addc(x[2], ra2[0], 0u, ca), y[2] = gf_addc(y[2], rb2[0], 0u, cb))
    BOTH(x[3] = gf_addc(x[3], ra2[1], ca, ca), y[3] = gf_addc(y[3], rb2[1], cb, cb))
    BOTH(x[4] = gf_addc(x[4], ra2[2], ca, ca), y[4] = gf_addc(y[4], rb2[2], cb, cb))
    BOTH(x[5] = gf_addc(x[5], ra2[3], ca, ca), y[5] = gf_addc(y[5], rb2[3], cb, cb))
    BOTH(x[6] = ra2[4] + ca, y[6] = rb2[4] + cb)
    BOTH(x[3] = gf_addc(x[3], ra3[0], 0u, ca), y[3] = gf_addc(y[3], rb3[0], 0u, cb))
    BOTH(x[4] = gf_addc(x[4], ra3[1], ca, ca), y[4] = gf_addc(y[4], rb3[1], cb, cb))
    BOTH(x[5] = gf_addc(x[5], ra3[2], ca, ca), y[5] = gf_addc(y[5], rb3[2], cb, cb))
    BOTH(x[6] = gf_addc(x[6], ra3[3], ca, ca), y[6] = gf_addc(y[6], rb3[3], cb, cb))
    BOTH(x[7] = ra3[4] + ca, y[7] = rb3[4] + cb)
    o0 = fe_reduce_wide(x);
    o1 = fe_reduce_wide(y);
}

template <int MODE>
__global__ void k_femul(fe *out, fe a, int iters) {
    fe acc[8];
    for (int i = 0; i < 8; i++) acc[i] = fe_make(threadIdx.x + 1, i + 7, blockIdx.x, 11);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = fe_mul(acc[i], a);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i += 2) fe_mul2(acc[i], a, acc[i + 1], a, acc[i], acc[i + 1]);
        }
    }
    fe s = acc[0];
    for (int i = 1; i < 8; i++) s = fe_add(s, acc[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int blocks = 2048, threads = 256, iters = 1000;
    fe *buf, *buf2;
    hipMalloc(&buf, (size_t)blocks * threads * 16);
    hipMalloc(&buf2, (size_t)blocks * threads * 16);
    fe a = fe_make(0x12345678, 0x9abcdef0, 0x0fedcba9, 0x76543210);
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int r = 0; r < 4; r++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_femul<0>, dim3(blocks), dim3(threads), 0, 0, buf, a, iters);
            else hipLaunchKernelGGL(k_femul<1>, dim3(blocks), dim3(threads), 0, 0, buf2, a, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double waves = (double)blocks * threads / 64.0;
        printf("mode %d: %.3f ms  %.1f cycles per wave-modmul\n", mode, best, best * 1e-3 * 2.4e9 * 1024.0 / (waves * iters * 8));
    }
    // results must agree
    size_t n = (size_t)blocks * threads;
    fe *h0 = (fe *)malloc(n * 16), *h1 = (fe *)malloc(n * 16);
    hipMemcpy(h0, buf, n * 16, hipMemcpyDeviceToHost); hipMemcpy(h1, buf2, n * 16, hipMemcpyDeviceToHost);
    printf("results equal: %d\n", memcmp(h0, h1, n * 16) == 0);
    return 0;
}
