// tools/microbench.hip — instruction-rate and modular-multiply roofs on gfx950 (the "second roof" beside HBM).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../genstark_amd/csrc/gf128.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ILP>
__global__ void k_mad(uint32_t *out, uint32_t a, uint32_t b, int iters) {
    uint64_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
    uint32_t x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = (uint64_t)x * (uint32_t)(acc[i] >> 7 | 1) + acc[i];
        x += y;
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int ILP>
__global__ void k_mullo(uint32_t *out, uint32_t a, int iters) {
    uint32_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = acc[i] * acc[i] + 1;
    }
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mulhi(uint32_t *out, uint32_t a, int iters) {
    uint32_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = __umulhi(acc[i], acc[i] | 0x80000000u) + 3;
    }
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mul24(uint32_t *out, uint32_t a, int iters) {
    uint32_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = __umul24(acc[i], acc[i] ^ 0x5555) + 1;
    }
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_add32(uint32_t *out, uint32_t a, int iters) {
    uint32_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = (acc[i] + a) ^ (acc[i] >> 3);   // 3 VALU ops (add, shift, xor) or fused
    }
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_dfma(double *out, double a, int iters) {
    double acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = __builtin_fma(acc[i], a, 1.0);
    }
    double s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_femul(fe *out, fe a, int iters) {
    fe acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = fe_make(threadIdx.x + 1, i + 7, blockIdx.x, 11);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = fe_mul(acc[i], a);
    }
    fe s = acc[0];
    for (int i = 1; i < ILP; i++) s = fe_add(s, acc[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_feadd(fe *out, fe a, int iters) {
    fe acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = fe_make(threadIdx.x + 1, i + 7, blockIdx.x, 11);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = fe_add(acc[i], a);
    }
    fe s = acc[0];
    for (int i = 1; i < ILP; i++) s = fe_sub(s, acc[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <typename F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    void *buf;
    CK(hipMalloc(&buf, (size_t)blocks * threads * 16));
    const double lanes = (double)blocks * threads;
    double ms;
#define RUN(NAME, KERNEL, OPS_PER_ITER, ...) \
    ms = time_ms([&] { hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(threads), 0, 0, __VA_ARGS__); }); \
    printf("%-28s %8.3f ms  %8.2f Gop/s (lane-ops)  => %.2f cycles/wave-instr @2.4GHz/1024 SIMDs\n", NAME, ms, lanes * iters * (OPS_PER_ITER) / ms / 1e6, \
           (ms * 1e-3 * 2.4e9 * 1024.0) / (lanes / 64.0 * iters * (OPS_PER_ITER)));
    RUN("v_mad_u64_u32 ilp8", k_mad<8>, 8, (uint32_t *)buf, 12345u, 7u, iters)
    RUN("v_mul_lo_u32(+add) ilp8", k_mullo<8>, 8, (uint32_t *)buf, 3u, iters)
    RUN("v_mul_hi_u32(+add) ilp8", k_mulhi<8>, 8, (uint32_t *)buf, 3u, iters)
    RUN("v_mul_u32_u24(+add) ilp8", k_mul24<8>, 8, (uint32_t *)buf, 3u, iters)
    RUN("add/shift/xor x3 ilp8", k_add32<8>, 24, (uint32_t *)buf, 3u, iters)
    RUN("v_fma_f64 ilp8", k_dfma<8>, 8, (double *)buf, 1.0000001, iters)
    fe a = fe_make(0x12345678, 0x9abcdef0, 0x0fedcba9, 0x76543210);
    RUN("fe_mul ilp4", k_femul<4>, 4, (fe *)buf, a, iters)
    RUN("fe_mul ilp8", k_femul<8>, 8, (fe *)buf, a, iters)
    RUN("fe_add ilp8", k_feadd<8>, 8, (fe *)buf, a, iters)
    // HBM copy
    const uint64_t n = 1ull << 26;  // 1 GiB in, 1 GiB out
    uint4 *in, *out;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16));
    CK(hipMemset(in, 1, n * 16));
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, in, out, n); });
    printf("%-28s %8.3f ms  %8.2f GB/s (read+write)\n", "uint4 copy 1GiB->1GiB", ms, 2.0 * n * 16 / ms / 1e6);
    return 0;
}
