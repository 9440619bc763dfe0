// tools/microbench_latency.hip — latency of a DEPENDENT chain of field products with one wave per SIMD (the regime of the trace-segment
// kernels: a few thousand threads on a chip of 1024 SIMDs), for the throughput-oriented fe_mul and for chains interleaved in one thread.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_latency.hip -o tools/microbench_latency && tools/microbench_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../genstark_amd/csrc/gf128.h"

template <int CHAINS>
__global__ void k_chain(const fe *in, fe *out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    fe x[CHAINS];
    const fe y = in[1];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = in[(t + c) & 7];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) x[c] = fe_mul(x[c], y);
    }
    fe s = x[0];
#pragma unroll
    for (int c = 1; c < CHAINS; c++) s = fe_add(s, x[c]);
    out[t] = s;
}
template <int CHAINS>
__global__ void k_chain_add(const fe *in, fe *out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    fe x[CHAINS];
    const fe y = in[1];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = in[(t + c) & 7];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) x[c] = fe_add(x[c], y);
    }
    fe s = x[0];
#pragma unroll
    for (int c = 1; c < CHAINS; c++) s = fe_add(s, x[c]);
    out[t] = s;
}

template <typename K>
static void run(const char *name, K kernel, int chains, int blocks, int threads, fe *din, fe *dout) {
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, din, dout, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-34s %4d blocks x %3d threads: %8.3f ms  %7.0f cycles per step of %d product(s) -> %6.0f cycles per product (2.4 GHz)\n", name, blocks, threads, ms,
           ms * 1e-3 * 2.4e9 / iters, chains, ms * 1e-3 * 2.4e9 / iters / chains);
}

int main() {
    fe h[8];
    for (int i = 0; i < 8; i++) h[i] = fe_make(0x12345678u + i, 0x9abcdef0u, 0x0fedcba9u, 0x07654321u);
    fe *din, *dout;
    hipMalloc(&din, sizeof h); hipMalloc(&dout, 1 << 24);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    run("fe_mul, 1 chain, 1 wave/SIMD", k_chain<1>, 1, 1024, 64, din, dout);
    run("fe_mul, 2 chains", k_chain<2>, 2, 1024, 64, din, dout);
    run("fe_mul, 4 chains", k_chain<4>, 4, 1024, 64, din, dout);
    run("fe_mul, 8 chains", k_chain<8>, 8, 1024, 64, din, dout);
    run("fe_mul, 1 chain, 16 waves only", k_chain<1>, 1, 16, 64, din, dout);
    run("fe_mul, 1 chain, 8 waves/SIMD", k_chain<1>, 1, 8192, 64, din, dout);
    run("fe_add, 1 chain, 1 wave/SIMD", k_chain_add<1>, 1, 1024, 64, din, dout);
    run("fe_add, 4 chains", k_chain_add<4>, 4, 1024, 64, din, dout);
    return 0;
}
