// tools/microbench_hash.hip — issue cost of one BLAKE2s / SHA-256 compression on gfx950 (registers only, no memory),
// 8 waves/SIMD resident.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_hash.hip -o tools/microbench_hash
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../genstark_amd/csrc/hash_core.h"

template <int ALG>
__global__ void k(uint32_t *out, int iters) {
    uint32_t h[8], m[16];
    for (int i = 0; i < 8; i++) h[i] = threadIdx.x * 8 + i;
    for (int i = 0; i < 16; i++) m[i] = blockIdx.x * 16 + i;
    for (int it = 0; it < iters; it++) {
        if (ALG == 1) b2s_compress(h, m, 64, true);
        else { uint32_t w[16]; for (int i = 0; i < 16; i++) w[i] = m[i]; sha256_compress(h, w); }
        m[it & 15] ^= h[it & 7];   // keep the message live and varying
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= h[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 2000;   // 8 waves per SIMD
    uint32_t *buf;
    hipMalloc(&buf, blocks * threads * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int alg = 0; alg < 2; alg++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (alg) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, buf, iters);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, buf, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            double hashes = (double)blocks * threads * iters;
            if (rep) printf("%-10s %8.3f ms  %7.2f G compressions/s  => %.0f cycles per wave-compression @2.4GHz/1024 SIMDs\n",
                            alg ? "blake2s" : "sha256", ms, hashes / ms / 1e6, 2.4e9 * 1024 * 64 / (hashes / (ms * 1e-3)));
        }
    }
    return 0;
}
