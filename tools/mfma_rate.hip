// tools/mfma_rate.hip — issue rate of v_mfma_i32_32x32x32_i8 under the shapes the NTT prototype uses: number of independent accumulator
// chains, operands from registers or from LDS (ds_read_b128 per pair), waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rate.hip -o tools/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int CHAINS, int LDS>
__global__ __launch_bounds__(256) void k_rate(const int4 *in, int *out, int iters) {
    __shared__ int4 tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) tab[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x & 63;
    v4i A[8], B[CHAINS];
    for (int s = 0; s < 8; s++) { const int4 t = tab[(l + 16 * s) & 255]; A[s] = v4i{t.x, t.y, t.z, t.w}; }
    for (int c = 0; c < CHAINS; c++) { const int4 t = in[256 + l + 64 * c]; B[c] = v4i{t.x, t.y, t.z, t.w}; }
    v16i acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = v16i{0};
    const int rr = (l & 3) + 4 * ((l & 31) >> 3);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            v4i a = A[s];
            if (LDS) { const int4 t = tab[(((it + s) & 15) << 4) + rr]; a = v4i{t.x, t.y, t.z, t.w}; }
#pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B[c], acc[c], 0, 0, 0);
        }
    }
    int sum = 0;
    for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 16; i++) sum += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <int CHAINS, int LDS>
static void run(const int4 *din, int *dout, int cus, int blocks_per_cu, const char *name) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate<CHAINS, LDS>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, din, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double mfmas_per_simd = (double)iters * 8 * CHAINS * blocks_per_cu;   // 4 waves per block, one per SIMD
    printf("%-44s %d waves/SIMD: %6.1f ns per MFMA per SIMD, %7.1f TOPS\n", name, blocks_per_cu, best * 1e6 / mfmas_per_simd,
           mfmas_per_simd * 4 * cus * 65536.0 / (best * 1e-3) / 1e12);
}

// Do the matrix pipe and the vector ALU run at the same time?  Every wave alternates a burst of 16 MFMAs with a block of NV plain vector
// instructions (8 independent chains); WHAT: 0 = MFMA bursts only, 1 = vector blocks only, 2 = both, every wave in the same order,
// 3 = both, odd waves of a SIMD start with the vector block (the two waves of a SIMD are then in opposite phases), 4 = both, interleaved
// inside each wave (one MFMA, then NV/16 vector instructions)
template <int WHAT, int NV, int KIND>
__global__ __launch_bounds__(256) void k_mix(const int4 *in, int *out, int iters) {
    const int l = threadIdx.x & 63;
    v4i A, B[2];
    { const int4 t = in[l]; A = v4i{t.x, t.y, t.z, t.w}; }
    for (int c = 0; c < 2; c++) { const int4 t = in[256 + l + 64 * c]; B[c] = v4i{t.x, t.y, t.z, t.w}; }
    v16i acc[2] = {v16i{0}, v16i{0}};
    uint32_t x[8];
    uint64_t y[8];
    for (int i = 0; i < 8; i++) { x[i] = in[512 + l].x + i; y[i] = x[i]; }
    const bool odd = (blockIdx.x & 1) != 0;   // two blocks per CU: wave w of block b sits on SIMD w with wave w of the other block
    auto mf = [&](int n) {
#pragma unroll
        for (int s = 0; s < n; s++) { acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[0], acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[1], acc[1], 0, 0, 0); }
    };
    auto va = [&](int n) {
#pragma unroll
        for (int k = 0; k < n / 8; k++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(k + 1));
                else if (KIND == 1) { uint64_t d, cy; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(x[i]), "v"(x[(i + 1) & 7]), "v"(y[i])); y[i] = d; }
                else asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(x[i]), "+v"(x[(i + 1) & 7]) : "v"(x[(i + 2) & 7]), "v"(x[(i + 3) & 7]) : "vcc");
            }
    };
    for (int it = 0; it < iters; it++) {
        if (WHAT == 0) mf(8);
        else if (WHAT == 1) va(NV);
        else if (WHAT == 2) { mf(8); va(NV); }
        else if (WHAT == 3) { if (odd) { va(NV); mf(8); } else { mf(8); va(NV); } }
        else {
#pragma unroll
            for (int s = 0; s < 16; s++) {
                acc[s & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[s & 1], acc[s & 1], 0, 0, 0);
                asm volatile("" ::: "memory");
                va(NV / 16);
            }
        }
    }
    int sum = 0;
    for (int c = 0; c < 2; c++) for (int i = 0; i < 16; i++) sum += acc[c][i];
    for (int i = 0; i < 8; i++) sum += x[i] + (int)y[i] + (int)(y[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <int WHAT, int NV, int KIND>
static float run_mix(const int4 *din, int *dout, int cus) {
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mix<WHAT, NV, KIND>), dim3(cus * 2), dim3(256), 0, 0, din, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e6f / iters;   // ns per iteration (16 MFMAs + NV vector instructions per wave, 2 waves per SIMD)
}
template <int NV, int KIND>
static void mix_row(const int4 *din, int *dout, int cus) {
    const char *kinds[3] = {"v_xad_u32", "v_mad_u64_u32", "v_add_co/v_addc_co pair"};
    printf("16 MFMA + %4d x %-24s per wave-iteration, 2 waves/SIMD: MFMA only %6.0f ns, vector only %6.0f ns, both/same order %6.0f ns, interleaved %6.0f ns\n", NV, kinds[KIND],
           run_mix<0, NV, KIND>(din, dout, cus), run_mix<1, NV, KIND>(din, dout, cus), run_mix<2, NV, KIND>(din, dout, cus), run_mix<4, NV, KIND>(din, dout, cus));
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int4 *din; int *dout;
    hipMalloc(&din, 1 << 20); hipMemset(din, 1, 1 << 20); hipMalloc(&dout, cus * 8 * 256 * 4);
    run<2, 0>(din, dout, cus, 1, "2 chains, operands in registers"); run<2, 0>(din, dout, cus, 2, "2 chains, operands in registers");
    run<4, 0>(din, dout, cus, 1, "4 chains, operands in registers"); run<4, 0>(din, dout, cus, 2, "4 chains, operands in registers");
    run<1, 0>(din, dout, cus, 2, "1 chain, operands in registers");
    run<2, 1>(din, dout, cus, 1, "2 chains, A from LDS per pair"); run<2, 1>(din, dout, cus, 2, "2 chains, A from LDS per pair");
    run<4, 1>(din, dout, cus, 2, "4 chains, A from LDS per 4");
    mix_row<256, 0>(din, dout, cus); mix_row<256, 1>(din, dout, cus); mix_row<256, 2>(din, dout, cus); mix_row<512, 1>(din, dout, cus);
    return 0;
}
