// tools/microbench2.hip — exact per-instruction issue cost on gfx950 via inline asm (8 independent chains per lane,
// 8 waves/SIMD resident).  build: hipcc --offload-arch=gfx950 -O3 tools/microbench2.hip -o tools/microbench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 4000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, BODY)                                                              \
    __global__ void NAME(uint32_t *out, uint32_t s) {                                    \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b0 = s, b1 = s + 1, b2 = s + 2, b3 = s + 3, b4 = s + 4, b5 = s + 5, b6 = s + 6, b7 = s + 7; \
        uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7; \
        for (int it = 0; it < ITERS; it++) { BODY }                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7 ^ \
            (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7);                            \
    }

#define X_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_add_u32, REP8(X_ADD))
#define X_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a##i) : "v"(b##i) : "vcc");
KERNEL(k_add_co, REP8(X_ADDCO))
#define X_ADDC(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##i) : "v"(b##i) : "vcc");
KERNEL(k_addc_co, REP8(X_ADDC))
// a 4-limb carry chain: add_co + 3 addc (the fe_add core), 2 independent chains per iteration
#define X_CHAIN(i, j, k, l) asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_addc_co_u32 %2, vcc, %2, %6, vcc\n v_addc_co_u32 %3, vcc, %3, %7, vcc" \
    : "+v"(a##i), "+v"(a##j), "+v"(a##k), "+v"(a##l) : "v"(b##i), "v"(b##j), "v"(b##k), "v"(b##l) : "vcc");
KERNEL(k_chain4, X_CHAIN(0, 1, 2, 3) X_CHAIN(4, 5, 6, 7))
#define X_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(a##i), "v"(b##i) : "vcc");
KERNEL(k_mad_u64_u32, REP8(X_MAD64))
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mul_lo, REP8(X_MULLO))
#define X_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mul_hi, REP8(X_MULHI))
#define X_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mad_u32_u24, REP8(X_MAD24))
#define X_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mul_hi_u24, REP8(X_MULHI24))
#define X_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b##i) : "vcc");
KERNEL(k_cndmask, REP8(X_CND))
#define X_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a##i) : "v"(b##i));
KERNEL(k_alignbit, REP8(X_ALIGN))
#define X_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q##i) : "v"(q##i));
KERNEL(k_lshl_add_u64, REP8(X_LSHLADD64))
#define X_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_add3, REP8(X_ADD3))
#define X_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_xor, REP8(X_XOR))
#define X_ROT(i) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a##i));
KERNEL(k_rotate, REP8(X_ROT))
// VOP3 carry forms with SGPR-pair carries (what the compiler emits for independent chains)
#define X_ADDCO64(i) asm volatile("v_add_co_u32_e64 %0, s[20:21], %0, %1" : "+v"(a##i) : "v"(b##i) : "s20", "s21");
KERNEL(k_add_co_e64, REP8(X_ADDCO64))

template <typename F>
static float time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int blocks = 256 * 8, threads = 256;
    uint32_t *buf;
    hipMalloc(&buf, (size_t)blocks * threads * 4);
    const double waves = (double)blocks * threads / 64.0;
#define RUN(K, PER_ITER) { float ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, buf, 3u); }); \
        printf("%-18s %7.3f ms  %6.2f cycles/wave-instr (2.4 GHz x 1024 SIMDs)\n", #K, ms, ms * 1e-3 * 2.4e9 * 1024.0 / (waves * ITERS * (PER_ITER))); }
    RUN(k_add_u32, 8) RUN(k_xor, 8) RUN(k_rotate, 8) RUN(k_add3, 8) RUN(k_alignbit, 8) RUN(k_cndmask, 8)
    RUN(k_add_co, 8) RUN(k_add_co_e64, 8) RUN(k_addc_co, 8) RUN(k_chain4, 8) RUN(k_lshl_add_u64, 8)
    RUN(k_mad_u64_u32, 8) RUN(k_mul_lo, 8) RUN(k_mul_hi, 8) RUN(k_mad_u32_u24, 8) RUN(k_mul_hi_u24, 8)
    return 0;
}
