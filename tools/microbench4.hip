// tools/microbench4.hip — per-instruction issue cost on gfx950 in SHADER CYCLES (s_memtime inside the kernel, not a
// wall-clock / nominal-frequency estimate), 8 independent chains per lane, at 1, 2, 4 and 8 waves per SIMD.
// The table this prints decides which instructions the field arithmetic of ntt.hip is built from (the "second roof").
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench4.hip -o tools/microbench4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define ITERS 2000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, BODY)                                                              \
    __global__ void NAME(uint32_t *out, uint64_t *cyc, uint32_t s) {                      \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b0 = s, b1 = s + 1, b2 = s + 2, b3 = s + 3, b4 = s + 4, b5 = s + 5, b6 = s + 6, b7 = s + 7; \
        uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7; \
        uint64_t r0 = b0, r1 = b1, r2 = b2, r3 = b3, r4 = b4, r5 = b5, r6 = b6, r7 = b7; \
        asm volatile("v_cmp_gt_u32 vcc, 7, %0" :: "v"(a0) : "vcc");                       \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int it = 0; it < ITERS; it++) { BODY }                                      \
        asm volatile("s_nop 0" ::: "memory");                                             \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7 ^ \
            (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32) ^ (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7); \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }

// ---- 32-bit, two sources (VOP2 encodings)
#define X1(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_add_u32, REP8(X1))
#define X2(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_sub_u32, REP8(X2))
#define X3(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_and_b32, REP8(X3))
#define X4(i) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a##i));
KERNEL(k_lshrrev_b32, REP8(X4))
#define X5(i) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a##i));
KERNEL(k_ashrrev_i32, REP8(X5))
#define X6(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a##i));
KERNEL(k_lshlrev_b32, REP8(X6))
#define X7(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mul_u32_u24, REP8(X7))
#define X8(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mul_hi_u32_u24, REP8(X8))
#define X9(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b##i));
KERNEL(k_cndmask_vcc, REP8(X9))
#define X10(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a##i) : "v"(b##i) : "vcc");
KERNEL(k_add_co_u32, REP8(X10))
#define X11(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##i) : "v"(b##i) : "vcc");
KERNEL(k_addc_co_u32, REP8(X11))
#define X12(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mov_b32, REP8(X12))
#define X13(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "s"(s));
KERNEL(k_add_u32_sgpr, REP8(X13))
#define X14(i) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_add_u32_e64, REP8(X14))
#define X15(i) asm volatile("v_and_b32 %0, 0x3ffffff, %0" : "+v"(a##i));
KERNEL(k_and_literal, REP8(X15))
#define X16(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_fmac_f32, REP8(X16))
#define X17(i) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(b##i));
KERNEL(k_add_u32_dpp, REP8(X17))
#define X18(i) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_subrev_u32, REP8(X18))

// ---- three sources / multipliers (VOP3 encodings)
#define Y1(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(a##i), "v"(b##i) : "vcc");
KERNEL(k_mad_u64_u32, REP8(Y1))
#define Y2(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(a##i), "s"(s) : "vcc");
KERNEL(k_mad_i64_i32_sgpr, REP8(Y2))
#define Y3(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, 1, %0" : "+v"(q##i) : "v"(a##i) : "vcc");
KERNEL(k_mad_i64_i32_by1, REP8(Y3))
#define Y4(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mul_lo_u32, REP8(Y4))
#define Y5(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mul_hi_u32, REP8(Y5))
#define Y6(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mad_u32_u24, REP8(Y6))
#define Y7(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_add3_u32, REP8(Y7))
#define Y8(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_lshl_add_u32, REP8(Y8))
#define Y9(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_and_or_b32, REP8(Y9))
#define Y10(i) asm volatile("v_bfe_u32 %0, %0, 3, 26" : "+v"(a##i));
KERNEL(k_bfe_u32, REP8(Y10))
#define Y11(i) asm volatile("v_alignbit_b32 %0, %0, %1, 26" : "+v"(a##i) : "v"(b##i));
KERNEL(k_alignbit_b32, REP8(Y11))
#define Y12(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q##i) : "v"(r##i));
KERNEL(k_lshl_add_u64, REP8(Y12))
#define Y13(i) asm volatile("v_lshrrev_b64 %0, 26, %0" : "+v"(q##i));
KERNEL(k_lshrrev_b64, REP8(Y13))
#define Y14(i) asm volatile("v_ashrrev_i64 %0, 26, %0" : "+v"(q##i));
KERNEL(k_ashrrev_i64, REP8(Y14))
#define Y15(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_fma_f32, REP8(Y15))
#define Y16(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(q##i) : "v"(r##i));
KERNEL(k_fma_f64, REP8(Y16))
#define Y17(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(q##i) : "v"(r##i));
KERNEL(k_add_f64, REP8(Y17))
#define Y18(i) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a##i) : "v"(b##i));
KERNEL(k_mad_u32_u16, REP8(Y18))
#define Y19(i) asm volatile("v_sub_co_u32_e64 %0, s[20:21], %0, %1" : "+v"(a##i) : "v"(b##i) : "s20", "s21");
KERNEL(k_sub_co_u32_e64, REP8(Y19))
#define Y20(i) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(q##i) : "v"(a##i));
KERNEL(k_cvt_f64_u32, REP8(Y20))
#define Y21(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_perm_b32, REP8(Y21))
#define Y22(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(a##i), "s"(s) : "vcc");
KERNEL(k_mad_u64_u32_sgpr, REP8(Y22))
#define Y23(i) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_xad_u32, REP8(Y23))
#define Y24(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(q##i) : "v"(r##i));
KERNEL(k_mul_f64, REP8(Y24))

// ---- packed / dot (VOP3P)
#define Z1(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q##i) : "v"(r##i));
KERNEL(k_pk_fma_f32, REP8(Z1))
#define Z2(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_pk_add_u16, REP8(Z2))
#define Z3(i) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_pk_mul_lo_u16, REP8(Z3))
#define Z4(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(a##i) : "v"(b##i));
KERNEL(k_pk_mad_u16, REP8(Z4))
#define Z5(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a##i) : "v"(b##i));
KERNEL(k_dot4_u32_u8, REP8(Z5))
#define Z6(i) asm volatile("v_dot2_u32_u16 %0, %0, %1, %0" : "+v"(a##i) : "v"(b##i));
KERNEL(k_dot2_u32_u16, REP8(Z6))
#define Z7(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q##i) : "v"(r##i));
KERNEL(k_pk_add_f32, REP8(Z7))

// ---- mixes that matter for the limb arithmetic
// (a) a mad and a plain add interleaved: do the two classes share one issue port?
#define M1(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_add_u32 %1, %1, %2" : "+v"(q##i), "+v"(a##i) : "v"(b##i) : "vcc");
KERNEL(k_mix_mad_add, REP8(M1))
// (b) carry propagation step of the 26-bit-limb form: alignbit + and + mad(c, 1, acc)
#define M2(i) asm volatile("v_alignbit_b32 %1, %H0, %L0, 26\n v_and_b32 %L0, 0x3ffffff, %L0\n v_mad_i64_i32 %2, vcc, %1, 1, %2" : "+v"(q##i), "+v"(a##i), "+v"(r##i) :: "vcc");

typedef void (*kern_t)(uint32_t *, uint64_t *, uint32_t);
struct Entry { const char *name; kern_t k; int per_iter; };

int main() {
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    uint32_t *buf; uint64_t *cyc;
    const int max_blocks = cus * 8;
    hipMalloc(&buf, (size_t)max_blocks * 256 * 4);
    hipMalloc(&cyc, (size_t)max_blocks * 4 * 8);
    std::vector<uint64_t> h(max_blocks * 4);
    Entry es[] = {
#define E(K, N) {#K, K, N},
        E(k_add_u32, 8) E(k_sub_u32, 8) E(k_subrev_u32, 8) E(k_and_b32, 8) E(k_and_literal, 8) E(k_lshrrev_b32, 8) E(k_ashrrev_i32, 8) E(k_lshlrev_b32, 8)
        E(k_mov_b32, 8) E(k_add_u32_sgpr, 8) E(k_add_u32_e64, 8) E(k_add_u32_dpp, 8) E(k_cndmask_vcc, 8) E(k_add_co_u32, 8) E(k_addc_co_u32, 8) E(k_sub_co_u32_e64, 8)
        E(k_mul_u32_u24, 8) E(k_mul_hi_u32_u24, 8) E(k_fmac_f32, 8) E(k_fma_f32, 8)
        E(k_mad_u64_u32, 8) E(k_mad_u64_u32_sgpr, 8) E(k_mad_i64_i32_sgpr, 8) E(k_mad_i64_i32_by1, 8) E(k_mul_lo_u32, 8) E(k_mul_hi_u32, 8) E(k_mad_u32_u24, 8) E(k_mad_u32_u16, 8)
        E(k_add3_u32, 8) E(k_lshl_add_u32, 8) E(k_and_or_b32, 8) E(k_xad_u32, 8) E(k_bfe_u32, 8) E(k_alignbit_b32, 8) E(k_perm_b32, 8)
        E(k_lshl_add_u64, 8) E(k_lshrrev_b64, 8) E(k_ashrrev_i64, 8)
        E(k_fma_f64, 8) E(k_add_f64, 8) E(k_mul_f64, 8) E(k_cvt_f64_u32, 8)
        E(k_pk_fma_f32, 8) E(k_pk_add_f32, 8) E(k_pk_add_u16, 8) E(k_pk_mul_lo_u16, 8) E(k_pk_mad_u16, 8) E(k_dot4_u32_u8, 8) E(k_dot2_u32_u16, 8)
        E(k_mix_mad_add, 16)
    };
    printf("%-22s %9s %9s %9s %9s   ns per wave-instruction per SIMD (wall clock, every CU holding exactly k 256-thread blocks = k waves per SIMD, forced by LDS size); x2.4 = cycles at 2.4 GHz\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    hipEvent_t ev0, ev1;
    hipEventCreate(&ev0); hipEventCreate(&ev1);
    for (auto &e : es) {
        printf("%-22s", e.name);
        hipFuncSetAttribute(reinterpret_cast<const void *>(e.k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * 8;   // 256-thread blocks: 4 waves, one per SIMD; LDS limits residency to wps blocks per CU
            const size_t lds = (size_t)(160 * 1024) / wps;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, buf, cyc, 3u);
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                hipEventRecord(ev0);
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, buf, cyc, 3u);
                hipEventRecord(ev1);
                hipEventSynchronize(ev1);
                float ms; hipEventElapsedTime(&ms, ev0, ev1);
                if (ms < best) best = ms;
            }
            // each SIMD executes (blocks / cus) waves in total, ITERS * per_iter instructions each
            printf(" %9.3f", best * 1e6 / ((double)(blocks / cus) * ITERS * e.per_iter));
        }
        hipMemcpy(h.data(), cyc, (size_t)cus * 8 * 4 * 8, hipMemcpyDeviceToHost);
        std::vector<uint64_t> v(h.begin(), h.begin() + cus * 8 * 4);
        std::sort(v.begin(), v.end());
        printf("   memtime ticks/instr/wave at 8 w/SIMD: %.2f\n", (double)v[v.size() / 2] / ((double)ITERS * e.per_iter));
    }
    return 0;
}
