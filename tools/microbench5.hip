// tools/microbench5.hip — the arithmetic cores of k_ntt_pass_lz in isolation (registers only, no memory traffic): how far is the
// pass kernel from what its own instruction stream can do at 1, 2, 3, 4 waves per SIMD?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 tools/microbench5.hip -o tools/microbench5
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../genstark_amd/csrc/gf128_lazy.h"
#include "gf128_f64.h"

__device__ __forceinline__ lzw load_w(const lzw *p) {
    lzw W;
    const __attribute__((address_space(4))) int32_t *q = (const __attribute__((address_space(4))) int32_t *)(const int32_t *)p;
#pragma unroll
    for (int g = 0; g < 25; g += 8) {
#pragma unroll
        for (int e = g; e < g + 8 && e < 25; e++) W.w[e / 5][e % 5] = q[e];
        asm volatile("" ::: "memory");
    }
    return W;
}
template <int S>
__device__ __forceinline__ void dif_level(lz (&x)[16], const lzw *w, const lzk &K) {
    constexpr int unit = 8 / S;
#pragma unroll
    for (int b = 0; b < 16; b += 2 * S)
#pragma unroll
        for (int i = 0; i < S; i++) {
            const lz u = x[b + i], v = x[b + i + S];
            x[b + i] = lz_add(u, v);
            const lz d = lz_sub(u, v);
            const int tw = i * unit;
            if (tw) { asm volatile("" ::: "memory"); const lzw W = load_w(w + tw - 1); x[b + i + S] = lz_mul_u(d, W, K); } else x[b + i + S] = d;
        }
    if constexpr (S > 1) dif_level<S / 2>(x, w, K);
}
#define ITERS 40
// (a) radix-16 DIF with tabulated multipliers, renormalised every round
__global__ __launch_bounds__(128) void k_dif(const fe *in, fe *out, const lzw *w) {
    const lzk K = lzk_make();
    lz v[16];
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = lz_unpack(in[threadIdx.x + 128 * m]);
    for (int it = 0; it < ITERS; it++) {
        dif_level<8>(v, w, K);
#pragma unroll
        for (int m = 0; m < 16; m++) v[m] = lz_norm(v[m]);
    }
#pragma unroll
    for (int m = 0; m < 16; m++) out[blockIdx.x * 2048 + threadIdx.x + 128 * m] = lz_pack(v[m]);
}
// (b) 16 per-lane products per round
__global__ __launch_bounds__(128) void k_mulv(const fe *in, fe *out, const lzw *w) {
    const lzk K = lzk_make();
    lz v[16], t[16];
#pragma unroll
    for (int m = 0; m < 16; m++) { v[m] = lz_unpack(in[threadIdx.x + 128 * m]); t[m] = lz_unpack(in[threadIdx.x + 128 * m + 64]); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) v[m] = lz_mul_v(v[m], t[m], K);
    }
#pragma unroll
    for (int m = 0; m < 16; m++) out[blockIdx.x * 2048 + threadIdx.x + 128 * m] = lz_pack(v[m]);
}
// (b') the same through the Montgomery form (REDC from the bottom: lz_mul_vm; multipliers premultiplied by 2^130)
__global__ __launch_bounds__(128) void k_mulvm(const fe *in, fe *out, const lzw *w) {
    const lzk K = lzk_make();
    lz v[16], t[16];
#pragma unroll
    for (int m = 0; m < 16; m++) { v[m] = lz_unpack(in[threadIdx.x + 128 * m]); t[m] = lz_unpack(in[threadIdx.x + 128 * m + 64]); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) v[m] = lz_mul_vm(v[m], t[m], K);
    }
#pragma unroll
    for (int m = 0; m < 16; m++) out[blockIdx.x * 2048 + threadIdx.x + 128 * m] = lz_pack(v[m]);
}
// (e) ONE dependent chain of squarings per lane (Rescue's inverse S-box: 127 squarings + 32 products that wait for each other): what a
// trace kernel with a single wave per SIMD pays per squaring — the floor of the compiled Rescue trace kernel
__global__ __launch_bounds__(128) void k_sqrchain(const fe *in, fe *out, const lzw *w) {
    const lzk K = lzk_make();
    lz v = lz_unpack(in[threadIdx.x]);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) v = lz_sqr(v, K);
    }
    out[blockIdx.x * 2048 + threadIdx.x] = lz_pack(v);
}
// (c) 16 packs + unpacks per round
__global__ __launch_bounds__(128) void k_pack(const fe *in, fe *out, const lzw *w) {
    lz v[16];
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = lz_unpack(in[threadIdx.x + 128 * m]);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) { v[m] = lz_add(v[m], v[(m + 1) & 15]); v[m] = lz_unpack(lz_pack(v[m])); }
    }
#pragma unroll
    for (int m = 0; m < 16; m++) out[blockIdx.x * 2048 + threadIdx.x + 128 * m] = lz_pack(v[m]);
}
// (d) canonical-limb products (the arithmetic of k_ntt_pass) for reference
__global__ __launch_bounds__(128) void k_femul(const fe *in, fe *out, const lzw *w) {
    fe v[16], t[16];
#pragma unroll
    for (int m = 0; m < 16; m++) { v[m] = in[threadIdx.x + 128 * m]; t[m] = in[threadIdx.x + 128 * m + 64]; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) v[m] = fe_mul(v[m], t[m]);
    }
#pragma unroll
    for (int m = 0; m < 16; m++) out[blockIdx.x * 2048 + threadIdx.x + 128 * m] = v[m];
}

// (f) EXPERIMENT: the per-lane product on the fp64 FMA pipe — three limbs of 43 bits, partial products split exactly
// by two FMAs and a subtraction (tools/gf128_f64.h; 73 fp64 operations per product, 3 x 3 = 6 VGPRs per element instead of 5)
__global__ __launch_bounds__(128) void k_mulf64(const fe *in, fe *out, const lzw *w) {
    fz v[16], t[16];
#pragma unroll
    for (int m = 0; m < 16; m++) { v[m] = fz_unpack(in[threadIdx.x + 128 * m]); t[m] = fz_unpack(in[threadIdx.x + 128 * m + 64]); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) v[m] = fz_mul(v[m], t[m]);
    }
#pragma unroll
    for (int m = 0; m < 16; m++) {          // (the limbs as they are: a timing experiment; values are checked on the host, tests/test_lazy_field.py)
        fe o;
        o.w0 = (uint32_t)(int64_t)v[m].l[0]; o.w1 = (uint32_t)(int64_t)v[m].l[1]; o.w2 = (uint32_t)(int64_t)v[m].l[2]; o.w3 = 0;
        out[blockIdx.x * 2048 + threadIdx.x + 128 * m] = o;
    }
}

// (g) the two candidates side by side at the pass kernels' REAL occupancy: eight products per round and a register budget of 128
// (four waves per SIMD); the sixteen-product kernels above need ~210 VGPRs and hold two
__global__ __launch_bounds__(128, 4) void k_mulvm8(const fe *in, fe *out, const lzw *w) {
    const lzk K = lzk_make();
    lz v[8], t[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { v[m] = lz_unpack(in[threadIdx.x + 128 * m]); t[m] = lz_unpack(in[threadIdx.x + 128 * m + 64]); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) v[m] = lz_mul_vm(v[m], t[m], K);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) out[blockIdx.x * 2048 + threadIdx.x + 128 * m] = lz_pack(v[m]);
}
// (h) what a WAVE-UNIFORM exchange twiddle would cost: the same eight products by multipliers in W-form out of
// scalar registers (lz_mul_u: five columns, no REDC) — the difference to k_mulvm8 is what a pass could save per element IF its exchange
// twiddles omega_R^(kk * qa) were the same for all lanes of a wave, i.e. if the lanes of a wave were 64 different sub-transforms
__global__ __launch_bounds__(128, 4) void k_mulu8(const fe *in, fe *out, const lzw *w) {
    const lzk K = lzk_make();
    lz v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = lz_unpack(in[threadIdx.x + 128 * m]);
    // best case for the uniform form: ONE multiplier, its 25 words resident in scalar registers for the whole loop (a real exchange
    // needs sixteen of them per pass: loads pipelined between the products as in the radix-16 network, ~5 % on top)
    const lzw W = load_w(w);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) v[m] = lz_mul_u(v[m], W, K);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) out[blockIdx.x * 2048 + threadIdx.x + 128 * m] = lz_pack(v[m]);
}
__global__ __launch_bounds__(128, 4) void k_mulf64_8(const fe *in, fe *out, const lzw *w) {
    fz v[8], t[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { v[m] = fz_unpack(in[threadIdx.x + 128 * m]); t[m] = fz_unpack(in[threadIdx.x + 128 * m + 64]); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) v[m] = fz_mul(v[m], t[m]);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
        fe o;
        o.w0 = (uint32_t)(int64_t)v[m].l[0]; o.w1 = (uint32_t)(int64_t)v[m].l[1]; o.w2 = (uint32_t)(int64_t)v[m].l[2]; o.w3 = 0;
        out[blockIdx.x * 2048 + threadIdx.x + 128 * m] = o;
    }
}

typedef void (*kern_t)(const fe *, fe *, const lzw *);
// `--json`: one JSON object on stdout (bench.py: the second roof of the NTT pass kernel, measured in the same run)
int main(int argc, char **argv) {
    const bool json = argc > 1 && !strcmp(argv[1], "--json");
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    fe *in, *out; lzw *w;
    const int blocks = cus * 16;
    hipMalloc(&in, 4096 * 16); hipMalloc(&out, (size_t)blocks * 2048 * 16); hipMalloc(&w, 8 * sizeof(lzw));
    hipMemset(in, 0x5a, 4096 * 16); hipMemset(w, 0x01, 8 * sizeof(lzw));
    struct { const char *name; kern_t k; double units; const char *unit; } es[] = {
        {"radix-16 DIF (17 mul_u + 64 add/sub) + 16 norm", k_dif, 1, "per network"},
        {"lz_mul_v", k_mulv, 16, "per product"},
        {"lz_pack + lz_unpack + add", k_pack, 16, "per element"},
        {"fe_mul (canonical limbs)", k_femul, 16, "per product"},
        {"lz_mul_vm (Montgomery REDC, radix 2^26)", k_mulvm, 16, "per product"},
        {"lz_sqr, ONE dependent chain per lane", k_sqrchain, 16, "per squaring"},
        {"fz_mul (fp64 FMA, radix 2^43: experiment)", k_mulf64, 16, "per product"},
        {"lz_mul_vm, 8 per round, <= 128 VGPRs", k_mulvm8, 8, "per product"},
        {"fz_mul, 8 per round, <= 128 VGPRs", k_mulf64_8, 8, "per product"},
        {"lz_mul_u (wave-uniform W-form), 8 per round", k_mulu8, 8, "per product"},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (!json) printf("%-50s %10s %10s %10s %10s   ns per wave per SIMD (wall clock; k waves per SIMD forced by LDS size, 128-thread blocks)\n", "core", "1 w/SIMD", "2 w/SIMD", "3 w/SIMD", "4 w/SIMD");
    double res[12][4];
    int ei = 0;
    for (auto &e : es) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(e.k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (!json) printf("%-50s", e.name);
        for (int wps : {1, 2, 3, 4}) {
            const size_t lds = (size_t)(160 * 1024) / (2 * wps);   // 2-wave blocks: 2*wps blocks per CU
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(128), lds, 0, in, out, w);
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                hipEventRecord(e0); hipLaunchKernelGGL(e.k, dim3(blocks), dim3(128), lds, 0, in, out, w); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            // waves per SIMD in total = blocks * 2 / (cus * 4)
            const double waves_per_simd = (double)blocks * 2 / (cus * 4);
            res[ei][wps - 1] = best * 1e6 / (waves_per_simd * ITERS * e.units);
            if (!json) printf(" %10.1f", res[ei][wps - 1]);
        }
        if (!json) printf("  %s\n", e.unit);
        ei++;
    }
    if (json) {
        const char *keys[10] = {"dif16_network_ns", "mul_v_ns", "pack_unpack_add_ns", "fe_mul_ns", "mul_vm_ns", "sqr_chain_ns", "mul_f64_ns", "mul_vm_128vgpr_ns", "mul_f64_128vgpr_ns", "mul_u_128vgpr_ns"};
        printf("{\"cus\": %d, \"unit\": \"ns per wave per SIMD at 1,2,3,4 waves per SIMD\"", cus);
        for (int k = 0; k < 10; k++) printf(", \"%s\": [%.2f, %.2f, %.2f, %.2f]", keys[k], res[k][0], res[k][1], res[k][2], res[k][3]);
        printf("}\n");
    }
    return 0;
}
