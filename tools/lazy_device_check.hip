// tools/lazy_device_check.hip — the lazy five-limb arithmetic of gf128_lazy.h executed ON THE GPU, every result compared with the
// canonical fe_mul / fe_add / fe_sub of gf128.h on the same lanes (tests/test_gpu_parity.py runs it; exit code 0 = all equal).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 tools/lazy_device_check.hip -o tools/lazy_device_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../genstark_amd/csrc/gf128_lazy.h"

__device__ __forceinline__ lzw load_w(const lzw *p) {
    lzw W;
    const __attribute__((address_space(4))) int32_t *q = (const __attribute__((address_space(4))) int32_t *)(const int32_t *)p;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) W.w[i][j] = q[i * 5 + j];
    return W;
}

// err[0] counts mismatches; every thread takes a (a, b) pair and a table multiplier
__global__ void k_check(const fe *__restrict__ a, const fe *__restrict__ b, const lzw *__restrict__ wtab, const fe *__restrict__ wcan, int nw,
                        uint64_t n, unsigned *err) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe x = a[i], y = b[i];
    const lz lx = lz_unpack(x), ly = lz_unpack(y);
    const lzk K = lzk_make();
    unsigned bad = 0, bit = 1;
#define CHK(cond) do { if (!(cond)) bad |= bit; bit <<= 1; } while (0)
    // 1. per-lane product
    CHK(fe_eq(lz_pack(lz_mul_v(lx, ly, K)), fe_mul(x, y)));
    // 2. tabulated multiplier (scalar loads)
    const int t = (int)(blockIdx.x % nw);
    asm volatile("" ::: "memory");
    const lzw W = load_w(wtab + t);
    CHK(fe_eq(lz_pack(lz_mul_u(lx, W, K)), fe_mul(x, wcan[t])));
    // 3. W-form built in registers from a per-lane value, running product of 5 steps
    lzw S;
    lz row = ly;
#pragma unroll
    for (int r = 0; r < 5; r++) {
#pragma unroll
        for (int c = 0; c < 5; c++) { S.w[r][c] = row.l[c]; asm volatile("" : "+v"(S.w[r][c])); }   // as ntt.hip does (see there)
        if (r < 4) row = lz_shift_limb(row, K);
    }
    lz cur = lx;
    fe ref = x;
#pragma unroll
    for (int s = 0; s < 5; s++) { cur = lz_mul_u(cur, S, K); ref = fe_mul(ref, y); }
    CHK(fe_eq(lz_pack(cur), ref));
    // 4. a small butterfly network: sums and differences of 16 values, then norm + product + pack
    lz acc = lx; fe racc = x;
    lz dif = lx; fe rdif = x;
#pragma unroll
    for (int s = 0; s < 15; s++) { acc = lz_add(acc, (s & 1) ? lx : ly); racc = fe_add(racc, (s & 1) ? x : y); }
#pragma unroll
    for (int s = 0; s < 3; s++) { dif = lz_sub(dif, (s & 1) ? lx : ly); rdif = fe_sub(rdif, (s & 1) ? x : y); }
    CHK(fe_eq(lz_pack(acc), racc));
    CHK(fe_eq(lz_pack(lz_mul_v(lz_norm(acc), ly, K)), fe_mul(racc, y)));
    CHK(fe_eq(lz_pack(lz_mul_v(dif, ly, K)), fe_mul(rdif, y)));
    CHK(fe_eq(lz_pack(lz_mul_u(dif, W, K)), fe_mul(rdif, wcan[t])));
    // 8. squaring, and an exponentiation chain as compiled AIR programs run it (5 squarings + a product, nothing packed in between)
    {
        CHK(fe_eq(lz_pack(lz_sqr(lx, K)), fe_mul(x, x)));
        lz c5 = lx; fe r5 = x;
#pragma unroll 1
        for (int q = 0; q < 5; q++) { c5 = lz_sqr(c5, K); r5 = fe_mul(r5, r5); }
        CHK(fe_eq(lz_pack(lz_mul_v(c5, ly, K)), fe_mul(r5, y)));
    }
    if (bad) atomicAdd(err, 1u);
    for (int k = 0; k < 9; k++) if (bad & (1u << k)) atomicAdd(err + 1 + k, 1u);
}

static uint64_t sm(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main() {
    const uint64_t n = 1 << 20;
    const int nw = 64;
    std::vector<fe> a(n), b(n), wc(nw);
    std::vector<lzw> wt(nw);
    uint64_t s = 12345;
    auto rnd = [&](uint64_t i) {
        fe r;
        if (i % 7 == 0) { const fe edge[6] = {fe_zero(), fe_one(), fe_make(0, GF_P1, GF_P2, GF_P3), fe_make(0xFFFFFFFFu, 8, 0, 0), fe_make(0xFFFFFFFFu, 0xFFFFFFF6u, GF_P2, GF_P3), fe_make(0, 0, 0, 0x80000000u)}; return edge[(i / 7) % 6]; }
        do { uint64_t lo = sm(s), hi = sm(s); r = fe_make((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)); } while (fe_ge_p(r));
        return r;
    };
    for (uint64_t i = 0; i < n; i++) { a[i] = rnd(i); b[i] = rnd(i + 3); }
    for (int i = 0; i < nw; i++) { wc[i] = rnd(i * 5 + 1); lz_wform(wc[i], wt[i]); }
    fe *da, *db, *dwc; lzw *dwt; unsigned *derr, herr[12] = {0};
    hipMalloc(&da, n * 16); hipMalloc(&db, n * 16); hipMalloc(&dwc, nw * 16); hipMalloc(&dwt, nw * sizeof(lzw)); hipMalloc(&derr, 48);
    hipMemcpy(da, a.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 16, hipMemcpyHostToDevice);
    hipMemcpy(dwc, wc.data(), nw * 16, hipMemcpyHostToDevice); hipMemcpy(dwt, wt.data(), nw * sizeof(lzw), hipMemcpyHostToDevice);
    hipMemset(derr, 0, 48);
    hipLaunchKernelGGL(k_check, dim3((unsigned)(n / 256)), dim3(256), 0, 0, da, db, dwt, dwc, nw, n, derr);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(herr, derr, 48, hipMemcpyDeviceToHost);
    printf("lazy_device_check: %llu lanes, %u mismatching (per test: mul_v %u, mul_u table %u, running W-form %u, sum16 %u, norm+mul_v %u, diff mul_v %u, diff mul_u %u, sqr %u, sqr chain %u)\n",
           (unsigned long long)n, herr[0], herr[1], herr[2], herr[3], herr[4], herr[5], herr[6], herr[7], herr[8], herr[9]);
    return herr[0] ? 1 : 0;
}
