// trace_bench.cpp — host-only latency experiments for the serial MiMC recurrence x <- x^3 + k (gs_mimc_trace).
// g++ -O3 -march=native tools/trace_bench.cpp -o tools/trace_bench && tools/trace_bench
#include <chrono>
#include <cstdio>
#include <vector>

#include "../genstark_amd/csrc/host_field.h"

// variant: (m1 >> 64) * C leaves the critical path; only the 0..2 carries are folded late (64-bit product)
static inline hu128 mul_weak_b(hu128 a, hu128 b) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    hu128 p00 = (hu128)a0 * b0, p01 = (hu128)a0 * b1, p10 = (hu128)a1 * b0, p11 = (hu128)a1 * b1;
    hu128 mid = p01 + p10;
    uint64_t midc = mid < p01;
    hu128 lo = p00 + (mid << 64);
    uint64_t c1 = lo < p00;
    hu128 hi = p11 + (mid >> 64) + ((hu128)midc << 64) + c1;
    const uint64_t cc = (uint64_t)HF_C;
    hu128 m0 = (hu128)(uint64_t)hi * cc, m1 = (hu128)(uint64_t)(hi >> 64) * cc;
    hu128 top = (hu128)(uint64_t)(m1 >> 64) * cc;     // < 2^72
    hu128 t = m0 + (m1 << 64);
    uint64_t k = t < m0;
    hu128 u = lo + top;                               // independent of t
    k += u < lo;
    hu128 s = u + t;
    k += s < u;
    hu128 r = s + (hu128)(k * cc);
    if (__builtin_expect(r < s, 0)) r += HF_C;
    return r;
}

template <int V>
static double run(uint64_t steps, const std::vector<hu128> &rc, hu128 seed, std::vector<hu128> &t) {
    auto t0 = std::chrono::steady_clock::now();
    hu128 x = seed;
    uint32_t ri = 0, nrc = (uint32_t)rc.size();
    for (uint64_t i = 0; i < steps; i++) {
        hu128 y = (V & 2) ? mul_weak_b(mul_weak_b(x, x), x) : hf_mul_weak(hf_mul_weak(x, x), x);
        hu128 sum = y + rc[ri];
        if (V & 1) {          // weak chain, canonicalisation off the critical path
            t[i] = hf_canon(x);
            if (__builtin_expect(sum < y, 0)) sum += HF_C;
            x = sum;
        } else {              // canonical chain
            t[i] = x;
            if (sum < y) sum += HF_C;
            x = hf_canon(sum);
        }
        if (++ri == nrc) ri = 0;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count();
}

int main() {
    const uint64_t steps = 1 << 20;
    std::vector<hu128> rc(64), t0(steps), t1(steps), t2(steps), t3(steps);
    for (int i = 0; i < 64; i++) rc[i] = hf_pow(3, 1000 + i);
    hu128 seed = 3;
    for (int rep = 0; rep < 4; rep++) {
        double a = run<0>(steps, rc, seed, t0), b = run<1>(steps, rc, seed, t1), c = run<2>(steps, rc, seed, t2), d = run<3>(steps, rc, seed, t3);
        bool ok = true;
        for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t1[i] && t0[i] == t2[i] && t0[i] == t3[i];
        printf("2^20 steps: A/canon-chain %.2f ms | A/weak-chain %.2f ms | B/canon-chain %.2f ms | B/weak-chain %.2f ms  (%s)\n", a, b, c, d,
               ok ? "all equal" : "DIFF");
    }
    return 0;
}
