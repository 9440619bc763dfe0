// trace_bench.cpp — host-only latency experiments for the serial MiMC recurrence x <- x^3 + k (gs_mimc_trace).
// g++ -O3 -march=native tools/trace_bench.cpp -o tools/trace_bench && tools/trace_bench
#include <chrono>
#include <cstdio>
#include <vector>

#include "../genstark_amd/csrc/host_field.h"

template <int V>
static double run(uint64_t steps, const std::vector<hfe> &rc, hfe seed, std::vector<hfe> &t) {
    auto t0 = std::chrono::steady_clock::now();
    hfe x = seed;
    uint32_t ri = 0, nrc = (uint32_t)rc.size();
    for (uint64_t i = 0; i < steps; i++) {
        hfe y = (V & 2) ? hf_cube_weak(x) : hf_mul_weak(hf_mul_weak(x, x), x);
        hfe sum = y + rc[ri];
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
    std::vector<hfe> rc(64), t0(steps), t1(steps), t2(steps), t3(steps);
    for (int i = 0; i < 64; i++) rc[i] = hf_pow(3, 1000 + i);
    hfe seed = 3;
    for (int rep = 0; rep < 4; rep++) {
        double a = run<0>(steps, rc, seed, t0), b = run<1>(steps, rc, seed, t1), c = run<2>(steps, rc, seed, t2), d = run<3>(steps, rc, seed, t3);
        bool ok = true;
        for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t1[i] && t0[i] == t2[i] && t0[i] == t3[i];
        printf("2^20 steps: A/canon-chain %.2f ms | A/weak-chain %.2f ms | cube/canon-chain %.2f ms | cube/weak-chain %.2f ms  (%s)\n", a, b, c, d,
               ok ? "all equal" : "DIFF");
    }
    return 0;
}
