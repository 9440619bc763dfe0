// trace_bench.cpp — host-only latency experiments for the serial MiMC recurrence x <- x^3 + k (gs_mimc_trace).
// g++ -O3 -march=native tools/trace_bench.cpp -o tools/trace_bench && tools/trace_bench
#include <chrono>
#include <cstdio>
#include <vector>

#include "../genstark_amd/csrc/host_field.h"

// ---- experiment: x^3 straight from the limbs' cubes (no x0*x1 product, no normalised square), columns split once (no carry chain),
// every high column folded by a constant of its own weight (one level of products), then the usual small second fold
static inline hfe hf_cube_direct(hfe x) {
    typedef uint64_t u64;
    const u64 C = (u64)HF_C, C20 = 0xFFFFFFEE00000001ull, C21 = 80, K5 = 80ull * 0x8FFFFFFFFull;   // 2^256 == C21*2^64 + C20, 2^320 == C20*2^64 + K5
    const u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    const hfe p00 = (hfe)x0 * x0, p11 = (hfe)x1 * x1;
    const u64 l0 = (u64)p00, h0 = (u64)(p00 >> 64), l1 = (u64)p11, h1 = (u64)(p11 >> 64);
    const hfe aL = (hfe)l0 * x0, aH = (hfe)h0 * x0, bL = (hfe)l0 * x1, bH = (hfe)h0 * x1;
    const hfe cL = (hfe)l1 * x0, cH = (hfe)h1 * x0, dL = (hfe)l1 * x1, dH = (hfe)h1 * x1;
#define LO(v) ((hfe)(u64)(v))
#define HI(v) ((hfe)(u64)((v) >> 64))
    const hfe col1 = HI(aL) + LO(aH) + 3 * LO(bL);
    const hfe col2 = HI(aH) + 3 * (HI(bL) + LO(bH) + LO(cL));
    const hfe col3 = 3 * (HI(bH) + HI(cL) + LO(cH)) + LO(dL);
    const hfe col4 = 3 * HI(cH) + HI(dL) + LO(dH);
    const u64 e0 = (u64)aL, e1 = (u64)col1;
    u64 e2, e3, e4, e5;
    const bool o2 = __builtin_add_overflow((u64)col2, (u64)(col1 >> 64), &e2);
    const bool o3 = __builtin_add_overflow((u64)col3, (u64)(col2 >> 64), &e3);
    const bool o4 = __builtin_add_overflow((u64)col4, (u64)(col3 >> 64), &e4);
    const bool o5 = __builtin_add_overflow((u64)(dH >> 64), (u64)(col4 >> 64), &e5);
    if (__builtin_expect(o2 | o3 | o4 | o5, 0)) return hf_cube_weak(x);     // a column's low word within 12 of 2^64: probability ~2^-58
    // weight 1: e0 + e2*C + e4*C20 + e5*K5      weight 2^64: e1 + e3*C + e4*C21 + e5*C20
    const hfe m2 = (hfe)e2 * C, m3 = (hfe)e3 * C, m40 = (hfe)e4 * C20, m41 = (hfe)e4 * C21, m50 = (hfe)e5 * K5, m51 = (hfe)e5 * C20;
    // limb sums (each a handful of 64-bit terms: no overflow of the 128-bit accumulators)
    const hfe s0 = (hfe)e0 + LO(m2) + LO(m40) + LO(m50);
    const hfe s1 = (hfe)e1 + HI(m2) + HI(m40) + HI(m50) + LO(m3) + LO(m41) + LO(m51) + HI(s0);
    const hfe T = HI(m3) + HI(m41) + HI(m51) + HI(s1);                      // weight 2^128: < 2^67
    const hfe R = ((hfe)(u64)s1 << 64) | (u64)s0;
    const hfe TC = (hfe)(u64)T * C + (((hfe)(u64)(T >> 64) * C) << 64);
    hfe r = R + TC;
    if (__builtin_expect(r < R, 0)) r += HF_C;
    return r;
#undef LO
#undef HI
}

template <int V>
static double run(uint64_t steps, const std::vector<hfe> &rc, hfe seed, std::vector<hfe> &t) {
    auto t0 = std::chrono::steady_clock::now();
    hfe x = seed;
    uint32_t ri = 0, nrc = (uint32_t)rc.size();
    for (uint64_t i = 0; i < steps; i++) {
        hfe y = (V & 4) ? hf_cube_direct(x) : (V & 2) ? hf_cube_weak(x) : hf_mul_weak(hf_mul_weak(x, x), x);
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
        double e = run<4>(steps, rc, seed, t1), f = run<5>(steps, rc, seed, t3);
        for (uint64_t i = 0; i < steps; i++) ok &= t0[i] == t1[i] && t0[i] == t3[i];
        printf("2^20 steps: A/canon-chain %.2f ms | A/weak-chain %.2f ms | cube/canon-chain %.2f ms | cube/weak-chain %.2f ms | direct/canon %.2f ms | direct/weak %.2f ms  (%s)\n", a, b, c, d, e, f,
               ok ? "all equal" : "DIFF");
    }
    return 0;
}
