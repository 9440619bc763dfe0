// air_jit.hip — AIR programs compiled instead of interpreted.  Default (auto): a program runs compiled whenever its code object already
// exists (this process or the on-disk cache) and is built in the background otherwise; gs_air_jit(ctx, 1) / GSTARK_AIR_JIT=1 compile on
// first use, gs_air_jit(ctx, 0) / GSTARK_AIR_JIT=0 always interpret.
//
// The reference's air-assembly GENERATES code for an AIR's transition function and constraint evaluator when a module is
// instantiated ("generated JS over BigInt").  The register machine of air_vm.hip is the portable form of the same
// programs; here they are turned into HIP source — static-register offsets and periods become literals, exponents become fixed
// addition chains — compiled once per program with hiprtc for gfx950 and cached for the process.
//   constraint programs (one thread per domain point, throughput-bound): VM registers become variables, straight-line code;
//   trace programs (a few thousand independent segments at most: one wave per SIMD, every dependent product paid at full
//   latency): SSA form, products scheduled by depth and spread over the 2-16 lanes that share a segment, S-box layers one member
//   per lane, results exchanged through LDS (see "trace programs" below).
// Same arithmetic (the field header the library itself is built from is embedded in the source), same values; any failure to
// compile falls back to the interpreter.  gs_air_jit_check generates + compiles without a device (CPU test tier).
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <thread>

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>

#include "common.h"
#include "host_sha256.h"

enum { J_LOADC = 0, J_LOADR = 1, J_LOADN = 2, J_LOADS = 3, J_ADDV = 4, J_SUBV = 5, J_MULV = 6, J_POW = 7, J_POWC = 8, J_OUT = 9 };

static const char *kFieldHeader =
#if defined(GS_SMALL_Q)
#include "jit_gf_small.inc"
#elif defined(GS_WIDE_BITS)
#include "jit_gf_wide.inc"
#else
#include "jit_gf128.inc"
#endif
    ;
#if !defined(GS_SMALL_Q) && !defined(GS_WIDE_BITS)
#define GS_JIT_LAZY 1          // the 128-bit field: long exponentiations run in the five-limb lazy form (gf128_lazy.h: lz_sqr, lz_mul_v)
static const char *kLazyHeader =
#include "jit_gf128_lazy.inc"
    ;
#endif

#define GS_STR2(x) #x
#define GS_STR(x) GS_STR2(x)

struct JitKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    bool failed = false;
    bool compiling = false;          // a background thread is building the code object (auto mode)
    std::vector<char> code;          // built (or read from the disk cache) but not loaded yet: the next launch loads it
};
// Process-wide state of the compiled programs.  Heap-allocated and never destroyed: a builder thread may still be running when the
// process reaches exit(), and function-/namespace-scope statics would be destroyed under it.  Background builds run in a helper
// process (jit_compile_helper); at exit / dlclose the threads waiting for one are released (their sockets shut down — the helper
// still finishes and fills the disk cache) and joined; a build running inside this process (no helper installed) is waited for.
static std::mutex &g_jit_mutex = *new std::mutex;
static std::map<std::string, JitKernel> &g_jit_cache = *new std::map<std::string, JitKernel>;   // per process: the lanes of a pool share the compiled programs
static std::vector<std::thread> &g_jit_builders = *new std::vector<std::thread>;                // guarded by g_jit_mutex
static std::vector<int> &g_jit_helper_fds = *new std::vector<int>;                              // sockets of the helpers being waited for; guarded by g_jit_mutex
static std::atomic<bool> g_jit_shutdown{false};
static std::atomic<int> g_jit_running{0};                   // builds inside this process that have not finished yet
static void jit_join_builders() {
    g_jit_shutdown.store(true);
    std::vector<std::thread> mine;
    {
        std::lock_guard<std::mutex> g(g_jit_mutex);
        mine.swap(g_jit_builders);
        for (int fd : g_jit_helper_fds) shutdown(fd, SHUT_RDWR);
    }
    // an in-process build takes a second to a minute and a half (the largest programs of the examples); its result still reaches the
    // disk cache.  Bounded: should a builder ever be stuck (a compiler that died under it), the process still ends — the thread is let go
    for (int waited = 0; g_jit_running.load() > 0 && waited < 200 * 100; waited++) usleep(10000);
    for (auto &t : mine) {
        if (!t.joinable()) continue;
        if (g_jit_running.load() > 0) t.detach(); else t.join();
    }
}
__attribute__((destructor)) static void jit_library_unload() { jit_join_builders(); }          // dlclose, or exit before the dependencies' finalisers

static std::string jit_preamble() {
    std::string s;
#if defined(GS_SMALL_Q)
    s += "#define GS_SMALL_Q " GS_STR(GS_SMALL_Q) "\n";
#elif defined(GS_WIDE_BITS)
    s += "#define GS_WIDE_BITS " GS_STR(GS_WIDE_BITS) "\n";
#endif
#ifdef GS_JIT_LAZY
    s += "#include \"gf128_lazy.h\"\n";        // brings gf128.h with it
#else
    s += "#include \"gs_field.h\"\n";
#endif
    // the straight-line products: a call in the multi-limb fields (a product is ~400 instructions there; inlining a hundred of
    // them costs minutes of compilation and buys nothing), inlined in the others
#if defined(GS_WIDE_BITS)
    s += "static __device__ __noinline__ fe gs_mul(const fe &a, const fe &b) { return fe_mul(a, b); }\n";
#else
    s += "#define gs_mul fe_mul\n";
#endif
#if defined(GS_WIDE_BITS)
    s += "static __device__ __noinline__ fe gs_sqr(const fe &a) { return fe_mul(a, a); }\n";
#else
    s += "#define gs_sqr fe_sqr\n";
#endif
    // c ? a : b limb by limb: written as a conditional copy of the struct, the compiler parks the candidates in scratch memory and
    // loads through a selected pointer (seen in the ISA: 64 scratch accesses per Poseidon step)
    s += "__device__ __forceinline__ fe gs_pick(bool c, fe a, const fe b) {\n"
         "    unsigned int *pa = reinterpret_cast<unsigned int *>(&a);\n"
         "    const unsigned int *pb = reinterpret_cast<const unsigned int *>(&b);\n"
         "#pragma unroll\n"
         "    for (int i = 0; i < (int)(sizeof(fe) / 4); i++) pa[i] = c ? pa[i] : pb[i];\n"
         "    return a;\n"
         "}\n";
#ifdef GS_JIT_LAZY
    // f * x for a selector f (a static register whose table is all 0 / 1 when the source is generated): a select; any other value of
    // f — the table is data, it may change without the program changing — takes the product, so the result never depends on the guess
    s += "__device__ __forceinline__ fe gs_blend(const fe f, const fe x) {\n"
         "    if (__builtin_expect((f.w1 | f.w2 | f.w3) != 0u || f.w0 > 1u, 0)) return gs_mul(f, x);\n"
         "    return gs_pick(f.w0 == 1u, x, fe_zero());\n"
         "}\n";
#endif
    // sum_k x_k * c_k on ONE lane (a fused row of a linear layer, ssa_fuse_dots)
#ifdef GS_JIT_LAZY
    // 128-bit field: the constants in W-form (five pre-shifted copies: no high columns), every term 25 v_mad into five shared 64-bit
    // columns, ONE fold for up to six terms (columns stay below 2^57: 6 * 5 * 2^52).  The rows of a W-form built in registers are
    // laundered through an empty asm: with their value ranges visible hipcc (ROCm 7.2) miscompiles the products (ntt.hip has the same note)
    s += "typedef lzw gs_dotk;\n"
         "__device__ __forceinline__ void gs_dotk_make(lzw &W, const fe c) {\n"
         "    lz_wform(c, W);\n"
         "#pragma unroll\n"
         "    for (int i = 0; i < 5; i++)\n"
         "#pragma unroll\n"
         "        for (int j = 0; j < 5; j++) asm volatile(\"\" : \"+v\"(W.w[i][j]));\n"
         "}\n"
         "struct gs_dot_t { long long c[5]; fe partial; bool has; };\n"
         "__device__ __forceinline__ void gs_dot_begin(gs_dot_t &a) { for (int j = 0; j < 5; j++) a.c[j] = 0; a.has = false; }\n"
         "__device__ __forceinline__ void gs_dot_acc(gs_dot_t &a, const fe x, const lzw &W) {\n"
         "    const lz u = lz_unpack(x);\n"
         "#pragma unroll\n"
         "    for (int j = 0; j < 5; j++) {\n"
         "        long long t = a.c[j];\n"
         "#pragma unroll\n"
         "        for (int i = 0; i < 5; i++) t += (long long)u.l[i] * W.w[i][j];\n"
         "        a.c[j] = t;\n"
         "    }\n"
         "}\n"
         "__device__ __forceinline__ fe gs_dot_fold(gs_dot_t &a) {\n"
         "    const lzk K = lzk_make();\n"
         "    int64_t c[5];\n"
         "    for (int j = 0; j < 5; j++) c[j] = a.c[j];\n"
         "    return lz_pack(lz_fold_columns(c, K));\n"
         "}\n"
         "__device__ __forceinline__ void gs_dot_flush(gs_dot_t &a) {\n"
         "    const fe v = gs_dot_fold(a);\n"
         "    a.partial = a.has ? fe_add(a.partial, v) : v;\n"
         "    a.has = true;\n"
         "    for (int j = 0; j < 5; j++) a.c[j] = 0;\n"
         "}\n"
         "__device__ __forceinline__ fe gs_dot_end(gs_dot_t &a) { const fe v = gs_dot_fold(a); return a.has ? fe_add(a.partial, v) : v; }\n"
         // the same with the constant as an ordinary element (lane-uniform in the constraint kernels: its limbs are scalar values):
         // nine shared columns, the four high ones folded once (lz_fold9)
         "struct gs_dot9_t { long long c[9]; fe partial; bool has; };\n"
         "__device__ __forceinline__ void gs_dot9_begin(gs_dot9_t &a) { for (int k = 0; k < 9; k++) a.c[k] = 0; a.has = false; }\n"
         "__device__ __forceinline__ void gs_dot9_acc(gs_dot9_t &a, const fe x, const fe w) {\n"
         "    const lz u = lz_unpack(x), v = lz_unpack(w);\n"
         "#pragma unroll\n"
         "    for (int k = 0; k < 9; k++) {\n"
         "        long long t = a.c[k];\n"
         "#pragma unroll\n"
         "        for (int i = 0; i < 5; i++) { const int j = k - i; if (j >= 0 && j < 5) t += (long long)u.l[i] * v.l[j]; }\n"
         "        a.c[k] = t;\n"
         "    }\n"
         "}\n"
         "__device__ __forceinline__ fe gs_dot9_fold(gs_dot9_t &a) {\n"
         "    const lzk K = lzk_make();\n"
         "    int64_t c[9];\n"
         "    for (int k = 0; k < 9; k++) c[k] = a.c[k];\n"
         "    return lz_pack(lz_fold9(c, K));\n"
         "}\n"
         "__device__ __forceinline__ void gs_dot9_flush(gs_dot9_t &a) {\n"
         "    const fe v = gs_dot9_fold(a);\n"
         "    a.partial = a.has ? fe_add(a.partial, v) : v;\n"
         "    a.has = true;\n"
         "    for (int k = 0; k < 9; k++) a.c[k] = 0;\n"
         "}\n"
         "__device__ __forceinline__ fe gs_dot9_end(gs_dot9_t &a) { const fe v = gs_dot9_fold(a); return a.has ? fe_add(a.partial, v) : v; }\n";
#else
    s += "typedef fe gs_dotk;\n"
         "__device__ __forceinline__ void gs_dotk_make(fe &k, const fe c) { k = c; }\n"
         "struct gs_dot_t { fe acc; };\n"
         "__device__ __forceinline__ void gs_dot_begin(gs_dot_t &a) { a.acc = fe_zero(); }\n"
         "__device__ __forceinline__ void gs_dot_acc(gs_dot_t &a, const fe x, const fe &k) { a.acc = fe_add(a.acc, gs_mul(x, k)); }\n"
         "__device__ __forceinline__ void gs_dot_flush(gs_dot_t &) {}\n"
         "__device__ __forceinline__ fe gs_dot_end(gs_dot_t &a) { return a.acc; }\n"
         "typedef gs_dot_t gs_dot9_t;\n"
         "__device__ __forceinline__ void gs_dot9_begin(gs_dot_t &a) { a.acc = fe_zero(); }\n"
         "__device__ __forceinline__ void gs_dot9_acc(gs_dot_t &a, const fe x, const fe k) { a.acc = fe_add(a.acc, gs_mul(x, k)); }\n"
         "__device__ __forceinline__ void gs_dot9_flush(gs_dot_t &) {}\n"
         "__device__ __forceinline__ fe gs_dot9_end(gs_dot_t &a) { return a.acc; }\n";
#endif
    return s;
}

// What the generator knows besides the program: the constant pool (exponents become addition chains at generation time) and, in
// the trace kernel, how many lanes work on one segment.
struct JitGen {
    const uint8_t *consts = nullptr;   // host copy, nconsts x GS_ELT bytes
    uint32_t nconsts = 0;
    uint32_t lanes = 1;                // 1, 2 or 4 consecutive lanes per segment (trace kernel only)
    // trace programs: host copy of the static registers' tables (element soff[s] + i), or null.  A static register whose table holds
    // only 0 and 1 is a SELECTOR (Poseidon's full / partial round flag): products by it are emitted as selects (SsaNode::BLEND)
    const uint8_t *statics = nullptr;
    const uint64_t *soff = nullptr, *slen = nullptr;
    uint32_t nstatic = 0;
    bool static_is_binary(uint32_t reg) const {
#ifdef GS_JIT_LAZY
        if (!statics || !soff || !slen || reg >= nstatic) return false;
        for (uint64_t i = 0; i < slen[reg]; i++) {
            const uint8_t *v = statics + (soff[reg] + i) * GS_ELT;
            if (v[0] > 1) return false;
            for (int b = 1; b < GS_ELT; b++) if (v[b]) return false;
        }
        return slen[reg] > 0;
#else
        (void)reg;
        return false;
#endif
    }
};

// x <- x^e as a fixed addition chain: left-to-right sliding windows over the bits of e (little-endian 32-bit limbs), the window
// width chosen by counting products.  The inverse S-box of Rescue (a 128-bit exponent) is 127 squarings + 32 products instead of
// the 127 + 64 of square-and-multiply; a Fermat inversion in the 224-bit field 224 + 53 instead of 224 + 222.  (Right to left the
// squarings and the running product are independent chains, but a wave issues in order and a loop does not interleave them: measured
// in the trace kernel, 2.2 ms against 1.8 ms for the windows.)
struct PowPlan {
    uint32_t table_max = 1;                                    // odd powers x^1 .. x^table_max are needed
    std::vector<std::pair<uint32_t, uint32_t>> steps;          // (squarings, odd power to multiply by; 0 = none); first: (0, start)
    uint32_t cost = 0;
};
static PowPlan pow_plan(const std::vector<uint32_t> &e, int nbits, int w) {
    PowPlan pl;
    auto bit = [&](int i) { return (e[i / 32] >> (i % 32)) & 1u; };
    int i = nbits - 1;
    uint32_t pending = 0;
    bool first = true;
    while (i >= 0) {
        if (!bit(i)) { pending++; i--; continue; }
        int j = i - w + 1 < 0 ? 0 : i - w + 1;
        while (!bit(j)) j++;
        uint32_t val = 0;
        for (int k = i; k >= j; k--) val = 2 * val + bit(k);
        if (first) pl.steps.push_back({0, val});
        else { pl.steps.push_back({pending + (uint32_t)(i - j + 1), val}); pl.cost += pending + (i - j + 1) + 1; }
        first = false;
        pending = 0;
        if (val > pl.table_max) pl.table_max = val;
        i = j - 1;
    }
    if (pending) { pl.steps.push_back({pending, 0}); pl.cost += pending; }
    if (pl.table_max > 1) pl.cost += 1 + (pl.table_max - 1) / 2;
    return pl;
}
// Exponents whose top bits repeat "10" — (2p - 1)/3, the inverse of the cube, is 1010...10 over its top 92 bits in the 128-bit field —
// have a much shorter chain than windows give: with r_k = x^("10" k times), r_{a+b} = r_a^(4^b) * r_b, so r_k costs 2k - 1 squarings and
// about log2 k + popcount k products (k = 46: 91 + 8, where five-bit windows spend 23 products on the same 92 bits), and the r_j met on
// the way are dictionary words for the remaining low bits (greedy longest match among them, x and x^3).  ops: (squarings, word to
// multiply by; "" = none); words are variable names: "p1", "p3", "g<j>" = r_j.  Returns false when the exponent has no such run.
struct PatOp { uint32_t sq; std::string word; };
struct PatPlan {
    uint32_t k = 0;                                            // pairs of the leading run
    std::vector<std::pair<uint32_t, uint32_t>> doubling;       // (j, 0): g_{2j} = g_j^(4^j) g_j;  (j, 1): g_{j+1} = g_j^4 g_1
    std::vector<PatOp> ops;                                    // after acc = g_k
    bool needs_p3 = false;
    uint32_t cost = 0;
};
static bool pattern_plan(const std::vector<uint32_t> &e, int nbits, PatPlan &pl) {
    auto bit = [&](int i) { return i >= 0 ? (e[i / 32] >> (i % 32)) & 1u : 0u; };
    int i = nbits - 1;
    while (i >= 1 && bit(i) && !bit(i - 1)) { pl.k++; i -= 2; }
    if (pl.k < 8) return false;
    std::vector<uint32_t> have = {1};                          // the r_j that exist, ascending
    pl.cost = 1;                                               // g1 = x^2
    {
        int top = 31;
        while (!((pl.k >> top) & 1u)) top--;
        uint32_t j = 1;
        for (int b = top - 1; b >= 0; b--) {
            pl.doubling.push_back({j, 0}); pl.cost += 2 * j + 1; j *= 2; have.push_back(j);
            if ((pl.k >> b) & 1u) { pl.doubling.push_back({j, 1}); pl.cost += 3; j += 1; have.push_back(j); }
        }
    }
    uint32_t pending = 0;
    while (i >= 0) {
        if (!bit(i)) { pending++; i--; continue; }
        // longest dictionary word that matches the bits from i down: "10" x j (needs 2j bits), "11" (x^3), "1" (x)
        uint32_t best_j = 0;
        for (uint32_t j : have) {
            if ((int)(2 * j) > i + 1) continue;
            bool ok = true;
            for (uint32_t t = 0; ok && t < j; t++) ok = bit(i - 2 * (int)t) && !bit(i - 2 * (int)t - 1);
            if (ok) best_j = j;
        }
        char w[16];
        uint32_t len;
        if (best_j) { snprintf(w, sizeof w, "g%u", best_j); len = 2 * best_j; }
        else if (i >= 1 && bit(i - 1)) { snprintf(w, sizeof w, "p3"); len = 2; pl.needs_p3 = true; }
        else { snprintf(w, sizeof w, "p1"); len = 1; }
        pl.ops.push_back({pending + len, w});
        pl.cost += pending + len + 1;
        pending = 0;
        i -= (int)len;
    }
    if (pending) { pl.ops.push_back({pending, ""}); pl.cost += pending; }
    if (pl.needs_p3) pl.cost += 1;
    return true;
}

static void emit_pow(std::string &s, const char *x, const std::vector<uint32_t> &e) {
    char buf[160];
    int nbits = 0;
    for (int i = (int)e.size() * 32 - 1; i >= 0 && !nbits; i--)
        if ((e[i / 32] >> (i % 32)) & 1u) nbits = i + 1;
    if (!nbits) { snprintf(buf, sizeof buf, "            %s = fe_one();\n", x); s += buf; return; }
    PowPlan best = pow_plan(e, nbits, 1);
    for (int w = 2; w <= 5; w++) {
        PowPlan pl = pow_plan(e, nbits, w);
        if (pl.cost < best.cost) best = pl;
    }
    if (best.steps.size() == 1 && best.steps[0].second == 1) return;     // x^1
#ifdef GS_JIT_LAZY
    // A long chain (Rescue's inverse S-box: 127 squarings + 32 products) runs in the lazy five-limb form from end to end: one unpack,
    // lz_sqr (15 products + fold: ~53 instructions against the 84 of a canonical fe_mul) and lz_mul_v, one pack.  On a single wave per
    // SIMD a chain of dependent products costs its instruction count, so this is the length of the trace kernel's critical path.
    PatPlan pat;
    if (pattern_plan(e, nbits, pat) && pat.cost < best.cost) {
        auto sqr_run = [&](const char *v, uint32_t n) {
            if (n >= 4) { snprintf(buf, sizeof buf, "#pragma nounroll\n                for (int q = 0; q < %u; q++) %s = lz_sqr(%s, K);\n", n, v, v); s += buf; }
            else for (uint32_t q = 0; q < n; q++) { snprintf(buf, sizeof buf, "                %s = lz_sqr(%s, K);\n", v, v); s += buf; }
        };
        s += "            {\n                const lzk K = lzk_make();\n";
        snprintf(buf, sizeof buf, "                const lz p1 = lz_unpack(%s);\n", x); s += buf;
        s += "                const lz g1 = lz_sqr(p1, K);\n";
        if (pat.needs_p3) s += "                const lz p3 = lz_mul_v(g1, p1, K);\n";
        s += "                lz acc;\n";
        for (auto &d : pat.doubling) {
            const uint32_t j = d.first;
            snprintf(buf, sizeof buf, "                acc = g%u;\n", j); s += buf;
            sqr_run("acc", d.second ? 2 : 2 * j);
            if (d.second) snprintf(buf, sizeof buf, "                const lz g%u = lz_mul_v(acc, g1, K);\n", j + 1);
            else snprintf(buf, sizeof buf, "                const lz g%u = lz_mul_v(acc, g%u, K);\n", 2 * j, j);
            s += buf;
        }
        snprintf(buf, sizeof buf, "                acc = g%u;\n", pat.k); s += buf;
        for (auto &o : pat.ops) {
            sqr_run("acc", o.sq);
            if (!o.word.empty()) { snprintf(buf, sizeof buf, "                acc = lz_mul_v(acc, %s, K);\n", o.word.c_str()); s += buf; }
        }
        snprintf(buf, sizeof buf, "                %s = lz_pack(acc);\n            }\n", x); s += buf;
        return;
    }
    if (best.cost >= 8) {
        s += "            {\n                const lzk K = lzk_make();\n";
        snprintf(buf, sizeof buf, "                const lz p1 = lz_unpack(%s);\n", x); s += buf;
        if (best.table_max > 1) {
            s += "                const lz p2 = lz_sqr(p1, K);\n";
            for (uint32_t v = 3; v <= best.table_max; v += 2) { snprintf(buf, sizeof buf, "                const lz p%u = lz_mul_v(p%u, p2, K);\n", v, v - 2); s += buf; }
        }
        snprintf(buf, sizeof buf, "                lz acc = p%u;\n", best.steps[0].second); s += buf;
        for (size_t k = 1; k < best.steps.size(); k++) {
            const uint32_t sq = best.steps[k].first, val = best.steps[k].second;
            if (sq >= 4) { snprintf(buf, sizeof buf, "#pragma nounroll\n                for (int q = 0; q < %u; q++) acc = lz_sqr(acc, K);\n", sq); s += buf; }
            else for (uint32_t q = 0; q < sq; q++) s += "                acc = lz_sqr(acc, K);\n";
            if (val) { snprintf(buf, sizeof buf, "                acc = lz_mul_v(acc, p%u, K);\n", val); s += buf; }
        }
        snprintf(buf, sizeof buf, "                %s = lz_pack(acc);\n            }\n", x); s += buf;
        return;
    }
#endif
    s += "            {\n";
    snprintf(buf, sizeof buf, "                const fe p1 = %s;\n", x); s += buf;
    if (best.table_max > 1) {
        s += "                const fe p2 = gs_sqr(p1);\n";
        for (uint32_t v = 3; v <= best.table_max; v += 2) { snprintf(buf, sizeof buf, "                const fe p%u = gs_mul(p%u, p2);\n", v, v - 2); s += buf; }
    }
    snprintf(buf, sizeof buf, "                fe acc = p%u;\n", best.steps[0].second); s += buf;
    for (size_t k = 1; k < best.steps.size(); k++) {
        const uint32_t sq = best.steps[k].first, val = best.steps[k].second;
        if (sq >= 4) { snprintf(buf, sizeof buf, "#pragma nounroll\n                for (int q = 0; q < %u; q++) acc = gs_sqr(acc);\n", sq); s += buf; }
        else for (uint32_t q = 0; q < sq; q++) s += "                acc = gs_sqr(acc);\n";
        if (val) { snprintf(buf, sizeof buf, "                acc = gs_mul(acc, p%u);\n", val); s += buf; }
    }
    snprintf(buf, sizeof buf, "                %s = acc;\n            }\n", x); s += buf;
}

// adjacent exponentiations with one exponent whose results do not feed each other (the S-boxes of a round)
static uint32_t pow_group_size(const uint32_t *code, uint32_t ninstr, uint32_t pc) {
    const uint32_t op = code[4 * pc], b = code[4 * pc + 3];
    uint32_t g = 1;
    while (g < 4 && pc + g < ninstr) {
        const uint32_t *nx = code + 4 * (pc + g);
        bool ok = nx[0] == op && nx[3] == b;
        for (uint32_t i = 0; ok && i < g; i++) ok = nx[2] != code[4 * (pc + i) + 1];
        if (!ok) break;
        g++;
    }
    return g;
}

// straight-line statements for a constraint program (one thread per domain point: throughput-bound, no lanes); `index` names the
// loop variable static tables are indexed by
static bool jit_body(std::string &s, const JitGen &gen, const uint32_t *code, uint32_t ninstr, const uint64_t *soff, const uint64_t *slen,
                     bool allow_statics, const char *cur, const char *nxt, const char *index, const char *sink) {
    char buf[256];
    for (uint32_t pc = 0; pc < ninstr; pc++) {
        const uint32_t op = code[4 * pc], d = code[4 * pc + 1], a = code[4 * pc + 2], b = code[4 * pc + 3];
        switch (op) {
            case J_LOADC: snprintf(buf, sizeof buf, "        t%u = consts[%u];\n", d, a); break;
            case J_LOADR: snprintf(buf, sizeof buf, "        t%u = %s%u;\n", d, cur, a); break;
            case J_LOADN:
                if (!nxt) return false;
                snprintf(buf, sizeof buf, "        t%u = %s%u;\n", d, nxt, a);
                break;
            case J_LOADS:
                if (!allow_statics) return false;
                if (slen[a] & (slen[a] - 1)) snprintf(buf, sizeof buf, "        t%u = statics[%lluull + %s %% %lluull];\n", d, (unsigned long long)soff[a], index, (unsigned long long)slen[a]);
                else snprintf(buf, sizeof buf, "        t%u = statics[%lluull + (%s & %lluull)];\n", d, (unsigned long long)soff[a], index, (unsigned long long)(slen[a] - 1));
                break;
            case J_ADDV: snprintf(buf, sizeof buf, "        t%u = fe_add(t%u, t%u);\n", d, a, b); break;
            case J_SUBV: snprintf(buf, sizeof buf, "        t%u = fe_sub(t%u, t%u);\n", d, a, b); break;
            case J_MULV: snprintf(buf, sizeof buf, "        t%u = gs_mul(t%u, t%u);\n", d, a, b); break;
            case J_POW:
            case J_POWC: {
                std::vector<uint32_t> e(GS_ELT / 4, 0u);
                if (op == J_POW) e[0] = b;
                else {
                    if (!gen.consts || b >= gen.nconsts) return false;
                    memcpy(e.data(), gen.consts + (size_t)b * GS_ELT, GS_ELT);
                }
                const uint32_t g = pow_group_size(code, ninstr, pc);
                s += "        {\n";
                for (uint32_t i = 0; i < g; i++) { snprintf(buf, sizeof buf, "            fe x%u = t%u;\n", i, code[4 * (pc + i) + 2]); s += buf; }
                for (uint32_t i = 0; i < g; i++) { snprintf(buf, sizeof buf, "x%u", i); emit_pow(s, std::string(buf).c_str(), e); }
                for (uint32_t i = 0; i < g; i++) { snprintf(buf, sizeof buf, "            t%u = x%u;\n", code[4 * (pc + i) + 1], i); s += buf; }
                snprintf(buf, sizeof buf, "        }\n");
                pc += g - 1;
                break;
            }
            case J_OUT: snprintf(buf, sizeof buf, "        %s%u = t%u;\n", sink, d, a); break;
            default: return false;
        }
        s += buf;
    }
    return true;
}

// ---- trace programs: products scheduled by depth and spread over the lanes of a segment ------------------------------------------
// A thread of the trace kernel has its SIMD to itself, so it pays the full latency of every product it waits for, and the compiler
// does not overlap inlined 128-bit products either: a compiled Poseidon round (36 MDS products + 18 S-box products) costs
// 54 x ~750 cycles.  The program is therefore put in SSA form and every product gets a depth (1 + the depth of its operands;
// additions, loads and outputs do not add depth): products of one depth are independent, and L of them run as ONE round — lane i of
// the segment's group multiplies the i-th pair, the results are broadcast with shuffles, every lane continues with all values.
// Cheap operations run redundantly on all lanes.  Long exponentiations of one depth and exponent are rounds of their own (one
// member per lane).  L = 1, 2, 4 or 8 by the widest depth.
struct SsaNode {
    enum Kind { ZERO, CONSTV, ROW, ROWN, STATICV, ADD, SUB, MUL, POWLONG, POWSHORT, OUT, DOT, DEAD, BLEND } kind = ZERO;
    int a = -1, b = -1;        // operand nodes (MUL/ADD/SUB/POWLONG/OUT: a; b for binary; BLEND: a = the selector, b = the value) or the index of a constant/register/static
    uint32_t aux = 0;          // POWLONG: constant index of the exponent; POWSHORT: the exponent; OUT: destination register
    int depth = 0;
    // DOT: sum_k value(tx[k]) * consts[tc[k]]  (tc[k] = -1: the literal one) — a linear layer's row (an MDS matrix times the state) fused
    // into ONE unit of work for ONE lane (ssa_fuse_dots)
    std::vector<int> tx, tc;
};
static void ssa_operands(const SsaNode &x, std::vector<int> &out) {
    out.clear();
    switch (x.kind) {
        case SsaNode::ADD: case SsaNode::SUB: case SsaNode::MUL: case SsaNode::BLEND: out.push_back(x.a); out.push_back(x.b); break;
        case SsaNode::POWLONG: case SsaNode::POWSHORT: case SsaNode::OUT: out.push_back(x.a); break;
        case SsaNode::DOT: out = x.tx; break;
        default: break;
    }
}
static void ssa_recompute_depths(std::vector<SsaNode> &nodes) {
    std::vector<int> ops;
    for (SsaNode &n : nodes) {
        ssa_operands(n, ops);
        int d = 0;
        for (int o : ops) d = std::max(d, nodes[o].depth);
        if (n.kind == SsaNode::MUL || n.kind == SsaNode::POWLONG || n.kind == SsaNode::POWSHORT || n.kind == SsaNode::DOT) d++;
        n.depth = d;
    }
}
// Sums of products by constants — the rows of a linear layer, y_j = sum_k m_jk x_k: an ADD tree whose leaves are MUL(value, constant)
// nodes used nowhere else — become ONE node each.  Spread over the lanes product by product such a layer costs a round per L products,
// a select chain per operand, an LDS read-back per product and, on every lane, the whole tree of additions (a Poseidon step: 36
// products in 2.25 rounds + 30 additions x 16 lanes); as a DOT node a row is one lane's work — in the 128-bit field K x 25 v_mad into
// five shared 64-bit columns and ONE fold (gf128_lazy.h: the W-forms of the row's constants are built before the step loop), the rows
// of a layer side by side on K lanes, one exchange for the whole layer.  Leaves that are not products by constants ride along with the
// constant one.  Trees with fewer than two genuine products are left alone.
static void ssa_fuse_dots(std::vector<SsaNode> &nodes) {
#ifdef GS_NTT_EXPERIMENTS
    if (const char *e = getenv("GSTARK_AIR_JIT_FUSE")) if (e[0] == '0') return;       // experiments build only: the unfused programs (A/B of the generator)
#endif
    const int n = (int)nodes.size();
    std::vector<int> uses(n, 0), ops;
    for (const SsaNode &x : nodes) { ssa_operands(x, ops); for (int o : ops) uses[o]++; }
    auto is_const = [&](int id) { return nodes[id].kind == SsaNode::CONSTV && nodes[id].a >= 0; };
    std::vector<bool> absorbed(n, false);
    for (int id = n - 1; id >= 0; id--) {
        if (nodes[id].kind != SsaNode::ADD || absorbed[id]) continue;
        // leaves of the maximal ADD tree rooted here (inner ADDs must have no other user)
        std::vector<int> leaves, inner, stack = {nodes[id].a, nodes[id].b};
        while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            if (nodes[v].kind == SsaNode::ADD && uses[v] == 1 && !absorbed[v]) { inner.push_back(v); stack.push_back(nodes[v].a); stack.push_back(nodes[v].b); }
            else leaves.push_back(v);
        }
        std::vector<int> tx, tc, dead;
        int products = 0;
        for (int v : leaves) {
            const SsaNode &m = nodes[v];
            if (m.kind == SsaNode::MUL && uses[v] == 1 && (is_const(m.a) != is_const(m.b))) {
                tx.push_back(is_const(m.a) ? m.b : m.a);
                tc.push_back(nodes[is_const(m.a) ? m.a : m.b].a);
                dead.push_back(v);
                products++;
            } else { tx.push_back(v); tc.push_back(-1); }
        }
        if (products < 2 || tx.size() > 64) continue;
        // operands in a canonical order: the rows of one layer then line up position by position (shared operands need no selects)
        std::vector<size_t> order(tx.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return tx[x] < tx[y]; });
        SsaNode &r = nodes[id];
        r.kind = SsaNode::DOT;
        r.a = r.b = -1;
        for (size_t k : order) { r.tx.push_back(tx[k]); r.tc.push_back(tc[k]); }
        for (int v : dead) { nodes[v].kind = SsaNode::DEAD; absorbed[v] = true; }
        for (int v : inner) { nodes[v].kind = SsaNode::DEAD; absorbed[v] = true; }
    }
    ssa_recompute_depths(nodes);
}

static bool ssa_build(std::vector<SsaNode> &nodes, const JitGen &gen, const uint32_t *code, uint32_t ninstr, uint32_t vm_regs, bool allow_statics,
                      bool allow_next = false) {
    std::vector<int> cur(vm_regs, -1);
    auto add = [&](SsaNode n) { nodes.push_back(n); return (int)nodes.size() - 1; };
    auto use = [&](uint32_t r) {
        if (cur[r] < 0) { SsaNode z; cur[r] = add(z); }
        return cur[r];
    };
    auto binary = [&](SsaNode::Kind k, int x, int y) {
        SsaNode n;
        n.kind = k; n.a = x; n.b = y;
        n.depth = std::max(nodes[x].depth, nodes[y].depth) + (k == SsaNode::MUL ? 1 : 0);
        return add(n);
    };
    for (uint32_t pc = 0; pc < ninstr; pc++) {
        const uint32_t op = code[4 * pc], d = code[4 * pc + 1], a = code[4 * pc + 2], b = code[4 * pc + 3];
        if (op != J_OUT && d >= vm_regs) return false;
        SsaNode n;
        switch (op) {
            case J_LOADC: n.kind = SsaNode::CONSTV; n.a = (int)a; cur[d] = add(n); break;
            case J_LOADR: n.kind = SsaNode::ROW; n.a = (int)a; cur[d] = add(n); break;
            case J_LOADN:
                if (!allow_next) return false;       // (does not occur in trace programs)
                n.kind = SsaNode::ROWN; n.a = (int)a; cur[d] = add(n);
                break;
            case J_LOADS:
                if (!allow_statics) return false;
                n.kind = SsaNode::STATICV; n.a = (int)a; cur[d] = add(n);
                break;
            case J_ADDV: case J_SUBV: case J_MULV: {
                if (a >= vm_regs || b >= vm_regs) return false;
                const int x = use(a), y = use(b);
                if (op == J_MULV) {
                    // a product by a 0/1 selector is a select: cheap, on every lane, no round of its own (gs_blend still multiplies
                    // should the table ever hold anything else)
                    const bool fx = nodes[x].kind == SsaNode::STATICV && gen.static_is_binary((uint32_t)nodes[x].a);
                    const bool fy = nodes[y].kind == SsaNode::STATICV && gen.static_is_binary((uint32_t)nodes[y].a);
                    if (fx || fy) {
                        SsaNode bl;
                        bl.kind = SsaNode::BLEND; bl.a = fx ? x : y; bl.b = fx ? y : x;
                        bl.depth = std::max(nodes[x].depth, nodes[y].depth);
                        cur[d] = add(bl);
                        break;
                    }
                }
                cur[d] = binary(op == J_ADDV ? SsaNode::ADD : (op == J_SUBV ? SsaNode::SUB : SsaNode::MUL), x, y);
                break;
            }
            case J_POW: case J_POWC: {
                if (a >= vm_regs) return false;
                const int x = use(a);
                uint64_t e = b;
                if (op == J_POWC) {
                    if (!gen.consts || b >= gen.nconsts) return false;
                    const uint8_t *eb = gen.consts + (size_t)b * GS_ELT;
                    bool wide_exponent = false;
                    for (int i = 4; i < GS_ELT; i++) wide_exponent |= eb[i] != 0;
                    if (wide_exponent) {
                        n.kind = SsaNode::POWLONG; n.a = x; n.aux = b; n.depth = nodes[x].depth + 1;
                        cur[d] = add(n);
                        break;
                    }
                    e = (uint64_t)eb[0] | ((uint64_t)eb[1] << 8) | ((uint64_t)eb[2] << 16) | ((uint64_t)eb[3] << 24);
                }
                if (e == 0) { n.kind = SsaNode::CONSTV; n.a = -1; cur[d] = add(n); break; }      // x^0 = 1 (a = -1: the literal one)
                if (e == 1) { cur[d] = x; break; }
                if (e >> 32) return false;
                // an S-box: a short chain of products that stays on ONE lane (x^5: square, square, multiply) — a round of its own
                // kind, so that the lanes exchange the powers only, not every intermediate square
                n.kind = SsaNode::POWSHORT; n.a = x; n.aux = (uint32_t)e; n.depth = nodes[x].depth + 1;
                cur[d] = add(n);
                break;
            }
            case J_OUT:
                if (a >= vm_regs) return false;
                n.kind = SsaNode::OUT; n.a = use(a); n.aux = d; n.depth = nodes[n.a].depth;
                add(n);
                break;
            default: return false;      // J_LOADN does not occur in trace programs
        }
    }
    return true;
}

// Long exponentiations that do not depend on each other should share a round even when their operands are ready at different depths
// (the two slopes of a point addition: one inversion waits for the doubled point, the other does not): the earlier one is delayed to
// the depth of the later one — a delay of a few products against a whole exponentiation saved — and the depths are recomputed.
static void ssa_align_pows(std::vector<SsaNode> &nodes) {
    const int n = (int)nodes.size();
    std::vector<int> pows;
    for (int id = 0; id < n; id++)
        if (nodes[id].kind == SsaNode::POWLONG) pows.push_back(id);
    if (pows.size() < 2) return;
    std::vector<int> op;
    // reach[k][id]: node id depends on pows[k]
    std::vector<std::vector<bool>> reach(pows.size(), std::vector<bool>(n, false));
    for (size_t k = 0; k < pows.size(); k++)
        for (int id = pows[k] + 1; id < n; id++) {
            ssa_operands(nodes[id], op);
            for (int o : op)
                if (o >= 0 && (o == pows[k] || reach[k][o])) reach[k][id] = true;
        }
    std::vector<int> forced(n, 0);
    std::vector<bool> grouped(pows.size(), false);
    for (size_t k = 0; k < pows.size(); k++) {
        if (grouped[k]) continue;
        std::vector<size_t> group = {k};
        grouped[k] = true;
        for (size_t j = k + 1; j < pows.size() && group.size() < 8; j++) {
            if (grouped[j] || nodes[pows[j]].aux != nodes[pows[k]].aux) continue;
            bool independent = true;
            for (size_t g : group) independent &= !reach[g][pows[j]] && !reach[j][pows[g]];
            if (independent) { group.push_back(j); grouped[j] = true; }
        }
        int depth = 0;
        for (size_t g : group) depth = std::max(depth, nodes[pows[g]].depth);
        for (size_t g : group) forced[pows[g]] = depth;
        // the delay moves everything behind the group: recompute before the next group is formed
        for (int id = 0; id < n; id++) {
            int d = 0;
            ssa_operands(nodes[id], op);
            for (int o : op)
                if (o >= 0) d = std::max(d, nodes[o].depth);
            if (nodes[id].kind == SsaNode::MUL || nodes[id].kind == SsaNode::POWLONG || nodes[id].kind == SsaNode::POWSHORT || nodes[id].kind == SsaNode::DOT) d++;
            nodes[id].depth = std::max(d, forced[id]);
        }
    }
}

static uint32_t ssa_lanes(const std::vector<SsaNode> &nodes) {
    std::map<int, uint32_t> per_depth;
    uint32_t widest = 1;
    for (const SsaNode &n : nodes) {
#if defined(GS_WIDE_BITS)
        // multi-limb flavours: the group is as wide as the layer of long exponentiations (a step is their chain of ~280 products;
        // more lanes for the sake of the few single products changes nothing: point multiplication 79.5 vs 79.7 ms)
        if (n.kind == SsaNode::MUL || n.kind == SsaNode::POWSHORT) continue;
#endif
        if (n.kind == SsaNode::MUL || n.kind == SsaNode::POWLONG || n.kind == SsaNode::POWSHORT || n.kind == SsaNode::DOT)
            widest = std::max(widest, ++per_depth[n.depth * 4 + (n.kind == SsaNode::POWLONG ? 1 : (n.kind == SsaNode::POWSHORT ? 2 : (n.kind == SsaNode::DOT ? 3 : 0)))]);
    }
    return widest >= 12 ? 16 : (widest >= 5 ? 8 : (widest >= 3 ? 4 : widest));
}

// `hoisted` receives declarations that belong before the step loop: the constants the body uses, and per round of products by
// constants the ONE constant each lane needs (a lane-dependent, step-independent load instead of L loads and selects per step).
static void ssa_emit(std::string &s, std::string &hoisted, std::vector<bool> &const_declared, const std::vector<SsaNode> &nodes, const JitGen &gen,
                     const uint64_t *soff, const uint64_t *slen, const char *index, const char *sink, const char *tag) {
    char buf[256];
    const uint32_t L = gen.lanes;
    int max_depth = 0;
    for (const SsaNode &n : nodes) max_depth = std::max(max_depth, n.depth);
    auto name = [&](int id) {
        const SsaNode &n = nodes[id];
        char t[48];
        if (n.kind == SsaNode::ZERO) return std::string("fe_zero()");
        if (n.kind == SsaNode::ROW) { snprintf(t, sizeof t, "r%d", n.a); return std::string(t); }
        if (n.kind == SsaNode::CONSTV) {
            if (n.a < 0) return std::string("fe_one()");
            if (!const_declared[n.a]) {
                snprintf(t, sizeof t, "    const fe c%d = consts[%d];\n", n.a, n.a);
                hoisted += t;
                const_declared[n.a] = true;
            }
            snprintf(t, sizeof t, "c%d", n.a);
            return std::string(t);
        }
        snprintf(t, sizeof t, "%s%d", tag, id);
        return std::string(t);
    };
    auto is_const = [&](int id) { return nodes[id].kind == SsaNode::CONSTV && nodes[id].a >= 0; };
    int round_no = 0;
    const bool spread = L > 1;
    for (int depth = 0; depth <= max_depth; depth++) {
        // the products of this depth, L per round
        std::vector<int> muls, pows, dots;
        for (int id = 0; id < (int)nodes.size(); id++) {
            if (nodes[id].depth != depth) continue;
            if (nodes[id].kind == SsaNode::MUL) muls.push_back(id);
            if (nodes[id].kind == SsaNode::POWLONG || nodes[id].kind == SsaNode::POWSHORT) pows.push_back(id);
            if (nodes[id].kind == SsaNode::DOT) dots.push_back(id);
        }
        if (spread) {   // products by constants first, so that rounds are all-constant where they can be
            std::stable_partition(muls.begin(), muls.end(), [&](int id) { return is_const(nodes[id].a) || is_const(nodes[id].b); });
        }
        for (size_t base = 0; base < muls.size(); base += spread ? L : 1) {
            const size_t m = std::min<size_t>(L, muls.size() - base);
            if (!spread) {
                const SsaNode &n = nodes[muls[base]];
                s += "        const fe " + name(muls[base]) + " = gs_mul(" + name(n.a) + ", " + name(n.b) + ");\n";
                continue;
            }
            // operands: (variable, constant) when every product of the round has a constant side
            std::vector<int> xa(m), xb(m);
            bool by_consts = true, squares = true;
            for (size_t i = 0; i < m; i++) {
                const SsaNode &n = nodes[muls[base + i]];
                xa[i] = n.a; xb[i] = n.b;
                if (is_const(xa[i]) && !is_const(xb[i])) std::swap(xa[i], xb[i]);
                by_consts &= is_const(xb[i]);
                squares &= n.a == n.b;
            }
            for (size_t i = 0; i < m; i++) s += "        fe " + name(muls[base + i]) + ";\n";
            s += "        {\n            fe xa = " + name(xa[0]) + ";\n";
            for (size_t i = 1; i < m; i++) {
                if (xa[i] == xa[0]) continue;                      // (the lanes that keep the default need no select)
                snprintf(buf, sizeof buf, "            xa = gs_pick(sub == %zuu, ", i); s += buf + name(xa[i]) + ", xa);\n";
            }
            if (squares) s += "            const fe xr = gs_sqr(xa);\n";
            else if (by_consts) {
                snprintf(buf, sizeof buf, "    const fe %sk%d = consts[", tag, round_no); hoisted += buf;
                for (size_t i = m - 1; i >= 1; i--) { snprintf(buf, sizeof buf, "sub == %zuu ? %du : ", i, nodes[xb[i]].a); hoisted += buf; }
                snprintf(buf, sizeof buf, "%du];\n", nodes[xb[0]].a); hoisted += buf;
                snprintf(buf, sizeof buf, "            const fe xr = gs_mul(xa, %sk%d);\n", tag, round_no); s += buf;
            } else {
                s += "            fe xb = " + name(xb[0]) + ";\n";
                for (size_t i = 1; i < m; i++) {
                    if (xb[i] == xb[0]) continue;
                    snprintf(buf, sizeof buf, "            xb = gs_pick(sub == %zuu, ", i); s += buf + name(xb[i]) + ", xb);\n";
                }
                s += "            const fe xr = gs_mul(xa, xb);\n";
            }
            s += "            gs_swap[threadIdx.x] = xr;\n            __syncthreads();\n";
            for (size_t i = 0; i < m; i++) { snprintf(buf, sizeof buf, " = gs_swap[gs_group + %zu];\n", i); s += "            " + name(muls[base + i]) + buf; }
            s += "            __syncthreads();\n        }\n";
            round_no++;
        }
        // long exponentiations of this depth: one member per lane, grouped by exponent
        std::vector<bool> done(pows.size(), false);
        for (size_t first = 0; first < pows.size(); first++) {
            if (done[first]) continue;
            std::vector<int> members;
            for (size_t k = first; k < pows.size() && members.size() < L; k++)
                if (!done[k] && nodes[pows[k]].kind == nodes[pows[first]].kind && nodes[pows[k]].aux == nodes[pows[first]].aux) { members.push_back(pows[k]); done[k] = true; }
            std::vector<uint32_t> e(GS_ELT / 4, 0u);
            if (nodes[pows[first]].kind == SsaNode::POWSHORT) e[0] = nodes[pows[first]].aux;
            else memcpy(e.data(), gen.consts + (size_t)nodes[pows[first]].aux * GS_ELT, GS_ELT);
            for (int id : members) s += "        fe " + name(id) + ";\n";
            s += "        {\n            fe x = " + name(nodes[members[0]].a) + ";\n";
            for (size_t i = 1; i < members.size(); i++) { snprintf(buf, sizeof buf, "            x = gs_pick(sub == %zuu, ", i); s += buf + name(nodes[members[i]].a) + ", x);\n"; }
            emit_pow(s, "x", e);
            if (L > 1) s += "            gs_swap[threadIdx.x] = x;\n            __syncthreads();\n";
            for (size_t i = 0; i < members.size(); i++) {
                if (L == 1) s += "            " + name(members[i]) + " = x;\n";
                else { snprintf(buf, sizeof buf, " = gs_swap[gs_group + %zu];\n", i); s += "            " + name(members[i]) + buf; }
            }
            if (L > 1) s += "            __syncthreads();\n";
            s += "        }\n";
        }
        // the fused rows of this depth (ssa_fuse_dots): one row per lane, rows with the same number of terms side by side
        std::vector<bool> dot_done(dots.size(), false);
        for (size_t first = 0; first < dots.size(); first++) {
            if (dot_done[first]) continue;
            std::vector<int> members;
            const size_t K = nodes[dots[first]].tx.size();
            for (size_t k = first; k < dots.size() && members.size() < (spread ? L : 1); k++)
                if (!dot_done[k] && nodes[dots[k]].tx.size() == K) { members.push_back(dots[k]); dot_done[k] = true; }
            const size_t m = members.size();
            for (int id : members) s += "        fe " + name(id) + ";\n";
            // the constants of position k, one per lane, in the form the accumulation wants (before the step loop)
            for (size_t k = 0; k < K; k++) {
                snprintf(buf, sizeof buf, "    gs_dotk %sd%d_%zu; gs_dotk_make(%sd%d_%zu, ", tag, round_no, k, tag, round_no, k); hoisted += buf;
                std::string sel;
                for (size_t i = m - 1; i >= 1; i--) {
                    const int c = nodes[members[i]].tc[k];
                    if (c < 0) snprintf(buf, sizeof buf, "sub == %zuu ? fe_one() : ", i); else snprintf(buf, sizeof buf, "sub == %zuu ? consts[%d] : ", i, c);
                    sel += buf;
                }
                const int c0 = nodes[members[0]].tc[k];
                if (c0 < 0) sel += "fe_one()"; else { snprintf(buf, sizeof buf, "consts[%d]", c0); sel += buf; }
                hoisted += sel + ");\n";
            }
            s += "        {\n            gs_dot_t acc;\n            gs_dot_begin(acc);\n";
            for (size_t k = 0; k < K; k++) {
                if (k && k % 6 == 0) s += "            gs_dot_flush(acc);\n";      // (128-bit field: six terms per fold keep the 64-bit columns below 2^57)
                bool shared = true;
                for (size_t i = 1; i < m; i++) shared &= nodes[members[i]].tx[k] == nodes[members[0]].tx[k];
                if (shared) { snprintf(buf, sizeof buf, ", %sd%d_%zu);\n", tag, round_no, k); s += "            gs_dot_acc(acc, " + name(nodes[members[0]].tx[k]) + buf; }
                else {
                    s += "            {\n                fe xo = " + name(nodes[members[0]].tx[k]) + ";\n";
                    for (size_t i = 1; i < m; i++) { snprintf(buf, sizeof buf, "                xo = gs_pick(sub == %zuu, ", i); s += buf + name(nodes[members[i]].tx[k]) + ", xo);\n"; }
                    snprintf(buf, sizeof buf, "                gs_dot_acc(acc, xo, %sd%d_%zu);\n            }\n", tag, round_no, k); s += buf;
                }
            }
            s += "            const fe xr = gs_dot_end(acc);\n";
            if (spread) {
                s += "            gs_swap[threadIdx.x] = xr;\n            __syncthreads();\n";
                for (size_t i = 0; i < m; i++) { snprintf(buf, sizeof buf, " = gs_swap[gs_group + %zu];\n", i); s += "            " + name(members[i]) + buf; }
                s += "            __syncthreads();\n";
            } else s += "            " + name(members[0]) + " = xr;\n";
            s += "        }\n";
            round_no++;
        }
        // everything cheap of this depth, in program order
        for (int id = 0; id < (int)nodes.size(); id++) {
            const SsaNode &n = nodes[id];
            if (n.depth != depth) continue;
            switch (n.kind) {
                case SsaNode::STATICV: {
                    // the value of the NEXT step is requested now and used one iteration later: nothing hides a load's latency here
                    char at[160], next_at[160];
                    if (slen[n.a] & (slen[n.a] - 1)) {
                        snprintf(at, sizeof at, "statics[%lluull + (g * seglen) %% %lluull]", (unsigned long long)soff[n.a], (unsigned long long)slen[n.a]);
                        snprintf(next_at, sizeof next_at, "statics[%lluull + (%s + 1ull) %% %lluull]", (unsigned long long)soff[n.a], index, (unsigned long long)slen[n.a]);
                    } else {
                        snprintf(at, sizeof at, "statics[%lluull + ((g * seglen) & %lluull)]", (unsigned long long)soff[n.a], (unsigned long long)(slen[n.a] - 1));
                        snprintf(next_at, sizeof next_at, "statics[%lluull + ((%s + 1ull) & %lluull)]", (unsigned long long)soff[n.a], index, (unsigned long long)(slen[n.a] - 1));
                    }
                    snprintf(buf, sizeof buf, "    fe %sq%d = %s;\n", tag, id, at); hoisted += buf;
                    snprintf(buf, sizeof buf, " = %sq%d;\n        %sq%d = %s;\n", tag, id, tag, id, next_at);
                    s += "        const fe " + name(id) + buf;
                    break;
                }
                case SsaNode::ADD: s += "        const fe " + name(id) + " = fe_add(" + name(n.a) + ", " + name(n.b) + ");\n"; break;
                case SsaNode::SUB: s += "        const fe " + name(id) + " = fe_sub(" + name(n.a) + ", " + name(n.b) + ");\n"; break;
                case SsaNode::BLEND: s += "        const fe " + name(id) + " = gs_blend(" + name(n.a) + ", " + name(n.b) + ");\n"; break;
                case SsaNode::OUT: snprintf(buf, sizeof buf, "        %s%u = ", sink, n.aux); s += buf + name(n.a) + ";\n"; break;
                default: break;
            }
        }
    }
}

// hiprtc: source -> gfx950 code object (no device needed); false + log on failure
static bool jit_compile(const std::string &source, const char *entry, std::vector<char> &code, std::string &log) {
    hiprtcProgram prog;
#ifdef GS_JIT_LAZY
    const char *header_names[] = {"gf128.h", "gf128_lazy.h"};
    const char *headers[] = {kFieldHeader, kLazyHeader};
    const int nheaders = 2;
#else
    const char *header_names[] = {"gs_field.h"};
    const char *headers[] = {kFieldHeader};
    const int nheaders = 1;
#endif
    if (hiprtcCreateProgram(&prog, source.c_str(), "gs_air_jit.hip", nheaders, headers, header_names) != HIPRTC_SUCCESS) { log = "hiprtcCreateProgram failed"; return false; }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    const char *verbose = getenv("GSTARK_AIR_JIT_VERBOSE");
    if (verbose) fprintf(stderr, "[gstark] compiling an AIR program (%zu bytes of source, entry %s)\n", source.size(), entry);
    if (const char *dump = getenv("GSTARK_AIR_JIT_DUMP")) {          // keep the generated source (<dir>/<entry>_<n>.hip) for inspection
        static std::atomic<int> dumped{0};
        char path[512];
        snprintf(path, sizeof path, "%s/%s_%d.hip", dump, entry, dumped++);
        if (FILE *f = fopen(path, "w")) { fputs(source.c_str(), f); fclose(f); }
    }
    hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        log.assign(ls, 0);
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        if (verbose) fprintf(stderr, "[gstark] hiprtc failed, interpreting instead:\n%s\n%.3000s\n", log.c_str(), verbose[0] == '2' ? source.c_str() : "");
        hiprtcDestroyProgram(&prog);
        return false;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return true;
}

// ---- background builds run in a helper process (jitc.cc explains why): <directory of this library>/gstark_jitc, or $GSTARK_JITC.
// 1 = built (code filled; the helper has written the cache file too), 0 = the compiler refused the source (log filled),
// -1 = no helper / could not be started (the caller compiles in-process, as gs_air_jit(ctx, 1) does)
extern char **environ;
static std::string jit_helper_path() {
    if (const char *e = getenv("GSTARK_JITC")) return e;
    Dl_info info;
    if (!dladdr((const void *)&jit_helper_path, &info) || !info.dli_fname) return "";
    std::string lib = info.dli_fname;
    const size_t slash = lib.rfind('/');
    return (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/gstark_jitc";
}
static bool sock_send(int fd, const void *src, size_t n) {
    const char *p = (const char *)src;
    while (n) {
        const ssize_t r = send(fd, p, n, MSG_NOSIGNAL);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static bool sock_recv(int fd, void *dst, size_t n) {
    char *p = (char *)dst;
    while (n) {
        const ssize_t r = recv(fd, p, n, 0);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static int jit_compile_helper(const std::string &source, const char *entry, const std::string &cache_path, std::vector<char> &code, std::string &log) {
    const std::string helper = jit_helper_path();
    if (helper.empty() || access(helper.c_str(), X_OK) != 0) return -1;
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0, sv) != 0) return -1;
    {
        std::lock_guard<std::mutex> g(g_jit_mutex);
        if (g_jit_shutdown.load()) { close(sv[0]); close(sv[1]); return 0; }
        g_jit_helper_fds.push_back(sv[0]);
    }
    auto forget = [&] { std::lock_guard<std::mutex> g(g_jit_mutex); g_jit_helper_fds.erase(std::find(g_jit_helper_fds.begin(), g_jit_helper_fds.end(), sv[0])); close(sv[0]); };
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, sv[1], 0);                 // dup2 clears close-on-exec on the copies
    posix_spawn_file_actions_adddup2(&fa, sv[1], 1);
    if (!getenv("GSTARK_AIR_JIT_VERBOSE")) posix_spawn_file_actions_addopen(&fa, 2, "/dev/null", O_WRONLY, 0);
#if defined(__GLIBC__) && (__GLIBC__ > 2 || (__GLIBC__ == 2 && __GLIBC_MINOR__ >= 34))
    posix_spawn_file_actions_addclosefrom_np(&fa, 3);                // nothing else of the host (device nodes, sockets) reaches the helper
#endif
    char *const argv[] = {const_cast<char *>(helper.c_str()), nullptr};
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, helper.c_str(), &fa, nullptr, argv, environ);
    posix_spawn_file_actions_destroy(&fa);
    close(sv[1]);
    if (rc != 0) { forget(); return -1; }
    if (getenv("GSTARK_AIR_JIT_VERBOSE")) fprintf(stderr, "[gstark] %s: built by %s (pid %d), %zu bytes of source\n", entry, helper.c_str(), (int)pid, source.size());
    const int fd = sv[0];
    std::vector<std::pair<std::string, const char *>> items;
    items.push_back({entry, source.c_str()});
#ifdef GS_JIT_LAZY
    items.push_back({"gf128.h", kFieldHeader});
    items.push_back({"gf128_lazy.h", kLazyHeader});
#else
    items.push_back({"gs_field.h", kFieldHeader});
#endif
    const uint32_t nitems = (uint32_t)items.size();
    bool ok = sock_send(fd, "GSJ1", 4) && sock_send(fd, &nitems, 4);
    for (size_t i = 0; ok && i < items.size(); i++) {
        const uint32_t nl = (uint32_t)items[i].first.size();
        const uint64_t dl = strlen(items[i].second);
        ok = sock_send(fd, &nl, 4) && sock_send(fd, items[i].first.data(), nl) && sock_send(fd, &dl, 8) && sock_send(fd, items[i].second, dl);
    }
    const uint32_t pl = (uint32_t)cache_path.size();
    ok = ok && sock_send(fd, &pl, 4) && sock_send(fd, cache_path.data(), pl);
    char tag[4];
    uint64_t len = 0;
    int result = -1;
    if (ok && sock_recv(fd, tag, 4) && sock_recv(fd, &len, 8) && len <= (256ull << 20)) {
        std::vector<char> body(len);
        if (sock_recv(fd, body.data(), len)) {
            if (memcmp(tag, "GSOK", 4) == 0 && len > 0) { code.swap(body); result = 1; }
            else if (memcmp(tag, "GSER", 4) == 0) { log.assign(body.begin(), body.end()); result = 0; }
        }
    }
    forget();
    int status = 0;
    (void)waitpid(pid, &status, g_jit_shutdown.load() ? WNOHANG : 0);          // reap; the verdict is what came over the socket
    if (result < 0 && g_jit_shutdown.load()) result = 0;            // released at exit: nothing to fall back to
    return result;
}

// ---- code objects on disk: <dir>/<sha256 of the generated source>.hsaco.  A compiled program outlives the process that built it, so
// "compile when an AIR is instantiated" costs its seconds once per machine, and the default mode (auto) can use compiled programs
// whenever they already exist without ever making a proof wait for the compiler.
// The directory is PRIVATE to the user: created 0700, and used only if it is a real directory (no symlink) owned by the effective
// user with no group / other write permission — a code object found there is loaded into the proving path, so nobody else may be
// able to plant one.  Without GSTARK_JIT_CACHE_DIR and without HOME (daemons, bare containers) there is no safe default: no cache.
// Files are opened without following symlinks and must be regular files of the same owner.
static bool jit_dir_is_private(const std::string &dir) {
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH));
}
static std::string jit_cache_path(const std::string &source, const char *entry) {
    const char *dir = getenv("GSTARK_JIT_CACHE_DIR");
    std::string base;
    if (dir && dir[0]) base = dir;
    else {
        const char *home = getenv("HOME");
        if (!home || !home[0]) return "";
        base = std::string(home) + "/.cache/gstark_jit";
    }
    if (base == "0" || base == "off") return "";
    while (base.size() > 1 && base.back() == '/') base.pop_back();
    std::string acc;
    for (size_t i = 1; i <= base.size(); i++)          // mkdir -p: parents with the usual mode, the cache directory itself private
        if (i == base.size() || base[i] == '/') { acc = base.substr(0, i); (void)mkdir(acc.c_str(), i == base.size() ? 0700 : 0755); }
    if (!jit_dir_is_private(base)) {
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true) && getenv("GSTARK_AIR_JIT_VERBOSE"))
            fprintf(stderr, "[gstark] code-object cache %s is not a private directory of this user (owner, mode, symlink): not used\n", base.c_str());
        return "";
    }
    // the key covers everything the code object depends on: the generated source AND the field headers it is compiled against
    std::string keyed = std::string(entry) + "|gfx950|v2|" + kFieldHeader;
#ifdef GS_JIT_LAZY
    keyed += kLazyHeader;
#endif
    keyed += "|" + source;
    uint8_t d[32];
    host_sha256((const uint8_t *)keyed.data(), keyed.size(), d);
    char hex[65];
    for (int i = 0; i < 32; i++) snprintf(hex + 2 * i, 3, "%02x", d[i]);
    return base + "/" + hex + ".hsaco";
}
static bool jit_disk_read(const std::string &path, std::vector<char> &code) {
    if (path.empty()) return false;
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH)) && st.st_size > 0;
    if (ok) {
        code.resize((size_t)st.st_size);
        size_t got = 0;
        while (got < code.size()) {
            const ssize_t r = read(fd, code.data() + got, code.size() - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        ok = got == code.size();
    }
    close(fd);
    if (!ok) code.clear();
    return ok;
}
static void jit_disk_write(const std::string &path, const std::vector<char> &code) {
    if (path.empty() || code.empty()) return;
    char tmp[32];
    snprintf(tmp, sizeof tmp, ".%d.tmp", (int)getpid());
    const std::string t = path + tmp;
    const int fd = open(t.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return;
    size_t put = 0;
    while (put < code.size()) {
        const ssize_t r = write(fd, code.data() + put, code.size() - put);
        if (r <= 0) break;
        put += (size_t)r;
    }
    close(fd);
    if (put == code.size()) rename(t.c_str(), path.c_str()); else remove(t.c_str());
}
static bool jit_load(JitKernel &k, const char *entry) {
    if (hipModuleLoadData(&k.module, k.code.data()) != hipSuccess) return false;
    if (hipModuleGetFunction(&k.fn, k.module, entry) != hipSuccess) return false;
    std::vector<char>().swap(k.code);
    return true;
}

// mode 1 (gs_air_jit(ctx, 1)): the kernel, compiling it now if nobody has yet.  mode 2 (auto, the default): the kernel if this process
// or the disk cache already holds it; otherwise null — the caller interprets this time — and ONE background thread builds it for the
// launches (and processes) to come.
static JitKernel *jit_get(gs_ctx *c, const std::string &source, const char *entry) {
    std::unique_lock<std::mutex> lock(g_jit_mutex);
    auto it = g_jit_cache.find(source);
    if (it != g_jit_cache.end()) {
        JitKernel &k = it->second;
        if (k.fn) return &k;
        if (k.failed) return nullptr;
        if (!k.code.empty()) { if (jit_load(k, entry)) return &k; k.failed = true; return nullptr; }
        if (k.compiling && c->air_jit != 1) return nullptr;           // still being built in the background: interpret
        if (k.compiling) {                                            // mode on: wait for the builder instead of compiling twice
            while (k.compiling) { lock.unlock(); usleep(2000); lock.lock(); }
            if (k.failed || k.code.empty()) return nullptr;
            if (jit_load(k, entry)) return &k;
            k.failed = true;
            return nullptr;
        }
    }
    JitKernel &k = g_jit_cache[source];
    const std::string path = jit_cache_path(source, entry);
    if (jit_disk_read(path, k.code)) {
        if (jit_load(k, entry)) return &k;
        k.failed = false;                                             // a stale / truncated file: fall through and rebuild
        std::vector<char>().swap(k.code);
    }
    if (c->air_jit == 1) {
        k.failed = true;
        std::string log;
        if (!jit_compile(source, entry, k.code, log)) { gs_fail(c, GS_ERR_DEVICE, "air jit: %.400s", log.c_str()); return nullptr; }
        jit_disk_write(path, k.code);
        if (!jit_load(k, entry)) return nullptr;
        k.failed = false;
        return &k;
    }
    if (g_jit_shutdown.load()) return nullptr;                        // the process is on its way out: interpret, start nothing
    k.compiling = true;
    // Background builds run in the helper process: a host that exits meanwhile leaves nothing of the compiler inside THIS process to be
    // torn down under a running thread.  Only without the helper (not installed next to the library) does the builder compile here, and
    // then exit waits for it (jit_join_builders, bounded).
    static std::once_flag at_exit_once;
    std::call_once(at_exit_once, [] { atexit(jit_join_builders); });
    g_jit_builders.emplace_back([source, path, entry]() {
        std::vector<char> code;
        std::string log;
        int built = g_jit_shutdown.load() ? 0 : jit_compile_helper(source, entry, path, code, log);
        if (built < 0) {
            static std::atomic<bool> told{false};
            if (!told.exchange(true)) fprintf(stderr, "[gstark] gstark_jitc not found next to the library (%s): AIR programs are compiled inside this process\n", jit_helper_path().c_str());
            g_jit_running.fetch_add(1);
            struct Done { ~Done() { g_jit_running.fetch_sub(1); } } done;
            static std::mutex &one_at_a_time = *new std::mutex;   // hiprtc builds one program at a time (two concurrent builds: the second came back empty)
            std::lock_guard<std::mutex> b(one_at_a_time);
            built = !g_jit_shutdown.load() && jit_compile(source, entry, code, log) ? 1 : 0;          // queued behind another build at exit: skip
            if (built == 1) jit_disk_write(path, code);
        }
        if (built != 1 && getenv("GSTARK_AIR_JIT_VERBOSE")) fprintf(stderr, "[gstark] background build of %s failed: %.600s\n", entry, log.c_str());
        std::lock_guard<std::mutex> g(g_jit_mutex);
        JitKernel &kk = g_jit_cache[source];
        if (built == 1) kk.code.swap(code); else kk.failed = true;
        kk.compiling = false;
    });
    return nullptr;
}

// ---- generated sources, remembered.  Turning a program into source (SSA, scheduling, ~10 KB of text) costs ~0.1 ms of host time — per
// proof, if done per call, and a Poseidon proof is 1.5 ms.  The inputs of the generator — program(s), constants, static layout and,
// for trace programs, which static registers are 0/1 selectors — are concatenated into a key (a ~10 KB memcpy + one map lookup);
// the source (itself the key of the code-object cache) and the lane count are generated once per key and process.
struct JitSource { std::string source; uint32_t lanes = 1; bool ok = false; };
static std::mutex &g_src_mutex = *new std::mutex;
static std::map<std::string, std::shared_ptr<JitSource>> &g_src_memo = *new std::map<std::string, std::shared_ptr<JitSource>>;
static void key_add(std::string &k, const void *p, size_t n) { const uint64_t len = n; k.append((const char *)&len, 8); if (n) k.append((const char *)p, n); }
template <class Make>
static std::shared_ptr<const JitSource> jit_source_memo(const std::string &key, Make make) {
    std::lock_guard<std::mutex> g(g_src_mutex);
    auto it = g_src_memo.find(key);
    if (it == g_src_memo.end()) {
        if (g_src_memo.size() > 256) g_src_memo.clear();          // a host cycling through hundreds of AIRs: start over rather than grow (callers hold their entry)
        it = g_src_memo.emplace(key, std::make_shared<JitSource>()).first;
        make(*it->second);
    }
    return it->second;
}

// ---- trace segments --------------------------------------------------------------------------------------------------------------
// returns GS_OK when the compiled kernel was launched, GS_ERR_UNSUPPORTED when the caller should interpret instead
static bool jit_trace_source(std::string &s, JitGen &gen, const uint32_t *code, uint32_t ninstr, const uint32_t *icode, uint32_t init_ninstr,
                             uint32_t vm_regs, uint32_t registers, const uint64_t *soff, const uint64_t *slen) {
    s = jit_preamble();
    char buf[256];
    std::vector<SsaNode> main_nodes, init_nodes;
    if (!ssa_build(main_nodes, gen, code, ninstr, vm_regs, true)) return false;
    if (init_ninstr && !ssa_build(init_nodes, gen, icode, init_ninstr, vm_regs, false)) return false;
    ssa_fuse_dots(main_nodes);
    ssa_fuse_dots(init_nodes);
    ssa_align_pows(main_nodes);
    ssa_align_pows(init_nodes);
    gen.lanes = std::max(ssa_lanes(main_nodes), ssa_lanes(init_nodes));
    snprintf(buf, sizeof buf, "#define GS_LANES %uu\n", gen.lanes); s += buf;
    s += "extern \"C\" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void gs_jit_trace(const fe *__restrict__ consts, const fe *__restrict__ statics, const fe *__restrict__ first_rows,\n"
         "                                         unsigned long long segments, unsigned long long seglen, fe *__restrict__ out) {\n"
         "    const unsigned long long tid = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;\n"
         "    // every thread runs every step (and reaches every barrier): the lanes past the last segment recompute it and store nothing\n"
         "    const bool gs_live = tid / GS_LANES < segments;\n"
         "    const unsigned long long g = gs_live ? tid / GS_LANES : segments - 1ull;\n"
         "    const unsigned int sub = (unsigned int)(tid % GS_LANES);\n"
         "    // results of a round change lanes through LDS: one 16-byte write per lane, one read per result (a block is one wave:\n"
         "    // the barriers only order the accesses)\n"
         "    __shared__ fe gs_swap[64];\n"
         "    const unsigned int gs_group = threadIdx.x & ~(GS_LANES - 1u);\n"
         "    const unsigned long long steps = segments * seglen;\n";
    for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "    fe r%u = first_rows[g * %uull + %uull], n%u;\n", r, registers, r, r); s += buf; }
    std::vector<bool> const_declared(gen.nconsts, false);
    std::string hoisted, init_body, main_body;
    if (init_ninstr) ssa_emit(init_body, hoisted, const_declared, init_nodes, gen, soff, slen, "0ull", "n", "u");
    ssa_emit(main_body, hoisted, const_declared, main_nodes, gen, soff, slen, "i", "n", "v");
    s += hoisted;
    if (init_ninstr) {
        s += "    {\n";
        for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        n%u = r%u;\n", r, r); s += buf; }
        s += init_body;
        for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        r%u = n%u;\n", r, r); s += buf; }
        s += "    }\n";
    }
    s += "    for (unsigned long long k = 0; k < seglen; k++) {\n"
         "        const unsigned long long i = g * seglen + k;\n";
    for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        if (gs_live && sub == %uu) out[%uull * steps + i] = r%u;\n", r % gen.lanes, r, r); s += buf; }
    s += "        if (k + 1 == seglen) break;\n";
    for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        n%u = r%u;\n", r, r); s += buf; }
    s += main_body;
    for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        r%u = n%u;\n", r, r); s += buf; }
    s += "    }\n}\n";
    return true;
}

int gs_jit_trace_segments(gs_ctx *c, const uint32_t *code, uint32_t ninstr, const uint32_t *icode, uint32_t init_ninstr, const uint8_t *consts_host,
                          uint32_t nconsts, uint32_t vm_regs, uint32_t registers, const uint64_t *soff, const uint64_t *slen, const uint8_t *statics_host,
                          uint32_t nstatic, const fe *dconst, const fe *dstat, const fe *drows, uint64_t segments, uint64_t seglen, fe *out) {
    JitGen gen;
    gen.consts = consts_host;
    gen.nconsts = nconsts;
    gen.statics = statics_host;
    gen.soff = soff;
    gen.slen = slen;
    gen.nstatic = nstatic;
    std::string key("T");
    key_add(key, code, (size_t)ninstr * 16);
    key_add(key, icode, (size_t)init_ninstr * 16);
    key_add(key, consts_host, (size_t)nconsts * GS_ELT);
    key_add(key, soff, sizeof(uint64_t) * GS_AIR_MAX_REGISTERS);
    key_add(key, slen, sizeof(uint64_t) * GS_AIR_MAX_REGISTERS);
    const uint32_t shape[3] = {vm_regs, registers, nstatic};
    key_add(key, shape, sizeof shape);
    for (uint32_t r = 0; r < nstatic; r++) key.push_back(gen.static_is_binary(r) ? '1' : '0');
    const std::shared_ptr<const JitSource> src = jit_source_memo(key, [&](JitSource &js) {
        js.ok = jit_trace_source(js.source, gen, code, ninstr, icode, init_ninstr, vm_regs, registers, soff, slen);
        js.lanes = gen.lanes;
    });
    if (!src->ok) return GS_ERR_UNSUPPORTED;
    gen.lanes = src->lanes;
    const std::string &s = src->source;
    JitKernel *k = jit_get(c, s, "gs_jit_trace");
    if (!k) return GS_ERR_UNSUPPORTED;
    unsigned long long a_segments = segments, a_seglen = seglen;
    void *args[] = {(void *)&dconst, (void *)&dstat, (void *)&drows, (void *)&a_segments, (void *)&a_seglen, (void *)&out};
    constexpr unsigned block = 64;
    static_assert(block == 64, "gs_jit_trace: gs_swap[64], __launch_bounds__(64) and the lane groups assume one wave64 per block");
    const unsigned grid = (unsigned)((segments * gen.lanes + block - 1) / block);
    gs_traffic(c, (uint64_t)registers * segments * (seglen + 1) * GS_ELT, (uint64_t)registers * segments * seglen, "gs_jit_trace");      // first rows in, the trace out
    if (hipModuleLaunchKernel(k->fn, grid, 1, 1, block, 1, 1, 0, c->stream, args, nullptr) != hipSuccess) return GS_ERR_UNSUPPORTED;
    c->jit_launches++;
    return GS_OK;
}

// ---- constraints -----------------------------------------------------------------------------------------------------------------
// A constraint program whose linear layers fuse (ssa_fuse_dots) is emitted from its SSA form, node by node: one thread per domain
// point, no lanes.  A fused row is K x 25 v_mad into nine shared columns and ONE fold (gs_dot9: the constants are lane-uniform, their
// limbs come out of scalar loads) instead of K products of 84 instructions and K - 1 additions: the 36 MDS products of a Poseidon
// round are 6 x ~300 instructions instead of 36 x 84 + 30 x 12.
static bool ssa_emit_flat(std::string &s, const std::vector<SsaNode> &nodes, const JitGen &gen, const uint64_t *soff, const uint64_t *slen) {
    char buf[256];
    auto name = [&](int id) {
        const SsaNode &n = nodes[id];
        char t[48];
        if (n.kind == SsaNode::ZERO) return std::string("fe_zero()");
        if (n.kind == SsaNode::CONSTV) {
            if (n.a < 0) return std::string("fe_one()");
            snprintf(t, sizeof t, "consts[%d]", n.a);
            return std::string(t);
        }
        snprintf(t, sizeof t, "v%d", id);
        return std::string(t);
    };
    for (int id = 0; id < (int)nodes.size(); id++) {
        const SsaNode &n = nodes[id];
        switch (n.kind) {
            case SsaNode::ROW: snprintf(buf, sizeof buf, "        const fe v%d = p[%dull * prow + jp];\n", id, n.a); s += buf; break;
            case SsaNode::ROWN: snprintf(buf, sizeof buf, "        const fe v%d = p[%dull * prow + jnp];\n", id, n.a); s += buf; break;
            case SsaNode::STATICV:
                if (slen[n.a] & (slen[n.a] - 1)) snprintf(buf, sizeof buf, "        const fe v%d = statics[%lluull + j %% %lluull];\n", id, (unsigned long long)soff[n.a], (unsigned long long)slen[n.a]);
                else snprintf(buf, sizeof buf, "        const fe v%d = statics[%lluull + (j & %lluull)];\n", id, (unsigned long long)soff[n.a], (unsigned long long)(slen[n.a] - 1));
                s += buf;
                break;
            case SsaNode::ADD: s += "        const fe " + name(id) + " = fe_add(" + name(n.a) + ", " + name(n.b) + ");\n"; break;
            case SsaNode::SUB: s += "        const fe " + name(id) + " = fe_sub(" + name(n.a) + ", " + name(n.b) + ");\n"; break;
            case SsaNode::MUL: s += "        const fe " + name(id) + " = gs_mul(" + name(n.a) + ", " + name(n.b) + ");\n"; break;
            case SsaNode::POWSHORT: case SsaNode::POWLONG: {
                std::vector<uint32_t> e(GS_ELT / 4, 0u);
                if (n.kind == SsaNode::POWSHORT) e[0] = n.aux;
                else memcpy(e.data(), gen.consts + (size_t)n.aux * GS_ELT, GS_ELT);
                s += "        fe " + name(id) + " = " + name(n.a) + ";\n        {\n";
                emit_pow(s, name(id).c_str(), e);
                s += "        }\n";
                break;
            }
            case SsaNode::DOT:
                s += "        fe " + name(id) + ";\n        {\n            gs_dot9_t acc;\n            gs_dot9_begin(acc);\n";
                for (size_t k = 0; k < n.tx.size(); k++) {
                    if (k && k % 6 == 0) s += "            gs_dot9_flush(acc);\n";
                    if (n.tc[k] < 0) s += "            gs_dot9_acc(acc, " + name(n.tx[k]) + ", fe_one());\n";
                    else { snprintf(buf, sizeof buf, ", consts[%d]);\n", n.tc[k]); s += "            gs_dot9_acc(acc, " + name(n.tx[k]) + buf; }
                }
                s += "            " + name(id) + " = gs_dot9_end(acc);\n        }\n";
                break;
            case SsaNode::OUT: snprintf(buf, sizeof buf, "        out[%uull * nc + j] = ", n.aux); s += buf + name(n.a) + ";\n"; break;
            default: break;      // CONSTV / ZERO are named in place, DEAD nodes were absorbed
        }
    }
    return true;
}

static bool jit_constraints_source(std::string &s, const JitGen &gen, const uint32_t *code, uint32_t ninstr, uint32_t vm_regs, const uint64_t *soff,
                                   const uint64_t *slen) {
    s = jit_preamble();
    char buf[256];
    // register r at point j is p[r * prow + j * pstride]: the columns may be those of a larger domain read with a stride (the composition
    // domain inside the evaluation domain: gs_air_constraints_strided) — no plucked copy of the trace extension
    s += "extern \"C\" __global__ __launch_bounds__(128) void gs_jit_constraints(const fe *__restrict__ consts, const fe *__restrict__ p, unsigned long long nc,\n"
         "                                               unsigned long long shift, const fe *__restrict__ statics, fe *__restrict__ out,\n"
         "                                               unsigned long long prow, unsigned long long pstride) {\n";
    for (uint32_t t = 0; t < vm_regs; t++) { snprintf(buf, sizeof buf, "    fe t%u;\n", t); s += buf; }
    s += "    for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < nc; j += (unsigned long long)gridDim.x * blockDim.x) {\n"
         "        unsigned long long jn = j + shift;\n"
         "        if (jn >= nc) jn -= nc;\n"
         "        const unsigned long long jp = j * pstride, jnp = jn * pstride;\n";
    {   // linear layers present: the SSA route with fused rows (everything else: the register-by-register generator below)
        std::vector<SsaNode> nodes;
        if (ssa_build(nodes, gen, code, ninstr, vm_regs, true, true)) {
            ssa_fuse_dots(nodes);
            bool any = false;
            for (const SsaNode &n : nodes) any |= n.kind == SsaNode::DOT;
            std::string body;
            if (any && ssa_emit_flat(body, nodes, gen, soff, slen)) {
                s += body + "    }\n}\n";
                return true;
            }
        }
    }
    // loads of trace registers and outputs are memory operations here: rewrite them on the fly
    std::vector<uint32_t> tmp(code, code + 4 * (size_t)ninstr);
    std::string body;
    for (uint32_t pc = 0; pc < ninstr; pc++) {
        const uint32_t op = tmp[4 * pc], d = tmp[4 * pc + 1], a = tmp[4 * pc + 2];
        if (op == J_LOADR) { snprintf(buf, sizeof buf, "        t%u = p[%uull * prow + jp];\n", d, a); body += buf; }
        else if (op == J_LOADN) { snprintf(buf, sizeof buf, "        t%u = p[%uull * prow + jnp];\n", d, a); body += buf; }
        else if (op == J_OUT) { snprintf(buf, sizeof buf, "        out[%uull * nc + j] = t%u;\n", d, a); body += buf; }
        else {
            // runs of arithmetic go through the common generator (so that exponentiation groups are found)
            uint32_t end = pc;
            while (end < ninstr && tmp[4 * end] != J_LOADR && tmp[4 * end] != J_LOADN && tmp[4 * end] != J_OUT) end++;
            if (!jit_body(body, gen, tmp.data() + 4 * pc, end - pc, soff, slen, true, "r", nullptr, "j", "n")) return false;
            pc = end - 1;
        }
    }
    s += body;
    s += "    }\n}\n";
    return true;
}

int gs_jit_constraints(gs_ctx *c, const uint32_t *code, uint32_t ninstr, const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs,
                       uint32_t registers, const uint64_t *soff, const uint64_t *slen, const fe *dconst, const fe *p, uint64_t nc, uint64_t shift,
                       const fe *statics, fe *out, uint64_t prow, uint64_t pstride) {
    JitGen gen;
    gen.consts = consts_host;
    gen.nconsts = nconsts;
    std::string key("C");
    key_add(key, code, (size_t)ninstr * 16);
    key_add(key, consts_host, (size_t)nconsts * GS_ELT);
    key_add(key, soff, sizeof(uint64_t) * GS_AIR_MAX_REGISTERS);
    key_add(key, slen, sizeof(uint64_t) * GS_AIR_MAX_REGISTERS);
    const uint32_t shape[2] = {vm_regs, registers};
    key_add(key, shape, sizeof shape);
    const std::shared_ptr<const JitSource> src = jit_source_memo(key, [&](JitSource &js) { js.ok = jit_constraints_source(js.source, gen, code, ninstr, vm_regs, soff, slen); });
    if (!src->ok) return GS_ERR_UNSUPPORTED;
    const std::string &s = src->source;
    JitKernel *k = jit_get(c, s, "gs_jit_constraints");
    if (!k) return GS_ERR_UNSUPPORTED;
    unsigned long long a_nc = nc, a_shift = shift, a_prow = prow, a_pstride = pstride;
    void *args[] = {(void *)&dconst, (void *)&p, (void *)&a_nc, (void *)&a_shift, (void *)&statics, (void *)&out, (void *)&a_prow, (void *)&a_pstride};
    const unsigned block = 128, grid = gs_grid(nc, block, 256 * 16);
    {   // every register read at nc points (the neighbour row is the same vector), one vector per constraint written
        uint64_t outs = 0;
        for (uint32_t pc = 0; pc < ninstr; pc++) outs += code[4 * pc] == J_OUT;
        gs_traffic(c, nc * ((uint64_t)registers + outs) * GS_ELT, nc * outs, "gs_jit_constraints");
    }
    if (hipModuleLaunchKernel(k->fn, grid, 1, 1, block, 1, 1, 0, c->stream, args, nullptr) != hipSuccess) return GS_ERR_UNSUPPORTED;
    c->jit_launches++;
    return GS_OK;
}

// Compile-only check (no device, no context): does the generated source for this program build for gfx950?  kind 0: trace segments
// (code + optional init program), kind 1: constraints.  The "does it build" tier of the tests runs it on machines without a GPU.
// Test hook (not part of include/gstark.h): where the code object of `source` would be cached under the current environment — the
// empty string when the cache is disabled or the directory is not private to this user.  Creates the directory like a real build.
// test hook: one program through the helper process, synchronously.  Returns jit_compile_helper's verdict; *size = bytes of code object
// (or of the compiler's log), the first min(cap, *size) of which are copied to out
extern "C" int gs_jit_helper_probe(const char *source, const char *entry, const char *cache_path, char *out, uint64_t cap, uint64_t *size) {
    std::vector<char> code;
    std::string log;
    const int r = jit_compile_helper(source, entry, cache_path ? cache_path : "", code, log);
    const char *src = r == 1 ? code.data() : log.data();
    const uint64_t n = r == 1 ? code.size() : log.size();
    if (size) *size = n;
    if (out && cap) memcpy(out, src, n < cap ? n : cap);
    return r;
}
extern "C" int gs_jit_cache_path_probe(const char *source, char *out, uint64_t cap) {
    if (!source || !out || !cap) return GS_ERR_ARG;
    const std::string path = jit_cache_path(source, "gs_jit_probe");
    snprintf(out, (size_t)cap, "%s", path.c_str());
    return GS_OK;
}

// gs_air_jit_check with the tables of the static registers (trace programs: products by 0/1 selectors become selects, so the source
// a prover compiles depends on them); not part of the ABI — a hook of this library for AirObject.compileCheck and the CPU test tier
extern "C" int gs_air_jit_check_statics(int kind, const uint32_t *code, uint32_t ninstr, const uint32_t *init_code, uint32_t init_ninstr,
                                        const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs, uint32_t registers, const uint64_t *static_lens,
                                        uint32_t nstatic, const uint8_t *static_values_host, char *log_out, uint64_t log_cap) {
    if (!code || !ninstr || nstatic > GS_AIR_MAX_REGISTERS || (nstatic && !static_lens) || (nconsts && !consts_host)) return GS_ERR_ARG;
    JitGen gen;
    gen.consts = consts_host;
    gen.nconsts = nconsts;
    uint64_t soff[GS_AIR_MAX_REGISTERS], slen[GS_AIR_MAX_REGISTERS], off = 0;
    for (uint32_t s = 0; s < GS_AIR_MAX_REGISTERS; s++) {
        soff[s] = off;
        slen[s] = s < nstatic ? static_lens[s] : 1;
        if (s < nstatic) off += static_lens[s];
    }
    if (kind == 0 && static_values_host) { gen.statics = static_values_host; gen.soff = soff; gen.slen = slen; gen.nstatic = nstatic; }
    std::string src, log;
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = kind == 0 ? jit_trace_source(src, gen, code, ninstr, init_code, init_ninstr, vm_regs, registers, soff, slen)
                              : jit_constraints_source(src, gen, code, ninstr, vm_regs, soff, slen);
    if (getenv("GSTARK_AIR_JIT_VERBOSE"))
        fprintf(stderr, "[gstark] source of the %s program generated in %.1f us (%zu bytes)\n", kind == 0 ? "trace" : "constraint",
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), src.size());
    std::vector<char> obj;
    const bool built = ok && jit_compile(src, kind == 0 ? "gs_jit_trace" : "gs_jit_constraints", obj, log);
    if (log_out && log_cap) snprintf(log_out, log_cap, "%s", !ok ? "the program has an instruction the generator does not know" : log.c_str());
    return built ? GS_OK : GS_ERR_UNSUPPORTED;
}
extern "C" int gs_air_jit_check(int kind, const uint32_t *code, uint32_t ninstr, const uint32_t *init_code, uint32_t init_ninstr,
                                const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs, uint32_t registers, const uint64_t *static_lens,
                                uint32_t nstatic, char *log_out, uint64_t log_cap) {
    return gs_air_jit_check_statics(kind, code, ninstr, init_code, init_ninstr, consts_host, nconsts, vm_regs, registers, static_lens, nstatic, nullptr, log_out, log_cap);
}
