// ntt.hip — forward / inverse number-theoretic transform over GF(p) for gfx950.
//
// Replaces FiniteField.evalPolyAtRoots / evalPolysAtRoots / interpolateRoots of @guildofweavers/galois
// (call sites: lib/Stark.ts:106,109; lib/components/CompositionPolynomial.ts:109-110;
// lib/components/BoundaryConstraints.ts:87-88).  Natural order in, natural order out.
//
// Structure (designed for CDNA4, not translated from anything):
//   * Stockham autosort decomposition n = R_0 * R_1 * ... with 2..3 HBM passes for n up to 2^24
//     (radix R_i = 2^L_i, 16 <= R_i <= 256).  Every pass reads and writes each element once with
//     128-bit (dwordx4) accesses in >= 256-byte contiguous segments.
//   * Inside a pass a 256-thread workgroup owns a tile of R x Wj elements (R*Wj = 4096 = 64 KiB).
//     Each thread keeps 16 elements (64 VGPRs) in registers and runs a fully unrolled radix-16
//     butterfly network on them (stage A), the tile is exchanged once through LDS (conflict-free:
//     consecutive lanes touch consecutive 16-byte slots), and a radix-RB network (RB = R/16) finishes
//     the pass (stage B).  8 butterfly levels per HBM round trip, one LDS round trip.
//   * Twiddles are never read from a domain vector: inter-pass twiddles come from a two-level
//     power table of omega (2 x <= 4096 entries, L2 resident) + a running product, the in-register
//     radix-16 twiddles sit in SGPRs (kernel arguments).
//   * Zero-extension (low-degree extension of a T-coefficient polynomial to n points) is folded
//     into the first pass: out-of-range loads are predicated off, nothing is padded in memory.
//   * The inverse transform is the same kernel on omega^-1 with the 1/n scale fused into the last
//     pass's stores.
#include <stdlib.h>

#include "common.h"

#if !defined(GS_SMALL_Q) && !defined(GS_WIDE_BITS)
#define GS_NTT_LAZY 1      // the 128-bit field of the hot path: butterflies in five-limb lazy form (gf128_lazy.h)
#include "gf128_lazy.h"
struct alignas(32) lz8 { int32_t l[8]; };   // an NN element as a table entry: five limbs, 32-byte stride
#endif

struct NttPlan {
    fe omega;
    uint64_t n = 0;
    int logn = 0, log_lo = 0;
    fe *tw_lo = nullptr;  // omega^i, i < 2^log_lo
    fe *tw_hi = nullptr;  // omega^(i << log_lo), i < n >> log_lo
    int npass = 0;
    int L[4] = {0, 0, 0, 0};
    fe *wR[4] = {nullptr, nullptr, nullptr, nullptr};  // (omega^(n/R))^i, i < R, per pass
    fe *twp[4] = {nullptr, nullptr, nullptr, nullptr}; // inter-pass twiddles omega_{Ns*R}^(jq*k) as a [k][jq] table (passes >= 1, when small)
    fe w16[8];                                         // omega_16^i
    fe *inv_table = nullptr;                           // 1 / (omega^j - 1), j < n, [0] = 0 (built on first use: gs_plan_inverse_table)
    std::map<std::string, fe *> inv_table_shifted;     // 1 / (shift * omega^j - 1) per shift != 1 (gs_plan_inverse_table_shifted: a rank's coset)
#ifdef GS_NTT_LAZY
    lzw *wtab = nullptr;                               // device: W-forms of omega_16^1..7 and of 1/n (read with scalar loads)
    lz8 *wRz[4] = {nullptr, nullptr, nullptr, nullptr};   // wR[i] * 2^130 as NN limbs (signed digits): the multipliers of lz_mul_vm
    fe *twpR[4] = {nullptr, nullptr, nullptr, nullptr};   // twp[i] * 2^130: per-lane twiddles go through the Montgomery product (gf128_lazy.h)
    fe *tw_loR = nullptr;                                 // tw_lo * 2^130 (start value of the running-product twiddles)
    lz8 *wRz_scaled = nullptr;                         // the last pass's table times 1/n (inverse transforms: the scale rides on the exchange twiddle)
#ifdef GS_NTT_EXPERIMENTS
    int4 *mf_tab = nullptr;                            // matrix-core passes (tools/ntt_mfma.h): the 4 KB operand table of omega_16
    fe *mf_wR_scaled = nullptr;                        // omega_256^e / n
    int mf_offs[16];
    fe mf_bias0;
#endif
#endif
};

struct PassArgs {
    uint64_t n, in_len, in_stride, out_stride;
    int logn, logNs, logWj, log_lo, scale;
    const fe *tw_lo, *tw_hi, *wR, *twp;
    fe w16[8];
    fe ninv;
};

int gs_power_series_dev(gs_ctx *c, const fe &base, uint64_t n, fe *out);  // pointwise.hip

__device__ __forceinline__ fe pow_lookup(const fe *__restrict__ tw_lo, const fe *__restrict__ tw_hi, int log_lo, int logn,
                                         uint64_t e) {
    fe x = tw_lo[e & ((1ull << log_lo) - 1)];
    if (logn > log_lo) x = fe_mul(x, tw_hi[e >> log_lo]);  // wave-uniform
    return x;
}

__host__ __device__ constexpr int brev(int i, int bits) {
    int r = 0;
    for (int b = 0; b < bits; b++) r |= ((i >> b) & 1) << (bits - 1 - b);
    return r;
}

// Fully unrolled decimation-in-frequency network on N = 2^LOGN registers.  w16[i] = omega_16^i.
// Result is bit-reversed: X[q] = x[brev(q, LOGN)].
template <int LOGN>
__device__ __forceinline__ void ntt_dif_reg(fe (&x)[1 << LOGN], const fe (&w16)[8]) {
    constexpr int N = 1 << LOGN;
#pragma unroll
    for (int s = N / 2; s >= 1; s >>= 1) {
#pragma unroll
        for (int b = 0; b < N; b += 2 * s) {
#pragma unroll
            for (int i = 0; i < s; i++) {
                fe u = x[b + i], v = x[b + i + s];
                x[b + i] = fe_add(u, v);
                fe d = fe_sub(u, v);
                const int tw = i * (N / 2 / s) * (16 / N);
                x[b + i + s] = tw ? fe_mul(d, w16[tw]) : d;
            }
        }
    }
}


#ifdef GS_NTT_LAZY
// ======================================================================================================================
// The pass kernel of the 128-bit field, butterflies in the lazy five-limb form of gf128_lazy.h.
//
// Same decomposition, indexing and LDS exchange as k_ntt_pass below (which stays for the other field flavours); what changes
// is the arithmetic and therefore the shape of a workgroup:
//   * elements are unpacked to 5 x 26-bit limbs right after the 16-byte load and packed (canonical) right before the store;
//     add/sub inside the networks are 5 plain 32-bit operations, no carries, no reduction;
//   * the radix-16 twiddles are kernel arguments in W-form (25 SGPR words each): lz_mul_u, no high columns;
//   * per-lane twiddles (inter-pass table / running product, exchange table) go through lz_mul_v;
//   * a thread still owns 16 elements (80 VGPRs); a workgroup is 128 threads on a tile of 2048 elements, the exchange buffer
//     holds the five limb planes (40 KiB: four workgroups = eight waves per CU), bank-conflict free on both sides
//     (consecutive lanes -> consecutive words on the write side, an XOR swizzle of the bank bits on the read side).
struct LzPassArgs {
    uint64_t n, in_len, in_stride, out_stride;
    int logn, logNs, logWj, log_lo;
    int scale;      // multiply every output by ninv (inverse transform, passes without an exchange stage)
    int exq0;       // the exchange table carries a common factor (1/n): output qa = 0 takes wR[0] as well
    int weak;       // not the last pass: store any representative below 2^128 (lz_pack_weak), the next pass unpacks it
    const fe *tw_lo, *tw_hi, *twp;      // twp: the [k][jq] table times 2^130 (lz_mul_vm)
    const fe *tw_loR;                   // tw_lo times 2^130
    const lz8 *wR;                      // exchange twiddles times 2^130 (times 1/n on the last pass of an inverse transform)
    const lzw *wtab;   // W-forms: [0..6] omega_16^1..7, [7] 1/n.  Read with scalar loads right before each use (see LZ_FENCE)
};

// The W-form of a lane-uniform multiplier is 25 SGPR words.  Left alone, the scheduler hoists the scalar loads of all seven
// radix-16 twiddles to the top of the unrolled network (175 live SGPRs -> spills to VGPR lanes, one v_readlane per use);
// a compiler-only memory fence before each product keeps every load next to its use.
#define LZ_FENCE() asm volatile("" ::: "memory")
// makes a value opaque where it stands (no instruction): it is computed before this point and cannot be rematerialised after it
#define LZ_PIN(x) asm volatile("" : "+v"((x).l[0]), "+v"((x).l[1]), "+v"((x).l[2]), "+v"((x).l[3]), "+v"((x).l[4]))
// ... and read through the constant address space: a lane-uniform load from it is always a scalar load, whatever stores and
// fences surround it (the table is written once, when the plan is made, long before any kernel that reads it is launched)
typedef const lzw *lzw_cptr;
__device__ __forceinline__ lzw lz_load_w(const lzw *p) {
    lzw W;
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) int32_t *q = (const __attribute__((address_space(4))) int32_t *)(const int32_t *)p;
    // 8 + 8 + 8 + 1 words: narrower scalar loads than the dwordx16 the compiler would merge these into leave the SGPR allocator
    // room (a dwordx16 destination needs 16 consecutive, aligned registers; two multipliers are live at a time)
#pragma unroll
    for (int g = 0; g < 25; g += 8) {
#pragma unroll
        for (int e = g; e < g + 8 && e < 25; e++) W.w[e / 5][e % 5] = q[e];
        asm volatile("" ::: "memory");
    }
#else
    W = *p;
#endif
    return W;
}

__device__ __forceinline__ lz lz_load8(const lz8 *__restrict__ p) {
    const int4 a = *reinterpret_cast<const int4 *>(p);
    lz r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = p->l[4];
    return r;
}

__device__ __forceinline__ lz lz_pow_lookup(const fe *__restrict__ tw_lo, const fe *__restrict__ tw_hi, int log_lo, int logn, uint64_t e, const lzk &K) {
    lz x = lz_unpack(tw_lo[e & ((1ull << log_lo) - 1)]);
    if (logn > log_lo) x = lz_mul_v(x, lz_unpack(tw_hi[e >> log_lo]), K);  // wave-uniform
    return x;
}

// the same power times 2^130 (a multiplier of lz_mul_vm): the low table is stored premultiplied, the ordinary product keeps the factor
__device__ __forceinline__ lz lz_pow_lookup_r(const fe *__restrict__ tw_loR, const fe *__restrict__ tw_hi, int log_lo, int logn, uint64_t e, const lzk &K) {
    lz x = lz_unpack(tw_loR[e & ((1ull << log_lo) - 1)]);
    if (logn > log_lo) x = lz_mul_v(x, lz_unpack(tw_hi[e >> log_lo]), K);  // wave-uniform
    return x;
}

// decimation in frequency on N = 2^LOGN lazy registers, w[i - 1] = W-form of omega_16^i; result bit-reversed.
// Inputs NN; a product is only ever applied to a difference (at most 2^3 NN values apart); outputs are sums of up to N NN values.
template <int LOGN, int S>
__device__ __forceinline__ void ntt_dif_lz_level(lz (&x)[1 << LOGN], lzw_cptr w, const lzk &K) {
    constexpr int N = 1 << LOGN;
    constexpr int unit = (N / 2 / S) * (16 / N);   // butterfly i of a block takes omega_16^(i * unit)
    // the butterflies of this level: sums in place, differences where the products will find them
#pragma unroll
    for (int b = 0; b < N; b += 2 * S) {
#pragma unroll
        for (int i = 0; i < S; i++) {
            const lz u = x[b + i], v = x[b + i + S];
            x[b + i] = lz_add(u, v);
            x[b + i + S] = lz_sub(u, v);
            // pin both results HERE: left alone, the compiler sinks each subtraction down to the product that consumes it, which keeps
            // the inputs of all the level's butterflies alive beside their sums (~50 extra VGPRs on the twiddled passes: round 2's spills)
            LZ_PIN(x[b + i]);
            LZ_PIN(x[b + i + S]);
        }
    }
    // the products, software-pipelined on the scalar side: the 25 words of the NEXT multiplier are requested before the
    // current product starts, and nothing moves across the end of a product — two multipliers (50 SGPRs) are live, never seven
    if constexpr (S > 1) {
        lzw Wn = lz_load_w(w + (unit - 1));
#pragma unroll
        for (int p = 1; p < N / 2; p++) {
            const int i = p % S;
            if (i == 0) continue;
            const int b = (p / S) * 2 * S;
            const lzw W = Wn;
            int pn = p + 1;
            if (pn % S == 0) pn++;
            if (pn < N / 2) Wn = lz_load_w(w + ((pn % S) * unit - 1));
            x[b + i + S] = lz_mul_u(x[b + i + S], W, K);
            __builtin_amdgcn_sched_barrier(0);
        }
        ntt_dif_lz_level<LOGN, S / 2>(x, w, K);   // one level per instantiation: a single loop over the levels is too big to unroll
    }
}
template <int LOGN>
__device__ __forceinline__ void ntt_dif_lz(lz (&x)[1 << LOGN], lzw_cptr w, const lzk &K) {
    if constexpr (LOGN > 0) ntt_dif_lz_level<LOGN, (1 << LOGN) / 2>(x, w, K);
}

// lz_pack with the canonical / weak choice as a lane-uniform runtime flag, branch-free: the final "- p when >= p" is computed
// either way (8 operations) and taken only when asked for — a branch per element would split the unrolled stage into 32 blocks
__device__ __forceinline__ fe lz_pack_flag(const lz &x, int weak) {
    const fe r = lz_pack_weak(x);
    uint32_t c;
    fe s2;
    s2.w0 = gf_addc(r.w0, 0xFFFFFFFFu, 0u, c);
    s2.w1 = gf_addc(r.w1, 8u, c, c);
    s2.w2 = gf_addc(r.w2, 0u, c, c);
    s2.w3 = gf_addc(r.w3, 0u, c, c);
    const bool take = c && !weak;
    fe o;
    o.w0 = take ? s2.w0 : r.w0;
    o.w1 = take ? s2.w1 : r.w1;
    o.w2 = take ? s2.w2 : r.w2;
    o.w3 = take ? s2.w3 : r.w3;
    return o;
}

// word index of element e in one limb plane of the exchange buffer (see the header comment)
__device__ __forceinline__ int lz_slot(int e, int logWj) {
    if (logWj >= 5 || logWj == 0) return e;
    return e ^ (((e >> (4 + logWj)) & ((32 >> logWj) - 1)) << logWj);
}

// TW: 0 = first pass (logNs == 0: no input twiddle; the tile's output is one contiguous block of R*Wj elements, written through an
// LDS transpose), 1 = input twiddles from the [k][jq] table, 2 = from the power tables + a running product
template <int LB, int TW>
__global__ __launch_bounds__(128, 2) void k_ntt_pass_lz(const fe *__restrict__ in, fe *__restrict__ out, LzPassArgs a) {
    constexpr int RB = 1 << LB, R = 16 * RB, GB = 16 / RB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int32_t *lds = reinterpret_cast<int32_t *>(lds_raw);      // five planes of R*Wj words
    fe *ldsf = reinterpret_cast<fe *>(lds_raw);               // first pass: the output transpose, Wj rows of R + 1 elements

    const lzk K = lzk_make();
    const int t = threadIdx.x;
    const int Wj = 1 << a.logWj;
    const int plane = R << a.logWj;
    const int jj = t & (Wj - 1);
    const int kk = t >> a.logWj;  // < RB
    const uint64_t j = (uint64_t)blockIdx.x * Wj + jj;
    const uint64_t nR = a.n >> (4 + LB);
    const fe *src = in + (uint64_t)blockIdx.y * a.in_stride;
    fe *dst = out + (uint64_t)blockIdx.y * a.out_stride;

    // ---- stage A: 16 strided loads (coalesced across jj), Stockham input twiddle, radix-16 in registers
    lz v[16];
    constexpr bool FIRST = (TW == 0);
    const bool pruned = FIRST && a.in_len <= nR * RB;   // low-degree extension by >= 16x: only m = 0 is in range (first pass, no twiddle)
    if (pruned) {
        const uint64_t idx = j + (uint64_t)kk * nR;
        fe x0 = idx < a.in_len ? src[idx] : fe_zero();
        v[0] = lz_unpack(x0);
#pragma unroll
        for (int m = 1; m < 16; m++) v[m] = v[0];   // the 16-point transform of (v0, 0, ..., 0) is v0 everywhere
    } else {
        fe raw[16];
        if (!FIRST || a.in_len >= a.n) {   // every pass but a zero-extending first one: 16 unconditional loads in flight
#pragma unroll
            for (int m = 0; m < 16; m++) raw[m] = src[j + (uint64_t)(kk + RB * m) * nR];
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const uint64_t idx = j + (uint64_t)(kk + RB * m) * nR;
                raw[m] = src[idx < a.in_len ? idx : 0];          // branch-free: an in-range address, the value dropped below
                if (idx >= a.in_len) raw[m] = fe_zero();
            }
        }
        const uint64_t Ns = 1ull << a.logNs;
        const uint64_t jq = j & (Ns - 1);
        if constexpr (TW == 1) {
            // precomputed [k][jq] table (cache resident)
#pragma unroll
            for (int m = 0; m < 16; m++) {
                if ((m & 3) == 0) { LZ_FENCE(); __builtin_amdgcn_sched_barrier(0); }   // four twiddle loads and four products at a time: the 16 data loads above already hold 64 VGPRs
                v[m] = lz_mul_vm(lz_unpack(raw[m]), lz_unpack(a.twp[((uint64_t)(kk + RB * m) << a.logNs) + jq]), K);
            }
        } else if constexpr (TW == 2) {
            // v[m] *= omega_{Ns*R}^(jq * (kk + RB*m)): start value + running product; the step is put in W-form once
            const uint64_t eu = a.n >> (a.logNs + 4 + LB);
            lz cur = lz_pow_lookup_r(a.tw_loR, a.tw_hi, a.log_lo, a.logn, jq * kk * eu, K);        // times 2^130, and stays so: the step is an ordinary multiplier
            lz row = lz_pow_lookup(a.tw_lo, a.tw_hi, a.log_lo, a.logn, jq * RB * eu, K);
            lzw step;
#pragma unroll
            for (int r = 0; r < 5; r++) {
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    step.w[r][c] = row.l[c];
                    // opaque to the optimizer: with the rows' value ranges visible, hipcc (ROCm 7.2) miscompiles the products below
                    // (tools/lazy_device_check.hip test 3 catches it); table multipliers come from memory and are opaque anyway
                    asm volatile("" : "+v"(step.w[r][c]));
                }
                if (r < 4) row = lz_shift_limb(row, K);
            }
#pragma unroll
            for (int m = 0; m < 16; m++) {
                v[m] = lz_mul_vm(lz_unpack(raw[m]), cur, K);
                if (m < 15) cur = lz_mul_u(cur, step, K);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++) v[m] = lz_unpack(raw[m]);
        }
        ntt_dif_lz<4>(v, (lzw_cptr)a.wtab, K);  // A[qa] = v[brev(qa,4)], sums of up to 16 NN values
    }

    const uint64_t Ns = 1ull << a.logNs;
    const uint64_t jq = j & (Ns - 1);
    const uint64_t jbase = (j - jq) * R + jq;
    constexpr bool first = FIRST;

    if constexpr (RB == 1) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            lz x = v[brev(q, 4)];
            if (a.scale) { LZ_FENCE(); const lzw W = lz_load_w(a.wtab + 7); x = lz_mul_u(lz_norm(x), W, K); }
            const fe y = a.weak ? lz_pack_weak(x) : lz_pack(x);
            if (first) ldsf[jj * (R + 1) + q] = y;
            else dst[jbase + (uint64_t)q * Ns] = y;
        }
        if (!first) return;
    } else {
        // ---- exchange through LDS with the A->B twiddle omega_R^(kk*qa): normalise (the network grew the limbs by four bits),
        //      multiply, store the five limbs into their planes
#pragma unroll
        for (int qa = 0; qa < 16; qa++) {
            if ((qa & 3) == 0) LZ_FENCE();   // same for the exchange twiddles
            // outputs 0 and 8 of the network are sums of all 16 inputs: too large for the product (and output 0 may skip it);
            // the other fourteen are at most 9 NN values apart, which the signed-digit table entries absorb
            lz x = v[brev(qa, 4)];
            if (qa == 0 || qa == 8) x = lz_norm(x);
            if (qa != 0) x = lz_mul_vm(x, lz_load8(a.wR + ((kk * qa) & (R - 1))), K);
            else if (a.exq0) x = lz_mul_vm(x, lz_load8(a.wR), K);
            const int slot = lz_slot((qa * RB + kk) * Wj + jj, a.logWj);
#pragma unroll
            for (int l = 0; l < 5; l++) lds[l * plane + slot] = x.l[l];
        }
        __syncthreads();
        // ---- stage B: GB radix-RB networks per thread (g = kk indexes the group of qa values)
        lz xb[GB][RB];
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const int qa = kk * GB + u;
#pragma unroll
            for (int k2 = 0; k2 < RB; k2++) {
                const int slot = lz_slot((qa * RB + k2) * Wj + jj, a.logWj);
#pragma unroll
                for (int l = 0; l < 5; l++) xb[u][k2].l[l] = lds[l * plane + slot];
            }
        }
        if (first) __syncthreads();  // the exchange buffer is reused for the output transpose below
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const int qa = kk * GB + u;
            ntt_dif_lz<LB>(xb[u], (lzw_cptr)a.wtab, K);
#pragma unroll
            for (int qb = 0; qb < RB; qb++) {
                const fe y = a.weak ? lz_pack_weak(xb[u][brev(qb, LB)]) : lz_pack(xb[u][brev(qb, LB)]);
                const int q = qa + 16 * qb;
                if (first) ldsf[jj * (R + 1) + q] = y;
                else dst[jbase + (uint64_t)q * Ns] = y;
            }
        }
        if (!first) return;
    }
    // ---- first pass only: y[j*R + q] for the tile is contiguous; stream it out of LDS coalesced
    __syncthreads();
    const int T = RB * Wj;  // threads in this block
    fe *tile = dst + (uint64_t)blockIdx.x * Wj * R;
#pragma unroll 4
    for (int e = t; e < R * Wj; e += T) {
        const int ej = e / R, eq = e % R;
        tile[e] = ldsf[ej * (R + 1) + eq];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same pass as ONE WAVE per tile (R x Wj elements with RB * Wj = 64, i.e. 1024 elements): the sixteen threads that exchange
// their radix-16 outputs sit in the same wave, so the exchange needs no workgroup barrier, and it goes through LDS one LIMB PLANE
// at a time (write the 16 words of limb l, read back the 16 words of limb l of the other side; an in-order LDS queue per wave
// makes that safe): 4 KiB of LDS per wave instead of 20.  What limits occupancy is then the register file alone: at <= 128 VGPRs
// four waves per SIMD are resident, where the 128-thread kernel above has two (tools/microbench5: a radix-16 network takes
// 2.3 us per wave at four waves per SIMD, 3.0 us at two).
// Global accesses are 64 >> LB elements wide per row (LB = 4: 64 bytes); adjacent tiles are given to workgroups b and b + 8, which
// the dispatcher places on the same XCD one after the other, so the two halves of a 128-byte line meet in that XCD's L2.
// The first pass writes its tile (contiguous in q for RB = 16) straight from registers; first passes of other radices keep the
// workgroup kernel (their stores would be 16-byte pieces).
// the conversions between the 16-byte elements of a pass's input / output and the five-limb form it computes in.  The experiments build
// can replace them by register moves (-DGS_EXP_NO_FORMAT: results are garbage, TIMINGS are what a pass costs without them — the part
// a plan with fewer passes would save one third of; tools/ntt_noformat.sh)
#ifdef GS_EXP_NO_FORMAT
__device__ __forceinline__ lz lz_unpack_fake(const fe &r) { lz x; x.l[0] = (int32_t)(r.w0 & LZ_M); x.l[1] = (int32_t)(r.w1 & LZ_M); x.l[2] = (int32_t)(r.w2 & LZ_M); x.l[3] = (int32_t)(r.w3 & LZ_M); x.l[4] = (int32_t)(r.w3 >> 26); return x; }
__device__ __forceinline__ fe lz_pack_fake(const lz &x) { return fe_make((uint32_t)x.l[0], (uint32_t)x.l[1], (uint32_t)x.l[2], (uint32_t)(x.l[3] ^ (x.l[4] << 26))); }
#define LZ_DATA_UNPACK(r) lz_unpack_fake(r)
#define LZ_DATA_PACK(x, w) lz_pack_fake(x)
#else
#define LZ_DATA_UNPACK(r) lz_unpack(r)
#define LZ_DATA_PACK(x, w) lz_pack_flag(x, w)
#endif
template <int LB, int TW>
__global__ __launch_bounds__(64, 4) void k_ntt_wave(const fe *__restrict__ in, fe *__restrict__ out, LzPassArgs a) {
    constexpr int RB = 1 << LB, R = 16 * RB, GB = 16 / RB, LOGWJ = 6 - LB, Wj = 64 >> LB;
    constexpr bool FIRST = (TW == 0);
    static_assert(!FIRST || LB == 4, "first passes of radix below 256 use k_ntt_pass_lz");
    // one limb plane of the tile.  Element (qa, k2, jj) lives at word g*17*Wj + (u*RB + k2)*Wj + jj with qa = g*GB + u: one padding
    // row of Wj words per reader group g keeps the reads conflict-free (g*17*Wj mod 32 = g*Wj), and both sides address it as
    // ONE per-lane base plus a compile-time offset (no per-element address registers)
    __shared__ int32_t plane[17 * 64];
    const lzk K = lzk_make();
    const int t = threadIdx.x;
    const int jj = t & (Wj - 1);
    const int kk = t >> LOGWJ;  // < RB
    uint32_t tile = blockIdx.x;
    if ((gridDim.x & 15) == 0) tile = (tile & ~15u) | ((tile & 7u) << 1) | ((tile >> 3) & 1u);   // blocks b, b + 8 (same XCD) <-> tiles 2i, 2i + 1
    const uint64_t j = (uint64_t)tile * Wj + jj;
    const uint64_t nR = a.n >> (4 + LB);
    const fe *src = in + (uint64_t)blockIdx.y * a.in_stride;
    const uint64_t Ns = 1ull << a.logNs;
    const uint64_t jq = j & (Ns - 1);

    lz v[16];
    const bool pruned = FIRST && a.in_len <= nR * RB;
    if (pruned) {
        const uint64_t idx = j + (uint64_t)kk * nR;
        const fe x0 = src[idx < a.in_len ? idx : 0];
        v[0] = lz_unpack(x0);
        if (idx >= a.in_len) v[0] = lz_unpack(fe_zero());
#pragma unroll
        for (int m = 1; m < 16; m++) v[m] = v[0];
    } else {
        fe raw[16];
        if (!FIRST || a.in_len >= a.n) {
#pragma unroll
            for (int m = 0; m < 16; m++) raw[m] = src[j + (uint64_t)(kk + RB * m) * nR];
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const uint64_t idx = j + (uint64_t)(kk + RB * m) * nR;
                raw[m] = src[idx < a.in_len ? idx : 0];
                if (idx >= a.in_len) raw[m] = fe_zero();
            }
        }
        if constexpr (TW == 1) {
            // twiddle [k][jq] for k = kk + RB*m, streamed FOUR at a time behind the sixteen data loads: raw[] shrinks by 16 registers per
            // group while v[] grows by 20, so the live set stays under the 128 VGPRs that four waves per SIMD leave (all sixteen
            // twiddles at once cost 64 more registers: three waves and 48-72 bytes of scratch per lane in round 2)
            const fe *tp = a.twp + ((uint64_t)kk << a.logNs) + jq;
            const uint64_t tstep = (uint64_t)RB << a.logNs;
            fe tws[4];
#pragma unroll
            for (int m = 0; m < 16; m++) {
                if ((m & 3) == 0) {
#pragma unroll
                    for (int u = 0; u < 4; u++) { tws[u] = *tp; tp += tstep; }
                }
                __builtin_amdgcn_sched_barrier(0);
                v[m] = lz_mul_vm(LZ_DATA_UNPACK(raw[m]), lz_unpack(tws[m & 3]), K);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (TW == 2) {
            // v[m] *= omega^(jq * (kk + RB*m) * eu): start value + running product.  The step is a per-lane value (jq differs between
            // the lanes of a tile), and its W-form would be 25 VGPRs: a plain five-limb multiplier and lz_mul_v keep the pass at four
            // waves per SIMD with nothing in scratch
            const uint64_t eu = a.n >> (a.logNs + 4 + LB);
            // (both times 2^130: multipliers of the Montgomery product, whose result is the ordinary product — the running value keeps its factor)
            lz cur = lz_pow_lookup_r(a.tw_loR, a.tw_hi, a.log_lo, a.logn, jq * kk * eu, K);
            const lz step = lz_pow_lookup_r(a.tw_loR, a.tw_hi, a.log_lo, a.logn, jq * RB * eu, K);
#pragma unroll
            for (int m = 0; m < 16; m++) {
                __builtin_amdgcn_sched_barrier(0);
                v[m] = lz_mul_vm(LZ_DATA_UNPACK(raw[m]), cur, K);
                if (m < 15) cur = lz_mul_vm(cur, step, K);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++) v[m] = LZ_DATA_UNPACK(raw[m]);
        }
        ntt_dif_lz<4>(v, (lzw_cptr)a.wtab, K);
    }

    // The radix-16 network above needs ~125 of the 128 VGPRs four waves per SIMD leave, and two W-forms (50 SGPRs) at a time:
    // nothing else may live across it, in either register file.  Lane and tile coordinates are therefore derived AGAIN from the
    // thread index here, and every kernel argument the second half uses is read AGAIN from the kernel-argument segment through a
    // pointer the compiler cannot see through (by-value arguments are preloaded at the top and stay live: with ~40 argument
    // words beside the W-forms the scalar file overflowed into v_writelane / v_readlane traffic and, with it, into scratch).
    typedef const __attribute__((address_space(4))) unsigned char *kseg_t;
    typedef const __attribute__((address_space(4))) LzPassArgs *kargs_t;
    kseg_t kp = (kseg_t)__builtin_amdgcn_kernarg_segment_ptr();
    int t2 = threadIdx.x;
    asm volatile("" : "+s"(kp), "+v"(t2) : : "memory");
    // the kernel-argument segment is the argument list laid out like a struct: (const fe *in, fe *out, LzPassArgs a)
    struct KernargMirror { const fe *in; fe *out; LzPassArgs a; };
    static_assert(offsetof(KernargMirror, out) == 8 && offsetof(KernargMirror, a) == 16 && alignof(LzPassArgs) <= 16 &&
                  sizeof(KernargMirror) == 16 + sizeof(LzPassArgs), "k_ntt_wave re-reads its arguments at these offsets: keep them in step with the signature");
    fe *const out2 = *(fe *const __attribute__((address_space(4))) *)(kp + offsetof(KernargMirror, out));
    const kargs_t b = (kargs_t)(kp + offsetof(KernargMirror, a));
    fe *const dst = out2 + (uint64_t)blockIdx.y * b->out_stride;
    const uint64_t Ns2 = 1ull << b->logNs;
    const int weak = b->weak;
    uint32_t tile2 = blockIdx.x;
    if ((gridDim.x & 15) == 0) tile2 = (tile2 & ~15u) | ((tile2 & 7u) << 1) | ((tile2 >> 3) & 1u);
    const int jj2 = t2 & (Wj - 1);
    const int kk2 = t2 >> LOGWJ;
    const uint64_t j2 = (uint64_t)tile2 * Wj + jj2;
    const uint64_t jq2 = j2 & (Ns2 - 1);
    const uint64_t jbase = (j2 - jq2) * R + jq2;
    if constexpr (RB == 1) {
        const int scale = b->scale;
        const lzw_cptr wt = b->wtab;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            lz x = v[brev(q, 4)];
            if (scale) { LZ_FENCE(); const lzw W = lz_load_w(wt + 7); x = lz_mul_u(lz_norm(x), W, K); }
            dst[jbase + (uint64_t)q * Ns2] = LZ_DATA_PACK(x, weak);
        }
    } else {
        // ---- exchange twiddles in place, then the exchange itself, limb plane by limb plane
        const lz8 *const wR = b->wR;
        const int exq0 = b->exq0;
#pragma unroll
        for (int qa = 0; qa < 16; qa++) {
            if ((qa & 1) == 0) { LZ_FENCE(); __builtin_amdgcn_sched_barrier(0); }
            lz x = v[brev(qa, 4)];
            if (qa == 0 || qa == 8) x = lz_norm(x);
            if (qa != 0) x = lz_mul_vm(x, lz_load8(wR + ((kk2 * qa) & (R - 1))), K);
            else if (exq0) x = lz_mul_vm(x, lz_load8(wR), K);
            v[brev(qa, 4)] = x;
        }
        lz xb[GB][RB];
#pragma unroll
        for (int l = 0; l < 5; l++) {
#pragma unroll
            for (int qa = 0; qa < 16; qa++) plane[(qa / GB) * 17 * Wj + (qa % GB) * RB * Wj + t2] = v[brev(qa, 4)].l[l];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < GB; u++)
#pragma unroll
                for (int k2 = 0; k2 < RB; k2++) xb[u][k2].l[l] = plane[kk2 * 17 * Wj + jj2 + (u * RB + k2) * Wj];
            __builtin_amdgcn_wave_barrier();
        }
        const lzw_cptr wt = b->wtab;
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const int qa = kk2 * GB + u;
            ntt_dif_lz<LB>(xb[u], wt, K);
            // output q = qa + 16*qb: ONE running pointer, stepped by 16*Ns elements (16 separate 64-bit addresses would cost 32 VGPRs)
            fe *o = FIRST ? dst + j2 * R + qa : dst + jbase + (uint64_t)qa * Ns2;
            const uint64_t ostep = FIRST ? 16 : (Ns2 << 4);
#pragma unroll
            for (int qb = 0; qb < RB; qb++) {
                *o = LZ_DATA_PACK(xb[u][brev(qb, LB)], weak);
                o += ostep;
                if ((qb & 3) == 3 || qb == RB - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// dst[i] = NN limbs of src[i] * mult (mult = 1: a plain change of layout)
__global__ void k_build_lz_table(const fe *__restrict__ src, lz8 *__restrict__ dst, uint64_t count, fe mult, int use_mult) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        fe x = src[i];
        if (use_mult) x = fe_mul(x, mult);
        lz u = lz_unpack(x);
        // signed digits, |limb| <= 2^25: the exchange stage multiplies sums of up to 9 NN values by these entries without
        // normalising them first (lz_mul_v needs |column| < 2^57: 2^25 * 9 * 2^28.1 is below that, 2^26 * ... is not)
        for (int l = 0; l < 4; l++)
            if (u.l[l] >= (1 << 25)) { u.l[l] -= 1 << 26; u.l[l + 1] += 1; }
        lz8 o;
        for (int l = 0; l < 5; l++) o.l[l] = u.l[l];
        o.l[5] = o.l[6] = o.l[7] = 0;
        dst[i] = o;
    }
}
#endif  // GS_NTT_LAZY

template <int LB>
__global__ __launch_bounds__(256) void k_ntt_pass(const fe *__restrict__ in, fe *__restrict__ out, PassArgs a) {
    constexpr int RB = 1 << LB, R = 16 * RB, GB = 16 / RB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    fe *lds = reinterpret_cast<fe *>(lds_raw);

    const int t = threadIdx.x;
    const int Wj = 1 << a.logWj;
    const int jj = t & (Wj - 1);
    const int kk = t >> a.logWj;  // < RB
    const uint64_t j = (uint64_t)blockIdx.x * Wj + jj;
    const uint64_t nR = a.n >> (4 + LB);
    const fe *src = in + (uint64_t)blockIdx.y * a.in_stride;
    fe *dst = out + (uint64_t)blockIdx.y * a.out_stride;

    // ---- stage A: 16 strided loads (coalesced across jj), Stockham input twiddle, radix-16 in registers
    fe v[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
        uint64_t idx = j + (uint64_t)(kk + RB * m) * nR;
        v[m] = idx < a.in_len ? src[idx] : fe_zero();
    }
    const uint64_t Ns = 1ull << a.logNs;
    const uint64_t jq = j & (Ns - 1);
    if (a.logNs > 0 && a.twp != nullptr) {
        // precomputed [k][jq] table (L2-resident): 16 coalesced loads replace the 15-step running product
#pragma unroll
        for (int m = 0; m < 16; m++) v[m] = fe_mul(v[m], a.twp[((uint64_t)(kk + RB * m) << a.logNs) + jq]);
    } else if (a.logNs > 0) {
        // v[m] *= omega_{Ns*R}^(jq * (kk + RB*m)) : start value + running product
        const uint64_t eu = a.n >> (a.logNs + 4 + LB);
        fe cur = pow_lookup(a.tw_lo, a.tw_hi, a.log_lo, a.logn, jq * kk * eu);
        const fe step = pow_lookup(a.tw_lo, a.tw_hi, a.log_lo, a.logn, jq * RB * eu);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            v[m] = fe_mul(v[m], cur);
            if (m < 15) cur = fe_mul(cur, step);
        }
    }
    // Low-degree extension by >= 16x (in_len <= n/16, e.g. trace polynomials onto the evaluation domain): only m = 0 was in
    // range above, and the 16-point transform of (v0, 0, ..., 0) is v0 everywhere — skip the butterfly network (wave-uniform)
    if (a.in_len <= nR * RB) {
#pragma unroll
        for (int m = 1; m < 16; m++) v[m] = v[0];
    } else {
        ntt_dif_reg<4>(v, a.w16);  // A[qa] = v[brev(qa,4)]
    }

    const uint64_t jbase = (j - jq) * R + jq;
    const bool first = (a.logNs == 0);  // first pass: the tile's output is one contiguous block of R*Wj elements

    if constexpr (RB == 1) {
        if (!first) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                fe x = v[brev(q, 4)];
                if (a.scale) x = fe_mul(x, a.ninv);
                dst[jbase + (uint64_t)q * Ns] = x;
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            fe x = v[brev(q, 4)];
            if (a.scale) x = fe_mul(x, a.ninv);
            lds[jj * (R + 1) + q] = x;
        }
    } else {
        // ---- exchange through LDS with the A->B twiddle omega_R^(kk*qa)
#pragma unroll
        for (int qa = 0; qa < 16; qa++) {
            fe x = v[brev(qa, 4)];
            if (qa != 0 && kk != 0) x = fe_mul(x, a.wR[(kk * qa) & (R - 1)]);
            lds[(qa * RB + kk) * Wj + jj] = x;
        }
        __syncthreads();
        // ---- stage B: GB radix-RB networks per thread (g = kk indexes the group of qa values)
        fe xb[GB][RB];
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const int qa = kk * GB + u;
#pragma unroll
            for (int k2 = 0; k2 < RB; k2++) xb[u][k2] = lds[(qa * RB + k2) * Wj + jj];
        }
        if (first) __syncthreads();  // the exchange buffer is reused for the output transpose below
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const int qa = kk * GB + u;
            ntt_dif_reg<LB>(xb[u], a.w16);
#pragma unroll
            for (int qb = 0; qb < RB; qb++) {
                fe x = xb[u][brev(qb, LB)];
                if (a.scale) x = fe_mul(x, a.ninv);
                const int q = qa + 16 * qb;
                if (first) lds[jj * (R + 1) + q] = x;
                else dst[jbase + (uint64_t)q * Ns] = x;
            }
        }
        if (!first) return;
    }
    // ---- first pass only: y[j*R + q] for the tile is contiguous; stream it out of LDS coalesced
    __syncthreads();
    const int T = RB * Wj;  // threads in this block
    fe *tile = dst + (uint64_t)blockIdx.x * Wj * R;
#pragma unroll 4
    for (int e = t; e < R * Wj; e += T) {
        const int ej = e / R, eq = e % R;
        tile[e] = lds[ej * (R + 1) + eq];
    }
}

// out[r][i] = scale * sum_c in[r][c] * (omega^i)^c  — short polynomials / tiny domains (Horner per point)
__global__ void k_eval_horner(const fe *__restrict__ in, fe *__restrict__ out, uint64_t n, uint64_t in_len, uint64_t in_stride,
                              const fe *__restrict__ tw_lo, const fe *__restrict__ tw_hi, int log_lo, int logn, int scale, fe ninv) {
    const fe *src = in + (uint64_t)blockIdx.y * in_stride;
    fe *dst = out + (uint64_t)blockIdx.y * n;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        fe x = pow_lookup(tw_lo, tw_hi, log_lo, logn, i);
        fe s = fe_zero();
        for (uint64_t c = in_len; c-- > 0;) s = fe_add(fe_mul(s, x), src[c]);
        if (scale) s = fe_mul(s, ninv);
        dst[i] = s;
    }
}

// twp[k * Ns + jq] = omega^(jq * k * eu), eu = n / (Ns * R)
__global__ void k_build_pass_twiddles(fe *__restrict__ out, uint64_t Ns, uint64_t R, uint64_t eu, const fe *__restrict__ tw_lo,
                                      const fe *__restrict__ tw_hi, int log_lo, int logn, fe scale, int use_scale) {
    const uint64_t total = Ns * R;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = t / Ns, jq = t % Ns;
        fe x = pow_lookup(tw_lo, tw_hi, log_lo, logn, jq * k * eu);
        if (use_scale) x = fe_mul(x, scale);
        out[t] = x;
    }
}
__global__ void k_scale_table(const fe *__restrict__ in, fe *__restrict__ out, uint64_t count, fe scale) {
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < count; t += (uint64_t)gridDim.x * blockDim.x) out[t] = fe_mul(in[t], scale);
}

// inter-pass twiddle tables of up to 2^20 entries (16 MiB, L2 resident); larger passes keep the running product.  A [k][jq] table of
// 2^24 entries for the third pass of a 2^24-point transform was measured (round 3): 0.697 ms against 0.670 ms per transform — the
// table read costs what the saved chain products bring (profiles/r03_a_ntt_time_twiddle_log24.txt)
static uint64_t max_pass_twiddle_entries() {
#ifdef GS_NTT_EXPERIMENTS
    static uint64_t v = 0;
    if (!v) { const char *e = getenv("GSTARK_NTT_TWIDDLE_LOG"); v = 1ull << (e ? atoi(e) : 20); }
    return v;
#else
    return 1ull << 20;
#endif
}

static std::string plan_key(const fe &omega, uint64_t n) {
    char buf[16 + 8 * GF_LIMBS + 24];
    int at = 0;
    for (int l = GF_LIMBS - 1; l >= 0; l--) at += snprintf(buf + at, sizeof buf - at, "%08x", fe_limb(omega, l));
    snprintf(buf + at, sizeof buf - at, ":%llu", (unsigned long long)n);
    return buf;
}

static int plan_get(gs_ctx *c, const fe &omega, uint64_t n, NttPlan **out) {
    std::string key = plan_key(omega, n);
    auto it = c->plans.find(key);
    if (it != c->plans.end()) { *out = it->second; return GS_OK; }
    // omega must be a primitive n-th root of unity for the decomposition to hold
    if (n > 1) {
        fe h = fe_pow_u64(omega, n / 2);
        fe m1 = fe_sub(fe_zero(), fe_one());
        if (!fe_eq(h, m1)) return gs_fail(c, GS_ERR_ARG, "ntt: omega is not a primitive %llu-th root of unity", (unsigned long long)n);
    } else if (!fe_eq(omega, fe_one())) return gs_fail(c, GS_ERR_ARG, "ntt: omega must be 1 for n = 1");
    NttPlan *p = new NttPlan();
    p->omega = omega;
    p->n = n;
    p->logn = gs_log2(n);
    p->log_lo = p->logn < 12 ? p->logn : 12;
    int rc;
    void *q = nullptr;
    if ((rc = gs_alloc(c, (1ull << p->log_lo) * GS_ELT, &q))) { delete p; return rc; }
    p->tw_lo = (fe *)q;
    if ((rc = gs_power_series_dev(c, omega, 1ull << p->log_lo, p->tw_lo))) { delete p; return rc; }
    if (p->logn > p->log_lo) {
        uint64_t nhi = n >> p->log_lo;
        if ((rc = gs_alloc(c, nhi * GS_ELT, &q))) { delete p; return rc; }
        p->tw_hi = (fe *)q;
        if ((rc = gs_power_series_dev(c, fe_pow_u64(omega, 1ull << p->log_lo), nhi, p->tw_hi))) { delete p; return rc; }
    } else {
        p->tw_hi = p->tw_lo;
    }
    if (p->logn >= 8) {
        int np = (p->logn + 7) / 8, base = p->logn / np, extra = p->logn % np;
        p->npass = np;
        fe w16 = fe_pow_u64(omega, n / 16), cur = fe_one();
        for (int i = 0; i < 8; i++) { p->w16[i] = cur; cur = fe_mul(cur, w16); }
#ifdef GS_NTT_LAZY
        {
            lzw host[8];
            for (int i = 1; i < 8; i++) lz_wform(p->w16[i], host[i - 1]);
            lz_wform(fe_inv(fe_from_u64(n)), host[7]);
            if ((rc = gs_alloc(c, sizeof host, &q))) { delete p; return rc; }
            p->wtab = (lzw *)q;
            if (hipMemcpyAsync(p->wtab, host, sizeof host, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess) {   // `host` is a stack buffer: the copy must be over before it goes
                delete p;
                return gs_fail(c, GS_ERR_DEVICE, "ntt: twiddle upload failed");
            }
        }
#endif
#ifdef GS_NTT_LAZY
        if ((rc = gs_alloc(c, (1ull << p->log_lo) * GS_ELT, &q))) { delete p; return rc; }
        p->tw_loR = (fe *)q;
        hipLaunchKernelGGL(k_scale_table, dim3(gs_grid(1ull << p->log_lo)), dim3(256), 0, c->stream, p->tw_lo, p->tw_loR, 1ull << p->log_lo, lz_mont_r());
#endif
        uint64_t Ns_acc = 1;
        for (int i = 0; i < np; i++) {
            p->L[i] = base + (i < extra ? 1 : 0);
            uint64_t R = 1ull << p->L[i];
            if (i > 0 && Ns_acc * R <= max_pass_twiddle_entries()) {
#if !defined(GS_NTT_LAZY) || defined(GS_NTT_EXPERIMENTS)
                if ((rc = gs_alloc(c, Ns_acc * R * GS_ELT, &q))) { delete p; return rc; }
                p->twp[i] = (fe *)q;
                hipLaunchKernelGGL(k_build_pass_twiddles, dim3(gs_grid(Ns_acc * R)), dim3(256), 0, c->stream, p->twp[i], Ns_acc, R,
                                   n / (Ns_acc * R), p->tw_lo, p->tw_hi, p->log_lo, p->logn, fe_one(), 0);
#endif
#ifdef GS_NTT_LAZY
                // the lazy kernels multiply by these through lz_mul_vm (x * w * 2^-130): the table holds w * 2^130
                if ((rc = gs_alloc(c, Ns_acc * R * GS_ELT, &q))) { delete p; return rc; }
                p->twpR[i] = (fe *)q;
                hipLaunchKernelGGL(k_build_pass_twiddles, dim3(gs_grid(Ns_acc * R)), dim3(256), 0, c->stream, p->twpR[i], Ns_acc, R,
                                   n / (Ns_acc * R), p->tw_lo, p->tw_hi, p->log_lo, p->logn, lz_mont_r(), 1);
#endif
            }
            Ns_acc *= R;
            if (R > 16) {
                // share tables between passes of equal radix
                for (int k = 0; k < i; k++)
                    if (p->L[k] == p->L[i]) p->wR[i] = p->wR[k];
                if (!p->wR[i]) {
                    if ((rc = gs_alloc(c, R * GS_ELT, &q))) { delete p; return rc; }
                    p->wR[i] = (fe *)q;
                    if ((rc = gs_power_series_dev(c, fe_pow_u64(omega, n / R), R, p->wR[i]))) { delete p; return rc; }
                }
#ifdef GS_NTT_LAZY
                for (int k = 0; k < i; k++)
                    if (p->L[k] == p->L[i]) p->wRz[i] = p->wRz[k];
                if (!p->wRz[i]) {
                    if ((rc = gs_alloc(c, R * sizeof(lz8), &q))) { delete p; return rc; }
                    p->wRz[i] = (lz8 *)q;
                    hipLaunchKernelGGL(k_build_lz_table, dim3(gs_grid(R)), dim3(256), 0, c->stream, p->wR[i], p->wRz[i], R, lz_mont_r(), 1);
                }
#endif
            }
        }
    }
    c->plans[key] = p;
    *out = p;
    return GS_OK;
}

void gs_plans_destroy(gs_ctx *c) {
    for (auto &kv : c->plans) delete kv.second;  // device tables are owned by the block cache
    c->plans.clear();
}

int gs_plan_pow_tables(gs_ctx *c, const fe &omega, uint64_t n, const fe **tw_lo, const fe **tw_hi, int *log_lo) {
    NttPlan *p;
    int rc = plan_get(c, omega, n, &p);
    if (rc) return rc;
    *tw_lo = p->tw_lo;
    *tw_hi = p->tw_hi;
    *log_lo = p->log_lo;
    return GS_OK;
}

__global__ void k_omega_minus_one(const fe *__restrict__ tw_lo, const fe *__restrict__ tw_hi, int log_lo, int logn, uint64_t n,
                                  fe shift, int has_shift, fe *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        fe x = pow_lookup(tw_lo, tw_hi, log_lo, logn, i);
        if (has_shift) x = fe_mul(x, shift);
        out[i] = fe_sub(x, fe_one());
    }
}

int gs_vec_inv_dev(gs_ctx *c, const fe *a, uint64_t n, fe *out);   // pointwise.hip

// u[j] = 1 / (omega^j - 1) over the whole domain, u[0] = 0: a constant of (omega, n) like the twiddle tables, cached with the plan.
// 1 / (omega^i - omega^k) = omega^-k * u[(i - k) mod n]: every division by a polynomial whose roots lie in the domain
// (boundary-constraint denominators) becomes table look-ups instead of a batch inversion per proof.
int gs_plan_inverse_table(gs_ctx *c, const fe &omega, uint64_t n, const fe **u) {
    NttPlan *p;
    int rc = plan_get(c, omega, n, &p);
    if (rc) return rc;
    if (!p->inv_table) {
        void *q = nullptr, *t = nullptr;
        if ((rc = gs_alloc(c, n * GS_ELT, &q))) return rc;
        if ((rc = gs_tmp_alloc(c, n * GS_ELT, &t))) { gs_free(c, q); return rc; }
        hipLaunchKernelGGL(k_omega_minus_one, dim3(gs_grid(n)), dim3(256), 0, c->stream, p->tw_lo, p->tw_hi, p->log_lo, p->logn, n, fe_one(), 0, (fe *)t);
        rc = gs_vec_inv_dev(c, (const fe *)t, n, (fe *)q);
        gs_tmp_free(c, t);
        if (rc) { gs_free(c, q); return rc; }
        p->inv_table = (fe *)q;
    }
    *u = p->inv_table;
    return GS_OK;
}
// u_s[j] = 1 / (shift * omega^j - 1): the same table for the coset {shift * omega^j} (one rank's share of a larger domain; shift != 1
// is not a power of omega there, so no entry is zero); cached per shift with the plan of (omega, n)
int gs_plan_inverse_table_shifted(gs_ctx *c, const fe &omega, uint64_t n, const fe &shift, const fe **u) {
    if (fe_eq(shift, fe_one())) return gs_plan_inverse_table(c, omega, n, u);
    NttPlan *p;
    int rc = plan_get(c, omega, n, &p);
    if (rc) return rc;
    const std::string key = plan_key(shift, 0);
    auto it = p->inv_table_shifted.find(key);
    if (it == p->inv_table_shifted.end()) {
        void *q = nullptr, *t = nullptr;
        if ((rc = gs_alloc(c, n * GS_ELT, &q))) return rc;
        if ((rc = gs_tmp_alloc(c, n * GS_ELT, &t))) { gs_free(c, q); return rc; }
        hipLaunchKernelGGL(k_omega_minus_one, dim3(gs_grid(n)), dim3(256), 0, c->stream, p->tw_lo, p->tw_hi, p->log_lo, p->logn, n, shift, 1, (fe *)t);
        rc = gs_vec_inv_dev(c, (const fe *)t, n, (fe *)q);
        gs_tmp_free(c, t);
        if (rc) { gs_free(c, q); return rc; }
        it = p->inv_table_shifted.emplace(key, (fe *)q).first;
    }
    *u = it->second;
    return GS_OK;
}

template <int LB>
static void launch_pass(gs_ctx *c, const fe *in, fe *out, const PassArgs &a, uint32_t rows) {
    constexpr int RB = 1 << LB, R = 16 * RB;
    const int Wj = 1 << a.logWj;
    const uint64_t tiles = (a.n / R) / Wj;
    const bool need_lds = (RB > 1) || (a.logNs == 0);
    const size_t lds = need_lds ? (size_t)Wj * (R + 1) * GS_ELT : 0;
    // the attribute is per device and the library may serve several contexts / devices from several threads: no process-wide
    // "already set" flag; only tiles above the 64 KiB default need it (the multi-limb flavours)
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_pass<LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_ntt_pass<LB>, dim3((unsigned)tiles, rows), dim3(RB * Wj), lds, c->stream, in, out, a);
}


#ifdef GS_NTT_LAZY
template <int LB>
static void launch_pass_lz(gs_ctx *c, const fe *in, fe *out, const LzPassArgs &a, uint32_t rows) {
    constexpr int RB = 1 << LB, R = 16 * RB;
    const int Wj = 1 << a.logWj;
    const uint64_t tiles = (a.n / R) / Wj;
    const bool first = a.logNs == 0;
    size_t lds = 0;
    if (RB > 1) lds = (size_t)5 * 4 * R * Wj;
    if (first) { const size_t tr = (size_t)Wj * (R + 1) * sizeof(fe); if (tr > lds) lds = tr; }
    // a pass reads every element of its input once (a zero-extending first pass: in_len of them) and writes every element once
    gs_traffic(c, (uint64_t)rows * ((first ? a.in_len : a.n) + a.n) * GS_ELT, (uint64_t)rows * a.n, "k_ntt_pass_lz<%d, %d>", LB, first ? 0 : (a.twp ? 1 : 2));
    if (first) hipLaunchKernelGGL((k_ntt_pass_lz<LB, 0>), dim3((unsigned)tiles, rows), dim3(RB * Wj), lds, c->stream, in, out, a);
    else if (a.twp) hipLaunchKernelGGL((k_ntt_pass_lz<LB, 1>), dim3((unsigned)tiles, rows), dim3(RB * Wj), lds, c->stream, in, out, a);
    else hipLaunchKernelGGL((k_ntt_pass_lz<LB, 2>), dim3((unsigned)tiles, rows), dim3(RB * Wj), lds, c->stream, in, out, a);
}

template <int LB>
static void launch_pass_wave(gs_ctx *c, const fe *in, fe *out, const LzPassArgs &a, uint32_t rows) {
    constexpr int RB = 1 << LB, R = 16 * RB, Wj = 64 >> LB;
    const uint64_t tiles = (a.n / R) / Wj;
    gs_traffic(c, (uint64_t)rows * ((a.logNs == 0 ? a.in_len : a.n) + a.n) * GS_ELT, (uint64_t)rows * a.n, "k_ntt_wave<%d, %d>", LB, a.logNs == 0 ? 0 : (a.twp ? 1 : 2));
    if (a.logNs == 0) {
        if constexpr (LB == 4) hipLaunchKernelGGL((k_ntt_wave<4, 0>), dim3((unsigned)tiles, rows), dim3(64), 0, c->stream, in, out, a);
    } else if (a.twp) hipLaunchKernelGGL((k_ntt_wave<LB, 1>), dim3((unsigned)tiles, rows), dim3(64), 0, c->stream, in, out, a);
    else hipLaunchKernelGGL((k_ntt_wave<LB, 2>), dim3((unsigned)tiles, rows), dim3(64), 0, c->stream, in, out, a);
}
// ---- A/B variants measured in rounds 1-2 and NOT adopted: compiled only into the experiments library
// (tools/build_experiments.sh -> tools/ab/libgstark_hip_exp.so, -DGS_NTT_EXPERIMENTS), where environment switches select them
// per call.  The product library contains one kernel family per field flavour and reads no environment variable here.
#ifdef GS_NTT_EXPERIMENTS
#include "../../tools/ntt_mfma.h"
__global__ void k_mf_scale_table(const fe *__restrict__ in, fe *__restrict__ out, fe k) { out[threadIdx.x] = fe_mul(in[threadIdx.x], k); }
static void launch_pass_mfma(gs_ctx *c, const fe *in, fe *out, const MfPassArgs &a, uint32_t rows) {
    const uint64_t tiles = (a.n >> 8) / MF_COLS;
    static_assert(MF_LDS_BYTES <= 64 * 1024, "below the default dynamic LDS limit: no attribute needed");
    const dim3 grid((unsigned)(tiles / MF_WAVES), rows), block(64 * MF_WAVES);
    if (a.logNs == 0) hipLaunchKernelGGL((k_ntt_mfma<0>), grid, block, MF_LDS_BYTES, c->stream, in, out, a);
    else if (a.twp) hipLaunchKernelGGL((k_ntt_mfma<1>), grid, block, MF_LDS_BYTES, c->stream, in, out, a);
    else hipLaunchKernelGGL((k_ntt_mfma<2>), grid, block, MF_LDS_BYTES, c->stream, in, out, a);
}
static bool ntt_mfma_enabled() {   // GSTARK_NTT_MFMA=1: radix-256 passes on the matrix cores (A/B measurements; read per call)
    const char *e = getenv("GSTARK_NTT_MFMA");
    return e && e[0] == '1';
}

static bool ntt_wave_enabled() {   // GSTARK_NTT_WAVE=0 keeps the 128-thread workgroup kernel everywhere (A/B measurements)
    const char *e = getenv("GSTARK_NTT_WAVE");
    return !(e && e[0] == '0');
}

static bool ntt_lazy_enabled() {   // GSTARK_NTT_LAZY=0 keeps the canonical-limb kernel (A/B measurements)
    const char *e = getenv("GSTARK_NTT_LAZY");   // read per call: tools/ntt_ab.py flips it inside one process
    return !(e && e[0] == '0');
}
#else
static constexpr bool ntt_mfma_enabled() { return false; }
static constexpr bool ntt_wave_enabled() { return true; }
static constexpr bool ntt_lazy_enabled() { return true; }
#endif  // GS_NTT_EXPERIMENTS
#endif

// rows transforms of size n; row r reads in[r*in_stride .. +in_len) (zero-extended) and writes out[r*n .. +n)
static int ntt_run(gs_ctx *c, const fe *in, uint32_t rows, uint64_t in_len, uint64_t in_stride, const fe &omega, uint64_t n,
                   bool inverse, fe *out) {
    if (!gs_is_pow2(n) || in_len > n) return gs_fail(c, GS_ERR_ARG, "ntt: n must be a power of two >= input length");
    if (rows == 0) return GS_OK;
    if (rows > 65535) return gs_fail(c, GS_ERR_ARG, "ntt: at most 65535 rows per call");
    const uint64_t total = (uint64_t)rows * n;
    {   // output must not overlap the input (the passes are out of place)
        const uint8_t *i0 = (const uint8_t *)in, *i1 = i0 + ((uint64_t)(rows - 1) * in_stride + in_len) * GS_ELT;
        const uint8_t *o0 = (const uint8_t *)out, *o1 = o0 + total * GS_ELT;
        if (i0 < o1 && o0 < i1) return gs_fail(c, GS_ERR_ARG, "ntt: output overlaps input");
    }
    // (inversions are 4 us of host time each: remembered per context, a prover asks for the same ones with every proof)
    fe w = inverse ? gs_memo(c, gs_memo_key("inv").add(omega), [&](std::vector<fe> &t) { t.push_back(fe_inv(omega)); })[0] : omega;
    NttPlan *p;
    int rc = plan_get(c, w, n, &p);
    if (rc) return rc;
    fe ninv = inverse ? gs_memo(c, gs_memo_key("inv").add(fe_from_u64(n)), [&](std::vector<fe> &t) { t.push_back(fe_inv(fe_from_u64(n))); })[0] : fe_one();

    if (n < 256 || in_len <= 8) {
        dim3 grid(gs_grid(n, 256, 1024), rows);
        gs_traffic(c, (uint64_t)rows * (in_len + n) * GS_ELT, (uint64_t)rows * n, "k_eval_horner");
        hipLaunchKernelGGL(k_eval_horner, grid, dim3(256), 0, c->stream, in, out, n, in_len, in_stride, p->tw_lo, p->tw_hi, p->log_lo,
                           p->logn, inverse ? 1 : 0, ninv);
        GS_LAUNCH_CHECK(c);
        return GS_OK;
    }

    fe *tmp = nullptr;
    if (p->npass >= 2) {
        void *q;
        if ((rc = gs_tmp_alloc(c, total * GS_ELT, &q))) return rc;
        tmp = (fe *)q;
    }
    // ping-pong so that the last pass lands in `out`:  1: in->out   2: in->tmp->out   3: in->out->tmp->out   4: in->tmp->out->tmp->out
    const fe *src = in;
    int logNs = 0;
    for (int i = 0; i < p->npass; i++) {
        const int remaining = p->npass - 1 - i;
        fe *dst = (remaining % 2 == 0) ? out : tmp;
        const int LB = p->L[i] - 4;
        const uint64_t R = 1ull << p->L[i];
        const bool last = (i == p->npass - 1);
#ifdef GS_NTT_LAZY
        if (ntt_lazy_enabled()) {
            LzPassArgs a;
            a.n = n;
            a.in_len = (i == 0) ? in_len : n;
            a.in_stride = (i == 0) ? in_stride : n;
            a.out_stride = n;
            a.logn = p->logn;
            a.logNs = logNs;
            uint64_t Wj = 128 >> LB;
            if (Wj > n / R) Wj = n / R;
            a.logWj = gs_log2(Wj);
            a.log_lo = p->log_lo;
            a.tw_lo = p->tw_lo;
            a.tw_hi = p->tw_hi;
            a.twp = p->twpR[i];
            a.tw_loR = p->tw_loR;
            a.wR = p->wRz[i];
            a.scale = 0;
            a.exq0 = 0;
            a.weak = last ? 0 : 1;
            if (inverse && last) {
                if (LB > 0) {   // the 1/n scale rides on the exchange twiddles of the last pass
                    if (!p->wRz_scaled) {
                        void *q;
                        if ((rc = gs_alloc(c, R * sizeof(lz8), &q))) { if (tmp) gs_tmp_free(c, tmp); return rc; }
                        p->wRz_scaled = (lz8 *)q;
                        hipLaunchKernelGGL(k_build_lz_table, dim3(gs_grid(R)), dim3(256), 0, c->stream, p->wR[i], p->wRz_scaled, R, fe_mul(ninv, lz_mont_r()), 1);
                    }
                    a.wR = p->wRz_scaled;
                    a.exq0 = 1;
                } else {
                    a.scale = 1;
                }
            }
            a.wtab = p->wtab;
            const bool wave = ntt_wave_enabled() && n >= 1024 && (logNs > 0 || LB == 4);
#ifdef GS_NTT_EXPERIMENTS
            if (LB == 4 && n >= (1ull << 16) && ntt_mfma_enabled()) {
                if (!p->mf_tab) {
                    int8_t host[4096];
                    mf_host_tables(p->w16[1], host, p->mf_offs, p->mf_bias0);
                    void *q;
                    if ((rc = gs_alloc(c, sizeof host, &q))) { if (tmp) gs_tmp_free(c, tmp); return rc; }
                    if (hipMemcpyAsync(q, host, sizeof host, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
                        if (tmp) gs_tmp_free(c, tmp);
                        return gs_fail(c, GS_ERR_DEVICE, "ntt: operand table upload failed");
                    }
                    p->mf_tab = (int4 *)q;
                }
                MfPassArgs m;
                m.n = n; m.in_len = a.in_len; m.in_stride = a.in_stride; m.out_stride = n;
                m.logn = p->logn; m.logNs = logNs; m.log_lo = p->log_lo; m.weak = last ? 0 : 1;
                m.tw_lo = p->tw_lo; m.tw_hi = p->tw_hi; m.twp = p->twp[i]; m.wR = p->wR[i]; m.atab = p->mf_tab;
                for (int k = 0; k < 16; k++) m.offs[k] = p->mf_offs[k];
                m.bias0 = p->mf_bias0;
                if (inverse && last) {
                    if (!p->mf_wR_scaled) {
                        void *q;
                        if ((rc = gs_alloc(c, 256 * GS_ELT, &q))) { if (tmp) gs_tmp_free(c, tmp); return rc; }
                        p->mf_wR_scaled = (fe *)q;
                        hipLaunchKernelGGL(k_mf_scale_table, dim3(1), dim3(256), 0, c->stream, p->wR[i], p->mf_wR_scaled, ninv);
                    }
                    m.wR = p->mf_wR_scaled;
                }
                launch_pass_mfma(c, src, dst, m, rows);
            } else
#endif
            if (wave) {
                a.logWj = 6 - LB;
                switch (LB) {
                    case 0: launch_pass_wave<0>(c, src, dst, a, rows); break;
                    case 1: launch_pass_wave<1>(c, src, dst, a, rows); break;
                    case 2: launch_pass_wave<2>(c, src, dst, a, rows); break;
                    case 3: launch_pass_wave<3>(c, src, dst, a, rows); break;
                    default: launch_pass_wave<4>(c, src, dst, a, rows); break;
                }
            } else
            switch (LB) {
                case 0: launch_pass_lz<0>(c, src, dst, a, rows); break;
                case 1: launch_pass_lz<1>(c, src, dst, a, rows); break;
                case 2: launch_pass_lz<2>(c, src, dst, a, rows); break;
                case 3: launch_pass_lz<3>(c, src, dst, a, rows); break;
                default: launch_pass_lz<4>(c, src, dst, a, rows); break;
            }
        } else
#endif
        {
        PassArgs a;
        a.n = n;
        a.in_len = (i == 0) ? in_len : n;
        a.in_stride = (i == 0) ? in_stride : n;
        a.out_stride = n;
        a.logn = p->logn;
        a.logNs = logNs;
        uint64_t Wj = 256 >> LB;
        if (Wj > n / R) Wj = n / R;
        a.logWj = gs_log2(Wj);
        a.log_lo = p->log_lo;
        a.scale = (inverse && last) ? 1 : 0;
        a.tw_lo = p->tw_lo;
        a.tw_hi = p->tw_hi;
        a.wR = p->wR[i];
        a.twp = p->twp[i];
        for (int k = 0; k < 8; k++) a.w16[k] = p->w16[k];
        a.ninv = ninv;
        switch (LB) {
            case 0: launch_pass<0>(c, src, dst, a, rows); break;
            case 1: launch_pass<1>(c, src, dst, a, rows); break;
            case 2: launch_pass<2>(c, src, dst, a, rows); break;
            case 3: launch_pass<3>(c, src, dst, a, rows); break;
            default: launch_pass<4>(c, src, dst, a, rows); break;
        }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            if (tmp) gs_tmp_free(c, tmp);
            return gs_fail(c, GS_ERR_DEVICE, "ntt pass %d launch: %s", i, hipGetErrorString(e));
        }
        src = dst;
        logNs += p->L[i];
    }
    if (tmp) gs_tmp_free(c, tmp);
    return GS_OK;
}

extern "C" {

int gs_eval_polys_at_roots(gs_ctx *c, const void *polys, uint32_t rows, uint64_t poly_len, const gs_elt *omega, uint64_t n,
                           void *out) {
    if (!c || !polys || !omega || !out) return GS_ERR_ARG;
    return ntt_run(c, (const fe *)polys, rows, poly_len, poly_len, fe_from_bytes(omega), n, false, (fe *)out);
}

int gs_interpolate_roots(gs_ctx *c, const void *ys, uint32_t rows, const gs_elt *omega, uint64_t n, void *out) {
    if (!c || !ys || !omega || !out) return GS_ERR_ARG;
    return ntt_run(c, (const fe *)ys, rows, n, n, fe_from_bytes(omega), n, true, (fe *)out);
}

}  // extern "C"
