// pointwise.hip — streaming vector kernels over GF(p): the FiniteField vector/matrix members that are
// embarrassingly parallel over the domain index.
//
// All of them are HBM-streaming kernels: one 128-bit load/store per element per operand, consecutive
// lanes on consecutive elements, grid capped at 2048 workgroups with a grid-stride loop.
#include "common.h"

// ---- power series: out[i] = base^i -----------------------------------------------------------------
// thread t owns i = t, t + TOT, t + 2*TOT, ...: one exponentiation for base^t, then one multiplication
// by base^TOT per element; stores stay coalesced.
__global__ void k_power_series(fe base, fe step, uint64_t n, uint64_t tot, fe *__restrict__ out) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= tot || t >= n) return;
    fe cur = fe_pow_u64(base, t);
    for (uint64_t i = t; i < n; i += tot) {
        out[i] = cur;
        cur = fe_mul(cur, step);
    }
}

int gs_power_series_dev(gs_ctx *c, const fe &base, uint64_t n, fe *out) {
    if (n == 0) return GS_OK;
    uint64_t tot = n < 65536 ? n : 65536;
    fe step = fe_pow_u64(base, tot);
    unsigned blocks = (unsigned)((tot + 255) / 256);
    gs_traffic(c, n * GS_ELT, n, "k_power_series");
    hipLaunchKernelGGL(k_power_series, dim3(blocks), dim3(256), 0, c->stream, base, step, n, tot, out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

// ---- elementwise ------------------------------------------------------------------------------------
enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2 };

template <int OP>
__device__ __forceinline__ fe apply_op(const fe &x, const fe &y) {
    if (OP == OP_ADD) return fe_add(x, y);
    if (OP == OP_SUB) return fe_sub(x, y);
    return fe_mul(x, y);
}

template <int OP>
__global__ void k_vec_vec(const fe *__restrict__ a, const fe *__restrict__ b, uint64_t n, fe *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = apply_op<OP>(a[i], b[i]);
}
template <int OP>
__global__ void k_vec_scalar(const fe *__restrict__ a, fe s, uint64_t n, fe *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = apply_op<OP>(a[i], s);
}
__global__ void k_vec_exp(const fe *__restrict__ a, fe e, uint64_t n, fe *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = fe_pow(a[i], e);
}

// ---- batch inversion (Montgomery's trick), 0 -> 0 ------------------------------------------------------
// thread t owns the strided subsequence i = t + m*tot (coalesced).  Pass 1 writes running products into
// out[]; the threads of a workgroup then share ONE Fermat inversion: prefix and suffix products of the lanes' totals (shuffles)
// and of the waves' totals (LDS + wave 0) give every thread the inverse of its own total from the inverse of the workgroup's.
// Pass 2 walks back.  `num` (optional) fuses the division out[i] = num[i] * a[i]^-1 (divVectorElements).
__device__ __forceinline__ fe fe_shfl_up(const fe &v, int d) {
    fe o;
#pragma unroll
    for (int l = 0; l < GF_LIMBS; l++) fe_set_limb(o, l, __shfl_up(fe_limb(v, l), d));
    return o;
}
__device__ __forceinline__ fe fe_shfl_down(const fe &v, int d) {
    fe o;
#pragma unroll
    for (int l = 0; l < GF_LIMBS; l++) fe_set_limb(o, l, __shfl_down(fe_limb(v, l), d));
    return o;
}
__device__ __forceinline__ fe fe_shfl(const fe &v, int lane) {
    fe o;
#pragma unroll
    for (int l = 0; l < GF_LIMBS; l++) fe_set_limb(o, l, __shfl(fe_limb(v, l), lane));
    return o;
}

#if !defined(GS_SMALL_Q) && !defined(GS_WIDE_BITS)
#include "gf128_lazy.h"
// a^(p-2) on ONE lane while the rest of the workgroup waits: 127 squarings + 12 products, all dependent.  In the lazy five-limb form a
// squaring is ~53 instructions instead of 84 (gf128_lazy.h: lz_sqr), and on a lone wave the chain costs its instruction count.
// p - 2 = 2^128 - 9*2^32 - 1: 64 ones | 28 ones, 0110 | 32 ones (most significant first), as fe_inv (gf128.h).  0 -> 0.
__device__ __forceinline__ fe fe_inv_chain(const fe &a) {
    const lzk K = lzk_make();
    auto sqn = [&](lz v, int n) {
#pragma unroll 1
        for (int i = 0; i < n; i++) v = lz_sqr(v, K);
        return v;
    };
    const lz x1 = lz_unpack(a);
    const lz x2 = lz_mul_v(lz_sqr(x1, K), x1, K);
    const lz x4 = lz_mul_v(sqn(x2, 2), x2, K);
    const lz x8 = lz_mul_v(sqn(x4, 4), x4, K);
    const lz x16 = lz_mul_v(sqn(x8, 8), x8, K);
    const lz x32 = lz_mul_v(sqn(x16, 16), x16, K);
    const lz x64 = lz_mul_v(sqn(x32, 32), x32, K);
    lz x28 = lz_mul_v(sqn(x16, 8), x8, K);
    x28 = lz_mul_v(sqn(x28, 4), x4, K);
    lz r = lz_mul_v(sqn(x64, 28), x28, K);
    r = lz_sqr(r, K);                                   // 0
    r = lz_mul_v(lz_sqr(r, K), x1, K);                  // 1
    r = lz_mul_v(lz_sqr(r, K), x1, K);                  // 1
    r = lz_sqr(r, K);                                   // 0
    r = lz_mul_v(sqn(r, 32), x32, K);
    return lz_pack(r);
}
#else
__device__ __forceinline__ fe fe_inv_chain(const fe &a) { return fe_inv(a); }
#endif

#define GS_BINV_THREADS 1024
__global__ __launch_bounds__(GS_BINV_THREADS) void k_batch_inv(const fe *__restrict__ a, const fe *__restrict__ num, uint64_t n, uint64_t tot,
                                                             fe *__restrict__ out) {
    __shared__ fe wave_total[GS_BINV_THREADS / 64], wave_inverse[GS_BINV_THREADS / 64];
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const bool active = t < tot && t < n;            // every lane takes part in the shuffles and barriers below
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    fe acc = fe_one();
    uint64_t last = t;
    if (active) {
        for (uint64_t i = t; i < n; i += tot) {
            out[i] = acc;
            fe v = a[i];
            if (!fe_is_zero(v)) acc = fe_mul(acc, v);
            last = i;
        }
    }
    // inclusive prefix / suffix products of the lanes' totals within the wave
    fe pre = acc, suf = acc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        fe up = fe_shfl_up(pre, d), down = fe_shfl_down(suf, d);
        if (lane >= d) pre = fe_mul(pre, up);
        if (lane + d < 64) suf = fe_mul(suf, down);
    }
    if (lane == 63) wave_total[wave] = pre;
    __syncthreads();
    // wave 0: the same over the waves' totals, ONE Fermat inversion for the whole workgroup (248 dependent products; one per
    // thread was 7.75 products per element of a 32-element subsequence), then the inverse of every wave's total.  Measured on 2^24
    // elements: 415 -> 390 us only — the kernel moves 1.3-1.5 GB (a twice, out[] three times, num) and sits at ~3.9 TB/s
    if (wave == 0) {
        fe w = lane < nwaves ? wave_total[lane] : fe_one();
        fe p2 = w, s2 = w;
#pragma unroll
        for (int d = 1; d < GS_BINV_THREADS / 64; d <<= 1) {
            fe up = fe_shfl_up(p2, d), down = fe_shfl_down(s2, d);
            if (lane >= d) p2 = fe_mul(p2, up);
            if (lane + d < 64) s2 = fe_mul(s2, down);
        }
        const fe block_inv = fe_inv_chain(fe_shfl(p2, GS_BINV_THREADS / 64 - 1));
        fe b2 = fe_shfl_up(p2, 1), a2 = fe_shfl_down(s2, 1);
        if (lane == 0) b2 = fe_one();
        if (lane >= GS_BINV_THREADS / 64 - 1) a2 = fe_one();
        if (lane < nwaves) wave_inverse[lane] = fe_mul(fe_mul(block_inv, b2), a2);
    }
    __syncthreads();
    fe before = fe_shfl_up(pre, 1), after = fe_shfl_down(suf, 1);
    if (lane == 0) before = fe_one();
    if (lane == 63) after = fe_one();
    fe inv = fe_mul(fe_mul(wave_inverse[wave], before), after);   // 1 / acc
    if (!active) return;
    for (uint64_t i = last;; i -= tot) {
        fe v = a[i];
        fe r = fe_zero();
        if (!fe_is_zero(v)) {
            r = fe_mul(out[i], inv);
            inv = fe_mul(inv, v);
            if (num) r = fe_mul(r, num[i]);
        }
        out[i] = r;
        if (i < tot) break;
    }
}

static int launch_batch_inv(gs_ctx *c, const fe *a, const fe *num, uint64_t n, fe *out) {
    if (n == 0) return GS_OK;
    if ((const void *)a == (const void *)out || (num && (const void *)num == (const void *)out))
        return gs_fail(c, GS_ERR_ARG, "vec_inv/vec_div: output must not alias an input");
    // 32 elements per thread once there is enough work to fill the chip
    uint64_t tot = n / 32;
    if (tot < 16384) tot = n < 16384 ? n : 16384;
    unsigned threads = tot >= GS_BINV_THREADS ? GS_BINV_THREADS : (unsigned)((tot + 63) / 64 * 64);
    unsigned blocks = (unsigned)((tot + threads - 1) / threads);
    gs_traffic(c, n * GS_ELT * (num ? 3 : 2), n, "k_batch_inv");
    hipLaunchKernelGGL(k_batch_inv, dim3(blocks), dim3(threads), 0, c->stream, a, num, n, tot, out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_vec_inv_dev(gs_ctx *c, const fe *a, uint64_t n, fe *out) { return launch_batch_inv(c, a, nullptr, n, out); }

// ---- divisions whose denominators are known in closed form over the domain {omega^i} ---------------------------------
// 1/Z(x) of the transition constraints (ZeroPolynomial.ts:36-44 + CompositionPolynomial.ts:117): the denominator x^T - 1 takes only
// n/T distinct values over the domain -> a table of n/T inverses (host), one product per point, no vector is read
#define GS_ZPOLY_MAX_PERIOD 32
struct ZTable { fe c[GS_ZPOLY_MAX_PERIOD]; };
__global__ void k_zero_poly_inverses(const fe *__restrict__ tw_lo, const fe *__restrict__ tw_hi, int log_lo, int logn, uint64_t n, ZTable tab,
                                     uint32_t period, fe x_last, fe shift, int has_shift, fe *__restrict__ out) {
    __shared__ fe c[GS_ZPOLY_MAX_PERIOD];
    if (threadIdx.x < period) c[threadIdx.x] = tab.c[threadIdx.x];
    __syncthreads();
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        fe x = tw_lo[i & ((1ull << log_lo) - 1)];
        if (logn > log_lo) x = fe_mul(x, tw_hi[i >> log_lo]);
        if (has_shift) x = fe_mul(x, shift);                       // a coset of the domain (wave-uniform flag)
        out[i] = fe_mul(fe_sub(x, x_last), c[i & (period - 1)]);
    }
}

// num[i] / prod_a (omega^i - omega^k_a): the roots are domain points, so each factor's inverse is omega^-k_a * u[(i - k_a) mod n]
// with u = 1/(omega^j - 1) (cached table): nroots table reads and products per point instead of a batch inversion
#define GS_DOMAIN_ROOTS_MAX 4
struct RootArgs { uint64_t k[GS_DOMAIN_ROOTS_MAX]; };
__global__ void k_div_by_domain_roots(const fe *__restrict__ num, const fe *__restrict__ u, uint64_t n, RootArgs roots, uint32_t nroots, fe scale,
                                      fe *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        fe r = fe_mul(num[i], scale);
        for (uint32_t a = 0; a < nroots; a++) r = fe_mul(r, u[(i + n - roots.k[a]) & (n - 1)]);
        out[i] = r;
    }
}

// ---- the tail of the composition polynomial and the linear combination in ONE pass (gs_composition_tail, include/gstark.h) ----------
//   l[i] = q[i] * zinv[i]  +  sum_b (k_b + pw[i] kp_b) (P_b[i] - I_b(w^i)) / prod_a (w^i - w^r_ba)  +  sum_v (k_v + pw[i] kp_v) e_v[i]
// Separately (gs_vec_mul, gs_eval_polys_at_roots, gs_sub_matrix_from_vectors, gs_div_by_domain_roots, two gs_combine_adjusted) these are
// seven passes over the domain with five intermediate matrices; here every input is read once and nothing but l (and, if asked for, C)
// is written.  The descriptors are uniform: they sit in device memory (too many for the 4 KB of kernel arguments) and come in through
// scalar loads.  Each quotient's constant prod_a w^-r_a is folded into its two coefficients on the host.
#define GS_TAIL_MAX_ROOTS 4
#define GS_TAIL_MAX_ILEN 4
#if !defined(GS_SMALL_Q) && !defined(GS_WIDE_BITS)
#define GS_TAIL_LAZY 1
// 128-bit field: both sums are dot products with UNIFORM constants — sum_t k_t x_t and sum_t k'_t x_t over the boundary quotients and
// the committed vectors alike, l = D + S + pw * S'.  A term is one lz_unpack of x_t and 2 x 25 v_mad into two sets of nine 64-bit
// columns (the constants' limbs, unpacked on the host, arrive through scalar loads); a set is folded every six terms (columns below
// 2^57).  ~60 instructions per term against two canonical products and two additions (2 x 84 + 2 x 12).
struct TailK { int32_t k[5], kp[5], pad[2]; };
#else
struct TailK { fe k, kp; };
#endif
struct TailRow { const fe *v; uint64_t root[GS_TAIL_MAX_ROOTS]; fe ipoly[GS_TAIL_MAX_ILEN]; TailK c; uint32_t nroots, pad[3]; };
struct TailVec { const fe *v; uint64_t pad; TailK c; };
#ifdef GS_TAIL_LAZY
struct TailSums {
    int64_t a[9], b[9];
    fe pa, pb;
    int terms;
};
template <int HAS_PW>
__device__ __forceinline__ void tail_flush(TailSums &s, const lzk &K) {
    s.pa = fe_add(s.pa, lz_pack(lz_fold9(s.a, K)));
    if (HAS_PW) s.pb = fe_add(s.pb, lz_pack(lz_fold9(s.b, K)));
#pragma unroll
    for (int k = 0; k < 9; k++) { s.a[k] = 0; s.b[k] = 0; }
    s.terms = 0;
}
template <int HAS_PW>
__device__ __forceinline__ void tail_term(TailSums &s, const fe &x, const TailK &c, const lzk &K) {
    const lz u = lz_unpack(x);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int64_t ta = s.a[k], tb = s.b[k];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = k - i;
            if (j >= 0 && j < 5) {
                ta += (int64_t)u.l[i] * c.k[j];
                if (HAS_PW) tb += (int64_t)u.l[i] * c.kp[j];
            }
        }
        s.a[k] = ta;
        s.b[k] = tb;
    }
    if (++s.terms == 6) tail_flush<HAS_PW>(s, K);
}
#endif
// HAS_PW: 0 no degree adjustment, 1 pw[i] read from a vector, 2 pw[i] = omega^(i * pw_exp) from the domain's power tables.
// ZC: 1/Z(x_i) = (x_i - x_last) * ztab[i mod period] computed here (k_zero_poly_inverses' formula) instead of read.  Both vectors are
// written once and read once otherwise — the kernel is bound by its ~12 HBM streams, and its VALU has room for the three products.
template <int HAS_PW, int HAS_X, int ZC>
__global__ __launch_bounds__(256) void k_composition_tail(const fe *__restrict__ q, const fe *__restrict__ zinv, const fe *__restrict__ pw, const fe *__restrict__ u,
                                                          const fe *__restrict__ tw_lo, const fe *__restrict__ tw_hi, int log_lo, int logn, uint64_t n,
                                                          const TailRow *__restrict__ rows, uint32_t bcount, uint32_t ilen,
                                                          const TailVec *__restrict__ vecs, uint32_t lcount, ZTable ztab, uint32_t period, fe x_last,
                                                          uint64_t pw_exp, fe pw_step, fe shift, fe pw_scale, int has_shift, fe *__restrict__ c_out,
                                                          fe *__restrict__ l_out) {
    static_assert(!ZC || HAS_X, "1/Z(x) needs x");
#ifdef GS_TAIL_LAZY
    const lzk K = lzk_make();
#endif
    __shared__ fe zc[GS_ZPOLY_MAX_PERIOD];
    if (ZC) {
        if (threadIdx.x < period) zc[threadIdx.x] = ztab.c[threadIdx.x];
        __syncthreads();
    }
    // HAS_PW == 2: the power of this thread's first point from the tables (scattered reads, once), then a running product by
    // pw_step = omega^(pw_exp * stride of the loop) — a look-up per point would be two uncoalesced 16-byte gathers per lane
    fe prun = fe_one();
    if (HAS_PW == 2) {
        const uint64_t e = ((blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * pw_exp) & (n - 1);
        prun = tw_lo[e & ((1ull << log_lo) - 1)];
        if (logn > log_lo) prun = fe_mul(prun, tw_hi[e >> log_lo]);
        if (has_shift) prun = fe_mul(prun, pw_scale);               // a coset: x_i = shift * omega^i, so x_i^e = shift^e * omega^(i e)
    }
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        fe p = fe_one(), x = fe_one();
        if (HAS_PW == 1) p = pw[i];
        if (HAS_PW == 2) { p = prun; prun = fe_mul(prun, pw_step); }
        if (HAS_X) {
            x = tw_lo[i & ((1ull << log_lo) - 1)];
            if (logn > log_lo) x = fe_mul(x, tw_hi[i >> log_lo]);
            if (has_shift) x = fe_mul(x, shift);                     // (wave-uniform flag)
        }
        const fe d = fe_mul(q[i], ZC ? fe_mul(fe_sub(x, x_last), zc[i & (period - 1)]) : zinv[i]);
#ifdef GS_TAIL_LAZY
        TailSums s;
#pragma unroll
        for (int k = 0; k < 9; k++) { s.a[k] = 0; s.b[k] = 0; }
        s.pa = d;
        s.pb = fe_zero();
        s.terms = 0;
#else
        fe acc = d;
#endif
        for (uint32_t b = 0; b < bcount; b++) {
            const TailRow &r = rows[b];
            fe iv = r.ipoly[ilen - 1];
            if (HAS_X) for (int t = (int)ilen - 2; t >= 0; t--) iv = fe_add(fe_mul(iv, x), r.ipoly[t]);
            fe t = fe_sub(r.v[i], iv);
            for (uint32_t a = 0; a < r.nroots; a++) t = fe_mul(t, u[(i + n - r.root[a]) & (n - 1)]);
#ifdef GS_TAIL_LAZY
            tail_term<HAS_PW>(s, t, r.c, K);
#else
            const fe cf = HAS_PW ? fe_add(r.c.k, fe_mul(p, r.c.kp)) : r.c.k;
            acc = fe_add(acc, fe_mul(t, cf));
#endif
        }
#ifdef GS_TAIL_LAZY
        if (c_out) {
            if (s.terms) tail_flush<HAS_PW>(s, K);
            c_out[i] = HAS_PW ? fe_add(s.pa, fe_mul(p, s.pb)) : s.pa;
        }
#else
        if (c_out) c_out[i] = acc;
#endif
        // six vectors at a time: their loads are issued together (one at a time behind a run-time trip count, every load's latency would
        // be paid separately: 0.96 ms against 0.7x for six registers at N = 2^24)
        for (uint32_t v0 = 0; v0 < lcount; v0 += 6) {
            fe xs[6];
#pragma unroll
            for (int k = 0; k < 6; k++) xs[k] = v0 + k < lcount ? vecs[v0 + k].v[i] : fe_zero();
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (v0 + k >= lcount) break;
                const TailVec &e = vecs[v0 + k];
#ifdef GS_TAIL_LAZY
                tail_term<HAS_PW>(s, xs[k], e.c, K);
#else
                const fe cf = HAS_PW ? fe_add(e.c.k, fe_mul(p, e.c.kp)) : e.c.k;
                acc = fe_add(acc, fe_mul(xs[k], cf));
#endif
            }
        }
#ifdef GS_TAIL_LAZY
        if (s.terms) tail_flush<HAS_PW>(s, K);
        l_out[i] = HAS_PW ? fe_add(s.pa, fe_mul(p, s.pb)) : s.pa;
#else
        l_out[i] = acc;
#endif
    }
}

// ---- linear combination of many vectors --------------------------------------------------------------
struct PtrArgs {
    const fe *v[GS_MAX_COMBINE];
};
struct CombineArgs {  // vector pointers and coefficients travel as kernel arguments (1.5 KB): no staging copy, no sync
    const fe *v[GS_MAX_COMBINE];
    fe k[GS_MAX_COMBINE];
};
// accumulate: out already holds the sum over the previous batch of GS_MAX_COMBINE vectors (gs_combine_many splits longer lists)
__global__ void k_combine_many(CombineArgs va, uint32_t count, uint64_t n, int accumulate, fe *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        fe s = fe_mul(va.v[0][i], va.k[0]);
        if (accumulate) s = fe_add(s, out[i]);
        for (uint32_t j = 1; j < count; j++) s = fe_add(s, fe_mul(va.v[j][i], va.k[j]));
        out[i] = s;
    }
}

// out[i] = sum_j k_j v_j[i]  +  powers[i] * sum_j kp_j v_j[i]  (+ plus[i]): a merge with its degree adjustment (gs_combine_adjusted)
struct CombineAdjArgs {
    const fe *v[GS_MAX_COMBINE];
    fe k[GS_MAX_COMBINE];
    fe kp[GS_MAX_COMBINE];
};
template <int HAS_K, int HAS_KP>
__global__ void k_combine_adjusted(CombineAdjArgs va, uint32_t count, uint64_t n, const fe *__restrict__ powers, const fe *plus, fe *out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        fe x = va.v[0][i];
        fe a = HAS_K ? fe_mul(x, va.k[0]) : x, b = HAS_KP ? fe_mul(x, va.kp[0]) : x;
        for (uint32_t j = 1; j < count; j++) {
            x = va.v[j][i];
            if (HAS_K) a = fe_add(a, fe_mul(x, va.k[j]));
            if (HAS_KP) b = fe_add(b, fe_mul(x, va.kp[j]));
        }
        fe s;
        if (HAS_K && HAS_KP) s = fe_add(a, fe_mul(b, powers[i]));
        else if (HAS_KP) s = fe_mul(b, powers[i]);
        else s = a;
        if (plus) s = fe_add(s, plus[i]);
        out[i] = s;
    }
}

// ---- dot product -> one element ---------------------------------------------------------------------
__device__ __forceinline__ fe block_reduce_add(fe s, fe *sh) {
    // wave reduce via shuffles, then one LDS step across the (<= 4) waves of the block
    for (int off = 32; off >= 1; off >>= 1) {
        fe o;
#pragma unroll
        for (int l = 0; l < GF_LIMBS; l++) fe_set_limb(o, l, __shfl_down(fe_limb(s, l), off));
        s = fe_add(s, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; w++) s = fe_add(s, sh[w]);
    }
    return s;
}
__global__ void k_dot_partial(const fe *__restrict__ a, const fe *__restrict__ b, uint64_t n, fe *__restrict__ partial) {
    __shared__ fe sh[4];
    fe s = fe_zero();
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        s = fe_add(s, fe_mul(a[i], b[i]));
    s = block_reduce_add(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void k_sum_final(const fe *__restrict__ partial, uint32_t count, fe *__restrict__ out) {
    __shared__ fe sh[4];
    fe s = fe_zero();
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) s = fe_add(s, partial[i]);
    s = block_reduce_add(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}

// ---- index shuffles -------------------------------------------------------------------------------------
__global__ void k_pluck(const fe *__restrict__ v, uint64_t vlen, uint64_t skip, uint64_t times, fe *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < times; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = v[(i * skip) % vlen];
}
// out[r*cols + k] = v[(r + k*rows)*step]; one thread per output element, lanes walk r for a fixed k so
// the (large) reads stay coalesced when step == 1
__global__ void k_transpose_vector(const fe *__restrict__ v, uint64_t rows, uint32_t cols, uint64_t step, fe *__restrict__ out) {
    uint64_t total = rows * cols;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = t / rows, r = t % rows;
        out[r * cols + k] = v[(r + k * rows) * step];
    }
}
// cols == 4 specialisation: thread per row, 4 coalesced 16-byte loads, one 64-byte row store
__global__ void k_transpose_vector4(const fe *__restrict__ v, uint64_t rows, uint64_t step, fe *__restrict__ out) {
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) {
        fe x0 = v[r * step], x1 = v[(r + rows) * step], x2 = v[(r + 2 * rows) * step], x3 = v[(r + 3 * rows) * step];
        fe *o = out + r * 4;
        o[0] = x0; o[1] = x1; o[2] = x2; o[3] = x3;
    }
}
__global__ void k_transpose_matrix(const fe *__restrict__ m, uint64_t rows, uint64_t cols, fe *__restrict__ out) {
    uint64_t total = rows * cols;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = t / cols, k = t % cols;
        out[k * rows + r] = m[t];
    }
}
__global__ void k_sub_matrix_from_vectors(PtrArgs va, const fe *__restrict__ m, uint64_t cols, fe *__restrict__ out) {
    const fe *v = va.v[blockIdx.y];
    const fe *mr = m + (uint64_t)blockIdx.y * cols;
    fe *o = out + (uint64_t)blockIdx.y * cols;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cols; i += (uint64_t)gridDim.x * blockDim.x)
        o[i] = fe_sub(v[i], mr[i]);
}

// ---- FRI row polynomials ------------------------------------------------------------------------------
// General Lagrange interpolation through 4 arbitrary points per row (verifier-side shape).
__global__ void k_quartic_interp_generic(const fe *__restrict__ xs, const fe *__restrict__ ys, uint64_t rows, fe *__restrict__ out) {
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) {
        fe x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { x[j] = xs[r * 4 + j]; y[j] = ys[r * 4 + j]; }
        // denominators d_j = prod_{m != j} (x_j - x_m); invert all four with one inversion
        fe d[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            fe p = fe_one();
#pragma unroll
            for (int m = 0; m < 4; m++)
                if (m != j) p = fe_mul(p, fe_sub(x[j], x[m]));
            d[j] = p;
        }
        fe p01 = fe_mul(d[0], d[1]), p012 = fe_mul(p01, d[2]), all = fe_mul(p012, d[3]);
        fe inv = fe_inv(all);
        fe i3 = fe_mul(inv, p012);
        inv = fe_mul(inv, d[3]);
        fe i2 = fe_mul(inv, p01);
        inv = fe_mul(inv, d[2]);
        fe i1 = fe_mul(inv, d[0]);
        fe i0 = fe_mul(inv, d[1]);
        fe s[4] = {fe_mul(y[0], i0), fe_mul(y[1], i1), fe_mul(y[2], i2), fe_mul(y[3], i3)};
        fe c0 = fe_zero(), c1 = fe_zero(), c2 = fe_zero(), c3 = fe_zero();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // numerator prod_{m != j} (X - x_m) = X^3 - e1 X^2 + e2 X - e3 over the three other points
            fe a = x[(j + 1) & 3], b = x[(j + 2) & 3], cc = x[(j + 3) & 3];
            fe e1 = fe_add(fe_add(a, b), cc);
            fe ab = fe_mul(a, b);
            fe e2 = fe_add(ab, fe_mul(cc, fe_add(a, b)));
            fe e3 = fe_mul(ab, cc);
            c3 = fe_add(c3, s[j]);
            c2 = fe_sub(c2, fe_mul(s[j], e1));
            c1 = fe_add(c1, fe_mul(s[j], e2));
            c0 = fe_sub(c0, fe_mul(s[j], e3));
        }
        fe *o = out + r * 4;
        o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
    }
}

// Prover-side shape: row r has x-coordinates x, zeta*x, zeta^2*x, zeta^3*x with x = omega^(r*step),
// zeta = omega^(n/4).  Then u_k = c_k x^k is the inverse 4-point DFT of the row and no inversion is needed:
// x^-1 = omega^(n - r*step) comes from the power tables.
__global__ void k_quartic_interp_domain(const fe *__restrict__ ys, uint64_t rows, uint64_t step, uint64_t n,
                                        const fe *__restrict__ tw_lo, const fe *__restrict__ tw_hi, int log_lo, int logn,
                                        fe zeta_inv, fe inv4, fe *__restrict__ out) {
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const fe *y = ys + r * 4;
        fe y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3];
        fe s0 = fe_add(y0, y2), s1 = fe_sub(y0, y2), s2 = fe_add(y1, y3);
        fe s3 = fe_mul(fe_sub(y1, y3), zeta_inv);
        fe u0 = fe_add(s0, s2), u2 = fe_sub(s0, s2), u1 = fe_add(s1, s3), u3 = fe_sub(s1, s3);
        uint64_t e = (r * step) & (n - 1);
        e = e ? n - e : 0;
        fe xi = e ? fe_make(0, 0, 0, 0) : fe_one();
        if (e) {
            xi = tw_lo[e & ((1ull << log_lo) - 1)];
            if (logn > log_lo) xi = fe_mul(xi, tw_hi[e >> log_lo]);
        }
        fe q1 = fe_mul(xi, inv4), q2 = fe_mul(q1, xi), q3 = fe_mul(q2, xi);
        fe *o = out + r * 4;
        o[0] = fe_mul(u0, inv4);
        o[1] = fe_mul(u1, q1);
        o[2] = fe_mul(u2, q2);
        o[3] = fe_mul(u3, q3);
    }
}

__global__ void k_quartic_eval(const fe *__restrict__ polys, uint64_t rows, fe x, fe *__restrict__ out) {
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const fe *p = polys + r * 4;
        fe s = p[3];
        s = fe_add(fe_mul(s, x), p[2]);
        s = fe_add(fe_mul(s, x), p[1]);
        s = fe_add(fe_mul(s, x), p[0]);
        out[r] = s;
    }
}

// One FRI folding step straight from a column (the transposed matrix is never written): row r of transposeVector(column, 4) is
// (column[r], column[r + rows], column[r + 2 rows], column[r + 3 rows]) at x_r zeta^c; the cubic through them evaluated at X is
// (u0 + u1 t + u2 t^2 + u3 t^3) / 4 with u = the inverse 4-point DFT above and t = X / x_r.
__global__ void k_fri_fold(const fe *__restrict__ column, uint64_t rows, uint64_t step, uint64_t n, const fe *__restrict__ tw_lo,
                           const fe *__restrict__ tw_hi, int log_lo, int logn, fe zeta_inv, fe inv4, fe X, const fe *__restrict__ Xdev, fe *__restrict__ out) {
    if (Xdev) X = fe_mul(*Xdev, X);                    // the evaluation point was derived on the device (gs_fri_fold_seeded): X holds its scale
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) {
        fe y0 = column[r], y1 = column[r + rows], y2 = column[r + 2 * rows], y3 = column[r + 3 * rows];
        fe s0 = fe_add(y0, y2), s1 = fe_sub(y0, y2), s2 = fe_add(y1, y3);
        fe s3 = fe_mul(fe_sub(y1, y3), zeta_inv);
        fe u0 = fe_add(s0, s2), u2 = fe_sub(s0, s2), u1 = fe_add(s1, s3), u3 = fe_sub(s1, s3);
        uint64_t e = (r * step) & (n - 1);
        e = e ? n - e : 0;
        fe t = X;                                      // X * x_r^-1
        if (e) {
            fe xi = tw_lo[e & ((1ull << log_lo) - 1)];
            if (logn > log_lo) xi = fe_mul(xi, tw_hi[e >> log_lo]);
            t = fe_mul(X, xi);
        }
        fe v = fe_add(fe_mul(u3, t), u2);
        v = fe_add(fe_mul(v, t), u1);
        v = fe_add(fe_mul(v, t), u0);
        out[r] = fe_mul(v, inv4);
    }
}

// ---- C ABI ----------------------------------------------------------------------------------------------
#define CHECK3(c, a, b, o) \
    if (!(c) || !(a) || !(b) || !(o)) return GS_ERR_ARG

static int dot_to_host(gs_ctx *c, const fe *a, const fe *b, uint64_t n, gs_elt *out_host) {
    unsigned blocks = gs_grid(n, 256, 1024);
    int rc = gs_stage_reserve(c, (uint64_t)(blocks + 1) * GS_ELT);
    if (rc) return rc;
    fe *partial = (fe *)c->d_stage;
    hipLaunchKernelGGL(k_dot_partial, dim3(blocks), dim3(256), 0, c->stream, a, b, n, partial + 1);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, c->stream, partial + 1, blocks, partial);
    GS_LAUNCH_CHECK(c);
    GS_HIP(c, hipMemcpyAsync(c->h_stage, partial, GS_ELT, hipMemcpyDeviceToHost, c->stream));
    GS_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(out_host, c->h_stage, GS_ELT);
    return GS_OK;
}

extern "C" {

int gs_power_series(gs_ctx *c, const gs_elt *base, uint64_t n, void *out) {
    if (!c || !base || (!out && n)) return GS_ERR_ARG;
    return gs_power_series_dev(c, fe_from_bytes(base), n, (fe *)out);
}

#define VEC_VEC(NAME, OP)                                                                                  \
    int NAME(gs_ctx *c, const void *a, const void *b, uint64_t n, void *out) {                             \
        CHECK3(c, a, b, out);                                                                              \
        if (!n) return GS_OK;                                                                              \
        gs_traffic(c, 3 * n * GS_ELT, n, "k_vec_vec<%d>", (int)OP);                                        \
        hipLaunchKernelGGL(k_vec_vec<OP>, dim3(gs_grid(n)), dim3(256), 0, c->stream, (const fe *)a, (const fe *)b, n, (fe *)out); \
        GS_LAUNCH_CHECK(c);                                                                                \
        return GS_OK;                                                                                      \
    }
#define VEC_SCALAR(NAME, OP)                                                                               \
    int NAME(gs_ctx *c, const void *a, const gs_elt *s, uint64_t n, void *out) {                       \
        CHECK3(c, a, s, out);                                                                              \
        if (!n) return GS_OK;                                                                              \
        gs_traffic(c, 2 * n * GS_ELT, n, "k_vec_scalar<%d>", (int)OP);                                     \
        hipLaunchKernelGGL(k_vec_scalar<OP>, dim3(gs_grid(n)), dim3(256), 0, c->stream, (const fe *)a, fe_from_bytes(s), n, (fe *)out); \
        GS_LAUNCH_CHECK(c);                                                                                \
        return GS_OK;                                                                                      \
    }
VEC_VEC(gs_vec_add, OP_ADD)
VEC_VEC(gs_vec_sub, OP_SUB)
VEC_VEC(gs_vec_mul, OP_MUL)
VEC_SCALAR(gs_vec_add_scalar, OP_ADD)
VEC_SCALAR(gs_vec_sub_scalar, OP_SUB)
VEC_SCALAR(gs_vec_mul_scalar, OP_MUL)

int gs_vec_inv(gs_ctx *c, const void *a, uint64_t n, void *out) {
    if (!c || !a || !out) return GS_ERR_ARG;
    return launch_batch_inv(c, (const fe *)a, nullptr, n, (fe *)out);
}
int gs_vec_div(gs_ctx *c, const void *a, const void *b, uint64_t n, void *out) {
    CHECK3(c, a, b, out);
    return launch_batch_inv(c, (const fe *)b, (const fe *)a, n, (fe *)out);
}
int gs_vec_exp(gs_ctx *c, const void *a, const gs_elt *e, uint64_t n, void *out) {
    CHECK3(c, a, e, out);
    if (!n) return GS_OK;
    hipLaunchKernelGGL(k_vec_exp, dim3(gs_grid(n)), dim3(256), 0, c->stream, (const fe *)a, fe_from_bytes(e), n, (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_combine_many(gs_ctx *c, const void *const *vecs_host, const uint8_t *coeffs_host, uint32_t count, uint64_t n, void *out) {
    if (!c || !vecs_host || !coeffs_host || !out) return GS_ERR_ARG;
    if (count == 0) return gs_fail(c, GS_ERR_ARG, "combine_many: no vectors");
    if (!n) return GS_OK;
    // pointers and coefficients travel as kernel arguments, GS_MAX_COMBINE per launch; a longer list (an AIR with more than 32
    // registers: LinearCombination.ts:36-64 combines 2*(registers) vectors) goes in batches that accumulate into `out`.
    // `out` must not alias an input of a LATER batch (it never does in the reference's call sites: the result is a new vector).
    for (uint32_t base = 0; base < count; base += GS_MAX_COMBINE) {
        const uint32_t m = count - base < GS_MAX_COMBINE ? count - base : GS_MAX_COMBINE;
        CombineArgs va;
        for (uint32_t j = 0; j < GS_MAX_COMBINE; j++) {
            va.v[j] = (const fe *)vecs_host[base + (j < m ? j : 0)];
            va.k[j] = j < m ? fe_from_bytes(coeffs_host + GS_ELT * (base + j)) : fe_zero();
        }
        gs_traffic(c, n * GS_ELT * (m + 1 + (base ? 1 : 0)), n, "k_combine_many");
        hipLaunchKernelGGL(k_combine_many, dim3(gs_grid(n)), dim3(256), 0, c->stream, va, m, n, base ? 1 : 0, (fe *)out);
        GS_LAUNCH_CHECK(c);
    }
    return GS_OK;
}

int gs_combine_adjusted(gs_ctx *c, const void *const *vecs_host, const uint8_t *coeffs_host, const uint8_t *adj_coeffs_host, uint32_t count,
                        const void *powers, const void *plus, uint64_t n, void *out) {
    if (!c || !vecs_host || !out || (!coeffs_host && !adj_coeffs_host) || (adj_coeffs_host && !powers)) return GS_ERR_ARG;
    if (count == 0) return gs_fail(c, GS_ERR_ARG, "combine_adjusted: no vectors");
    if (!n) return GS_OK;
    // GS_MAX_COMBINE vectors per launch (pointers and both coefficient lists as kernel arguments); later batches add to `out`
    for (uint32_t base = 0; base < count; base += GS_MAX_COMBINE) {
        const uint32_t m = count - base < GS_MAX_COMBINE ? count - base : GS_MAX_COMBINE;
        CombineAdjArgs va;
        for (uint32_t j = 0; j < GS_MAX_COMBINE; j++) {
            va.v[j] = (const fe *)vecs_host[base + (j < m ? j : 0)];
            va.k[j] = j < m && coeffs_host ? fe_from_bytes(coeffs_host + GS_ELT * (base + j)) : fe_zero();
            va.kp[j] = j < m && adj_coeffs_host ? fe_from_bytes(adj_coeffs_host + GS_ELT * (base + j)) : fe_zero();
        }
        const fe *pl = base ? (const fe *)out : (const fe *)plus;
        const dim3 grid(gs_grid(n)), block(256);
        // m vectors + the power series (when there are adjusted terms) + the addend read, one vector written
        gs_traffic(c, n * GS_ELT * (m + (adj_coeffs_host ? 1 : 0) + (pl ? 1 : 0) + 1), n, "k_combine_adjusted<%d, %d>", coeffs_host ? 1 : 0, adj_coeffs_host ? 1 : 0);
        if (coeffs_host && adj_coeffs_host)
            hipLaunchKernelGGL((k_combine_adjusted<1, 1>), grid, block, 0, c->stream, va, m, n, (const fe *)powers, pl, (fe *)out);
        else if (adj_coeffs_host)
            hipLaunchKernelGGL((k_combine_adjusted<0, 1>), grid, block, 0, c->stream, va, m, n, (const fe *)powers, pl, (fe *)out);
        else
            hipLaunchKernelGGL((k_combine_adjusted<1, 0>), grid, block, 0, c->stream, va, m, n, (const fe *)powers, pl, (fe *)out);
        GS_LAUNCH_CHECK(c);
    }
    return GS_OK;
}

int gs_combine(gs_ctx *c, const void *a, const void *b, uint64_t n, gs_elt *out_host) {
    CHECK3(c, a, b, out_host);
    if (!n) { memset(out_host, 0, GS_ELT); return GS_OK; }
    return dot_to_host(c, (const fe *)a, (const fe *)b, n, out_host);
}

int gs_zero_poly_inverses(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t steps, const gs_elt *x_last, void *out) {
    uint8_t one[sizeof(fe)];
    fe_to_bytes(one, fe_one());
    return gs_zero_poly_inverses_coset(c, omega, n, one, steps, x_last, out);
}
int gs_zero_poly_inverses_coset(gs_ctx *c, const gs_elt *omega, uint64_t n, const gs_elt *shift_bytes, uint64_t steps, const gs_elt *x_last, void *out) {
    if (!c || !omega || !shift_bytes || !x_last || !out) return GS_ERR_ARG;
    if (!gs_is_pow2(n) || !gs_is_pow2(steps) || steps > n) return gs_fail(c, GS_ERR_ARG, "zero_poly_inverses: n and steps must be powers of two, steps <= n");
    const fe shift = fe_from_bytes(shift_bytes);
    const int has_shift = fe_eq(shift, fe_one()) ? 0 : 1;
    const uint64_t period = n / steps;
    if (period > GS_ZPOLY_MAX_PERIOD) return gs_fail(c, GS_ERR_UNSUPPORTED, "zero_poly_inverses: n / steps above %d", GS_ZPOLY_MAX_PERIOD);
    const fe w = fe_from_bytes(omega);
    const fe *lo, *hi;
    int log_lo;
    int rc = gs_plan_pow_tables(c, w, n, &lo, &hi, &log_lo);      // also checks that omega is a primitive n-th root of unity
    if (rc) return rc;
    ZTable tab;
    const std::vector<fe> &inv = gs_memo(c, gs_memo_key("zpoly").add(w).add(n).add(steps).add(shift), [&](std::vector<fe> &t) {
        const fe g = fe_pow_u64(w, steps);                        // x_i^steps = shift^steps * g^(i mod period)
        fe cur = fe_pow_u64(shift, steps);
        for (uint64_t j = 0; j < period; j++) {
            t.push_back(fe_inv(fe_sub(cur, fe_one())));           // j = 0: 0^-1 = 0
            cur = fe_mul(cur, g);
        }
    });
    for (uint64_t j = 0; j < GS_ZPOLY_MAX_PERIOD; j++) tab.c[j] = j < period ? inv[j] : fe_zero();
    hipLaunchKernelGGL(k_zero_poly_inverses, dim3(gs_grid(n)), dim3(256), 0, c->stream, lo, hi, log_lo, gs_log2(n), n, tab, (uint32_t)period,
                       fe_from_bytes(x_last), shift, has_shift, (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_div_by_domain_roots(gs_ctx *c, const void *num, uint32_t rows, uint64_t n, const gs_elt *omega, const uint64_t *root_index_host,
                           const uint32_t *roots_per_row_host, uint32_t max_roots, void *out) {
    uint8_t one[sizeof(fe)];
    fe_to_bytes(one, fe_one());
    return gs_div_by_domain_roots_coset(c, num, rows, n, omega, one, root_index_host, roots_per_row_host, max_roots, out);
}
int gs_div_by_domain_roots_coset(gs_ctx *c, const void *num, uint32_t rows, uint64_t n, const gs_elt *omega, const gs_elt *shift_bytes,
                                 const uint64_t *root_index_host, const uint32_t *roots_per_row_host, uint32_t max_roots, void *out) {
    if (!c || !num || !omega || !shift_bytes || !out || !roots_per_row_host || (max_roots && !root_index_host)) return GS_ERR_ARG;
    if (!gs_is_pow2(n)) return gs_fail(c, GS_ERR_ARG, "div_by_domain_roots: n must be a power of two");
    if (!rows) return GS_OK;
    for (uint32_t r = 0; r < rows; r++)
        if (roots_per_row_host[r] > max_roots || roots_per_row_host[r] > GS_DOMAIN_ROOTS_MAX)
            return gs_fail(c, GS_ERR_UNSUPPORTED, "div_by_domain_roots: at most %d roots per row", GS_DOMAIN_ROOTS_MAX);
    const fe w = fe_from_bytes(omega);
    const fe *u;
    int rc = gs_plan_inverse_table_shifted(c, w, n, fe_from_bytes(shift_bytes), &u);      // (the unshifted table when shift = 1)
    if (rc) return rc;
    for (uint32_t r = 0; r < rows; r++) {
        RootArgs ra;
        uint64_t ksum = 0;
        for (uint32_t a = 0; a < GS_DOMAIN_ROOTS_MAX; a++) {
            ra.k[a] = a < roots_per_row_host[r] ? root_index_host[(uint64_t)r * max_roots + a] & (n - 1) : 0;
            if (a < roots_per_row_host[r]) ksum = (ksum + ra.k[a]) & (n - 1);
        }
        const fe scale = gs_memo_pow(c, w, (n - ksum) & (n - 1));     // prod omega^-k_a
        hipLaunchKernelGGL(k_div_by_domain_roots, dim3(gs_grid(n)), dim3(256), 0, c->stream, (const fe *)num + (uint64_t)r * n, u, n, ra,
                           roots_per_row_host[r], scale, (fe *)out + (uint64_t)r * n);
    }
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

static void tail_coeffs(TailK &c, const fe &k, const fe &kp) {
#ifdef GS_TAIL_LAZY
    const lz a = lz_unpack(k), b = lz_unpack(kp);
    for (int i = 0; i < 5; i++) { c.k[i] = a.l[i]; c.kp[i] = b.l[i]; }
#else
    c.k = k;
    c.kp = kp;
#endif
}
int gs_composition_tail(gs_ctx *c, uint64_t n, const gs_elt *omega, const void *q, const void *z_inv, uint64_t z_steps, const gs_elt *x_last,
                        const void *const *b_vecs_host, uint32_t bcount, const uint8_t *ipolys_host, uint32_t ilen, const uint64_t *root_index_host,
                        const uint32_t *roots_per_row_host, uint32_t max_roots, const uint8_t *b_coeffs_host, const uint8_t *b_adj_host,
                        const void *const *l_vecs_host, uint32_t lcount, const uint8_t *l_coeffs_host, const uint8_t *l_adj_host, const void *powers,
                        uint64_t powers_exponent, void *c_out, void *l_out) {
    uint8_t one[sizeof(fe)];
    fe_to_bytes(one, fe_one());
    return gs_composition_tail_coset(c, n, omega, one, q, z_inv, z_steps, x_last, b_vecs_host, bcount, ipolys_host, ilen, root_index_host, roots_per_row_host,
                                     max_roots, b_coeffs_host, b_adj_host, l_vecs_host, lcount, l_coeffs_host, l_adj_host, powers, powers_exponent, c_out, l_out);
}
int gs_composition_tail_coset(gs_ctx *c, uint64_t n, const gs_elt *omega, const gs_elt *shift_bytes, const void *q, const void *z_inv, uint64_t z_steps,
                              const gs_elt *x_last, const void *const *b_vecs_host, uint32_t bcount, const uint8_t *ipolys_host, uint32_t ilen,
                              const uint64_t *root_index_host, const uint32_t *roots_per_row_host, uint32_t max_roots, const uint8_t *b_coeffs_host,
                              const uint8_t *b_adj_host, const void *const *l_vecs_host, uint32_t lcount, const uint8_t *l_coeffs_host,
                              const uint8_t *l_adj_host, const void *powers, uint64_t powers_exponent, void *c_out, void *l_out) {
    if (!c || !omega || !shift_bytes || !q || (!z_inv && !x_last) || !l_out) return GS_ERR_ARG;
    const fe shift = fe_from_bytes(shift_bytes);
    const int has_shift = fe_eq(shift, fe_one()) ? 0 : 1;
    if (bcount && (!b_vecs_host || !ipolys_host || !roots_per_row_host || !b_coeffs_host || (max_roots && !root_index_host))) return GS_ERR_ARG;
    if (lcount && (!l_vecs_host || !l_coeffs_host)) return GS_ERR_ARG;
    if ((b_adj_host || l_adj_host) && !powers && !powers_exponent) return GS_ERR_ARG;
    if (!gs_is_pow2(n)) return gs_fail(c, GS_ERR_ARG, "composition_tail: n must be a power of two");
    ZTable ztab;
    uint64_t period = 1;
    if (!z_inv) {
        if (!gs_is_pow2(z_steps) || z_steps > n) return gs_fail(c, GS_ERR_ARG, "composition_tail: steps must be a power of two <= n");
        period = n / z_steps;
        if (period > GS_ZPOLY_MAX_PERIOD) return gs_fail(c, GS_ERR_UNSUPPORTED, "composition_tail: n / steps above %d (pass 1/Z as a vector)", GS_ZPOLY_MAX_PERIOD);
    }
    if (bcount > 64) return gs_fail(c, GS_ERR_UNSUPPORTED, "composition_tail: at most 64 boundary rows");
    if (bcount && (ilen == 0 || ilen > GS_TAIL_MAX_ILEN)) return gs_fail(c, GS_ERR_UNSUPPORTED, "composition_tail: 1..%d interpolant coefficients per row", GS_TAIL_MAX_ILEN);
    for (uint32_t r = 0; r < bcount; r++)
        if (roots_per_row_host[r] > max_roots || roots_per_row_host[r] > GS_TAIL_MAX_ROOTS)
            return gs_fail(c, GS_ERR_UNSUPPORTED, "composition_tail: at most %d roots per row", GS_TAIL_MAX_ROOTS);
    const fe w = fe_from_bytes(omega);
    const fe *lo = nullptr, *hi = nullptr, *u = nullptr;
    int log_lo = 0;
    int rc = gs_plan_pow_tables(c, w, n, &lo, &hi, &log_lo);
    if (!rc && bcount) rc = gs_plan_inverse_table_shifted(c, w, n, shift, &u);        // 1 / (shift * omega^j - 1): the unshifted table when shift = 1
    if (rc) return rc;
    if (!z_inv) {                                                  // gs_zero_poly_inverses[_coset]' table: 1 / (shift^steps g^j - 1), g = omega^steps, 0^-1 = 0
        const std::vector<fe> &inv = gs_memo(c, gs_memo_key("zpoly").add(w).add(n).add(z_steps).add(shift), [&](std::vector<fe> &t) {
            const fe g = fe_pow_u64(w, z_steps);
            fe cur = fe_pow_u64(shift, z_steps);
            for (uint64_t j = 0; j < period; j++) {
                t.push_back(fe_inv(fe_sub(cur, fe_one())));
                cur = fe_mul(cur, g);
            }
        });
        for (uint64_t j = 0; j < GS_ZPOLY_MAX_PERIOD; j++) ztab.c[j] = j < period ? inv[j] : fe_zero();
    } else {
        for (uint64_t j = 0; j < GS_ZPOLY_MAX_PERIOD; j++) ztab.c[j] = fe_zero();
    }
    // descriptors -> device scratch through the upload ring (asynchronous, the caller's arrays are free at once)
    const uint64_t rows_b = ((uint64_t)bcount * sizeof(TailRow) + 255) & ~(uint64_t)255, vecs_b = (uint64_t)lcount * sizeof(TailVec);
    const uint64_t total = rows_b + vecs_b ? rows_b + vecs_b : 256;
    void *d = nullptr, *h = nullptr;
    if ((rc = gs_tmp_alloc(c, total, &d))) return rc;
    if ((rc = gs_push_reserve(c, total, &h))) { gs_tmp_free(c, d); return rc == GS_ERR_UNSUPPORTED ? gs_fail(c, GS_ERR_UNSUPPORTED, "composition_tail: too many vectors") : rc; }
    memset(h, 0, total);
    TailRow *hr = (TailRow *)h;
    TailVec *hv = (TailVec *)((uint8_t *)h + rows_b);
    for (uint32_t r = 0; r < bcount; r++) {
        hr[r].v = (const fe *)b_vecs_host[r];
        hr[r].nroots = roots_per_row_host[r];
        uint64_t ksum = 0;
        for (uint32_t a = 0; a < roots_per_row_host[r]; a++) {
            hr[r].root[a] = root_index_host[(uint64_t)r * max_roots + a] & (n - 1);
            ksum = (ksum + hr[r].root[a]) & (n - 1);
        }
        const fe scale = gs_memo_pow(c, w, (n - ksum) & (n - 1));          // prod_a omega^-r_a: 1/(w^i - w^r) = w^-r * u[i - r]
        tail_coeffs(hr[r].c, fe_mul(fe_from_bytes(b_coeffs_host + (size_t)r * GS_ELT), scale),
                    b_adj_host ? fe_mul(fe_from_bytes(b_adj_host + (size_t)r * GS_ELT), scale) : fe_zero());
        for (uint32_t t = 0; t < ilen; t++) hr[r].ipoly[t] = fe_from_bytes(ipolys_host + ((size_t)r * ilen + t) * GS_ELT);
    }
    for (uint32_t v = 0; v < lcount; v++) {
        hv[v].v = (const fe *)l_vecs_host[v];
        tail_coeffs(hv[v].c, fe_from_bytes(l_coeffs_host + (size_t)v * GS_ELT), l_adj_host ? fe_from_bytes(l_adj_host + (size_t)v * GS_ELT) : fe_zero());
    }
    if ((rc = gs_push_commit(c, d, h, total))) { gs_tmp_free(c, d); return rc; }
    const TailRow *dr = (const TailRow *)d;
    const TailVec *dv = (const TailVec *)((uint8_t *)d + rows_b);
    const bool adjusted = b_adj_host || l_adj_host;
    const int pwk = !adjusted ? 0 : (powers ? 1 : 2);
    const bool zc = !z_inv, has_x = zc || (bcount && ilen > 1);
    // x_i^e = shift^e * omega^(i e): omega's exponent reduces mod n, the shift's does NOT (shift is a root of unity of a larger order)
    const fe pw_scale = has_shift && pwk == 2 ? gs_memo_pow(c, shift, powers_exponent) : fe_one();
    const fe xl = x_last ? fe_from_bytes(x_last) : fe_zero();
    const dim3 grid(gs_grid(n)), block(256);
    const fe pw_step = gs_memo_pow(c, w, (((powers_exponent & (n - 1)) * ((uint64_t)grid.x * block.x)) & (n - 1)));      // n <= 2^32: no overflow
#define GS_TAIL_LAUNCH(PW, X, ZC)                                                                                                                    \
    hipLaunchKernelGGL((k_composition_tail<PW, X, ZC>), grid, block, 0, c->stream, (const fe *)q, (const fe *)z_inv, (const fe *)powers, u, lo, hi,   \
                       log_lo, gs_log2(n), n, dr, bcount, ilen ? ilen : 1u, dv, lcount, ztab, (uint32_t)period, xl, powers_exponent & (n - 1),         \
                       pw_step, shift, pw_scale, has_shift, (fe *)c_out, (fe *)l_out)
#define GS_TAIL_PW(X, ZC)                                                                                                                              \
    do { if (pwk == 0) GS_TAIL_LAUNCH(0, X, ZC); else if (pwk == 1) GS_TAIL_LAUNCH(1, X, ZC); else GS_TAIL_LAUNCH(2, X, ZC); } while (0)
    // Q + (1/Z when it is not computed per point) + the power series when materialised + the bcount asserted registers' extensions + the
    // lcount committed vectors read, L (and C when asked for) written.  The asserted registers are among the committed vectors: counted once
    gs_traffic(c, n * GS_ELT * (1 + (z_inv ? 1 : 0) + (pwk == 1 ? 1 : 0) + (lcount ? lcount : bcount) + (c_out ? 1 : 0) + (l_out ? 1 : 0)), n,
               "k_composition_tail<%d, %d, %d>", pwk, (zc || has_x) ? 1 : 0, zc ? 1 : 0);
    if (zc) GS_TAIL_PW(1, 1);
    else if (has_x) GS_TAIL_PW(1, 0);
    else GS_TAIL_PW(0, 0);
#undef GS_TAIL_PW
#undef GS_TAIL_LAUNCH
    hipError_t e = hipGetLastError();
    gs_tmp_free(c, d);      // stream-ordered reuse: later users of the block are queued behind this kernel
    if (e != hipSuccess) return gs_fail(c, GS_ERR_DEVICE, "composition_tail launch: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_pluck(gs_ctx *c, const void *v, uint64_t vlen, uint64_t skip, uint64_t times, void *out) {
    if (!c || !v || !out) return GS_ERR_ARG;
    if (!vlen) return gs_fail(c, GS_ERR_ARG, "pluck: empty vector");
    if (!times) return GS_OK;
    hipLaunchKernelGGL(k_pluck, dim3(gs_grid(times)), dim3(256), 0, c->stream, (const fe *)v, vlen, skip, times, (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_transpose_vector(gs_ctx *c, const void *v, uint64_t n, uint32_t cols, uint64_t step, void *out) {
    if (!c || !v || !out) return GS_ERR_ARG;
    if (!cols || !step || n % ((uint64_t)cols * step)) return gs_fail(c, GS_ERR_ARG, "transpose_vector: n %% (cols*step) != 0");
    uint64_t rows = n / ((uint64_t)cols * step);
    if (!rows) return GS_OK;
    if (cols == 4)
        hipLaunchKernelGGL(k_transpose_vector4, dim3(gs_grid(rows)), dim3(256), 0, c->stream, (const fe *)v, rows, step, (fe *)out);
    else
        hipLaunchKernelGGL(k_transpose_vector, dim3(gs_grid(rows * cols)), dim3(256), 0, c->stream, (const fe *)v, rows, cols, step,
                           (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_transpose_matrix(gs_ctx *c, const void *m, uint64_t rows, uint64_t cols, void *out) {
    if (!c || !m || !out) return GS_ERR_ARG;
    if (!rows || !cols) return GS_OK;
    hipLaunchKernelGGL(k_transpose_matrix, dim3(gs_grid(rows * cols)), dim3(256), 0, c->stream, (const fe *)m, rows, cols, (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_sub_matrix_from_vectors(gs_ctx *c, const void *const *vecs_host, const void *m, uint32_t rows, uint64_t cols, void *out) {
    if (!c || !vecs_host || !m || !out) return GS_ERR_ARG;
    if (rows == 0) return gs_fail(c, GS_ERR_ARG, "sub_matrix_from_vectors: no rows");
    if (!cols) return GS_OK;
    for (uint32_t base = 0; base < rows; base += GS_MAX_COMBINE) {       // rows are independent: GS_MAX_COMBINE pointers per launch
        const uint32_t r = rows - base < GS_MAX_COMBINE ? rows - base : GS_MAX_COMBINE;
        PtrArgs va;
        for (uint32_t j = 0; j < GS_MAX_COMBINE; j++) va.v[j] = (const fe *)vecs_host[base + (j < r ? j : 0)];
        hipLaunchKernelGGL(k_sub_matrix_from_vectors, dim3(gs_grid(cols), r), dim3(256), 0, c->stream, va, (const fe *)m + (uint64_t)base * cols, cols,
                           (fe *)out + (uint64_t)base * cols);
        GS_LAUNCH_CHECK(c);
    }
    return GS_OK;
}

int gs_eval_poly_at(gs_ctx *c, const void *poly, uint64_t len, const gs_elt *x, gs_elt *out_host) {
    if (!c || !x || !out_host || (!poly && len)) return GS_ERR_ARG;
    if (!len) { memset(out_host, 0, GS_ELT); return GS_OK; }
    // sum_i poly[i] * x^i as a dot product with the power series of x
    void *pw;
    int rc = gs_tmp_alloc(c, len * GS_ELT, &pw);
    if (rc) return rc;
    rc = gs_power_series_dev(c, fe_from_bytes(x), len, (fe *)pw);
    if (!rc) rc = dot_to_host(c, (const fe *)poly, (const fe *)pw, len, out_host);
    gs_tmp_free(c, pw);
    return rc;
}

int gs_interpolate_quartic_batch(gs_ctx *c, const void *xs, const void *ys, uint64_t rows, void *out) {
    CHECK3(c, xs, ys, out);
    if (!rows) return GS_OK;
    hipLaunchKernelGGL(k_quartic_interp_generic, dim3(gs_grid(rows)), dim3(256), 0, c->stream, (const fe *)xs, (const fe *)ys, rows,
                       (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_interpolate_quartic_domain(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *ys, uint64_t rows,
                                  void *out) {
    CHECK3(c, omega, ys, out);
    if (!gs_is_pow2(n) || n < 4 || rows * 4 * step != n) return gs_fail(c, GS_ERR_ARG, "interpolate_quartic_domain: rows*4*step != n");
    fe w = fe_from_bytes(omega);
    const fe *lo, *hi;
    int log_lo;
    int rc = gs_plan_pow_tables(c, w, n, &lo, &hi, &log_lo);
    if (rc) return rc;
    fe zeta = gs_memo_pow(c, w, n / 4);
    fe zeta_inv = fe_mul(fe_mul(zeta, zeta), zeta);  // zeta^3 = zeta^-1
    static const fe inv4 = fe_inv(fe_from_u64(4));      // (a constant of the field: computed once)
    hipLaunchKernelGGL(k_quartic_interp_domain, dim3(gs_grid(rows)), dim3(256), 0, c->stream, (const fe *)ys, rows, step, n, lo, hi,
                       log_lo, gs_log2(n), zeta_inv, inv4, (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

static int fri_fold_launch(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, fe x, const fe *x_dev, void *out) {
    if (!gs_is_pow2(n) || n < 4 || m < 4 || m * step != n) return gs_fail(c, GS_ERR_ARG, "fri_fold: column length * step != n");
    if (column == out) return gs_fail(c, GS_ERR_ARG, "fri_fold: output must not alias the column");
    fe w = fe_from_bytes(omega);
    const fe *lo, *hi;
    int log_lo;
    int rc = gs_plan_pow_tables(c, w, n, &lo, &hi, &log_lo);
    if (rc) return rc;
    fe zeta = gs_memo_pow(c, w, n / 4);
    fe zeta_inv = fe_mul(fe_mul(zeta, zeta), zeta);  // zeta^3 = zeta^-1
    static const fe inv4 = fe_inv(fe_from_u64(4));      // (a constant of the field: computed once)
    gs_traffic(c, (m + m / 4) * GS_ELT, m / 4, "k_fri_fold");
    hipLaunchKernelGGL(k_fri_fold, dim3(gs_grid(m / 4)), dim3(256), 0, c->stream, (const fe *)column, m / 4, step, n, lo, hi, log_lo, gs_log2(n),
                       zeta_inv, inv4, x, x_dev, (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}
int gs_fri_fold(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const gs_elt *x, void *out) {
    if (!c || !omega || !column || !x || !out) return GS_ERR_ARG;
    return fri_fold_launch(c, omega, n, step, column, m, fe_from_bytes(x), nullptr, out);
}
extern "C++" int gs_prng_point_dev(gs_ctx *c, const void *seed32_dev, fe *out_dev);   // hash.hip
int gs_fri_fold_seeded_scaled(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const void *seed32_dev,
                              const gs_elt *scale, void *out);
int gs_fri_fold_seeded(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const void *seed32_dev, void *out) {
    uint8_t one[sizeof(fe)];
    fe_to_bytes(one, fe_one());
    return gs_fri_fold_seeded_scaled(c, omega, n, step, column, m, seed32_dev, one, out);
}
int gs_fri_fold_seeded_scaled(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const void *seed32_dev,
                              const gs_elt *scale, void *out) {
    if (!c || !omega || !column || !seed32_dev || !scale || !out) return GS_ERR_ARG;
    if (!c->fri_x) {   // 16 bytes that live as long as the context: every use is ordered on the context's stream
        int rc = gs_alloc(c, GS_ELT, &c->fri_x);
        if (rc) return rc;
    }
    int rc = gs_prng_point_dev(c, seed32_dev, (fe *)c->fri_x);
    if (rc) return rc;
    return fri_fold_launch(c, omega, n, step, column, m, fe_from_bytes(scale), (const fe *)c->fri_x, out);
}

int gs_fri_fold_at(gs_ctx *c, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const void *x_dev, void *out) {
    if (!c || !omega || !column || !x_dev || !out) return GS_ERR_ARG;
    if ((uintptr_t)x_dev & 15) return gs_fail(c, GS_ERR_ARG, "fri_fold_at: the point must be 16-byte aligned");
    return fri_fold_launch(c, omega, n, step, column, m, fe_one(), (const fe *)x_dev, out);
}

int gs_eval_quartic_batch(gs_ctx *c, const void *polys, uint64_t rows, const gs_elt *x, void *out) {
    CHECK3(c, polys, x, out);
    if (!rows) return GS_OK;
    hipLaunchKernelGGL(k_quartic_eval, dim3(gs_grid(rows)), dim3(256), 0, c->stream, (const fe *)polys, rows, fe_from_bytes(x), (fe *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

}  // extern "C"
