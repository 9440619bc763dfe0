// ntt_mfma.h — radix-256 Stockham passes whose two radix-16 butterfly networks run on the matrix cores (included by ntt.hip, 128-bit field only).
//
// A 16-point DFT over GF(p) is linear, so on the BYTES of its inputs it is one fixed 256 x 256 integer matrix:
//     acc[(q, r)] = sum over (e, t) of  digit_r( w16^(q e) * 2^(8t) mod p ) * byte_t(x_e)          q, e, r, t in 0..15
// with the constants written in balanced base-256 digits (every entry fits an i8) and the data bytes recoded to signed by ^ 0x80
// (x - 128 * 0x0101..01: by linearity that only shifts output q = 0 by a constant).  Each accumulator is a sum of 256 products of at
// most 2^14: exact in i32.  v_mfma_i32_32x32x32_i8 produces 32 rows of the matrix for 32 independent 16-point groups per instruction;
// 64 of them are one network over the 512 elements a wave holds.  What stays on the vector ALU per element and network: one carry/fold
// pass over 16 accumulators (mf_norm_weak, ~35 instructions) and the per-lane twiddle products between networks (fe_mul) — against
// ~17/16 modular products + 4 levels of five-limb butterflies per element for the register network of k_ntt_wave.
// The matrix entry depends on (q e mod 16, r, t) only: the whole operand table is 4 KB and lives in LDS.
//
// STATUS: opt-in (GSTARK_NTT_MFMA=1).  Byte-identical to the default kernels (tests/test_gpu_parity.py), but not faster: measured
// on MI355X at 2^24 the three passes take 218 / 295 / 309 us against 179 / 264 / 276 us for k_ntt_wave.  The networks this design
// moves to the matrix pipe were ~45 % of a pass; the five per-lane twiddle layers (~100 vector instructions per product) are the
// other half and stay; an MFMA costs the vector ALU ~40 % of its own duration in issue slots (tools/mfma_rate.hip), and the
// workgroup-shared table + exchange buffer leave 3 waves per SIMD.  DESIGN.md section 3.1 has the counters.
//
// Lane roles (lane l: half g = l >> 5, column j5 = l & 31).  B operand of MFMA (u, s): lane (g, j5) supplies the 16 bytes of element
// e = 2s + g of group j5.  A operand: lane (g, rho = j5) supplies row rho of block u for the same element slot, where row rho stands
// for output q = 2u + h(rho), digit r(rho), h = (rho >> 2) & 1, r = (rho & 3) + 4 (rho >> 3) — chosen so that the D layout
// (lane half g', register i holds row (i & 3) + 8 (i >> 2) + 4 g') hands lane (g', j5) ALL 16 digit sums of output q = 2u + g' of its
// own group j5 in its 16 accumulator registers.  A tile is 2 adjacent columns x 256 points = 32 groups: one wave, 8 elements per lane.
#pragma once

typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));

struct MfPassArgs {
    uint64_t n, in_len, in_stride, out_stride;
    int logn, logNs, log_lo;
    int weak;                 // not the last pass: any representative below 2^128 may be stored
    const fe *tw_lo, *tw_hi, *twp;
    const fe *wR;             // omega_256^e, e < 256 (times 1/n on the last pass of an inverse transform)
    const int4 *atab;         // [c][r] -> 16 bytes over t: digit_r(omega_16^c * 2^(8t) mod p), balanced digits
    int offs[16];             // accumulator start values: digits of a multiple of p that keeps every sum non-negative
    fe bias0;                 // what the ^ 0x80 recode takes from output q = 0 of a network: 128 * 16 * sum_t 2^(8t) mod p
};

__device__ __forceinline__ uint64_t mf_mad(uint32_t a, uint32_t k, uint32_t lo) {   // a * k + lo
    uint64_t d, carry;
    const uint64_t c = lo;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(k), "v"(c));
    return d;
}

// 16 digit sums a[i] (a[0] in [2^22, 2^24], the others in [0, 2^23 + 2^22]; value sum a[i] 2^(8i)) -> the value mod p as ANY
// representative below 2^128.  The top limb goes first: what it carries past 2^128 (e < 2^16) is folded into the low limbs before
// their carry chain (e * 2^128 == 9e * 2^32 - e); the chain is one v_mad_u64_u32 per limb; the last carry is 0 or 1.
__device__ __forceinline__ fe mf_norm_weak(const mf_v16i &acc) {
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (uint32_t)acc[i];
    const uint64_t L3 = mf_mad((a[15] << 8) + a[14], 65536u, (a[13] << 8) + a[12]);
    const uint32_t e = (uint32_t)(L3 >> 32);
    a[4] += 9u * e;
    a[0] -= e;
    uint32_t r[4], cin = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t p0 = (a[4 * k + 1] << 8) + a[4 * k] + cin;
        const uint32_t p1 = (a[4 * k + 3] << 8) + a[4 * k + 2];
        const uint64_t L = mf_mad(p1, 65536u, p0);
        r[k] = (uint32_t)L;
        cin = (uint32_t)(L >> 32);
    }
    uint32_t c, b;
    r[3] = gf_addc((uint32_t)L3, cin, 0u, c);
    // c = 1: the sum passed 2^128 and what is left is tiny (r3 < 2^17): + (9 * 2^32 - 1) cannot pass it again
    fe w;
    w.w0 = gf_subc(r[0], c, 0u, b);
    w.w1 = gf_addc(r[1], 9u * c - b, 0u, c);
    w.w2 = gf_addc(r[2], 0u, c, c);
    w.w3 = r[3] + c;
    return w;
}

// x < 2^128 -> x or x - p, whichever is canonical; `keep` (lane-uniform) leaves x as it is.  Branch-free like lz_pack_flag.
__device__ __forceinline__ fe mf_canonical(const fe &x, int keep) {
    uint32_t c;
    fe t;
    t.w0 = gf_addc(x.w0, 0xFFFFFFFFu, 0u, c);
    t.w1 = gf_addc(x.w1, 8u, c, c);
    t.w2 = gf_addc(x.w2, 0u, c, c);
    t.w3 = gf_addc(x.w3, 0u, c, c);
    const bool take = c && !keep;
    fe o;
    o.w0 = take ? t.w0 : x.w0;
    o.w1 = take ? t.w1 : x.w1;
    o.w2 = take ? t.w2 : x.w2;
    o.w3 = take ? t.w3 : x.w3;
    return o;
}

// x + k for x < 2^128 and a canonical constant k: any representative below 2^128 (a carry out of 2^128 comes back as 9 * 2^32 - 1; the
// sum was below 2^128 + p, so that cannot carry again)
__device__ __forceinline__ fe mf_add_weak(const fe &x, const fe &k) {
    uint32_t c, b;
    fe w;
    w.w0 = gf_addc(x.w0, k.w0, 0u, c);
    w.w1 = gf_addc(x.w1, k.w1, c, c);
    w.w2 = gf_addc(x.w2, k.w2, c, c);
    w.w3 = gf_addc(x.w3, k.w3, c, c);
    const uint32_t nine = 9u * c;
    w.w0 = gf_subc(w.w0, c, 0u, b);
    w.w1 = gf_addc(w.w1, nine - b, 0u, c);
    w.w2 = gf_addc(w.w2, 0u, c, c);
    w.w3 += c;
    return w;
}

__device__ __forceinline__ mf_v4i mf_recode(const fe &x) {
    return mf_v4i{(int)(x.w0 ^ 0x80808080u), (int)(x.w1 ^ 0x80808080u), (int)(x.w2 ^ 0x80808080u), (int)(x.w3 ^ 0x80808080u)};
}

// element (column jj, first index a, second index b) of a wave's exchange buffer.  The XOR spreads both access patterns over the
// 16-byte slots of a bank row: the 16 readers of one ds_read_b128 lane group differ in (jj, a) and share b, the 8 writers of one
// ds_write_b128 lane group differ in (jj, b mod 4) and share a
__device__ __forceinline__ int mf_slot(int jj, int a, int b) {
    const int phi = (a & 1) | (((a >> 2) & 3) << 1);
    return ((jj * 16 + a) << 4) + (b ^ ((jj << 2) | (phi & 3) | ((phi >> 2) << 3)));
}

#define MF_WAVES 4               // waves (= tiles) per workgroup
#define MF_COLS 2                // adjacent columns per tile: a wave holds 2 x 256 elements, 8 per lane
#define MF_LDS_BYTES (4096 + MF_WAVES * MF_COLS * 256 * 16)

// the eight A operands of output block u: lane (g, rho) reads row r(rho) of the 16 x 16 byte block of omega_16^(q e), q = 2u + h, e = 2s + g
__device__ __forceinline__ void mf_load_a(mf_v4i (&A)[8], const int4 *tab, int q, int g, int r) {
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int4 a4 = tab[(((q * (2 * s + g)) & 15) << 4) + r];
        A[s] = mf_v4i{a4.x, a4.y, a4.z, a4.w};
    }
}

__device__ __forceinline__ mf_v16i mf_chain(const mf_v4i (&A)[8], const mf_v4i (&B)[8], const mf_v16i &C0) {
    mf_v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], B[0], C0, 0, 0, 0);
#pragma unroll
    for (int s = 1; s < 8; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s], B[s], acc, 0, 0, 0);
    return acc;
}

// One radix-16 network over the 32 groups of a wave: eight output blocks; the operand reads of block u + 1 are put in flight right behind
// the MFMA chain of block u, so that the LDS latency runs under the carry/twiddle arithmetic of block u (the other waves of the SIMD fill
// the time a wave waits for its own chain: issuing chain u + 1 before the vector work of block u was measured, and costs more in registers
// — 2 waves per SIMD instead of 3 — than it hides).  pre(u, u & 1) may start loads block u's `out` will need;
// out(u, y, u & 1) receives output q = 2u + g of the lane's group as a weak residue (the parity is a literal at every call).
template <class Pre, class Out>
__device__ __forceinline__ void mf_network(const int4 *tab, const mf_v4i (&B)[8], const mf_v16i &C0, int g, int h, int r, const fe &bias_g, Pre pre,
                                           Out out) {
    mf_v4i A[8];
    mf_load_a(A, tab, h, g, r);
    pre(0, 0);
#pragma unroll 1
    for (int u = 0; u < 8; u += 2) {
        __builtin_amdgcn_sched_barrier(0);
        mf_v16i acc = mf_chain(A, B, C0);                   // block u
        __builtin_amdgcn_sched_barrier(0);
        mf_load_a(A, tab, 2 * (u + 1) + h, g, r);           // the operands of block u + 1 travel while block u is finished
        pre(u + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        {
            fe y = mf_norm_weak(acc);
            if (u == 0) y = mf_add_weak(y, bias_g);         // wave-uniform branch: output q = 0 lives in block 0 (lanes g = 0; the others add 0)
            out(u, y, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = mf_chain(A, B, C0);                           // block u + 1
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < 8) { mf_load_a(A, tab, 2 * (u + 2) + h, g, r); pre(u + 2, 0); }
        __builtin_amdgcn_sched_barrier(0);
        out(u + 1, mf_norm_weak(acc), 1);
    }
}

// TW: 0 = first pass (no input twiddle; may zero-extend), 1 = input twiddles from the [k][jq] table, 2 = from the power tables + a running product
template <int TW>
__global__ __launch_bounds__(64 * MF_WAVES, 3) void k_ntt_mfma(const fe *__restrict__ in, fe *__restrict__ out, MfPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mf_lds[];
    int4 *tab = reinterpret_cast<int4 *>(mf_lds);
    const int4 tab_word = a.atab[threadIdx.x];        // stored (and the workgroup synchronised) once the tile's own loads are in flight
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    fe *buf = reinterpret_cast<fe *>(mf_lds + 4096) + wv * (MF_COLS * 256);
    const int g = l >> 5, j5 = l & 31, jj = j5 & 1, k1 = j5 >> 1;      // stage A: group (jj, k1); stage B: group (jj, qa = k1)
    const int h = (j5 >> 2) & 1, r = (j5 & 3) + 4 * (j5 >> 3);
    // workgroups go to the 8 XCDs round-robin: give each XCD one contiguous eighth of the columns, so that the lines its L2 holds are
    // consecutive (every channel of that L2 in use) instead of every eighth 128-byte line
    const uint32_t nb = gridDim.x, wg = (nb & 7) ? blockIdx.x : (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
    const uint64_t tile = (uint64_t)wg * MF_WAVES + wv;
    const uint64_t j = tile * MF_COLS + jj;
    const uint64_t nR = a.n >> 8;
    const fe *src = in + (uint64_t)blockIdx.y * a.in_stride;
    fe *dst = out + (uint64_t)blockIdx.y * a.out_stride;
    const uint64_t Ns = 1ull << a.logNs, jq = j & (Ns - 1);
    mf_v16i C0;
#pragma unroll
    for (int i = 0; i < 16; i++) C0[i] = a.offs[i];
    const fe bias_g = g ? fe_zero() : a.bias0;

    // ---- stage A inputs: element m = 2s + g of group (jj, k1)
    mf_v4i B[8];
    fe x0 = fe_zero();
    const bool pruned = TW == 0 && a.in_len <= nR * 16;   // low-degree extension by >= 16x: only m = 0 is in range, its transform is constant
    if (pruned) {
        const uint64_t idx = j + (uint64_t)k1 * nR;
        x0 = src[idx < a.in_len ? idx : 0];
        if (idx >= a.in_len) x0 = fe_zero();
        tab[threadIdx.x] = tab_word;
        __syncthreads();
    } else {
        fe cur, step;
        if constexpr (TW == 2) {
            const uint64_t eu = a.n >> (a.logNs + 8);
            cur = pow_lookup(a.tw_lo, a.tw_hi, a.log_lo, a.logn, jq * (uint64_t)(k1 + 16 * g) * eu);
            step = pow_lookup(a.tw_lo, a.tw_hi, a.log_lo, a.logn, jq * 32 * eu);
        }
        fe raw[8];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const uint64_t idx = j + (uint64_t)(k1 + 16 * (2 * s + g)) * nR;
            if (TW != 0 || a.in_len >= a.n) raw[s] = src[idx];
            else { raw[s] = src[idx < a.in_len ? idx : 0]; if (idx >= a.in_len) raw[s] = fe_zero(); }
        }
        tab[threadIdx.x] = tab_word;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; s++) {
            fe x = raw[s];
            if constexpr (TW == 1) x = fe_mul(x, a.twp[((uint64_t)(k1 + 16 * (2 * s + g)) << a.logNs) + jq]);
            if constexpr (TW == 2) { x = fe_mul(x, cur); if (s < 7) cur = fe_mul(cur, step); }
            B[s] = mf_recode(x);
        }
    }

    // ---- network A (over m) for every group, exchange twiddle omega_256^(k1 * qa), into the exchange buffer as [jj][qa][k1]
    {
        fe w[2];                                             // twiddle of block u in w[u & 1], loaded one block ahead
        auto pre = [&](int u, int par) { w[par] = a.wR[(k1 * (2 * u + g)) & 255]; };
        if (pruned) {
#pragma unroll 1
            for (int u = 0; u < 8; u++) buf[mf_slot(jj, 2 * u + g, k1)] = fe_mul(x0, a.wR[(k1 * (2 * u + g)) & 255]);
        } else {
            mf_network(tab, B, C0, g, h, r, bias_g, pre, [&](int u, const fe &y, int par) { buf[mf_slot(jj, 2 * u + g, k1)] = fe_mul(y, w[par]); });
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stage B: group (jj, qa = k1), element index 2s + g
#pragma unroll
    for (int s = 0; s < 8; s++) B[s] = mf_recode(buf[mf_slot(jj, k1, 2 * s + g)]);

    fe *o = dst + (j - jq) * 256 + jq;
    mf_network(tab, B, C0, g, h, r, bias_g, [](int, int) {}, [&](int u, const fe &y, int) { o[(uint64_t)(k1 + 16 * (2 * u + g)) << a.logNs] = mf_canonical(y, a.weak); });
}

// ---- host side: the operand table and the offset digits of a plan --------------------------------------------------------------
// balanced base-256 digits (each in [-128, 127]) of the representative of v (mod p) that fits 16 of them: v itself, or v - p
static void mf_balanced_digits(fe v, int8_t d[16]) {
    for (int attempt = 0; attempt < 2; attempt++) {
        const uint32_t w[4] = {v.w0, v.w1, v.w2, v.w3};
        int carry = 0;
        for (int i = 0; i < 16; i++) {
            int t = (int)((w[i >> 2] >> (8 * (i & 3))) & 0xFF) + carry;
            if (t >= 128) { t -= 256; carry = 1; } else carry = 0;
            d[i] = (int8_t)t;
        }
        if (!carry) { if (attempt == 0) return; break; }   // attempt 1 must end with the carry (it stands for the - 2^128)
        if (attempt == 1) return;
        // v + (2^128 - p) < 2^128: its digits with the final carry dropped are those of v - p
        uint32_t c;
        fe u;
        u.w0 = gf_addc(v.w0, 0xFFFFFFFFu, 0u, c);
        u.w1 = gf_addc(v.w1, 8u, c, c);
        u.w2 = gf_addc(v.w2, 0u, c, c);
        u.w3 = gf_addc(v.w3, 0u, c, c);
        v = u;
    }
}

static void mf_host_tables(const fe &w16, int8_t table[4096], int offs[16], fe &bias0) {
    fe wp[16];
    wp[0] = fe_one();
    for (int i = 1; i < 16; i++) wp[i] = fe_mul(wp[i - 1], w16);
    for (int c = 0; c < 16; c++) {
        fe sh = wp[c];                        // omega^c * 2^(8t)
        for (int t = 0; t < 16; t++) {
            int8_t d[16];
            mf_balanced_digits(sh, d);
            for (int r = 0; r < 16; r++) table[(c * 16 + r) * 16 + t] = d[r];
            sh = fe_mul(sh, fe_from_u64(256));
        }
    }
    // offsets b_i + byte_i(delta): sum b_i 2^(8i) + delta == target (mod p), delta < p < 2^128
    auto base = [](int i) { return i == 0 ? (1u << 23) + (1u << 20) : (1u << 22) + (1u << 20); };
    fe Bm = fe_zero(), S = fe_zero(), sh = fe_one();
    for (int i = 15; i >= 0; i--) Bm = fe_add(fe_mul(Bm, fe_from_u64(256)), fe_from_u64(base(i)));
    for (int t = 0; t < 16; t++) { S = fe_add(S, sh); sh = fe_mul(sh, fe_from_u64(256)); }
    bias0 = fe_mul(S, fe_from_u64(2048));            // 128 * 16 * sum_t 2^(8t): what the ^ 0x80 recode takes from output 0
    const fe d0 = fe_sub(fe_zero(), Bm);
    const uint32_t w0[4] = {d0.w0, d0.w1, d0.w2, d0.w3};
    for (int i = 0; i < 16; i++) offs[i] = (int)(base(i) + ((w0[i >> 2] >> (8 * (i & 3))) & 0xFF));
}
