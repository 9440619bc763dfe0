// hash_core.h — BLAKE2s-256 and SHA-256 compression functions for gfx950 device code: everything in registers,
// rounds fully unrolled so the message-schedule indices are compile-time constants.  Used by hash.hip (leaf / row /
// node hashing, Merkle construction) and tools/microbench_hash.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ------------------------------------------------------------------------------------------- BLAKE2s
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int r) { return __builtin_rotateright32(x, r); }

#define B2S_G(a, b, c, d, x, y)        \
    do {                               \
        a = a + b + (x);               \
        d = rotr32(d ^ a, 16);         \
        c = c + d;                     \
        b = rotr32(b ^ c, 12);         \
        a = a + b + (y);               \
        d = rotr32(d ^ a, 8);          \
        c = c + d;                     \
        b = rotr32(b ^ c, 7);          \
    } while (0)

#define B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    B2S_G(v0, v4, v8, v12, m[s0], m[s1]);                                                 \
    B2S_G(v1, v5, v9, v13, m[s2], m[s3]);                                                 \
    B2S_G(v2, v6, v10, v14, m[s4], m[s5]);                                                \
    B2S_G(v3, v7, v11, v15, m[s6], m[s7]);                                                \
    B2S_G(v0, v5, v10, v15, m[s8], m[s9]);                                                \
    B2S_G(v1, v6, v11, v12, m[s10], m[s11]);                                              \
    B2S_G(v2, v7, v8, v13, m[s12], m[s13]);                                               \
    B2S_G(v3, v4, v9, v14, m[s14], m[s15]);

// one compression; t = bytes hashed so far including this block; last = final block
__device__ __forceinline__ void b2s_compress(uint32_t h[8], const uint32_t m[16], uint32_t t, bool last) {
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
    uint32_t v12 = 0x510E527Fu ^ t, v13 = 0x9B05688Cu, v14 = last ? ~0x1F83D9ABu : 0x1F83D9ABu, v15 = 0x5BE0CD19u;
    B2S_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2S_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2S_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2S_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2S_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2S_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2S_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2S_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2S_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2S_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    h[0] ^= v0 ^ v8;  h[1] ^= v1 ^ v9;  h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// ---- one 64-byte block (a Merkle node: two child digests) compressed by FOUR lanes: lane l of a quad holds column l of the 4 x 4
// state (v[l], v[4+l], v[8+l], v[12+l]); a column step is one G per lane, a diagonal step the same after rotating rows b, c, d by
// 1, 2, 3 lanes (DPP quad_perm, no LDS).  ~420 instructions per lane instead of 989: the LATENCY of a compression drops 2.3x, which is
// what the narrow levels of a tree are bound by (one dependent compression per level).  The 16 message words are read from LDS
// (`m`, the same 64 bytes for the four lanes) by per-lane index: the sigma entries of the four lanes packed into one constant.
// Unkeyed BLAKE2s-256, single final block (t = 64).  Lane l returns digest words l (lo) and 4 + l (hi).
#define B2S_QP(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 4) | ((uint32_t)(c) << 8) | ((uint32_t)(d) << 12))
#define B2S_QDPP(x, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xF, 0xF, true))
#define B2S_QROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                     \
    {                                                                                                         \
        const uint32_t x0 = m[(B2S_QP(s0, s2, s4, s6) >> sh4) & 15u], y0 = m[(B2S_QP(s1, s3, s5, s7) >> sh4) & 15u];          \
        const uint32_t x1 = m[(B2S_QP(s8, s10, s12, s14) >> sh4) & 15u], y1 = m[(B2S_QP(s9, s11, s13, s15) >> sh4) & 15u];    \
        B2S_G(a, b, c, d, x0, y0);                                                                            \
        b = B2S_QDPP(b, 0x39); c = B2S_QDPP(c, 0x4E); d = B2S_QDPP(d, 0x93);     /* lane l takes b[l+1], c[l+2], d[l+3] */  \
        B2S_G(a, b, c, d, x1, y1);                                                                            \
        b = B2S_QDPP(b, 0x93); c = B2S_QDPP(c, 0x4E); d = B2S_QDPP(d, 0x39);     /* and back */               \
    }
__device__ __forceinline__ void b2s_node_quad(const uint32_t *m, int l, uint32_t &lo, uint32_t &hi) {
    const int sh4 = 4 * l;
    const uint32_t iv_lo = l == 0 ? 0x6A09E667u : (l == 1 ? 0xBB67AE85u : (l == 2 ? 0x3C6EF372u : 0xA54FF53Au));
    const uint32_t iv_hi = l == 0 ? 0x510E527Fu : (l == 1 ? 0x9B05688Cu : (l == 2 ? 0x1F83D9ABu : 0x5BE0CD19u));
    const uint32_t h_lo = iv_lo ^ (l == 0 ? 0x01010020u : 0u), h_hi = iv_hi;
    uint32_t a = h_lo, b = h_hi, c = iv_lo, d = iv_hi ^ (l == 0 ? 64u : 0u) ^ (l == 2 ? 0xFFFFFFFFu : 0u);
    B2S_QROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2S_QROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2S_QROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2S_QROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2S_QROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2S_QROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2S_QROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2S_QROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2S_QROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2S_QROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    lo = h_lo ^ a ^ c;
    hi = h_hi ^ b ^ d;
}

// ------------------------------------------------------------------------------------------- SHA-256
__constant__ const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// one compression over 16 big-endian message words (w is clobbered: rolling 16-word schedule)
__device__ __forceinline__ void sha256_compress(uint32_t h[8], uint32_t w[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + SHA_K[i] + w[i & 15];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

