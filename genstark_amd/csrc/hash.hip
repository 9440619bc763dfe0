// hash.hip — SHA-256 / BLAKE2s-256 leaf, row and node hashing and Merkle tree construction on gfx950.
//
// Replaces Hash.digest / mergeVectorRows / digestValues and MerkleTree.create of @guildofweavers/merkle
// (call sites: lib/Stark.ts:115,118; lib/components/LowDegreeProver.ts:45-46,163-164,201-202;
// lib/utils/index.ts:37).  One thread per digest: the message words are loaded with 128-bit accesses,
// the compression runs entirely in registers (rounds fully unrolled so the message schedule indices
// are compile-time constants), the 32-byte digest is stored as two 128-bit words.
#include "common.h"

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

// Digest of a message made of `nwords16` 16-byte words; word16(i) returns the i-th one (little-endian
// memory order).  ALG: 0 = sha256, 1 = blake2s256.  Output: 8 x u32 in memory (byte) order.
template <int ALG, typename Loader>
__device__ __forceinline__ void digest_words16(Loader word16, uint32_t nwords16, uint32_t out[8]) {
    const uint32_t len = nwords16 * 16;
    if (ALG == 1) {
        uint32_t h[8] = {0x6A09E667u ^ 0x01010020u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                         0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        const uint32_t nblocks = nwords16 ? (nwords16 + 3) / 4 : 1;
        for (uint32_t blk = 0; blk < nblocks; blk++) {
            uint32_t m[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t wi = blk * 4 + q;
                uint4 x = wi < nwords16 ? word16(wi) : make_uint4(0, 0, 0, 0);
                m[4 * q] = x.x; m[4 * q + 1] = x.y; m[4 * q + 2] = x.z; m[4 * q + 3] = x.w;
            }
            const bool last = blk == nblocks - 1;
            b2s_compress(h, m, last ? len : (blk + 1) * 64, last);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = h[i];
    } else {
        uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
        // len is a multiple of 16, so the 0x80 pad byte starts a fresh 16-byte word and the 64-bit length
        // sits in the last 8 bytes of the final block
        const uint32_t nblocks = (len + 9 + 63) / 64;
        for (uint32_t blk = 0; blk < nblocks; blk++) {
            uint32_t w[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t wi = blk * 4 + q;
                uint4 x = make_uint4(0, 0, 0, 0);
                if (wi < nwords16) x = word16(wi);
                else if (wi == nwords16) x.x = 0x80u;  // byte 0x80 then zeros (little-endian load order)
                w[4 * q] = bswap32(x.x); w[4 * q + 1] = bswap32(x.y); w[4 * q + 2] = bswap32(x.z); w[4 * q + 3] = bswap32(x.w);
            }
            if (blk == nblocks - 1) { w[14] = 0; w[15] = len * 8; }
            sha256_compress(h, w);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = bswap32(h[i]);
    }
}

__device__ __forceinline__ void store_digest(uint4 *out, uint64_t i, const uint32_t d[8]) {
    out[2 * i] = make_uint4(d[0], d[1], d[2], d[3]);
    out[2 * i + 1] = make_uint4(d[4], d[5], d[6], d[7]);
}

struct HashPtrArgs {
    const uint4 *v[GS_MAX_COMBINE];
};

// mergeVectorRows: digest i = H(v_0[i] || v_1[i] || ...): each column load is coalesced across lanes
template <int ALG>
__global__ void k_hash_merge_rows(HashPtrArgs va, uint32_t count, uint64_t n, uint4 *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t d[8];
        digest_words16<ALG>([&](uint32_t w) { return va.v[w][i]; }, count, d);
        store_digest(out, i, d);
    }
}
// single-column fast path (MiMC: one register): no pointer-table indirection
template <int ALG>
__global__ void k_hash_merge_rows1(const uint4 *__restrict__ v, uint64_t n, uint4 *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t d[8];
        digest_words16<ALG>([&](uint32_t) { return v[i]; }, 1, d);
        store_digest(out, i, d);
    }
}
// digestValues / Merkle level: digest i = H(buf[i*vs .. (i+1)*vs)), vs = 16*w16
template <int ALG>
__global__ void k_hash_values(const uint4 *__restrict__ buf, uint32_t w16, uint64_t count, uint4 *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 *p = buf + i * w16;
        uint32_t d[8];
        digest_words16<ALG>([&](uint32_t w) { return p[w]; }, w16, d);
        store_digest(out, i, d);
    }
}
// 64-byte messages (FRI rows, Merkle nodes): fully unrolled 4-word loader
template <int ALG>
__global__ void k_hash_values64(const uint4 *__restrict__ buf, uint64_t count, uint4 *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 *p = buf + i * 4;
        uint4 x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
        uint32_t d[8];
        digest_words16<ALG>([&](uint32_t w) { return w == 0 ? x0 : (w == 1 ? x1 : (w == 2 ? x2 : x3)); }, 4, d);
        store_digest(out, i, d);
    }
}
// top of the tree: nodes[1 .. 2*width) for width <= 256 computed by one workgroup through LDS
template <int ALG>
__global__ void k_merkle_top(uint4 *__restrict__ nodes, uint32_t width) {
    __shared__ uint4 sh[2 * 512];  // sh[2*i], sh[2*i+1] = digest of heap node i (i < 512)
    // load level [width, 2*width)
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) {
        sh[2 * (width + i)] = nodes[2 * (uint64_t)(width + i)];
        sh[2 * (width + i) + 1] = nodes[2 * (uint64_t)(width + i) + 1];
    }
    __syncthreads();
    for (uint32_t w = width / 2; w >= 1; w >>= 1) {
        for (uint32_t i = threadIdx.x; i < w; i += blockDim.x) {
            const uint32_t node = w + i;
            uint4 x0 = sh[4 * node], x1 = sh[4 * node + 1], x2 = sh[4 * node + 2], x3 = sh[4 * node + 3];
            uint32_t d[8];
            digest_words16<ALG>([&](uint32_t k) { return k == 0 ? x0 : (k == 1 ? x1 : (k == 2 ? x2 : x3)); }, 4, d);
            sh[2 * node] = make_uint4(d[0], d[1], d[2], d[3]);
            sh[2 * node + 1] = make_uint4(d[4], d[5], d[6], d[7]);
            store_digest(nodes, node, d);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { nodes[0] = make_uint4(0, 0, 0, 0); nodes[1] = make_uint4(0, 0, 0, 0); }
}

template <int ALG>
static int hash_values_launch(gs_ctx *c, const void *buf, uint64_t value_size, uint64_t count, void *out) {
    if (!count) return GS_OK;
    if (value_size == 64)
        hipLaunchKernelGGL(k_hash_values64<ALG>, dim3(gs_grid(count)), dim3(256), 0, c->stream, (const uint4 *)buf, count, (uint4 *)out);
    else
        hipLaunchKernelGGL(k_hash_values<ALG>, dim3(gs_grid(count)), dim3(256), 0, c->stream, (const uint4 *)buf,
                           (uint32_t)(value_size / 16), count, (uint4 *)out);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

static int check_alg(gs_ctx *c, gs_hash_alg alg) {
    if (alg != GS_HASH_SHA256 && alg != GS_HASH_BLAKE2S256) return gs_fail(c, GS_ERR_ARG, "unknown hash algorithm %d", (int)alg);
    return GS_OK;
}

extern "C" {

int gs_hash_digest_values(gs_ctx *c, gs_hash_alg alg, const void *buf, uint64_t value_size, uint64_t count, void *out) {
    if (!c || !buf || !out) return GS_ERR_ARG;
    int rc = check_alg(c, alg);
    if (rc) return rc;
    if (value_size == 0 || value_size % 16 || value_size > (1u << 20))
        return gs_fail(c, GS_ERR_UNSUPPORTED, "digest_values: value size must be a non-zero multiple of 16 bytes");
    if (((uintptr_t)buf | (uintptr_t)out) & 15) return gs_fail(c, GS_ERR_ARG, "digest_values: buffers must be 16-byte aligned");
    return alg == GS_HASH_SHA256 ? hash_values_launch<0>(c, buf, value_size, count, out) : hash_values_launch<1>(c, buf, value_size, count, out);
}

int gs_hash_merge_rows(gs_ctx *c, gs_hash_alg alg, const void *const *vecs_host, uint32_t count, uint64_t n, void *out) {
    if (!c || !vecs_host || !out) return GS_ERR_ARG;
    int rc = check_alg(c, alg);
    if (rc) return rc;
    if (count == 0 || count > GS_MAX_COMBINE) return gs_fail(c, GS_ERR_ARG, "hash_merge_rows: count must be in 1..%d", GS_MAX_COMBINE);
    if (!n) return GS_OK;
    if (count == 1) {
        if (alg == GS_HASH_SHA256)
            hipLaunchKernelGGL(k_hash_merge_rows1<0>, dim3(gs_grid(n)), dim3(256), 0, c->stream, (const uint4 *)vecs_host[0], n, (uint4 *)out);
        else
            hipLaunchKernelGGL(k_hash_merge_rows1<1>, dim3(gs_grid(n)), dim3(256), 0, c->stream, (const uint4 *)vecs_host[0], n, (uint4 *)out);
    } else {
        HashPtrArgs va;
        for (uint32_t j = 0; j < GS_MAX_COMBINE; j++) va.v[j] = (const uint4 *)vecs_host[j < count ? j : 0];
        if (alg == GS_HASH_SHA256)
            hipLaunchKernelGGL(k_hash_merge_rows<0>, dim3(gs_grid(n)), dim3(256), 0, c->stream, va, count, n, (uint4 *)out);
        else
            hipLaunchKernelGGL(k_hash_merge_rows<1>, dim3(gs_grid(n)), dim3(256), 0, c->stream, va, count, n, (uint4 *)out);
    }
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_merkle_build(gs_ctx *c, gs_hash_alg alg, const void *leaves, uint64_t n, void *nodes) {
    if (!c || !leaves || !nodes) return GS_ERR_ARG;
    int rc = check_alg(c, alg);
    if (rc) return rc;
    if (!gs_is_pow2(n) || n < 2) return gs_fail(c, GS_ERR_ARG, "merkle_build: n must be a power of two >= 2");
    uint4 *nd = (uint4 *)nodes;
    // bottom level: parents of leaf pairs -> nodes[n/2 .. n)
    if (alg == GS_HASH_SHA256) rc = hash_values_launch<0>(c, leaves, 64, n / 2, nd + 2 * (n / 2));
    else rc = hash_values_launch<1>(c, leaves, 64, n / 2, nd + 2 * (n / 2));
    if (rc) return rc;
    // wide levels: one streaming launch each; nodes[w .. 2w) <- pairs of nodes[2w .. 4w)
    uint64_t w = n / 4;
    for (; w >= 256; w >>= 1) {
        if (alg == GS_HASH_SHA256) rc = hash_values_launch<0>(c, nd + 2 * (2 * w), 64, w, nd + 2 * w);
        else rc = hash_values_launch<1>(c, nd + 2 * (2 * w), 64, w, nd + 2 * w);
        if (rc) return rc;
    }
    // the last <= 8 levels in one workgroup: level [have, 2*have) is complete, have <= 256
    uint32_t have = (uint32_t)(2 * w);
    if (have < 1) have = 1;
    if (have > n / 2) have = (uint32_t)(n / 2);
    if (alg == GS_HASH_SHA256) hipLaunchKernelGGL(k_merkle_top<0>, dim3(1), dim3(256), 0, c->stream, nd, have);
    else hipLaunchKernelGGL(k_merkle_top<1>, dim3(1), dim3(256), 0, c->stream, nd, have);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

int gs_hash_digest(gs_ctx *c, gs_hash_alg alg, const uint8_t *msg_host, uint64_t len, uint8_t out_host[32]) {
    if (!c || (!msg_host && len) || !out_host) return GS_ERR_ARG;
    int rc = check_alg(c, alg);
    if (rc) return rc;
    if (len == 0 || len % 16 || len > 4096)
        return gs_fail(c, GS_ERR_UNSUPPORTED, "hash_digest: message length must be a non-zero multiple of 16 bytes (<= 4096)");
    uint64_t off = (len + 255) & ~(uint64_t)255;
    if ((rc = gs_stage_reserve(c, off + 32))) return rc;
    memcpy(c->h_stage, msg_host, len);
    GS_HIP(c, hipMemcpyAsync(c->d_stage, c->h_stage, len, hipMemcpyHostToDevice, c->stream));
    uint8_t *d_out = (uint8_t *)c->d_stage + off;
    if (alg == GS_HASH_SHA256) rc = hash_values_launch<0>(c, c->d_stage, len, 1, d_out);
    else rc = hash_values_launch<1>(c, c->d_stage, len, 1, d_out);
    if (rc) return rc;
    uint8_t *h_out = (uint8_t *)c->h_stage + off;
    GS_HIP(c, hipMemcpyAsync(h_out, d_out, 32, hipMemcpyDeviceToHost, c->stream));
    GS_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(out_host, h_out, 32);
    return GS_OK;
}

}  // extern "C"
