// hash.hip — SHA-256 / BLAKE2s-256 leaf, row and node hashing and Merkle tree construction on gfx950.
//
// Replaces Hash.digest / mergeVectorRows / digestValues and MerkleTree.create of @guildofweavers/merkle
// (call sites: lib/Stark.ts:115,118; lib/components/LowDegreeProver.ts:45-46,163-164,201-202;
// lib/utils/index.ts:37).  One thread per digest: the message words are loaded with 128-bit accesses,
// the compression runs entirely in registers (rounds fully unrolled so the message schedule indices
// are compile-time constants), the 32-byte digest is stored as two 128-bit words.
#include "common.h"

#include "hash_core.h"

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
        digest_words16<ALG>([&](uint32_t w) { return va.v[w / GS_EW][i * GS_EW + w % GS_EW]; }, count * GS_EW, d);
        store_digest(out, i, d);
    }
}
// more than GS_MAX_COMBINE columns: the pointer table sits in device memory (one uniform scalar load per column)
template <int ALG>
__global__ void k_hash_merge_rows_table(const uint4 *const *__restrict__ tab, uint32_t count, uint64_t n, uint4 *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t d[8];
        digest_words16<ALG>([&](uint32_t w) { return tab[w / GS_EW][i * GS_EW + w % GS_EW]; }, count * GS_EW, d);
        store_digest(out, i, d);
    }
}
// single-column fast path (MiMC: one register): no pointer-table indirection
template <int ALG>
__global__ void k_hash_merge_rows1(const uint4 *__restrict__ v, uint64_t n, uint4 *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t d[8];
        digest_words16<ALG>([&](uint32_t w) { return v[i * GS_EW + w]; }, GS_EW, d);
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
// ---------------------------------------------------------------------------------------------------------------------------
// Merkle construction, fused (round 3).  A tree is built by at most a handful of launches instead of one per level:
//   k_merkle_fused   — wide layers, streaming.  One thread owns 2^(lv-1) consecutive digests of a first layer and the lv - 1 layers
//                      of nodes above them (lv <= 4: 8 + 4 + 2 + 1 = 15 compressions), depth first: it keeps one pending digest
//                      per height (in LDS, [height][word][thread]: conflict-free, and indexable — a register array indexed by the
//                      height would live in scratch), so a node is hashed the moment its second child exists and no layer is read
//                      back from memory.  The code is a LOOP around two compression sites (first layer, node): unrolled, fifteen
//                      inlined compressions would be 120 KB of straight-line code per wave against a 64 KB instruction cache.
//   k_merkle_subtree — layers of at most 2^15 digests.  One workgroup owns `chunk` (<= 1024; merkle_run gives it 256) consecutive digests of a layer and the
//                      whole subtree above them, level by level through LDS (heap layout: levels occupy disjoint slots, one barrier
//                      per level); with chunk = layer width it finishes the tree (root at nodes[1], nodes[0] = 0).
// The first layer of either kernel is (SRC 0) a node layer, H(child[2i] || child[2i+1]) of the digests below, or the tree's LEAVES
// hashed from the committed vectors themselves: (1) H(v[i]), (2) H(v_0[i] || ... || v_{count-1}[i]) — mergeVectorRows fused into
// the construction (gs_merkle_commit_rows).
template <int ALG, int SRC>
__device__ __forceinline__ void first_layer_digest(const HashPtrArgs &va, uint32_t count, const uint4 *__restrict__ child, uint64_t i, uint32_t d[8]) {
    if constexpr (SRC == 0) {
        const uint4 *p = child + 4 * i;
        const uint4 x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
        digest_words16<ALG>([&](uint32_t w) { return w == 0 ? x0 : (w == 1 ? x1 : (w == 2 ? x2 : x3)); }, 4, d);
    } else if constexpr (SRC == 1) {
        const uint4 *p = va.v[0] + i * GS_EW;
        digest_words16<ALG>([&](uint32_t w) { return p[w]; }, GS_EW, d);
    } else {
        digest_words16<ALG>([&](uint32_t w) { return va.v[w / GS_EW][i * GS_EW + w % GS_EW]; }, count * GS_EW, d);
    }
}
template <int ALG>
__device__ __forceinline__ void merge_digests(const uint32_t l[8], const uint32_t r[8], uint32_t d[8]) {
    const uint4 x0 = make_uint4(l[0], l[1], l[2], l[3]), x1 = make_uint4(l[4], l[5], l[6], l[7]);
    const uint4 x2 = make_uint4(r[0], r[1], r[2], r[3]), x3 = make_uint4(r[4], r[5], r[6], r[7]);
    digest_words16<ALG>([&](uint32_t w) { return w == 0 ? x0 : (w == 1 ? x1 : (w == 2 ? x2 : x3)); }, 4, d);
}

#define GS_MERKLE_MAX_LV 4
template <int ALG, int SRC>
__global__ __launch_bounds__(256) void k_merkle_fused(HashPtrArgs va, uint32_t count, const uint4 *__restrict__ child, uint64_t groups, int lv,
                                                      uint4 *__restrict__ outA, uint4 *__restrict__ nodes, uint64_t wA) {
    __shared__ uint32_t pending[GS_MERKLE_MAX_LV - 1][8][256];
    const int K = 1 << (lv - 1), tid = threadIdx.x;
    for (uint64_t g = blockIdx.x * (uint64_t)256 + tid; g < groups; g += (uint64_t)gridDim.x * 256) {
        for (int k = 0; k < K; k++) {
            uint64_t idx = g * K + k;
            uint32_t cur[8];
            first_layer_digest<ALG, SRC>(va, count, child, idx, cur);
            store_digest(outA, idx, cur);
            int h = 0;
            for (; (k >> h) & 1; h++) {       // the left sibling at this height is waiting: hash the pair, go one layer up
                uint32_t l[8];
#pragma unroll
                for (int w = 0; w < 8; w++) l[w] = pending[h][w][tid];
                merge_digests<ALG>(l, cur, cur);
                idx >>= 1;
                store_digest(nodes, (wA >> (h + 1)) + idx, cur);
            }
            if (h < lv - 1) {
#pragma unroll
                for (int w = 0; w < 8; w++) pending[h][w][tid] = cur[w];
            }
        }
    }
}

// What the launch that produces a tree's ROOT can do on the side (gs_merkle_commit_rows_seed): hand the root to the host — 32 bytes
// into a slot of the posted-read-back ring, then the slot's flag (common.h: gs_readback_reserve) — and derive the FRI evaluation
// point prng(root) (LowDegreeProver.ts:194) where the next fold will read it.  Two launches fewer per FRI layer.
struct RootTail {
    uint4 *slot;
    unsigned long long *flag;
    unsigned long long value;
    fe *point;
};
__device__ __forceinline__ fe prng_point(const uint32_t *seed);

#define GS_MERKLE_CHUNK 1024
#define GS_MERKLE_SUBTREE_THREADS 1024      // one first-layer digest per thread, the 512-node level in one step, quads from 256 nodes down
// The levels of a subtree whose first layer (2 * first_m digests) sits in LDS at heap slots [2 first_m, 4 first_m): one barrier per level,
// nodes stored at nodes[wl + offset * m + i] (wl: width of the layer being produced, halving; offset: this workgroup's subtree).  Levels
// no wider than a quarter of the workgroup are bound by the latency of ONE compression per level: four lanes per node (hash_core.h:
// b2s_node_quad) cut it 2.3x; quads are wholly active or wholly idle (4 m threads), as the lane rotations need.
template <int ALG>
__device__ __forceinline__ void subtree_levels(uint4 *sh, uint32_t first_m, uint64_t wl, uint64_t offset, uint4 *__restrict__ nodes) {
    for (uint32_t m = first_m; m >= 1; m >>= 1, wl >>= 1) {
        bool done = false;
        if constexpr (ALG == 1) {
            if (m <= blockDim.x / 4) {
                if (threadIdx.x < 4 * m) {
                    const uint32_t i = threadIdx.x >> 2, l = threadIdx.x & 3u, s = m + i;
                    uint32_t *w = reinterpret_cast<uint32_t *>(sh);
                    uint32_t lo, hi;
                    b2s_node_quad(w + 16 * s, (int)l, lo, hi);
                    w[8 * s + l] = lo;
                    w[8 * s + 4 + l] = hi;
                    uint32_t *g = reinterpret_cast<uint32_t *>(nodes) + 8 * (wl + offset * m + i);
                    g[l] = lo;
                    g[4 + l] = hi;
                }
                done = true;
            }
        }
        if (!done) {
            for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
                const uint32_t s = m + i;
                const uint4 x0 = sh[4 * s], x1 = sh[4 * s + 1], x2 = sh[4 * s + 2], x3 = sh[4 * s + 3];
                uint32_t d[8];
                digest_words16<ALG>([&](uint32_t k) { return k == 0 ? x0 : (k == 1 ? x1 : (k == 2 ? x2 : x3)); }, 4, d);
                sh[2 * s] = make_uint4(d[0], d[1], d[2], d[3]);
                sh[2 * s + 1] = make_uint4(d[4], d[5], d[6], d[7]);
                store_digest(nodes, wl + offset * m + i, d);
            }
        }
        __syncthreads();
    }
}

template <int ALG, int SRC>
__global__ __launch_bounds__(GS_MERKLE_SUBTREE_THREADS) void k_merkle_subtree(HashPtrArgs va, uint32_t count, const uint4 *__restrict__ layerA, uint4 *__restrict__ outA,
                                                        uint4 *__restrict__ nodes, uint64_t wA, uint32_t chunk, RootTail tail, unsigned int *counter) {
    __shared__ uint4 sh[4 * GS_MERKLE_CHUNK];   // digest of subtree-heap node s at sh[2s], sh[2s + 1]; the first layer is s in [chunk, 2 chunk)
    const uint64_t base = (uint64_t)blockIdx.x * chunk;
    for (uint32_t i = threadIdx.x; i < chunk; i += blockDim.x) {
        if constexpr (SRC == 0) {
            sh[2 * (chunk + i)] = layerA[2 * (base + i)];
            sh[2 * (chunk + i) + 1] = layerA[2 * (base + i) + 1];
        } else {
            uint32_t d[8];
            first_layer_digest<ALG, SRC>(va, count, nullptr, base + i, d);
            store_digest(outA, base + i, d);
            sh[2 * (chunk + i)] = make_uint4(d[0], d[1], d[2], d[3]);
            sh[2 * (chunk + i) + 1] = make_uint4(d[4], d[5], d[6], d[7]);
        }
    }
    __syncthreads();
    subtree_levels<ALG>(sh, chunk / 2, wA >> 1, blockIdx.x, nodes);
    bool at_root = chunk == wA;
    if (counter && gridDim.x > 1) {
        // Several workgroups and a counter: the tree over their subtree roots is built HERE, by the workgroup that arrives last (k_fri_layers'
        // pattern: agent-scope release, count, acquire — the eight XCD L2s are not coherent with each other; nobody waits for anybody).
        // One launch per tree of <= 2^15 first-layer digests instead of two.
        __shared__ int last_sh;
        const uint32_t G = gridDim.x;                 // a power of two <= 1024
        // (the last level's barrier has passed: every store of this workgroup has reached its XCD's L2.  ONE thread then writes the L2
        //  back — the release is cumulative over what the barrier ordered before it — counts the workgroup, and in the last workgroup
        //  invalidates before anybody loads: sixteen waves each issuing the write-back / invalidate cost 8 us per tree top)
        if (threadIdx.x == 0) {
            __threadfence();
            last_sh = atomicAdd(counter, 1u) == G - 1;
            if (last_sh) __threadfence();
        }
        __syncthreads();
        if (!last_sh) return;
        for (uint32_t i = threadIdx.x; i < G; i += blockDim.x) {                     // heap nodes G .. 2G - 1 of the tree: the subtree roots
            sh[2 * (G + i)] = nodes[2 * (G + (uint64_t)i)];
            sh[2 * (G + i) + 1] = nodes[2 * (G + (uint64_t)i) + 1];
        }
        if (threadIdx.x == 0) atomicExch(counter, 0u);                               // ready for the next launch on this context's stream
        __syncthreads();
        subtree_levels<ALG>(sh, G / 2, G >> 1, 0, nodes);
        at_root = true;
    }
    if (at_root && threadIdx.x == 0) {
        nodes[0] = make_uint4(0, 0, 0, 0);
        nodes[1] = make_uint4(0, 0, 0, 0);
        // the root is heap node 1: sh[2], sh[3] (the last level's barrier has passed)
        if (tail.slot) {
            tail.slot[0] = sh[2];
            tail.slot[1] = sh[3];
            __threadfence_system();
            __hip_atomic_store(tail.flag, tail.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (tail.point) {
            const uint4 r0 = sh[2], r1 = sh[3];
            const uint32_t seed[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            *tail.point = prng_point(seed);
        }
    }
}

#define GS_FRI_MAX_LAYERS 16
// per-context device scratch of the launches whose workgroups count themselves done (k_fri_layers, k_merkle_subtree): the points between
// FRI layers, then 64 zeroed bytes — word 0: gs_fri_layers' arrival counter, word 1: k_merkle_subtree's
static int arrival_counters(gs_ctx *c, unsigned int **counters) {
    if (!c->fri_points) {
        int rc = gs_alloc(c, (GS_FRI_MAX_LAYERS + 1) * GS_ELT + 64, &c->fri_points);
        if (rc) return rc;
        GS_HIP(c, hipMemsetAsync((uint8_t *)c->fri_points + (GS_FRI_MAX_LAYERS + 1) * GS_ELT, 0, 64, c->stream));
    }
    *counters = (unsigned int *)((uint8_t *)c->fri_points + (GS_FRI_MAX_LAYERS + 1) * GS_ELT);
    return GS_OK;
}

// compressions one message of `bytes` bytes costs (gs_traffic: the work unit of the hash kernels): BLAKE2s absorbs 64-byte blocks, the
// last one together with the finalisation; SHA-256 pads with 0x80 and a 64-bit length
static inline uint64_t hash_blocks(int alg, uint64_t bytes) { return alg == 1 ? (bytes + 63) / 64 : (bytes + 9 + 63) / 64; }
#define GS_MERKLE_SUBW (1ull << 15)   // layers at most this wide go to k_merkle_subtree (below it a streaming launch is latency-bound)
// leaves (n digests, given or hashed from `count` columns when src != 0) -> nodes (heap order).  src as SRC above.
template <int ALG>
static int merkle_run(gs_ctx *c, int src, const HashPtrArgs &va, uint32_t count, const void *leaves_in, void *leaves_out, uint64_t n, uint4 *nd,
                      const RootTail *root_tail = nullptr) {
    const RootTail none = {nullptr, nullptr, 0, nullptr};
    // the launch that ends at the root: a subtree launch of one workgroup, or of several that finish the tree themselves (arrival counter)
    auto tail_for = [&](uint64_t width, uint32_t chunk) { return root_tail && (width == chunk || width / chunk <= GS_MERKLE_CHUNK) ? *root_tail : none; };
    // digests per workgroup of a subtree launch: 256, not the 1 024 the kernel can take — every level of a subtree then runs at four lanes
    // per node on at most two waves per SIMD (one compression deep: ~0.7-1.3 us), where a 1 024-digest subtree pays 3.8 + 2.5 us for its
    // two widest levels on one CU; the levels moved to the tree over the subtree roots cost the same there.  Trees back to back
    // (tools/merkle_top.py, us at 1 024 / 512 / 256 / 128 per workgroup): 2^10 digests 17.0 / 16.0 / 14.4 / 14.1, 2^13 22.2 / 19.6 /
    // 17.8 / 17.5, 2^15 24.3 / 21.8 / 21.3 / 24.0, 2^18 41.8 / 39.3 / 38.3 / 41.3
    const uint32_t CHUNK_X = 256;
    unsigned int *counters = nullptr;
    if (n > CHUNK_X) {
        int rc = arrival_counters(c, &counters);
        if (rc) return rc;
        counters += 1;                           // (word 0 is gs_fri_layers')
    }
    const dim3 blk(256), sub_blk(GS_MERKLE_SUBTREE_THREADS);      // (512 threads per subtree workgroup measure the same as 1 024)
    uint64_t w = n;                              // the widest complete layer and where its digests are
    const uint4 *cur = (const uint4 *)leaves_in;
    if (src != 0) {
        uint4 *lo = (uint4 *)leaves_out;
        if (n > GS_MERKLE_SUBW) {
            // a thread's 2^(lv-1) consecutive leaves are 128-byte-strided loads per column, re-touched on every iteration: with several
            // columns the re-reads saturate the L2 at lv = 4 (2^22 rows x 4 columns: 448 us against 316 us at lv = 2 and 356 us
            // unfused, profiles/r03_b_merkle_fused_depth.txt); one column streams a single line per thread and takes all four layers
            int lv = 1;
            const int maxlv = src == 2 ? 2 : GS_MERKLE_MAX_LV;
            while (lv < maxlv && (n >> lv) >= GS_MERKLE_SUBW) lv++;         // layers n, n/2, ..., n >> (lv - 1)
            const uint64_t groups = n >> (lv - 1);
            {   // n leaves hashed from `count` columns (digests written: 32 B each) + the lv - 1 node layers above them (written only)
                const uint64_t above = n - (n >> (lv - 1));
                gs_traffic(c, n * ((uint64_t)count * GS_ELT + 32) + above * 32, n * hash_blocks(ALG, (uint64_t)count * GS_ELT) + above, "k_merkle_fused<%d, %d>", ALG, src);
            }
            if (src == 1) hipLaunchKernelGGL((k_merkle_fused<ALG, 1>), dim3(gs_grid(groups)), blk, 0, c->stream, va, count, nullptr, groups, lv, lo, nd, n);
            else hipLaunchKernelGGL((k_merkle_fused<ALG, 2>), dim3(gs_grid(groups)), blk, 0, c->stream, va, count, nullptr, groups, lv, lo, nd, n);
            w = n >> (lv - 1);
            cur = lv == 1 ? lo : nd + 2 * w;
        } else {
            const uint32_t chunk = (uint32_t)(n <= CHUNK_X ? n : CHUNK_X);
            gs_traffic(c, n * ((uint64_t)count * GS_ELT + 32) + n * 32, n * hash_blocks(ALG, (uint64_t)count * GS_ELT) + n - 1, "k_merkle_subtree<%d, %d>", ALG, src);
            if (src == 1) hipLaunchKernelGGL((k_merkle_subtree<ALG, 1>), dim3((unsigned)(n / chunk)), sub_blk, 0, c->stream, va, count, nullptr, lo, nd, n, chunk, tail_for(n, chunk), counters);
            else hipLaunchKernelGGL((k_merkle_subtree<ALG, 2>), dim3((unsigned)(n / chunk)), sub_blk, 0, c->stream, va, count, nullptr, lo, nd, n, chunk, tail_for(n, chunk), counters);
            GS_LAUNCH_CHECK(c);                  // (n <= 2^15: at most 128 subtrees, whose roots the last workgroup finishes)
            return GS_OK;
        }
    }
    while (w > GS_MERKLE_SUBW) {                  // node layers w/2, ..., w >> lv from the digests of layer w
        int lv = 1;
        while (lv < GS_MERKLE_MAX_LV && (w >> (lv + 1)) >= GS_MERKLE_SUBW) lv++;
        // (round 5 measured the other choice — fewer layers per launch so that a launch keeps >= 2^17 threads, one thread owning 2^lv digests
        // depth first: trees built back to back gain 10-18 % from 2^19 digests up (tools/merkle_lv.py, profiles/r05_f_*), inside a proof
        // the extra launches cost what the better occupancy gains: C5 3.36 -> 3.36 ms of kernels in 40 launches instead of 35, C4-long
        // 9.53 -> 9.62.  Not adopted; the experiments build keeps the switch)
#ifdef GS_NTT_EXPERIMENTS
        // experiments build only (tools/merkle_lv.py): keep at least 2^K threads per node-layer launch (one thread owns 2^lv digests of
        // layer w, depth first), i.e. fewer layers per launch while the launch would otherwise run at a wave or two per SIMD
        if (const char *e = getenv("GSTARK_MERKLE_NODE_MIN_GROUPS")) {      // 0 = round 4's rule (as many layers as stay above 2^15 digests)
            const int K = atoi(e);
            lv = 1;
            while (lv < GS_MERKLE_MAX_LV && (w >> (lv + 1)) >= GS_MERKLE_SUBW) lv++;
            while (lv > 1 && (w >> lv) < (1ull << K)) lv--;
        }
#endif
        const uint64_t groups = w >> lv;
        gs_traffic(c, w * 32 + (w - (w >> lv)) * 32, w - (w >> lv), "k_merkle_fused<%d, 0>", ALG);      // reads layer w, writes layers w/2 .. w >> lv
        hipLaunchKernelGGL((k_merkle_fused<ALG, 0>), dim3(gs_grid(groups)), blk, 0, c->stream, va, 0u, cur, groups, lv, nd + 2 * (w / 2), nd, w / 2);
        w >>= lv;
        cur = nd + 2 * w;
    }
    while (w > 1) {                               // subtrees of <= 1024 digests, then the subtree over their roots
        const uint32_t chunk = (uint32_t)(w <= CHUNK_X ? w : (w / CHUNK_X <= GS_MERKLE_CHUNK ? CHUNK_X : GS_MERKLE_CHUNK));
        gs_traffic(c, w * 32 + (w - 1) * 32, w - 1, "k_merkle_subtree<%d, 0>", ALG);
        hipLaunchKernelGGL((k_merkle_subtree<ALG, 0>), dim3((unsigned)(w / chunk)), sub_blk, 0, c->stream, va, 0u, cur, nullptr, nd, w, chunk, tail_for(w, chunk), counters);
        w /= chunk;
        cur = nd + 2 * w;
        if (counters && w <= GS_MERKLE_CHUNK) break;       // the last workgroup of that launch has built the tree over the subtree roots
    }
    GS_LAUNCH_CHECK(c);
    return GS_OK;
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

// field.prng(seed) for a 32-byte seed already on the device: sha256(seed) as a big-endian integer, mod p (byte-wise Horner: the same
// code serves every field flavour).  One lane; the point of it is that the host does not have to see the seed.
__device__ __forceinline__ fe prng_point(const uint32_t *seed);
__global__ void k_prng_point(const uint32_t *__restrict__ seed, fe *__restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    *out = prng_point(seed);
}
__device__ __forceinline__ fe prng_point(const uint32_t *seed) {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = bswap32(seed[i]);
    w[8] = 0x80000000u;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 256;                                     // message length in bits
    sha256_compress(h, w);
#if !defined(GS_SMALL_Q) && !defined(GS_WIDE_BITS)
    // the digest as a big-endian integer IS a 256-bit value in eight 32-bit limbs: the fold every product ends with reduces it
    // (2^128 == 9 * 2^32 - 1 mod p) — no multiplication at all, instead of the 32 dependent byte-wise Horner steps of the generic
    // form below (15 -> 4 us on the critical path of every FRI layer)
    const uint32_t r[8] = {h[7], h[6], h[5], h[4], h[3], h[2], h[1], h[0]};
    return fe_reduce_wide(r);
#else
    fe x = fe_zero();
    const fe b = fe_make(256u, 0u, 0u, 0u);
    for (int i = 0; i < 8; i++)
        for (int k = 3; k >= 0; k--) x = fe_add(fe_mul(x, b), fe_make((h[i] >> (8 * k)) & 0xFFu, 0u, 0u, 0u));
    return x;
#endif
}
int gs_prng_point_dev(gs_ctx *c, const void *seed32_dev, fe *out_dev) {
    if (((uintptr_t)seed32_dev) & 3) return gs_fail(c, GS_ERR_ARG, "prng_point: the seed must be 4-byte aligned");
    hipLaunchKernelGGL(k_prng_point, dim3(1), dim3(64), 0, c->stream, (const uint32_t *)seed32_dev, out_dev);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}


// ---- a RUN of FRI layers (gs_fri_layers; LowDegreeProver.ts:176-221) ------------------------------------------------------------------
// One launch per layer, and ONE launch for all layers from the first that fits a single workgroup.  A workgroup owns GS_FRI_CH
// consecutive leaves of the layer's tree: 4 * GS_FRI_CH threads fold the rows those leaves hash (leaf i = the folded values at i,
// i + L, i + 2L, i + 3L: each thread one row of transposeVector(column, 4), exactly k_fri_fold's arithmetic), the values go to the
// next column in memory and to LDS, GS_FRI_CH threads hash the leaves from LDS, the subtree above them is built in LDS (the levels of
// k_merkle_subtree).  With several workgroups, each counts itself done on a device counter after an agent-scope release fence; the one
// that arrives LAST reads the subtree roots of all (acquire fence first: the L2s of the eight XCDs are not coherent with each other
// without it) and builds the tree over them.  The workgroup that holds the root posts it to the host, derives prng(root) — the point
// of the next layer — and, when it is the only workgroup, goes on with the next layer itself: its inputs are what it has just written.
#define GS_FRI_CH 256
#define GS_FRI_THREADS (4 * GS_FRI_CH)
#define GS_FRI_ONE_LAUNCH_LEAVES (1ull << 15)      // wider layers: the streaming fold + the fused tree launches (throughput-bound)
struct FriLayerDesc {
    const fe *column;          // 4 * rows values
    fe *next;                  // rows values
    uint4 *leaves, *nodes;     // rows / 4 digests each
    fe *point_out;             // may be null
    uint4 *slot;               // posted root
    unsigned long long *flag;
    unsigned long long value;
    uint64_t rows, step;
};
struct FriLayersArgs {
    FriLayerDesc layer[GS_FRI_MAX_LAYERS];
    uint32_t first, count;
    const fe *point_in;
    const fe *tw_lo, *tw_hi;
    int log_lo, logn;
    uint64_t n;
    fe zeta_inv, inv4;
    unsigned int *counter;
};

// the levels m = first_m, first_m / 2, ..., 1 of a subtree whose first layer lies in the LDS heap `sh` (node s at sh[2s], sh[2s + 1]);
// level of width m is stored to nodes[wl + offset * m + i], wl halving with m.  blockDim.x threads, one barrier per level.
template <int ALG>
__global__ __launch_bounds__(GS_FRI_THREADS) void k_fri_layers(FriLayersArgs a) {
    __shared__ fe colbuf[4 * GS_FRI_CH];           // this workgroup's folded values: quarter q of leaf i at [q * CH + i]
    __shared__ uint4 sh[4 * GS_FRI_CH];            // digest heap of the subtree (and, in the last workgroup, of the tree over the subtree roots)
    __shared__ fe point_sh;
    __shared__ int last_sh;
    const uint32_t tid = threadIdx.x, G = gridDim.x;
    fe t = *a.point_in;
    for (uint32_t li = a.first; li < a.first + a.count; li++) {
        const FriLayerDesc &D = a.layer[li];
        const uint64_t rows = D.rows, L = rows >> 2;
        const uint32_t CH = L < GS_FRI_CH ? (uint32_t)L : (uint32_t)GS_FRI_CH;
        const uint64_t base = (uint64_t)blockIdx.x * CH;
        if (tid < 4 * CH) {                          // one row of transposeVector(column, 4) per thread (k_fri_fold)
            const uint32_t q = tid / CH, i = tid - q * CH;
            const uint64_t r = base + i + (uint64_t)q * L;
            const fe *col = D.column;
            const fe y0 = col[r], y1 = col[r + rows], y2 = col[r + 2 * rows], y3 = col[r + 3 * rows];
            const fe s0 = fe_add(y0, y2), s1 = fe_sub(y0, y2), s2 = fe_add(y1, y3);
            const fe s3 = fe_mul(fe_sub(y1, y3), a.zeta_inv);
            const fe u0 = fe_add(s0, s2), u2 = fe_sub(s0, s2), u1 = fe_add(s1, s3), u3 = fe_sub(s1, s3);
            uint64_t e = (r * D.step) & (a.n - 1);
            e = e ? a.n - e : 0;
            fe tx = t;                               // X * x_r^-1
            if (e) {
                fe xi = a.tw_lo[e & ((1ull << a.log_lo) - 1)];
                if (a.logn > a.log_lo) xi = fe_mul(xi, a.tw_hi[e >> a.log_lo]);
                tx = fe_mul(t, xi);
            }
            fe v = fe_add(fe_mul(u3, tx), u2);
            v = fe_add(fe_mul(v, tx), u1);
            v = fe_add(fe_mul(v, tx), u0);
            v = fe_mul(v, a.inv4);
            D.next[r] = v;
            colbuf[q * CH + i] = v;
        }
        __syncthreads();
        if (tid < CH) {                              // leaf = H(next[i] || next[i + L] || next[i + 2L] || next[i + 3L])
            const uint4 *cb = reinterpret_cast<const uint4 *>(colbuf);
            uint32_t d[8];
            digest_words16<ALG>([&](uint32_t w) { return cb[(uint64_t)((w / GS_EW) * CH + tid) * GS_EW + w % GS_EW]; }, 4 * GS_EW, d);
            store_digest(D.leaves, base + tid, d);
            sh[2 * (CH + tid)] = make_uint4(d[0], d[1], d[2], d[3]);
            sh[2 * (CH + tid) + 1] = make_uint4(d[4], d[5], d[6], d[7]);
        }
        __syncthreads();
        subtree_levels<ALG>(sh, CH / 2, L >> 1, blockIdx.x, D.nodes);
        if (G > 1) {
            // (subtree_levels ended in a barrier: every store of this workgroup — D.next, leaves, nodes — has reached its XCD's L2)
            if (tid == 0) {
                __threadfence();                     // release: this workgroup's nodes (its subtree root among them) reach memory
                last_sh = atomicAdd(a.counter, 1u) == G - 1;
                if (last_sh) __threadfence();        // acquire: the other workgroups' subtree roots, not a stale line of this XCD's L2
            }
            __syncthreads();
            if (!last_sh) return;
            for (uint32_t i = tid; i < G; i += GS_FRI_THREADS) {        // heap nodes G .. 2G - 1 of the layer's tree
                sh[2 * (G + i)] = D.nodes[2 * (G + (uint64_t)i)];
                sh[2 * (G + i) + 1] = D.nodes[2 * (G + (uint64_t)i) + 1];
            }
            if (tid == 0) atomicExch(a.counter, 0u);                    // ready for the next launch on this context's stream
            __syncthreads();
            subtree_levels<ALG>(sh, G / 2, G >> 1, 0, D.nodes);
        }
        if (tid == 0) {                              // the root is heap node 1: sh[2], sh[3]
            D.nodes[0] = make_uint4(0, 0, 0, 0);
            D.nodes[1] = make_uint4(0, 0, 0, 0);
            const uint4 r0 = sh[2], r1 = sh[3];
            if (D.slot) {
                D.slot[0] = r0;
                D.slot[1] = r1;
                __threadfence_system();
                __hip_atomic_store(D.flag, D.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            const uint32_t seed[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            const fe x = prng_point(seed);
            if (D.point_out) *D.point_out = x;
            point_sh = x;
        }
        if (G > 1) return;                           // a launch of several workgroups is one layer
        __syncthreads();                             // the next layer's column is what this workgroup has just stored (D.next)
        t = point_sh;
    }
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
    if (count == 0) return gs_fail(c, GS_ERR_ARG, "hash_merge_rows: no vectors");
    if (!n) return GS_OK;
    if (count > GS_MAX_COMBINE) {
        // an AIR with more registers than fit the kernel-argument table: the pointers go to device memory first (a blocking copy of
        // 8 bytes per register; every column of a row is still hashed in one pass)
        void *tab = nullptr;
        if ((rc = gs_tmp_alloc(c, (uint64_t)count * sizeof(void *), &tab))) return rc;
        if ((rc = gs_push(c, tab, vecs_host, (uint64_t)count * sizeof(void *)))) { gs_tmp_free(c, tab); return rc; }      // vecs_host belongs to the caller
        if (alg == GS_HASH_SHA256)
            hipLaunchKernelGGL(k_hash_merge_rows_table<0>, dim3(gs_grid(n)), dim3(256), 0, c->stream, (const uint4 *const *)tab, count, n, (uint4 *)out);
        else
            hipLaunchKernelGGL(k_hash_merge_rows_table<1>, dim3(gs_grid(n)), dim3(256), 0, c->stream, (const uint4 *const *)tab, count, n, (uint4 *)out);
        gs_tmp_free(c, tab);      // stream-ordered reuse (common.h)
        GS_LAUNCH_CHECK(c);
        return GS_OK;
    }
    gs_traffic(c, n * ((uint64_t)count * GS_ELT + 32), n * hash_blocks(alg == GS_HASH_SHA256 ? 0 : 1, (uint64_t)count * GS_ELT),
               count == 1 ? "k_hash_merge_rows1<%d>" : "k_hash_merge_rows<%d>", alg == GS_HASH_SHA256 ? 0 : 1);
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
    if (((uintptr_t)leaves | (uintptr_t)nodes) & 15) return gs_fail(c, GS_ERR_ARG, "merkle_build: buffers must be 16-byte aligned");
    HashPtrArgs va;
    for (uint32_t j = 0; j < GS_MAX_COMBINE; j++) va.v[j] = nullptr;
    return alg == GS_HASH_SHA256 ? merkle_run<0>(c, 0, va, 0, leaves, nullptr, n, (uint4 *)nodes) : merkle_run<1>(c, 0, va, 0, leaves, nullptr, n, (uint4 *)nodes);
}

static int commit_rows(gs_ctx *c, gs_hash_alg alg, const void *const *vecs_host, uint32_t count, uint64_t n, void *leaves, void *nodes, const RootTail *tail);
int gs_merkle_commit_rows(gs_ctx *c, gs_hash_alg alg, const void *const *vecs_host, uint32_t count, uint64_t n, void *leaves, void *nodes) {
    return commit_rows(c, alg, vecs_host, count, n, leaves, nodes, nullptr);
}
int gs_merkle_commit_rows_seed(gs_ctx *c, gs_hash_alg alg, const void *const *vecs_host, uint32_t count, uint64_t n, void *leaves, void *nodes,
                               void *point_out, uint64_t *ticket) {
    if (!c || (!point_out && !ticket)) return GS_ERR_ARG;
    if (point_out && ((uintptr_t)point_out & 15)) return gs_fail(c, GS_ERR_ARG, "merkle_commit_rows_seed: point_out must be 16-byte aligned");
    RootTail tail = {nullptr, nullptr, 0, (fe *)point_out};
    // every argument is checked BEFORE a read-back slot is reserved: a rejected call must not leave a ticket nobody will ever deliver
    if (!vecs_host || !leaves || !nodes) return GS_ERR_ARG;
    if (int rc0 = check_alg(c, alg)) return rc0;
    if (count == 0) return gs_fail(c, GS_ERR_ARG, "merkle_commit_rows: no vectors");
    if (!gs_is_pow2(n) || n < 2) return gs_fail(c, GS_ERR_ARG, "merkle_commit_rows: n must be a power of two >= 2");
    if (((uintptr_t)leaves | (uintptr_t)nodes) & 15) return gs_fail(c, GS_ERR_ARG, "merkle_commit_rows: buffers must be 16-byte aligned");
    if (ticket) {
        void *slot;
        int rc = gs_readback_reserve(c, 32, &slot, &tail.flag, &tail.value, ticket);
        if (rc) return rc;
        tail.slot = (uint4 *)slot;
    }
    return commit_rows(c, alg, vecs_host, count, n, leaves, nodes, &tail);
}
static int commit_rows(gs_ctx *c, gs_hash_alg alg, const void *const *vecs_host, uint32_t count, uint64_t n, void *leaves, void *nodes, const RootTail *tail) {
    if (!c || !vecs_host || !leaves || !nodes) return GS_ERR_ARG;
    int rc = check_alg(c, alg);
    if (rc) return rc;
    if (count == 0) return gs_fail(c, GS_ERR_ARG, "merkle_commit_rows: no vectors");
    if (!gs_is_pow2(n) || n < 2) return gs_fail(c, GS_ERR_ARG, "merkle_commit_rows: n must be a power of two >= 2");
    if (((uintptr_t)leaves | (uintptr_t)nodes) & 15) return gs_fail(c, GS_ERR_ARG, "merkle_commit_rows: buffers must be 16-byte aligned");
    HashPtrArgs va;
    if (count > GS_MAX_COMBINE) {      // more columns than fit the kernel-argument table: the two members one after the other
        if ((rc = gs_hash_merge_rows(c, alg, vecs_host, count, n, leaves))) return rc;
        for (uint32_t j = 0; j < GS_MAX_COMBINE; j++) va.v[j] = nullptr;
        return alg == GS_HASH_SHA256 ? merkle_run<0>(c, 0, va, 0, leaves, nullptr, n, (uint4 *)nodes, tail) : merkle_run<1>(c, 0, va, 0, leaves, nullptr, n, (uint4 *)nodes, tail);
    }
    for (uint32_t j = 0; j < GS_MAX_COMBINE; j++) va.v[j] = (const uint4 *)vecs_host[j < count ? j : 0];
    const int src = count == 1 ? 1 : 2;
    return alg == GS_HASH_SHA256 ? merkle_run<0>(c, src, va, count, nullptr, leaves, n, (uint4 *)nodes, tail)
                                 : merkle_run<1>(c, src, va, count, nullptr, leaves, n, (uint4 *)nodes, tail);
}

int gs_fri_layers(gs_ctx *c, gs_hash_alg alg, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t len, const void *x_dev,
                  uint32_t nlayers, struct gs_fri_layer *layers) {
    if (!c || !omega || !column || !x_dev || !layers || !nlayers) return GS_ERR_ARG;
    int rc = check_alg(c, alg);
    if (rc) return rc;
    if (!gs_is_pow2(n) || len < 32 || !gs_is_pow2(len) || len * step != n) return gs_fail(c, GS_ERR_ARG, "fri_layers: len must be a power of two >= 32 with len * step == n");
    if (nlayers > GS_FRI_MAX_LAYERS || (len >> (2 * nlayers)) < 8) return gs_fail(c, GS_ERR_ARG, "fri_layers: too many layers for this column");
    if (((uintptr_t)column | (uintptr_t)x_dev) & 15) return gs_fail(c, GS_ERR_ARG, "fri_layers: buffers must be 16-byte aligned");
    for (uint32_t i = 0; i < nlayers; i++) {
        const gs_fri_layer &L = layers[i];
        if (!L.next || !L.leaves || !L.nodes) return GS_ERR_ARG;
        if (((uintptr_t)L.next | (uintptr_t)L.leaves | (uintptr_t)L.nodes | (uintptr_t)L.point_out) & 15)
            return gs_fail(c, GS_ERR_ARG, "fri_layers: buffers must be 16-byte aligned");
        if (L.next == (i ? layers[i - 1].next : column)) return gs_fail(c, GS_ERR_ARG, "fri_layers: a layer's output must not alias its column");
    }
    // points between layers that the caller does not ask for live in a per-context scratch (ordered on the context's stream)
    unsigned int *counter = nullptr;
    if ((rc = arrival_counters(c, &counter))) return rc;
    fe *scratch = (fe *)c->fri_points;
    const fe w = fe_from_bytes(omega);
    const fe *lo, *hi;
    int log_lo;
    if ((rc = gs_plan_pow_tables(c, w, n, &lo, &hi, &log_lo))) return rc;
    const fe zeta = gs_memo_pow(c, w, n / 4);
    FriLayersArgs a;
    memset(&a, 0, sizeof a);
    a.tw_lo = lo; a.tw_hi = hi; a.log_lo = log_lo; a.logn = gs_log2(n); a.n = n;
    a.zeta_inv = fe_mul(fe_mul(zeta, zeta), zeta);
    static const fe inv4_const = fe_inv(fe_from_u64(4));  // (a constant of the field: computed once)
    a.inv4 = inv4_const;
    a.counter = counter;
    const void *col = column;
    const void *point = x_dev;
    uint64_t m = len, st = step;
    for (uint32_t i = 0; i < nlayers; i++, m >>= 2, st <<= 2) {
        gs_fri_layer &L = layers[i];
        FriLayerDesc &D = a.layer[i];
        fe *pout = L.point_out ? (fe *)L.point_out : scratch + i;
        D.column = (const fe *)col; D.next = (fe *)L.next; D.leaves = (uint4 *)L.leaves; D.nodes = (uint4 *)L.nodes; D.point_out = pout;
        D.rows = m / 4; D.step = st;
        void *slot;
        if ((rc = gs_readback_reserve(c, 32, &slot, &D.flag, &D.value, &L.ticket))) return rc;
        D.slot = (uint4 *)slot;
        const uint64_t leaves = m / 16;
        if (leaves > GS_FRI_ONE_LAUNCH_LEAVES) {
            // throughput-bound layer: the streaming fold, then the tree launches (root posted and point derived by the last of them)
            if ((rc = gs_fri_fold_at(c, omega, n, st, col, m, point, L.next))) return rc;
            const void *quarters[4];
            for (uint64_t k = 0; k < 4; k++) quarters[k] = (const uint8_t *)L.next + k * leaves * GS_ELT;
            const RootTail tail = {D.slot, D.flag, D.value, pout};
            if ((rc = commit_rows(c, alg, quarters, 4, leaves, L.leaves, L.nodes, &tail))) return rc;
        } else {
            const uint32_t grid = (uint32_t)(leaves <= GS_FRI_CH ? 1 : leaves / GS_FRI_CH);
            a.first = i;
            a.count = grid == 1 ? nlayers - i : 1;
            a.point_in = (const fe *)point;
            if (grid == 1)         // every remaining layer in this launch: their descriptors first
                for (uint32_t j = i + 1; j < nlayers; j++) {
                    gs_fri_layer &Lj = layers[j];
                    FriLayerDesc &Dj = a.layer[j];
                    const uint64_t mj = len >> (2 * j);
                    Dj.column = (const fe *)layers[j - 1].next; Dj.next = (fe *)Lj.next; Dj.leaves = (uint4 *)Lj.leaves; Dj.nodes = (uint4 *)Lj.nodes;
                    Dj.point_out = Lj.point_out ? (fe *)Lj.point_out : scratch + j;
                    Dj.rows = mj / 4; Dj.step = step << (2 * j);
                    void *sj;
                    if ((rc = gs_readback_reserve(c, 32, &sj, &Dj.flag, &Dj.value, &Lj.ticket))) return rc;
                    Dj.slot = (uint4 *)sj;
                }
            {   // per layer of mj values: the column read, the folded column (mj/4 elements), mj/16 leaf digests and mj/16 - 1 nodes written;
                // one 64-byte row per leaf = one compression, one per node
                uint64_t by = 0, un = 0;
                for (uint32_t j = i; j < i + a.count; j++) {
                    const uint64_t mj = len >> (2 * j);
                    by += mj * GS_ELT + mj / 4 * GS_ELT + mj / 16 * 64;
                    un += mj / 16 * hash_blocks(alg == GS_HASH_SHA256 ? 0 : 1, 4 * GS_ELT) + mj / 16;
                }
                gs_traffic(c, by, un, "k_fri_layers<%d>", alg == GS_HASH_SHA256 ? 0 : 1);
            }
            if (alg == GS_HASH_SHA256) hipLaunchKernelGGL(k_fri_layers<0>, dim3(grid), dim3(GS_FRI_THREADS), 0, c->stream, a);
            else hipLaunchKernelGGL(k_fri_layers<1>, dim3(grid), dim3(GS_FRI_THREADS), 0, c->stream, a);
            GS_LAUNCH_CHECK(c);
            if (grid == 1) break;
        }
        col = L.next;
        point = pout;
    }
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
