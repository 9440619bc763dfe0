// small.hip — host-side pieces of the galois / merkle surface that only ever see a few hundred values:
//   * FiniteField.interpolate(xs, ys) — general Lagrange interpolation (BoundaryConstraints.ts:42: one point per
//     assertion; LowDegreeProver.ts:243: <= 256 remainder values) and evalPolyAt over a point list (:248).
//     O(n^2) on native 64-bit limbs; the reference does the same work in its JS layer.
//   * MerkleTree.proveBatch(indexes) — the authentication-path PLAN is host logic over <= 256 indexes; the digests it
//     selects are fetched from the device-resident tree by ONE gather kernel and one device->host copy.
#include <algorithm>

#include "common.h"
#include "host_field.h"
#include "host_sha256.h"
#ifndef HF_CHAIN_MUL          // flavours without a weak form: the canonical operations
#define HF_CHAIN_MUL hf_mul
#define HF_CHAIN_END(x) (x)
#endif
#ifndef HF_CHAIN_ADD
#define HF_CHAIN_ADD hf_add
#endif

extern "C" {

int gs_small_interpolate(const uint8_t *xs_host, const uint8_t *ys_host, uint32_t n, uint8_t *coeffs_out) {
    if (!xs_host || !ys_host || !coeffs_out || n == 0 || n > 4096) return GS_ERR_ARG;
    std::vector<hfe> x(n), y(n), master(n + 1, 0), out(n, 0);
    for (uint32_t i = 0; i < n; i++) { x[i] = hf_load(xs_host + GS_ELT * i); y[i] = hf_load(ys_host + GS_ELT * i); }
    // The three O(n^2) loops below run on WEAK values (any 128-bit representative: HF_CHAIN_MUL / HF_CHAIN_ADD, no compare-and-subtract
    // per operation) and are canonicalised once at the end; independent chains are interleaved four at a time (a dependent product is
    // ~15 ns of latency, an independent one ~4 ns of issue).  112 points (the remainder of a degree-6 AIR): 0.70 -> 0.19 ms, the 128 evaluations after it 0.23 -> 0.05 ms.
    // master polynomial M(X) = prod (X - x_i)
    master[0] = 1;
    for (uint32_t i = 0; i < n; i++) {
        hfe nx = hf_sub(0, x[i]);
        for (uint32_t d = i + 1; d >= 1; d--) master[d] = HF_CHAIN_ADD(master[d - 1], HF_CHAIN_MUL(master[d], nx));
        master[0] = HF_CHAIN_MUL(master[0], nx);
    }
    // Lagrange denominators q_j(x_j) = M'(x_j), all inverted with ONE field inversion (Montgomery's trick)
    const bool keep = n <= 512;                                // keep the n quotients (n coefficients each: 4 MB at most) or recompute them
    std::vector<hfe> den(n), pre(n), qs(keep ? (size_t)n * n : n);
    // q_j = M / (X - x_j) by synthetic division, evaluated at x_j on the fly (Horner): four j's side by side
    for (uint32_t j0 = 0; j0 < n; j0 += 4) {
        const uint32_t w = n - j0 < 4 ? n - j0 : 4;
        hfe carry[4] = {0, 0, 0, 0}, dj[4] = {0, 0, 0, 0}, xj[4];
        for (uint32_t u = 0; u < 4; u++) xj[u] = x[j0 + (u < w ? u : 0)];
        for (uint32_t d = n; d >= 1; d--) {
            for (uint32_t u = 0; u < 4; u++) {
                carry[u] = HF_CHAIN_ADD(master[d], HF_CHAIN_MUL(carry[u], xj[u]));
                dj[u] = HF_CHAIN_ADD(HF_CHAIN_MUL(dj[u], xj[u]), carry[u]);
            }
            if (keep)
                for (uint32_t u = 0; u < w; u++) qs[(size_t)(j0 + u) * n + d - 1] = carry[u];
        }
        for (uint32_t u = 0; u < w; u++) den[j0 + u] = HF_CHAIN_END(dj[u]);
    }
    hfe acc = 1;
    for (uint32_t j = 0; j < n; j++) { pre[j] = acc; if (!hf_is_zero(den[j])) acc = hf_mul(acc, den[j]); }
    hfe inv_all = hf_inv(acc);
    for (uint32_t j = n; j-- > 0;) {
        hfe inv_j = 0;                            // a repeated x makes its denominator zero: 0^-1 = 0, as before
        if (!hf_is_zero(den[j])) { inv_j = hf_mul(inv_all, pre[j]); inv_all = hf_mul(inv_all, den[j]); }
        const hfe sc = hf_mul(y[j], inv_j);
        if (!keep) {
            hfe carry = 0;
            for (uint32_t d = n; d >= 1; d--) {
                carry = HF_CHAIN_ADD(master[d], HF_CHAIN_MUL(carry, x[j]));
                qs[d - 1] = carry;
            }
        }
        const hfe *qj = keep ? &qs[(size_t)j * n] : qs.data();
        for (uint32_t d = 0; d < n; d++) out[d] = HF_CHAIN_ADD(out[d], HF_CHAIN_MUL(qj[d], sc));
    }
    for (uint32_t d = 0; d < n; d++) out[d] = HF_CHAIN_END(out[d]);
    for (uint32_t d = 0; d < n; d++) hf_store(coeffs_out + GS_ELT * d, out[d]);
    return GS_OK;
}

int gs_small_eval_poly(const uint8_t *poly_host, uint32_t len, const uint8_t *xs_host, uint32_t m, uint8_t *out_host) {
    if ((!poly_host && len) || (!xs_host && m) || (!out_host && m)) return GS_ERR_ARG;
    std::vector<hfe> p(len);
    for (uint32_t i = 0; i < len; i++) p[i] = hf_load(poly_host + GS_ELT * i);
    for (uint32_t i0 = 0; i0 < m; i0 += 4) {          // Horner for four points at a time (independent chains: see gs_small_interpolate)
        const uint32_t w = m - i0 < 4 ? m - i0 : 4;
        hfe x[4], acc[4] = {0, 0, 0, 0};
        for (uint32_t u = 0; u < 4; u++) x[u] = hf_load(xs_host + GS_ELT * (i0 + (u < w ? u : 0)));
        for (uint32_t k = len; k-- > 0;)
            for (uint32_t u = 0; u < 4; u++) acc[u] = HF_CHAIN_ADD(HF_CHAIN_MUL(acc[u], x[u]), p[k]);
        for (uint32_t u = 0; u < w; u++) hf_store(out_host + GS_ELT * (i0 + u), HF_CHAIN_END(acc[u]));
    }
    return GS_OK;
}

}  // extern "C"

// digest j of the output comes from leaves (src_sel = 0) or nodes (src_sel = 1) at index idx
__global__ void k_gather_digests(const uint4 *__restrict__ leaves, const uint4 *__restrict__ nodes,
                                 const uint64_t *__restrict__ sel_idx, uint32_t count, uint4 *__restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * 2) return;
    uint64_t e = sel_idx[t >> 1];
    const uint4 *src = (e >> 63) ? nodes : leaves;
    out[t] = src[((e & ~(1ull << 63)) << 1) + (t & 1)];
}

extern "C" int gs_merkle_prove_batch(gs_ctx *c, const void *leaves, const void *nodes, uint64_t n, const uint64_t *idx_host,
                                     uint32_t count, uint8_t *values_out, uint32_t *ncols_out, uint32_t *col_lens_out,
                                     uint8_t *nodes_out, uint64_t nodes_cap) {
    if (!c || !leaves || !nodes || !idx_host || !values_out || !ncols_out || !col_lens_out || !nodes_out) return GS_ERR_ARG;
    if (!gs_is_pow2(n) || n < 2 || count == 0) return gs_fail(c, GS_ERR_ARG, "merkle_prove_batch: n must be a power of two >= 2, count > 0");
    const int depth = gs_log2(n);
    const uint64_t NODE = 1ull << 63;
    std::vector<uint64_t> sorted(idx_host, idx_host + count);
    for (uint32_t i = 0; i < count; i++)
        if (idx_host[i] >= n) return gs_fail(c, GS_ERR_ARG, "merkle_prove_batch: index %llu out of range", (unsigned long long)idx_host[i]);
    std::sort(sorted.begin(), sorted.end());
    for (uint32_t i = 1; i < count; i++)
        if (sorted[i] == sorted[i - 1]) return gs_fail(c, GS_ERR_ARG, "merkle_prove_batch: repeating indexes");
    // fetch list: first the requested leaves (request order), then the column entries in column-major order
    std::vector<uint64_t> fetch(idx_host, idx_host + count);
    std::vector<uint64_t> cur;
    std::vector<std::vector<uint64_t>> cols;
    for (uint32_t i = 0; i < count;) {
        uint64_t e = sorted[i] & ~1ull;
        bool has0 = false, has1 = false;
        while (i < count && (sorted[i] & ~1ull) == e) { (sorted[i] & 1 ? has1 : has0) = true; i++; }
        cols.emplace_back();
        cols.back().reserve(depth);
        if (has0 && !has1) cols.back().push_back(e + 1);
        else if (!has0 && has1) cols.back().push_back(e);
        cur.push_back((e + n) >> 1);
    }
    std::vector<uint64_t> nxt;
    for (int d = depth - 1; d > 0; d--) {
        nxt.clear();
        for (size_t i = 0; i < cur.size(); i++) {
            uint64_t sib = cur[i] ^ 1;
            if (i + 1 < cur.size() && cur[i + 1] == sib) i++;
            else cols[i].push_back(sib | NODE);
            nxt.push_back(sib >> 1);
        }
        cur.swap(nxt);
    }
    uint64_t total = 0;
    for (auto &col : cols) total += col.size();
    if (total > nodes_cap) return gs_fail(c, GS_ERR_ARG, "merkle_prove_batch: nodes_out too small (%llu digests needed)", (unsigned long long)total);
    for (auto &col : cols) fetch.insert(fetch.end(), col.begin(), col.end());
    const uint64_t nf = fetch.size();
    const uint64_t idx_bytes = (nf * 8 + 255) & ~(uint64_t)255, data_bytes = nf * 32;
    *ncols_out = (uint32_t)cols.size();                           // the shape is host knowledge; only the digests come from the device
    for (size_t i = 0; i < cols.size(); i++) col_lens_out[i] = (uint32_t)cols[i].size();
    if (c->defer) {
        // inside a deferral window: note the two 16-byte words of every digest; gs_defer_end fetches them all at once
        c->defer_copies.push_back({values_out, (uint64_t)c->defer_addrs.size(), (uint64_t)count * 32});
        if (total) c->defer_copies.push_back({nodes_out, (uint64_t)c->defer_addrs.size() + 2ull * count, total * 32});
        for (uint64_t e : fetch) {
            const uint64_t at = (uint64_t)(uintptr_t)((e >> 63) ? nodes : leaves) + (e & ~NODE) * 32;
            c->defer_addrs.push_back(at);
            c->defer_addrs.push_back(at + 16);
        }
        return GS_OK;
    }
    int rc = gs_stage_reserve(c, idx_bytes + data_bytes);
    if (rc) return rc;
    memcpy(c->h_stage, fetch.data(), nf * 8);   // zero-copy through mapped pinned memory (see gs_gather)
    uint8_t *d_out = (uint8_t *)c->h_stage_dev + idx_bytes;
    hipLaunchKernelGGL(k_gather_digests, dim3((unsigned)((nf * 2 + 255) / 256)), dim3(256), 0, c->stream, (const uint4 *)leaves,
                       (const uint4 *)nodes, (const uint64_t *)c->h_stage_dev, (uint32_t)nf, (uint4 *)d_out);
    GS_LAUNCH_CHECK(c);
    uint8_t *h_out = (uint8_t *)c->h_stage + idx_bytes;
    GS_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(values_out, h_out, (size_t)count * 32);
    memcpy(nodes_out, h_out + (size_t)count * 32, (size_t)total * 32);
    return GS_OK;
}

// ---- Fiat-Shamir query positions (QueryIndexGenerator.ts:39-67): host only -------------------------------------------------
#define SHA256(m, l, o) host_sha256((m), (l), (o))
extern "C" int gs_pseudorandom_indexes(const uint8_t *seed, uint32_t seed_len, uint32_t count, uint64_t max_, uint32_t exclude, uint64_t *out) {
    if (!seed || !out || !max_ || (count && !out)) return GS_ERR_ARG;
    uint64_t max_count = exclude ? max_ - max_ / exclude : max_;
    if (max_count < count) return GS_ERR_ARG;
    uint8_t st[32], msg[33], dg[32];
    SHA256(seed, seed_len, st);                       /* state: 256-bit big-endian integer */
    uint32_t found = 0;
    for (uint64_t i = 0; i < (uint64_t)count * 1000 && found < count; i++) {
        /* v = state + i, 33 bytes big-endian */
        uint64_t carry = i;
        msg[0] = 0;
        for (int k = 31; k >= 0; k--) { uint64_t t = (uint64_t)st[k] + (carry & 0xFF); msg[k + 1] = (uint8_t)t; carry = (carry >> 8) + (t >> 8); }
        msg[0] = (uint8_t)carry;
        /* hex digits without leading zeros; an odd count drops the last nibble: bytes = (v >> 4) then */
        int lead = 0;
        while (lead < 33 && msg[lead] == 0) lead++;
        int nhex = lead == 33 ? 0 : (33 - lead) * 2 - ((msg[lead] >> 4) == 0 ? 1 : 0);
        uint8_t buf[33];
        int nbytes = nhex / 2;
        if (nhex & 1) {                               /* shift right by one nibble */
            for (int k = 0; k < nbytes; k++) {
                /* byte k of the result = nibbles 2k, 2k+1 of the hex string */
                int hi_n = 2 * k, lo_n = 2 * k + 1;   /* nibble index from the most significant nibble of the string */
                int first = lead * 2 + 1;             /* string starts at the low nibble of msg[lead] */
                int a = first + hi_n, b = first + lo_n;
                uint8_t na = (a & 1) ? (msg[a >> 1] & 15) : (msg[a >> 1] >> 4);
                uint8_t nb = (b & 1) ? (msg[b >> 1] & 15) : (msg[b >> 1] >> 4);
                buf[k] = (uint8_t)((na << 4) | nb);
            }
        } else {
            for (int k = 0; k < nbytes; k++) buf[k] = msg[lead + k];
        }
        SHA256(buf, (size_t)nbytes, dg);
        /* index = dg (big-endian 256-bit) mod max */
        uint64_t index;
        if ((max_ & (max_ - 1)) == 0) {               /* domain sizes are powers of two: the low bits of the digest */
            uint64_t low = 0;
            for (int k = 24; k < 32; k++) low = (low << 8) | dg[k];
            index = low & (max_ - 1);
        } else {
            unsigned __int128 r = 0;
            for (int k = 0; k < 32; k++) r = ((r << 8) | dg[k]) % max_;
            index = (uint64_t)r;
        }
        if (exclude && index % exclude == 0) continue;
        int dup = 0;
        for (uint32_t k = 0; k < found; k++) if (out[k] == index) { dup = 1; break; }
        if (dup) continue;
        out[found++] = index;
    }
    return found == count ? GS_OK : GS_ERR_ARG;
}
#undef SHA256
