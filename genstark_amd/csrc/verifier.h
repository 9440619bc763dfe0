// verifier.h — Stark.verify() (lib/Stark.ts:167-248) and LowDegreeProver.verify (lib/components/LowDegreeProver.ts:70-172) as
// native host code (included once, by prover.cc: it shares the driver's scalar helpers, prng and index generator).
//
// CPU-side by design, like the reference's verifier: a proof is a few hundred field elements and a few thousand digests, there is
// nothing for a GPU to do.  Input: the statement as a gs_prover_job (the fields a verifier needs: sizes, query counts, hash,
// root of unity, assertions, and of the AIR its kind, register counts, constraint degrees and — kind 0 — the round constants or —
// kind 1 — the constraint evaluator program with its constants and the PUBLIC static registers' values) + the serialized proof.
// Every check of the reference is made, in its order, with its messages; what the reference computes on BigInt / wasm runs here on
// the build flavour's host arithmetic (host_field*.h).  Hashing: host SHA-256 (SHA-NI when present) and a portable BLAKE2s.
#pragma once

namespace {

// ---- BLAKE2s-256, unkeyed (RFC 7693) — the merkle package's 'blake2s256' (SURVEY appendix A.7).  A verification is ~5 000
// compressions: the state lives in sixteen locals and the ten rounds are written out (the message schedule as compile-time indices).
#define B2S_ROTR(x, r) (((x) >> (r)) | ((x) << (32 - (r))))
#define B2S_G(a, b, c, d, x, y)                                                                       \
    a = a + b + (x); d = B2S_ROTR(d ^ a, 16); c = c + d; b = B2S_ROTR(b ^ c, 12);                        \
    a = a + b + (y); d = B2S_ROTR(d ^ a, 8); c = c + d; b = B2S_ROTR(b ^ c, 7);
#define B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)             \
    B2S_G(v0, v4, v8, v12, m[s0], m[s1]) B2S_G(v1, v5, v9, v13, m[s2], m[s3]) B2S_G(v2, v6, v10, v14, m[s4], m[s5]) B2S_G(v3, v7, v11, v15, m[s6], m[s7]) \
    B2S_G(v0, v5, v10, v15, m[s8], m[s9]) B2S_G(v1, v6, v11, v12, m[s10], m[s11]) B2S_G(v2, v7, v8, v13, m[s12], m[s13]) B2S_G(v3, v4, v9, v14, m[s14], m[s15])
inline void host_blake2s_compress(uint32_t h[8], const uint8_t b[64], uint64_t t, bool last) {
    uint32_t m[16];
    memcpy(m, b, 64);                          // little-endian host
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
    uint32_t v12 = 0x510E527Fu ^ (uint32_t)t, v13 = 0x9B05688Cu ^ (uint32_t)(t >> 32), v14 = last ? ~0x1F83D9ABu : 0x1F83D9ABu, v15 = 0x5BE0CD19u;
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
    h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}
#undef B2S_ROUND
#undef B2S_G
#undef B2S_ROTR
inline void host_blake2s(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6A09E667u ^ 0x01010020u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    const size_t nblocks = len ? (len + 63) / 64 : 1;
    for (size_t blk = 0; blk + 1 < nblocks; blk++) host_blake2s_compress(h, msg + 64 * blk, (uint64_t)(blk + 1) * 64, false);
    uint8_t b[64] = {0};
    const size_t off = (nblocks - 1) * 64;
    memcpy(b, msg + off, len - off);
    host_blake2s_compress(h, b, len, true);
    memcpy(out, h, 32);                        // little-endian host
}
inline void host_digest(int alg, const uint8_t *msg, size_t len, uint8_t out[32]) {
    if (alg == GS_HASH_SHA256) host_sha256(msg, len, out); else host_blake2s(msg, len, out);
}

// ---- the wire format, read side (lib/Serializer.ts:83-144, lib/utils/serialization.ts:44-127): bounds-checked
struct Reader {
    const uint8_t *p;
    uint64_t len, at = 0;
    void need(uint64_t n) const { if (at + n > len || at + n < at) fail(GS_ERR_ARG, "malformed proof: truncated"); }
    uint8_t byte() { need(1); return p[at++]; }
    const uint8_t *take(uint64_t n) { need(n); const uint8_t *q = p + at; at += n; return q; }
};
struct ParsedMerkleProof {
    std::vector<Bytes> values;                 // the leaves' contents (rows)
    std::vector<std::vector<Bytes>> nodes;     // authentication columns
    uint32_t depth = 0;
};
ParsedMerkleProof read_merkle_proof(Reader &r, uint64_t leaf_size) {
    ParsedMerkleProof mp;
    uint32_t n = r.byte();
    if (!n) n = (uint32_t)MAX_ARRAY;
    for (uint32_t i = 0; i < n; i++) { const uint8_t *v = r.take(leaf_size); mp.values.emplace_back(v, v + leaf_size); }
    uint32_t cols = r.byte();
    if (!cols) cols = (uint32_t)MAX_ARRAY;
    std::vector<uint8_t> heads(cols);
    for (uint32_t i = 0; i < cols; i++) heads[i] = r.byte();
    for (uint8_t head : heads) {
        std::vector<Bytes> col;
        for (uint32_t j = 0; j < (uint32_t)(head >> 1); j++) {
            const uint64_t size = (j == 0 && (head & 1)) ? leaf_size : DIGEST;          // serialization.ts:93-101: a "leaf" column starts with a leaf-sized item
            const uint8_t *v = r.take(size);
            col.emplace_back(v, v + size);
        }
        mp.nodes.push_back(std::move(col));
    }
    mp.depth = r.byte();
    return mp;
}

// MerkleTree.verifyBatch (merkle package; restated with proveBatch's layout, genstark_amd/merkle.py:151-199): the leaves' DIGESTS at
// `indexes` (request order) + the columns of `proof` must hash up to `root`.  The nodes of a level are visited in ascending order, and so
// are their parents: two parallel arrays per level, no look-up structure.
struct Dg { uint8_t b[32]; };
bool merkle_verify_batch(int alg, const uint8_t *root, const std::vector<uint64_t> &indexes, const std::vector<Dg> &leaf_digests,
                         const std::vector<std::vector<Bytes>> &columns, uint32_t depth) {
    if (depth == 0 || depth > 62 || leaf_digests.size() != indexes.size()) return false;
    const uint64_t offset = 1ull << depth;
    std::vector<std::pair<uint64_t, size_t>> srt;         // (index, position in the request)
    for (size_t i = 0; i < indexes.size(); i++) {
        if (indexes[i] >= offset) return false;
        srt.push_back({indexes[i], i});
    }
    std::sort(srt.begin(), srt.end());
    for (size_t i = 1; i < srt.size(); i++) if (srt[i].first == srt[i - 1].first) return false;      // repeating indexes
    auto merge = [&](const uint8_t *a, const uint8_t *b) {
        uint8_t buf[64];
        memcpy(buf, a, 32); memcpy(buf + 32, b, 32);
        Dg d;
        host_digest(alg, buf, 64, d.b);
        return d;
    };
    auto column_item = [&](size_t col, size_t k) -> const uint8_t * {
        if (col >= columns.size() || k >= columns[col].size()) return nullptr;
        if (columns[col][k].size() != DIGEST) fail(GS_ERR_ARG, "malformed proof: a Merkle node is not a digest");
        return columns[col][k].data();
    };
    std::vector<uint64_t> ix;
    std::vector<Dg> dg;
    std::vector<size_t> ptr;
    size_t col = 0;
    for (size_t i = 0; i < srt.size(); col++) {           // the leaf pairs: one column each
        const uint64_t e = srt[i].first & ~1ull;
        const uint8_t *v1 = nullptr, *v2 = nullptr;
        size_t used = 0;
        if (srt[i].first == e) { v1 = leaf_digests[srt[i].second].b; i++; }
        if (i < srt.size() && srt[i].first == e + 1) { v2 = leaf_digests[srt[i].second].b; i++; }
        if (!v1) { v1 = column_item(col, 0); used = 1; }
        else if (!v2) { v2 = column_item(col, 0); used = 1; }
        if (!v1 || !v2) return false;
        ix.push_back((offset + e) >> 1);
        dg.push_back(merge(v1, v2));
        ptr.push_back(used);
    }
    if (col != columns.size()) return false;
    for (uint32_t lvl = depth - 1; lvl > 0; lvl--) {
        std::vector<uint64_t> nix;
        std::vector<Dg> ndg;
        for (size_t i = 0; i < ix.size(); i++) {
            const uint64_t node = ix[i];
            const uint8_t *self = dg[i].b, *sib;
            if (i + 1 < ix.size() && ix[i + 1] == (node ^ 1)) { sib = dg[i + 1].b; i++; ndg.push_back(merge(self, sib)); }
            else {
                // (the path entries of the i-th node of the CURRENT level sit in column i: proveBatch's layout)
                const size_t c = i;
                sib = column_item(c, ptr[c]);
                if (!sib) return false;
                ptr[c]++;
                ndg.push_back((node & 1) ? merge(sib, self) : merge(self, sib));
            }
            nix.push_back(node >> 1);
        }
        ix.swap(nix);
        dg.swap(ndg);
    }
    return ix.size() == 1 && ix[0] == 1 && !memcmp(dg[0].b, root, DIGEST);
}
// rehashMerkleProofValues (lib/utils/index.ts:34-45) + verifyBatch
bool merkle_check(int alg, const Bytes &root, const std::vector<uint64_t> &indexes, const ParsedMerkleProof &mp) {
    std::vector<Dg> digests(mp.values.size());
    for (size_t i = 0; i < mp.values.size(); i++) host_digest(alg, mp.values[i].data(), mp.values[i].size(), digests[i].b);
    return merkle_verify_batch(alg, root.data(), indexes, digests, mp.nodes, mp.depth);
}

// FiniteField.interpolate(xs, ys) for the handful of points of an assertion set / FiniteField.evalPolyAt, on host scalars
std::vector<F> lagrange(const std::vector<F> &xs, const std::vector<F> &ys) {
    const size_t n = xs.size();
    Bytes xb(n * ELEM), yb(n * ELEM), cb(n * ELEM);
    for (size_t i = 0; i < n; i++) { le16(xs[i], xb.data() + ELEM * i); le16(ys[i], yb.data() + ELEM * i); }
    if (A.gs_small_interpolate(xb.data(), yb.data(), (uint32_t)n, cb.data())) fail(GS_ERR_ARG, "gs_small_interpolate failed");
    std::vector<F> out(n);
    for (size_t i = 0; i < n; i++) out[i] = from16(cb.data() + ELEM * i);
    return out;
}
F horner(const std::vector<F> &poly, F x) {
    F r = 0;
    for (size_t k = poly.size(); k-- > 0;) r = hf_add(hf_mul(r, x), poly[k]);
    return r;
}
// a canonical product through the weak form where the field has one (the 128-bit flavour: no data-dependent reduction loops; the
// other flavours: their hf_mul) — for the two loops below that run over trace-length columns
#ifndef HF_CHAIN_MUL
#define HF_CHAIN_MUL hf_mul
#define HF_CHAIN_END(x) (x)
#endif
#ifndef HF_CHAIN_ADD
#define HF_CHAIN_ADD hf_add
#endif
static inline F vf_mul(F a, F b) { return HF_CHAIN_END(HF_CHAIN_MUL(a, b)); }
// one polynomial at many points: the coefficient loop outside, so that the points' chains are independent work for the core (a public
// input register of an air-assembly component is a polynomial as long as the trace, and one serial chain per query is what verifying costs)
std::vector<F> horner_many(const std::vector<F> &poly, const std::vector<F> &xs) {
    std::vector<F> r(xs.size(), (F)0);
    for (size_t k = poly.size(); k-- > 0;) {
        const F c = poly[k];
        for (size_t q = 0; q < xs.size(); q++) r[q] = HF_CHAIN_ADD(HF_CHAIN_MUL(r[q], xs[q]), c);      // weak values inside the chain
    }
    for (F &v : r) v = HF_CHAIN_END(v);
    return r;
}
F f_div(F a, F b) { return hf_mul(a, hf_inv(b)); }
// every element inverted with ONE field inversion (Montgomery's trick); 0 stays 0 (galois' convention)
void batch_invert(std::vector<F> &v) {
    std::vector<F> pre(v.size());
    F acc = 1;
    for (size_t i = 0; i < v.size(); i++) { pre[i] = acc; if (!hf_is_zero(v[i])) acc = hf_mul(acc, v[i]); }
    F inv = hf_inv(acc);
    for (size_t i = v.size(); i-- > 0;) {
        if (hf_is_zero(v[i])) continue;
        const F vi = v[i];
        v[i] = hf_mul(inv, pre[i]);
        inv = hf_mul(inv, vi);
    }
}
std::vector<uint64_t> augmented_rows(const std::vector<uint64_t> &positions, uint64_t column_length) {      // LowDegreeProver.ts:302-309
    std::vector<uint64_t> out;
    for (uint64_t p : positions) out.push_back(p % (column_length / 4));
    return unique_in_order(out);
}

// the constraint evaluator of an AIR given as a register-machine program (kind 1), on host scalars: genstark_amd/air_generic.py
// Program.run (what the device runs in k_air_constraints).  Opcodes as in include/gstark.h.
std::vector<F> run_program(const gs_prover_air &air, const std::vector<F> &cur, const std::vector<F> &nxt, const std::vector<F> &statics) {
    enum { LOADC, LOADR, LOADN, LOADS, ADD, SUB, MUL, POW, POWC, OUT };
    std::vector<F> vm(air.vm_regs ? air.vm_regs : 1, (F)0), out(air.nconstraints, (F)0);
    auto konst = [&](uint32_t i) { if (i >= air.nconsts) fail(GS_ERR_ARG, "program: constant index out of range"); return from16(air.consts + ELEM * i); };
    auto r = [&](uint32_t i) -> F & { if (i >= vm.size()) fail(GS_ERR_ARG, "program: register out of range"); return vm[i]; };
    for (uint32_t k = 0; k < air.e_ninstr; k++) {
        const uint32_t op = air.e_code[4 * k], d = air.e_code[4 * k + 1], a = air.e_code[4 * k + 2], b = air.e_code[4 * k + 3];
        switch (op) {
            case LOADC: r(d) = konst(a); break;
            case LOADR: if (a >= cur.size()) fail(GS_ERR_ARG, "program: trace register out of range"); r(d) = cur[a]; break;
            case LOADN: if (a >= nxt.size()) fail(GS_ERR_ARG, "program: trace register out of range"); r(d) = nxt[a]; break;
            case LOADS: if (a >= statics.size()) fail(GS_ERR_ARG, "program: static register out of range"); r(d) = statics[a]; break;
            case ADD: r(d) = hf_add(r(a), r(b)); break;
            case SUB: r(d) = hf_sub(r(a), r(b)); break;
            case MUL: r(d) = hf_mul(r(a), r(b)); break;
            case POW: r(d) = hf_pow(r(a), (hfe)(uint64_t)b); break;
            case POWC: r(d) = hf_pow(r(a), konst(b)); break;
            case OUT: if (d >= out.size()) fail(GS_ERR_ARG, "program: output out of range"); out[d] = r(a); break;
            default: fail(GS_ERR_ARG, "program: unknown opcode %u", op);
        }
    }
    return out;
}

// coefficients of the polynomial through `values` on the m-th roots of unity {g^i} (m a power of two): an inverse DFT of size m on host
// scalars — the O(m^2) sum for the short periods of cyclic registers, a radix-2 transform for the columns of input registers (a public
// input register of an air-assembly component can be as long as the trace)
std::vector<F> cyclic_poly(const std::vector<F> &values, F g) {
    const size_t m = values.size();
    const F ginv = hf_inv(g), minv = hf_inv((F)(uint64_t)m);
    std::vector<F> out(m);
    if (m <= 32) {
        std::vector<F> pw(m);
        F cur = 1;
        for (size_t i = 0; i < m; i++) { pw[i] = cur; cur = hf_mul(cur, ginv); }
        for (size_t j = 0; j < m; j++) {
            F s = 0;
            for (size_t i = 0; i < m; i++) s = hf_add(s, hf_mul(values[i], pw[(i * j) % m]));
            out[j] = hf_mul(s, minv);
        }
        return out;
    }
    uint32_t logm = 0;
    while ((1ull << logm) < m) logm++;
    for (size_t i = 0; i < m; i++) {                              // bit-reversed copy, then decimation-in-time butterflies with g^-1
        size_t r = 0;
        for (uint32_t b = 0; b < logm; b++) r |= ((i >> b) & 1) << (logm - 1 - b);
        out[r] = values[i];
    }
    for (size_t half = 1; half < m; half <<= 1) {
        const F wlen = hf_pow(ginv, (hfe)(uint64_t)(m / (2 * half)));
        std::vector<F> tw(half);
        F cur = 1;
        for (size_t k = 0; k < half; k++) { tw[k] = cur; cur = hf_mul(cur, wlen); }
        for (size_t base = 0; base < m; base += 2 * half)
            for (size_t k = 0; k < half; k++) {
                const F u = out[base + k], v = vf_mul(out[base + k + half], tw[k]);
                out[base + k] = hf_add(u, v);
                out[base + k + half] = hf_sub(u, v);
            }
    }
    for (size_t j = 0; j < m; j++) out[j] = vf_mul(out[j], minv);
    return out;
}
// the shortest power-of-two period of a column (a cyclic register of that length denotes the same polynomial): airassembly.py _shrink
void shrink_column(std::vector<F> &col) {
    while (col.size() > 1 && col.size() % 2 == 0) {
        const size_t half = col.size() / 2;
        bool same = true;
        for (size_t i = 0; i < half && same; i++) same = col[i] == col[half + i];
        if (!same) break;
        col.resize(half);
    }
}
// a column rotated right by `shift` steps: airassembly.py _rotate (new[i] = old[(i - shift) mod len])
void rotate_column(std::vector<F> &col, int32_t shift) {
    if (col.empty()) return;
    const int64_t len = (int64_t)col.size();
    const int64_t k = (((-(int64_t)shift) % len) + len) % len;
    std::rotate(col.begin(), col.begin() + k, col.end());
}

void verify_impl(const gs_prover_job &job, const uint8_t *proof, uint64_t proof_len) {
    const gs_prover_air &air = job.air;
    const uint64_t E = job.extension_factor;
    const uint32_t R = air.registers, S = air.nsecret;
    const int alg = job.hash_alg;
    if (!E || (E & (E - 1)) || !R || !air.nconstraints) fail(GS_ERR_ARG, "invalid job");
    if (job.nassertions < 1) fail(GS_ERR_ARG, "At least one assertion must be provided");
    if (alg != GS_HASH_SHA256 && alg != GS_HASH_BLAKE2S256) fail(GS_ERR_ARG, "unknown hash algorithm");
    if (air.ninputs && (air.kind != 1 || !air.inputs)) fail(GS_ERR_ARG, "invalid job: input registers belong to a program AIR and need their declarations");

    // ----- parse (lib/Serializer.ts:83-144)
    Reader r{proof, proof_len};
    const uint8_t *evr = r.take(DIGEST);
    Bytes evRoot(evr, evr + DIGEST);
    ParsedMerkleProof evProof = read_merkle_proof(r, (uint64_t)(R + S) * ELEM);
    const uint8_t *lcr = r.take(DIGEST);
    Bytes lcRoot(lcr, lcr + DIGEST);
    ParsedMerkleProof lcProof = read_merkle_proof(r, 4 * ELEM);
    const uint32_t ncomp = r.byte();
    struct Comp { Bytes columnRoot; ParsedMerkleProof columnProof, polyProof; };
    std::vector<Comp> comps(ncomp);
    for (auto &c : comps) {
        const uint8_t *cr = r.take(DIGEST);
        c.columnRoot.assign(cr, cr + DIGEST);
        c.columnProof = read_merkle_proof(r, 4 * ELEM);
        c.polyProof = read_merkle_proof(r, 4 * ELEM);
    }
    uint32_t rlen = r.byte();
    if (!rlen) rlen = (uint32_t)MAX_ARRAY;
    std::vector<F> remainder(rlen);
    for (uint32_t i = 0; i < rlen; i++) remainder[i] = from16(r.take(ELEM));
    // input shapes (lib/Serializer.ts:127-141) — and with them the trace length: the reference sizes the trace from the proof's shapes
    // (initVerificationContext(proof.iShapes, publicInputs), lib/Stark.ts:176).  An AIR without input registers has a fixed trace
    // (job.steps) and a proof of it carries no shapes.
    Shapes shapes;
    {
        const uint32_t nshapes = r.byte();
        shapes.resize(nshapes);
        for (auto &sh : shapes) {
            const uint32_t rank = r.byte();
            const uint8_t *q = r.take(4ull * rank);
            sh.resize(rank);
            for (uint32_t k = 0; k < rank; k++) sh[k] = (uint32_t)q[4 * k] | ((uint32_t)q[4 * k + 1] << 8) | ((uint32_t)q[4 * k + 2] << 16) | ((uint32_t)q[4 * k + 3] << 24);
        }
    }
    // (bytes after the shapes are ignored, as lib/Serializer.ts:83-144 ignores them)
    uint64_t T = job.steps;
    InputLayout layout;
    if (air.ninputs) {
        layout = input_layout(air, shapes);
        if (job.steps && job.steps != layout.length)
            fail(GS_ERR_ARG, "the proof's input shapes lay out a trace of %llu steps, the statement is about %llu", (unsigned long long)layout.length, (unsigned long long)job.steps);
        T = layout.length;
    } else if (!shapes.empty()) {
        fail(GS_ERR_ARG, "malformed proof: %zu input shapes for an AIR without input registers", shapes.size());
    }
    if (!T || (T & (T - 1))) fail(GS_ERR_ARG, "invalid job");
    if (T > (1ull << 32) / E) fail(GS_ERR_ARG, "a trace of %llu steps at extension factor %llu is beyond this verifier", (unsigned long long)T, (unsigned long long)E);
    const uint64_t N = T * E;
    const F omega = domain_root(job, N);
    uint32_t max_degree = 1;
    for (uint32_t i = 0; i < air.nconstraints; i++) max_degree = std::max(max_degree, air.degrees[i]);
    uint64_t cf = 1;
    while (cf < max_degree) cf <<= 1;
    const uint64_t combination_degree = cf * T, composition_degree = std::max(combination_degree - T, T), b_inc = composition_degree - T;
    // the number of FRI layers is a function of the domain size alone (LowDegreeProver.ts:179: fold while more than 256 values are
    // left); a proof with any other count is malformed — in particular one with extra layers, which would floor the degree bound of
    // the remainder to zero and leave the low-degree test with nothing to check
    {
        uint32_t want = 0;
        for (uint64_t len = N; len > MAX_ARRAY; len /= 4) want++;
        if (ncomp != want) fail(GS_ERR_ARG, "malformed proof: %u FRI components, %u expected for a domain of %llu", ncomp, want, (unsigned long long)N);
        const uint64_t rem_want = N >> (2 * want);
        if (rlen != rem_want) fail(GS_ERR_ARG, "malformed proof: remainder of %u values, %llu expected", rlen, (unsigned long long)rem_want);
    }

    // ----- composition polynomial set-up (CompositionPolynomial.ts:29-69): the same coefficient stream as the prover's
    struct RegData { uint32_t reg; std::vector<F> xs, ys, ipoly, zpoly; };
    std::vector<RegData> rdata;
    for (uint32_t i = 0; i < job.nassertions; i++) {
        const gs_assertion &a = job.assertions[i];
        if (a.reg >= R) fail(GS_ERR_ARG, "Invalid assertion: register %u is outside of register bank", a.reg);
        if (a.step >= T) fail(GS_ERR_ARG, "Invalid assertion: step %llu is outside of execution trace", (unsigned long long)a.step);
        RegData *d = nullptr;
        for (auto &e : rdata) if (e.reg == a.reg) d = &e;
        if (!d) { rdata.push_back(RegData{a.reg, {}, {}, {}, {}}); d = &rdata.back(); }
        d->xs.push_back(hf_pow(omega, (hfe)(a.step * E)));
        d->ys.push_back(from16(a.value));
    }
    for (auto &d : rdata) {
        d.ipoly = lagrange(d.xs, d.ys);                                    // BoundaryConstraints.ts:42
        d.zpoly = {(F)1};
        for (F x : d.xs) {                                                 // :24-30
            std::vector<F> nz(d.zpoly.size() + 1, (F)0);
            const F nx = hf_sub(0, x);
            for (size_t k = 0; k < d.zpoly.size(); k++) { nz[k] = hf_add(nz[k], hf_mul(d.zpoly[k], nx)); nz[k + 1] = hf_add(nz[k + 1], d.zpoly[k]); }
            d.zpoly.swap(nz);
        }
    }
    const uint32_t bcount = (uint32_t)rdata.size();
    std::vector<std::pair<uint64_t, std::vector<uint32_t>>> groups;
    for (uint32_t i = 0; i < air.nconstraints; i++) {
        const uint64_t d = (uint64_t)air.degrees[i] * T;
        bool found = false;
        for (auto &g : groups) if (g.first == d) { g.second.push_back(i); found = true; }
        if (!found) groups.push_back({d, {i}});
    }
    uint32_t dcount = air.nconstraints;
    for (auto &g : groups) if (g.first < combination_degree) dcount += (uint32_t)g.second.size();
    const uint32_t bcoef = bcount * (composition_degree > T ? 2 : 1);
    const uint32_t V = R + S, lccount = b_inc > 0 ? 2 * V : V;
    const std::vector<F> coeffs = prng_many(evRoot, dcount + bcoef + lccount);       // d, then b, then the linear combination's (LinearCombination.ts:58-59)
    const F x_last = hf_pow(omega, (hfe)((T - 1) * E));

    // static registers of the AIR at a point (kind 1: K_s(x^(T/period)); kind 0: the round-constant register)
    std::vector<std::vector<F>> static_polys;
    std::vector<uint64_t> static_periods;
    if (air.kind == 0) {
        if (!air.nrc || !air.round_constants) fail(GS_ERR_ARG, "the MiMC AIR needs its round constants");
        if ((air.nrc & (air.nrc - 1)) || T % air.nrc) fail(GS_ERR_ARG, "invalid job: the number of round constants must be a power of two dividing the trace length");
        std::vector<F> rc(air.nrc);
        for (uint32_t i = 0; i < air.nrc; i++) rc[i] = from16(air.round_constants + ELEM * i);
        static_polys.push_back(cyclic_poly(rc, hf_pow(omega, (hfe)(E * (T / air.nrc)))));
        static_periods.push_back(air.nrc);
    } else {
        if (air.nstatic < S) fail(GS_ERR_ARG, "static register list shorter than the secret register count");
        if (air.ninputs && !air.static_sources) fail(GS_ERR_ARG, "invalid job: an AIR with input registers says where each static register takes its values from");
        // which public input register's values are the k-th list of public_inputs (declaration order, lib/Stark.ts:167)
        std::vector<int64_t> public_at(air.ninputs, -1);
        std::vector<uint64_t> public_off;
        if (air.ninputs) {
            uint32_t k = 0;
            uint64_t off = 0;
            for (uint32_t j = 0; j < air.ninputs; j++) {
                if (air.inputs[j].secret) continue;
                if (k >= air.npublic_inputs || !air.public_input_counts || (!air.public_inputs && air.public_input_counts[k]))
                    fail(GS_ERR_ARG, "the values of %u public input registers are needed", (unsigned)std::count_if(air.inputs, air.inputs + air.ninputs, [](const gs_input_register &d) { return !d.secret; }));
                if (air.public_input_counts[k] != layout.count[j]) fail(GS_ERR_ARG, "public input register %u: %llu values expected (the shape in the proof), %llu given", j, (unsigned long long)layout.count[j], (unsigned long long)air.public_input_counts[k]);
                public_at[j] = k++;
                public_off.push_back(off);
                off += layout.count[j];
            }
        }
        uint64_t off = 0;
        uint32_t cyc = 0;
        for (uint32_t s = 0; s + S < air.nstatic; s++) {                     // the public ones (the secret ones follow them and arrive in the leaves)
            const gs_static_source src = air.static_sources ? air.static_sources[s] : gs_static_source{GS_STATIC_CYCLE, 0};
            std::vector<F> vals;
            if (src.kind == GS_STATIC_CYCLE) {
                const uint32_t m = air.static_periods[cyc++];
                if (!m || (m & (m - 1)) || T % m) fail(GS_ERR_ARG, "a static register's period must be a power of two dividing the trace length");
                vals.resize(m);
                for (uint32_t i = 0; i < m; i++) vals[i] = from16(air.static_values + ELEM * (off + i));
                off += m;
            } else if (src.kind == GS_STATIC_INPUT || src.kind == GS_STATIC_MASK) {
                // the column the loader lays out (airassembly.py: _Layout.column / .mask): every value held for `span` steps — or a 1
                // on the first of them —, the whole rotated by the register's shift, then cut to its shortest period
                const uint32_t j = src.index;
                if (j >= air.ninputs) fail(GS_ERR_ARG, "invalid job: static register %u names input register %u of %u", s, j, air.ninputs);
                if (T > (1ull << 26)) fail(GS_ERR_ARG, "a trace of %llu steps is beyond this verifier's input-register columns", (unsigned long long)T);
                const uint64_t span = layout.span[j];
                if (src.kind == GS_STATIC_MASK) {
                    // one period of the mask is all there is to it (span divides the power of two T, so it is a power of two itself): a
                    // proof cannot make the verifier build a trace-length column out of its shapes alone
                    vals.assign(span, (F)0);
                    vals[0] = (F)1;
                } else {
                    vals.resize(T);
                    if (air.inputs[j].secret || public_at[j] < 0) fail(GS_ERR_ARG, "invalid job: static register %u is a public one, input register %u is secret", s, j);
                    const uint8_t *src_vals = air.public_inputs + ELEM * public_off[public_at[j]];
                    for (uint64_t v = 0; v < layout.count[j]; v++) {
                        const F val = from16(src_vals + ELEM * v);
                        for (uint64_t i = 0; i < span; i++) vals[v * span + i] = val;
                    }
                }
                rotate_column(vals, air.inputs[j].shift);
                shrink_column(vals);
            } else {
                fail(GS_ERR_ARG, "invalid job: static register %u has unknown source %u", s, src.kind);
            }
            const uint64_t m = vals.size();
            static_polys.push_back(cyclic_poly(vals, hf_pow(omega, (hfe)(E * (T / m)))));
            static_periods.push_back(m);
        }
    }
    // statics_at[k][pi]: static register k at the pi-th queried point (filled below, once the positions are known)
    std::vector<std::vector<F>> statics_at;
    auto constraints_at = [&](size_t pi, const std::vector<F> &p, const std::vector<F> &n, const std::vector<F> &s) {
        std::vector<F> statics;
        for (size_t k = 0; k < static_polys.size(); k++) statics.push_back(statics_at[k][pi]);
        if (air.kind == 0) {                                                 // examples/mimc/mimc128Assembly.ts:46-51
            const F x3 = hf_mul(hf_mul(p[0], p[0]), p[0]);
            return std::vector<F>{hf_sub(n[0], hf_add(x3, statics[0]))};
        }
        for (F v : s) statics.push_back(v);
        return run_program(air, p, n, statics);
    };

    // ----- spot-check positions and the evaluation tree (lib/Stark.ts:183-216)
    const uint32_t exe_count = (uint32_t)std::min<uint64_t>(job.exe_query_count, N - N / E);
    const std::vector<uint64_t> positions = query_indexes(lcRoot, exe_count, N, (uint32_t)E);
    std::vector<uint64_t> aug;
    for (uint64_t p : positions) { aug.push_back(p); aug.push_back((p + E) % N); }
    aug = unique_in_order(aug);
    if (evProof.values.size() != aug.size()) fail(GS_ERR_ARG, "malformed proof: the evaluation proof does not hold one leaf per queried position");
    std::map<uint64_t, size_t> at_leaf;
    for (size_t i = 0; i < aug.size(); i++) at_leaf[aug[i]] = i;
    auto leaf_values = [&](uint64_t pos, std::vector<F> &p, std::vector<F> &s) {
        const Bytes &b = evProof.values[at_leaf.at(pos)];
        p.resize(R); s.resize(S);
        for (uint32_t k = 0; k < R; k++) p[k] = from16(b.data() + ELEM * k);
        for (uint32_t k = 0; k < S; k++) s[k] = from16(b.data() + ELEM * (R + k));
    };
    if (!merkle_check(alg, evRoot, aug, evProof)) fail(GS_ERR_ARG, "Verification of evaluation Merkle proof failed");

    // ----- transition and boundary constraints at every position (:218-234; CompositionPolynomial.ts:150-191; LinearCombination.ts:66-88)
    // (the divisions of all positions share one inversion: first the denominators — x^T - 1 of Z(x), Z_r(x) of every asserted register —
    //  then the values)
    std::vector<F> lcValues, xsq, dens;
    for (uint64_t step : positions) {
        const F x = hf_pow(omega, (hfe)step);
        xsq.push_back(x);
        dens.push_back(hf_sub(hf_pow(x, (hfe)T), 1));                                                  // ZeroPolynomial.ts:28-34: Z = (x^T - 1) / (x - x_last)
        for (auto &d : rdata) dens.push_back(horner(d.zpoly, x));                                      // BoundaryConstraints.ts:55-69
    }
    batch_invert(dens);
    for (size_t k = 0; k < static_polys.size(); k++) {                        // K_s(x^(T/period)) at every queried x
        std::vector<F> at(xsq.size());
        for (size_t pi = 0; pi < xsq.size(); pi++) at[pi] = hf_pow(xsq[pi], (hfe)(T / static_periods[k]));
        statics_at.push_back(horner_many(static_polys[k], at));
    }
    size_t di = 0;
    for (size_t pi = 0; pi < positions.size(); pi++) {
        const uint64_t step = positions[pi];
        const F x = xsq[pi];
        std::vector<F> p, n, s, unused;
        leaf_values(step, p, s);
        leaf_values((step + E) % N, n, unused);
        std::vector<F> q = constraints_at(pi, p, n, s);
        if (q.size() != air.nconstraints) fail(GS_ERR_ARG, "constraint evaluator returned the wrong number of values");
        for (auto &g : groups) {
            if (g.first == combination_degree) continue;
            const F power = hf_pow(x, (hfe)(combination_degree - g.first));
            for (uint32_t i : g.second) q.push_back(hf_mul(q[i], power));
        }
        F qc = 0;
        for (size_t k = 0; k < q.size(); k++) qc = hf_add(qc, hf_mul(q[k], coeffs[k]));
        const F dValue = hf_mul(hf_mul(qc, hf_sub(x, x_last)), dens[di++]);                             // Q / Z = Q (x - x_last) / (x^T - 1)
        std::vector<F> b;
        for (auto &d : rdata) b.push_back(hf_mul(hf_sub(p[d.reg], horner(d.ipoly, x)), dens[di++]));
        const F xb = hf_pow(x, (hfe)b_inc);
        if (b_inc > 0) for (uint32_t i = 0; i < bcount; i++) b.push_back(hf_mul(b[i], xb));
        F bValue = 0;
        for (size_t k = 0; k < b.size(); k++) bValue = hf_add(bValue, hf_mul(b[k], coeffs[dcount + k]));
        const F cValue = hf_add(dValue, bValue);
        std::vector<F> ps(p);
        ps.insert(ps.end(), s.begin(), s.end());
        if (b_inc > 0) for (uint32_t i = 0; i < V; i++) ps.push_back(hf_mul(ps[i], xb));
        F comb = 0;
        for (size_t k = 0; k < ps.size(); k++) comb = hf_add(comb, hf_mul(ps[k], coeffs[dcount + bcoef + k]));
        lcValues.push_back(hf_add(cValue, comb));
    }

    // ----- low-degree proof (LowDegreeProver.ts:70-172)
    uint64_t column_length = N;
    auto column_values = [&](const ParsedMerkleProof &mp, const std::vector<uint64_t> &pos, const std::vector<uint64_t> &rows, uint64_t clen) {   // :264-282
        const uint64_t row_len = clen / 4;
        std::vector<F> out;
        for (uint64_t p : pos) {
            size_t idx = rows.size();
            for (size_t k = 0; k < rows.size(); k++) if (rows[k] == p % row_len) { idx = k; break; }
            if (idx >= mp.values.size()) fail(GS_ERR_ARG, "malformed proof: a queried row is missing");
            out.push_back(from16(mp.values[idx].data() + ELEM * (p / row_len)));
        }
        return out;
    };
    {
        const std::vector<uint64_t> lc_rows = augmented_rows(positions, column_length);
        if (lcProof.values.size() != lc_rows.size()) fail(GS_ERR_ARG, "malformed proof: the linear-combination proof does not hold one row per queried position");
        const std::vector<F> checks = column_values(lcProof, positions, lc_rows, column_length);
        if (!merkle_check(alg, lcRoot, lc_rows, lcProof)) fail(GS_ERR_ARG, "Verification of linear combination Merkle proof failed");
        for (size_t i = 0; i < lcValues.size(); i++)
            if (lcValues[i] != checks[i]) fail(GS_ERR_ARG, "Verification of linear combination correctness failed");
    }
    Bytes pRoot = lcRoot;
    F rou = omega;
    uint64_t max_degree_plus1 = composition_degree;
    column_length /= 4;
    const F zeta[4] = {(F)1, hf_pow(omega, (hfe)(N / 4)), hf_pow(omega, (hfe)(N / 2)), hf_pow(omega, (hfe)(N / 4 * 3))};      // :75-77
    const F inv4 = hf_inv((F)4);
    uint64_t domain_size = N;                  // order of `rou`
    for (uint32_t depth = 0; depth < ncomp; depth++) {
        Comp &c = comps[depth];
        if (column_length < 4) fail(GS_ERR_ARG, "malformed proof: too many FRI components");
        const std::vector<uint64_t> pos = query_indexes(c.columnRoot, job.fri_query_count, column_length, (uint32_t)E);
        const std::vector<uint64_t> rows = augmented_rows(pos, column_length);
        if (c.columnProof.values.size() != rows.size() || c.polyProof.values.size() != pos.size()) fail(GS_ERR_ARG, "malformed proof: wrong number of rows at depth %u", depth);
        const std::vector<F> col = column_values(c.columnProof, pos, rows, column_length);
        if (!merkle_check(alg, c.columnRoot, rows, c.columnProof)) fail(GS_ERR_ARG, "Verification of column Merkle proof failed at depth %u", depth);
        if (!merkle_check(alg, pRoot, pos, c.polyProof)) fail(GS_ERR_ARG, "Verification of polynomial Merkle proof failed at depth %u", depth);
        const F special = prng_one(pRoot);                                                            // :132
        // the cubic through (x zeta^k, y_k), k < 4 (:123-137: interpolateQuarticBatch + evalQuarticBatch), evaluated at `special`:
        // (u0 + u1 t + u2 t^2 + u3 t^3) / 4 with u the inverse 4-point DFT of y and t = special / x — the same polynomial, no
        // inversion per position (x^-1 = rou^(order - position))
        for (size_t i = 0; i < pos.size(); i++) {
            const F xinv = hf_pow(rou, (hfe)((domain_size - pos[i]) % domain_size));
            F y[4];
            for (int k = 0; k < 4; k++) y[k] = from16(c.polyProof.values[i].data() + ELEM * k);
            const F s0 = hf_add(y[0], y[2]), s1 = hf_sub(y[0], y[2]), s2 = hf_add(y[1], y[3]), s3 = hf_mul(hf_sub(y[1], y[3]), zeta[3]);     // zeta^-1 = zeta^3
            const F u0 = hf_add(s0, s2), u2 = hf_sub(s0, s2), u1 = hf_add(s1, s3), u3 = hf_sub(s1, s3);
            const F t = hf_mul(special, xinv);
            F v = hf_add(hf_mul(u3, t), u2);
            v = hf_add(hf_mul(v, t), u1);
            v = hf_add(hf_mul(v, t), u0);
            if (hf_mul(v, inv4) != col[i]) fail(GS_ERR_ARG, "Degree 4 polynomial didn't evaluate to column value at depth %u", depth);
        }
        domain_size /= 4;
        pRoot = c.columnRoot;
        rou = hf_pow(rou, (hfe)4);
        max_degree_plus1 /= 4;
        column_length /= 4;
    }
    // ----- remainder (:155-171)
    if (max_degree_plus1 > remainder.size()) fail(GS_ERR_ARG, "Remainder degree is greater than number of remainder values");
    // the bound is floor(floor(d / 4) / 4 ...): with a tiny trace under a large extension factor it floors to 0 although a fold never
    // takes a polynomial below a constant.  Untrusted input never gets the prover's `m == 0: nothing to check` shortcut: the remainder
    // must then be a constant (the ceiling of the same divisions)
    if (!max_degree_plus1) max_degree_plus1 = 1;
    if (remainder.size() < 4 || (remainder.size() & 3)) fail(GS_ERR_ARG, "malformed proof: remainder length");
    {
        // the tree over the rows of transposeVector(remainder, 4) must be the last column's tree
        const uint64_t rows = remainder.size() / 4;
        std::vector<Bytes> level(rows, Bytes(32));
        for (uint64_t i = 0; i < rows; i++) {
            uint8_t msg[4 * GS_PROVER_ELT_MAX];
            for (int k = 0; k < 4; k++) le16(remainder[i + k * rows], msg + ELEM * k);
            host_digest(alg, msg, 4 * ELEM, level[i].data());
        }
        if (rows & (rows - 1)) fail(GS_ERR_ARG, "malformed proof: remainder length");
        while (level.size() > 1) {
            std::vector<Bytes> up(level.size() / 2, Bytes(32));
            for (size_t i = 0; i < up.size(); i++) {
                uint8_t buf[64];
                memcpy(buf, level[2 * i].data(), 32); memcpy(buf + 32, level[2 * i + 1].data(), 32);
                host_digest(alg, buf, 64, up[i].data());
            }
            level.swap(up);
        }
        if (level[0] != pRoot) fail(GS_ERR_ARG, "Remainder values do not match Merkle root of the last column");
    }
    if (!remainder_is_low_degree(remainder, E, max_degree_plus1, rou, 1))                              // :223-252
        fail(GS_ERR_ARG, "Remainder is not a valid degree %llu polynomial", (unsigned long long)(max_degree_plus1 - 1));
}

}  // namespace

extern "C" {

static int verify_entry(const struct gs_prover_job *job, const uint8_t *proof, uint64_t len, char *err, uint64_t errcap) {
    if (!job || !proof) return GS_ERR_ARG;
    try {
        verify_impl(*job, proof, len);
        return GS_OK;
    } catch (const Fail &f) {
        if (err && errcap) snprintf(err, (size_t)errcap, "%s", f.msg.c_str());
        return f.code ? f.code : GS_ERR_ARG;
    } catch (const std::exception &e) {
        if (err && errcap) snprintf(err, (size_t)errcap, "%s", e.what());
        return GS_ERR_ARG;
    }
}
int gs_prover_verify(const struct gs_prover_job *job, const uint8_t *proof, uint64_t len, char *err, uint64_t errcap) {
    if (!g_bound) return GS_ERR_UNSUPPORTED;
    UseApi use(&g_default_api);
    return verify_entry(job, proof, len, err, errcap);
}
int gs_prover_verify_on(const gs_prover_binding *b, const struct gs_prover_job *job, const uint8_t *proof, uint64_t len, char *err, uint64_t errcap) {
    if (!b) return GS_ERR_ARG;
    UseApi use(reinterpret_cast<const Api *>(b));
    return verify_entry(job, proof, len, err, errcap);
}

}  // extern "C"
