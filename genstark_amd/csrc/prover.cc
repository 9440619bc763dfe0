// prover.cc — the whole of Stark.prove() (lib/Stark.ts:81-163) as NATIVE host code above the C ABI.
//
// The Python package mirrors lib/Stark.ts and lib/components/*.ts call by call (genstark_amd/stark.py, components/); that
// mirror is the readable reference for the sequence below and stays the thing the parity tests compare against.  This file
// is the same sequence — context set-up, trace, P(x), low-degree extension, evaluation tree, CompositionPolynomial
// (CompositionPolynomial.ts:29-146, BoundaryConstraints.ts:15-95, ZeroPolynomial.ts:36-44), LinearCombination (:36-64),
// LowDegreeProver (:39-68, :176-252), QueryIndexGenerator, the spot checks and Serializer.serializeProof (:35-79) — without an
// interpreter between two launches: ~250 ABI calls per proof issue back to back, the few host-side steps (Fiat-Shamir
// hashing of roots, Lagrange interpolation of <= 256 points, the authentication-path plans) run on native limbs.
//
// It is written against include/gstark.h ONLY (plain C++, no HIP): gs_prover_bind() resolves the entry points from whatever
// implementation of the ABI the caller loaded, so the product binds libgstark_hip.so and the CPU tests bind the oracle's
// implementation, exactly like the Python mirror.  Output: the serialized proof (the bytes Serializer.serializeProof gives).
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>

#include <algorithm>
#include <map>
#include <chrono>
#include <deque>
#include <string>
#include <vector>

#include "../../include/gstark.h"
#include "../../include/gstark_prover.h"
#include "../../include/gstark_comm.h"
// One build of this file per field flavour of the ABI library (csrc/build.sh: the same -DGS_SMALL_Q / -DGS_WIDE_BITS): the host-side
// scalars (domain roots, Fiat-Shamir coefficients, boundary interpolants, the remainder check) use the flavour's host arithmetic,
// elements are gs_element_size() bytes on the ABI and in the proof.
#if defined(GS_WIDE_BITS)
#include "gf_wide.h"
#endif
#include "host_field.h"
#include "host_sha256.h"

namespace {

// ---- the slice of the ABI this driver uses, resolved at bind time ------------------------------------------------------
#define GS_API_LIST(X)                                                                                                        \
    X(gs_alloc) X(gs_free) X(gs_upload) X(gs_download) X(gs_gather) X(gs_last_error) X(gs_power_series) X(gs_vec_add) X(gs_vec_mul) \
    X(gs_vec_sub_scalar) X(gs_vec_div) X(gs_combine_many) X(gs_combine_adjusted) X(gs_pluck) X(gs_transpose_vector) X(gs_sub_matrix_from_vectors)     \
    X(gs_eval_polys_at_roots) X(gs_interpolate_roots) X(gs_interpolate_quartic_domain) X(gs_eval_quartic_batch)                \
    X(gs_hash_merge_rows) X(gs_hash_digest_values) X(gs_merkle_build) X(gs_merkle_commit_rows) X(gs_merkle_prove_batch) X(gs_small_interpolate)          \
    X(gs_small_eval_poly) X(gs_pseudorandom_indexes) X(gs_mimc_trace) X(gs_mimc_constraints) X(gs_air_trace)                    \
    X(gs_air_trace_segments) X(gs_air_constraints) X(gs_air_constraints_strided) X(gs_composition_tail) X(gs_composition_tail_coset) X(gs_zero_poly_inverses) X(gs_div_by_domain_roots) X(gs_mimc_composition) X(gs_fri_fold) X(gs_fri_fold_seeded) X(gs_defer_begin) X(gs_defer_end) X(gs_readback_post) X(gs_readback_wait) X(gs_merkle_commit_rows_seed) X(gs_fri_fold_at) \
    X(gs_vec_mul_scalar) X(gs_copy) X(gs_gather_words) X(gs_transpose_records) X(gs_fri_fold_seeded_scaled) X(gs_fri_layers) X(gs_sync) X(gs_zero_poly_inverses_coset) X(gs_div_by_domain_roots_coset)
struct Api {
#define X(name) decltype(&::name) name = nullptr;
    GS_API_LIST(X)
#undef X
};
// The driver is written against `A.gs_xxx(...)`: A is the binding of the CURRENT call on this thread — the process-wide default
// (gs_prover_bind) unless the entry point was handed a binding of its own (gs_prover_open, the `_on` entry points).
Api g_default_api;
bool g_bound = false;
thread_local const Api *g_api = &g_default_api;
#define A (*g_api)
struct UseApi {          // scope of one entry point
    const Api *saved;
    explicit UseApi(const Api *a) : saved(g_api) { g_api = a; }
    ~UseApi() { g_api = saved; }
};

struct Fail {
    int code;
    std::string msg;
};
[[noreturn]] void fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Fail{code, buf};
}

typedef hfe F;
#if defined(GS_WIDE_BITS)
const uint64_t ELEM = sizeof(fe);               // 32: the 256- / 224-bit fields
inline bool operator==(const F &a, const F &b) { return fe_eq(a.v, b.v); }
inline bool operator!=(const F &a, const F &b) { return !fe_eq(a.v, b.v); }
#else
const uint64_t ELEM = 16;
#endif
const uint64_t DIGEST = 32, MAX_ARRAY = 256;
const uint64_t WORD = 16, EW = ELEM / WORD;     // gs_gather_words / gs_defer_* move 16-byte words
static_assert(ELEM <= GS_PROVER_ELT_MAX, "element wider than the job's scalar fields");
typedef std::vector<uint8_t> Bytes;

struct Ctx {
    gs_ctx *c;
    void check(int rc, const char *what) {
        if (rc) fail(rc, "%s: %s", what, c ? A.gs_last_error(c) : "error");
    }
};

// device block owned for the duration of a proof (gs_free parks it in the context's cache: no synchronisation)
struct Buf {
    Ctx *x = nullptr;
    void *p = nullptr;
    Buf() {}
    Buf(Ctx &cx, uint64_t bytes) : x(&cx) { cx.check(A.gs_alloc(cx.c, bytes ? bytes : 16, &p), "gs_alloc"); }
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    Buf(Buf &&o) noexcept : x(o.x), p(o.p) { o.p = nullptr; }
    Buf &operator=(Buf &&o) noexcept {
        if (this != &o) { release(); x = o.x; p = o.p; o.p = nullptr; }
        return *this;
    }
    void release() { if (p) { A.gs_free(x->c, p); p = nullptr; } }
    ~Buf() { release(); }
    uint8_t *at(uint64_t byte_offset) const { return (uint8_t *)p + byte_offset; }
};

void le16(F v, uint8_t *out) { hf_store(out, v); }          // ELEM bytes, little-endian (the names date from the 128-bit-only driver)
F from16(const uint8_t *b) { return hf_load(b); }

// ---- galois prng (genstark_amd/field.py: prng — restated) and the index generator -----------------
F digest_mod_p(const uint8_t d[32]) {          // 256-bit big-endian integer mod p
#if !defined(GS_WIDE_BITS) && !defined(GS_SMALL_Q)
    F hi = 0, lo = 0;
    for (int i = 0; i < 16; i++) hi = (hi << 8) | d[i];
    for (int i = 16; i < 32; i++) lo = (lo << 8) | d[i];
    return hf_reduce(hi, lo);
#else
    F x = 0;                                    // byte-wise Horner: the same code for every other modulus
    const F b = 256;
    for (int i = 0; i < 32; i++) x = hf_add(hf_mul(x, b), (F)(uint64_t)d[i]);
    return x;
#endif
}
std::vector<F> prng_many(const Bytes &seed, size_t count) {
    std::vector<F> out(count);
    uint8_t st[32], msg[32];
    host_sha256(seed.data(), seed.size(), st);
    for (size_t i = 0; i < count; i++) {
        out[i] = digest_mod_p(st);
        int n = host_bigint_bytes(st, 32, msg);
        host_sha256(msg, (size_t)n, st);
    }
    return out;
}
F prng_one(const Bytes &seed) {
    uint8_t st[32];
    host_sha256(seed.data(), seed.size(), st);
    return digest_mod_p(st);
}
std::vector<uint64_t> query_indexes(const Bytes &seed, uint32_t count, uint64_t max, uint32_t exclude) {
    uint64_t max_count = exclude ? max - max / exclude : max;
    if (max_count < count) fail(GS_ERR_ARG, "Cannot select %u unique pseudorandom indexes from %llu values", count, (unsigned long long)max);
    std::vector<uint64_t> out(count ? count : 1);
    if (A.gs_pseudorandom_indexes(seed.data(), (uint32_t)seed.size(), count, max, exclude, out.data()))
        fail(GS_ERR_ARG, "Could not generate %u pseudorandom indexes", count);
    out.resize(count);
    return out;
}

// ---- Merkle batch proofs and the wire format (lib/Serializer.ts:35-79, lib/utils/serialization.ts) ------------------
struct MerkleProof {          // flat: one allocation per part instead of one per digest (a proof holds ~10^4 of them)
    Bytes values;              // nvalues records of value_size bytes (the queried rows)
    uint64_t value_size = 0;
    uint32_t nvalues = 0;
    Bytes nodes;               // digests, column after column
    std::vector<uint32_t> lens;
    uint8_t depth = 0;
};
void write_merkle_proof(Bytes &out, const MerkleProof &p, uint64_t leaf_size) {
    // values: lib/utils/serialization.ts writeArray
    if (!p.nvalues) fail(GS_ERR_ARG, "Array cannot be zero-length");
    if (p.nvalues > MAX_ARRAY) fail(GS_ERR_ARG, "Array length (%u) cannot exceed 256", p.nvalues);
    out.push_back(p.nvalues == MAX_ARRAY ? 0 : (uint8_t)p.nvalues);
    out.insert(out.end(), p.values.begin(), p.values.begin() + (uint64_t)p.nvalues * p.value_size);
    // nodes: writeMatrix
    if (p.lens.size() > MAX_ARRAY) fail(GS_ERR_ARG, "Matrix column count (%zu) cannot exceed 256", p.lens.size());
    out.push_back(p.lens.size() == MAX_ARRAY ? 0 : (uint8_t)p.lens.size());
    uint64_t total = 0;
    for (uint32_t len : p.lens) {
        if (len >= 128) fail(GS_ERR_ARG, "Matrix column length (%u) cannot exceed 127", len);
        const uint8_t type = (len && DIGEST == leaf_size) ? 1 : 0;
        out.push_back((uint8_t)((len << 1) | type));
        total += len;
    }
    out.insert(out.end(), p.nodes.begin(), p.nodes.begin() + total * DIGEST);
    out.push_back(p.depth);
}

struct Tree {
    Buf leaves, nodes;   // digests; nodes in heap order
    uint64_t n = 0;
    Bytes root;
    Buf point;           // FRI trees: prng(root), derived by the launch that produced the root (gs_merkle_commit_rows_seed)
    uint64_t ticket = 0; // ... which also posted the root to the host
};
// Hash.mergeVectorRows(vectors) + MerkleTree.create (lib/Stark.ts:115-118; LowDegreeProver.ts:45-46, 201-202) as one ABI call.
// read_root = false leaves the root on the device (nodes + DIGEST): the FRI layers derive their evaluation points from it there
// (gs_fri_fold_seeded) and all roots come back in one round trip
Tree commit_rows(Ctx &x, int alg, const void *const *vecs, uint32_t count, uint64_t n, bool read_root = true) {
    Tree t;
    t.n = n;
    t.leaves = Buf(x, n * DIGEST);
    t.nodes = Buf(x, n * DIGEST);
    x.check(A.gs_merkle_commit_rows(x.c, (gs_hash_alg)alg, vecs, count, n, t.leaves.p, t.nodes.p), "gs_merkle_commit_rows");
    t.root.resize(DIGEST);
    if (read_root) x.check(A.gs_download(x.c, t.root.data(), t.nodes.at(DIGEST), DIGEST), "gs_download(root)");
    return t;
}
// gs_defer_begin ... gs_defer_end as a scope: when a call inside the window throws, the window is still closed (the context would
// otherwise stay in deferral mode and the next prove() on it would fail with "already deferring")
struct DeferWindow {
    Ctx &x;
    bool open = false;
    explicit DeferWindow(Ctx &cx) : x(cx) { x.check(A.gs_defer_begin(x.c), "gs_defer_begin"); open = true; }
    void end() { open = false; x.check(A.gs_defer_end(x.c), "gs_defer_end"); }
    ~DeferWindow() { if (open) A.gs_defer_end(x.c); }
};
// The query answers of a proof are issued inside one deferral window (gs_defer_begin / gs_defer_end): the calls below queue the
// device work into buffers that stay put (std::deque) and `unpack` builds the proof objects after the single synchronisation.
struct Readbacks {
    struct Batch { MerkleProof *mp; Bytes digests; uint32_t ncols = 0; };
    struct Rows { MerkleProof *mp; Bytes raw; uint64_t rec = 0; uint32_t n = 0; };
    std::deque<Batch> batches;
    std::deque<Rows> rows;
    void prove_batch(Ctx &x, const Tree &t, const std::vector<uint64_t> &idx, MerkleProof *mp) {
        int depth = 0;
        while ((1ull << depth) < t.n) depth++;
        mp->depth = (uint8_t)depth;
        const uint32_t count = (uint32_t)idx.size();
        if (!count) return;
        batches.emplace_back();
        Batch &b = batches.back();
        b.mp = mp;
        const uint64_t cap = (uint64_t)count * (depth ? depth : 1);
        b.digests.resize(count * DIGEST);         // the leaf digests: the proof carries the rows themselves instead (gather below)
        mp->nodes.resize(cap * DIGEST);
        mp->lens.assign(count, 0);
        x.check(A.gs_merkle_prove_batch(x.c, t.leaves.p, t.nodes.p, t.n, idx.data(), count, b.digests.data(), &b.ncols, mp->lens.data(), mp->nodes.data(), cap),
                "gs_merkle_prove_batch");
        mp->lens.resize(b.ncols);                 // the shape is known at once; the digests arrive with gs_defer_end
    }
    void gather(Ctx &x, const void *src, uint64_t rec, const std::vector<uint64_t> &idx, uint64_t per_row, MerkleProof *mp) {
        if (idx.empty()) return;
        mp->values.resize(idx.size() * rec);
        mp->value_size = rec * per_row;
        mp->nvalues = (uint32_t)(idx.size() / per_row);
        x.check(A.gs_gather(x.c, src, rec, idx.data(), idx.size(), mp->values.data()), "gs_gather");
    }
    // rows of transposeVector(column, 4), read strided (see gather_rows4)
    void gather_rows4(Ctx &x, const void *column, uint64_t nrows, const std::vector<uint64_t> &positions, MerkleProof *mp) {
        std::vector<uint64_t> idx;
        idx.reserve(positions.size() * 4);
        for (uint64_t r : positions)
            for (uint64_t c = 0; c < 4; c++) idx.push_back(r + c * nrows);
        gather(x, column, ELEM, idx, 4, mp);
    }
};

MerkleProof prove_batch(Ctx &x, const Tree &t, const std::vector<uint64_t> &idx) {
    MerkleProof mp;
    Readbacks rb;
    rb.prove_batch(x, t, idx, &mp);               // outside a deferral window the call returns with the digests in place
    return mp;
}
// rows of transposeVector(column, 4) without the transposed copy: row r = column[r], column[r + rows], column[r + 2 rows], column[r + 3 rows]
void gather_rows4(Ctx &x, const void *column, uint64_t rows, const std::vector<uint64_t> &positions, MerkleProof *mp) {
    Readbacks rb;
    rb.gather_rows4(x, column, rows, positions, mp);
}
// tree over the rows of transposeVector(column, 4) (Hash.digestValues of the transposed matrix + MerkleTree.create, LowDegreeProver.ts:45-46,
// 201-202): the row digests are mergeVectorRows of the four quarters of the column (the same 64-byte messages)
// want_point: a layer will be folded at prng(this tree's root); the root is posted to the host either way (t.ticket)
Tree commit_rows4(Ctx &x, int alg, const void *column, uint64_t rows, bool want_point) {
    const void *quarters[4];
    for (uint64_t c = 0; c < 4; c++) quarters[c] = (const uint8_t *)column + c * rows * ELEM;
    Tree t;
    t.n = rows;
    t.leaves = Buf(x, rows * DIGEST);
    t.nodes = Buf(x, rows * DIGEST);
    if (want_point) t.point = Buf(x, ELEM);
    x.check(A.gs_merkle_commit_rows_seed(x.c, (gs_hash_alg)alg, quarters, 4, rows, t.leaves.p, t.nodes.p, want_point ? t.point.p : nullptr, &t.ticket),
            "gs_merkle_commit_rows_seed");
    t.root.resize(DIGEST);
    return t;
}
// ---- input registers of an air-assembly component: where every value of every input register sits in the trace, from the registers'
// declarations and the SHAPES of their values alone (genstark_amd/airassembly.py: _Layout, restated).  prove() checks the shapes it
// serializes with this; verify() sizes the trace from the shapes a proof carries (initVerificationContext(proof.iShapes, ...),
// lib/Stark.ts:176).  Messages as the Python loader words them.
typedef std::vector<std::vector<uint32_t>> Shapes;
struct InputLayout {
    std::vector<uint32_t> depth;
    std::vector<uint64_t> span, count;      // steps one value is held; values of the register
    uint64_t length = 0;                    // trace steps the inputs lay out
};
const uint64_t MAX_LAYOUT = 1ull << 40;     // products are capped here: nothing this driver proves or verifies is longer
uint64_t capped_mul(uint64_t a, uint64_t b) {
    if (a && b > MAX_LAYOUT / a) return MAX_LAYOUT;
    return std::min(a * b, MAX_LAYOUT);
}
InputLayout input_layout(const gs_prover_air &air, const Shapes &shapes) {
    const uint32_t n = air.ninputs;
    const gs_input_register *in = air.inputs;
    if (shapes.size() != n) fail(GS_ERR_ARG, "%u input registers: one entry (shape) for each is needed, got %zu", n, shapes.size());
    InputLayout L;
    L.depth.resize(n); L.span.assign(n, 0); L.count.resize(n);
    for (uint32_t j = 0; j < n; j++) {
        const int32_t ref = in[j].parent >= 0 ? in[j].parent : in[j].peer;
        if ((in[j].parent >= 0 || in[j].peer >= 0) && !(ref >= 0 && (uint32_t)ref < j))
            fail(GS_ERR_ARG, "input register: childof / peerof must name an earlier input register");
        if (in[j].parent >= 0 && in[j].peer >= 0 && (uint32_t)in[j].peer >= j) fail(GS_ERR_ARG, "input register: childof / peerof must name an earlier input register");
        L.depth[j] = (in[j].parent < 0 && in[j].peer < 0) ? 0 : L.depth[ref] + (in[j].parent >= 0 ? 1 : 0);
        if (shapes[j].size() != (size_t)L.depth[j] + 1) fail(GS_ERR_ARG, "input register %u: values nested %u deep expected", j, L.depth[j] + 1);
        if (in[j].peer >= 0 && shapes[j] != shapes[in[j].peer]) fail(GS_ERR_ARG, "input register %u: the shape of its peer %d expected", j, in[j].peer);
        if (in[j].parent >= 0 && !std::equal(shapes[j].begin(), shapes[j].end() - 1, shapes[in[j].parent].begin(), shapes[in[j].parent].end()))
            fail(GS_ERR_ARG, "input register %u: one list per value of register %d expected", j, in[j].parent);
    }
    // steps one value of a register is held: its own (steps n), or what its children take (`known`: a span of 0 steps — a child list of
    // zero values — is a value like any other here, as in the loader; the length rule below refuses it)
    std::vector<char> known(n, 0);
    for (uint32_t j = 0; j < n; j++) if (in[j].steps) { L.span[j] = in[j].steps; known[j] = 1; }
    for (bool changed = true; changed;) {
        changed = false;
        for (uint32_t j = 0; j < n; j++) {
            if (known[j] && in[j].parent >= 0) {
                const uint64_t want = capped_mul(L.span[j], shapes[j].back());
                const uint32_t root = (uint32_t)in[j].parent;
                if (!known[root]) { L.span[root] = want; known[root] = 1; changed = true; }
                else if (L.span[root] != want && !in[root].steps) fail(GS_ERR_ARG, "input registers: the children of one register take different numbers of steps");
            }
            if (!known[j] && in[j].peer >= 0 && known[in[j].peer]) { L.span[j] = L.span[in[j].peer]; known[j] = 1; changed = true; }
            if (known[j] && in[j].peer >= 0 && !known[in[j].peer]) { L.span[in[j].peer] = L.span[j]; known[in[j].peer] = 1; changed = true; }
        }
    }
    for (uint32_t j = 0; j < n; j++)
        if (!known[j]) fail(GS_ERR_ARG, "input registers: cannot tell how many steps a value is held (no (steps n) below it)");
    for (uint32_t j = 0; j < n; j++) {
        uint64_t count = 1;
        for (uint32_t d : shapes[j]) count = capped_mul(count, d);
        L.count[j] = count;
        const uint64_t len = capped_mul(count, L.span[j]);
        if (L.length && len != L.length) fail(GS_ERR_ARG, "input registers imply different trace lengths");
        L.length = len;
    }
    for (uint32_t j = 0; j < n; j++)            // (a register of no values, or of values held for 0 steps, that came first was skipped above)
        if (capped_mul(L.count[j], L.span[j]) != L.length) fail(GS_ERR_ARG, "input registers imply different trace lengths");
    if (n && (L.length < 2 || (L.length & (L.length - 1)) || L.length >= MAX_LAYOUT))
        fail(GS_ERR_ARG, "the inputs make a trace of %llu steps: a power of 2 is required", (unsigned long long)L.length);
    return L;
}
// iShapes as the job carries them (per register: rank, then the dimensions) -> Shapes
Shapes job_shapes(const gs_prover_air &air) {
    Shapes out(air.ninputs);
    if (air.ninputs && !air.input_shapes) fail(GS_ERR_ARG, "an AIR with input registers needs the shapes of its inputs");
    const uint32_t *q = air.input_shapes;
    for (uint32_t j = 0; j < air.ninputs; j++) {
        const uint32_t rank = *q++;
        if (rank > 255) fail(GS_ERR_ARG, "input register %u: values nested %u deep", j, rank);
        out[j].assign(q, q + rank);
        q += rank;
    }
    return out;
}
// Serializer.serializeProof's last part (lib/Serializer.ts:70-78): count, then per shape its rank and the dimensions as uint32 LE
void write_input_shapes(Bytes &out, const Shapes &shapes) {
    if (shapes.size() > 255) fail(GS_ERR_ARG, "too many input registers");
    out.push_back((uint8_t)shapes.size());
    for (const auto &sh : shapes) {
        out.push_back((uint8_t)sh.size());
        for (uint32_t d : sh) for (int b = 0; b < 4; b++) out.push_back((uint8_t)(d >> (8 * b)));
    }
}
// what prove() does with a job's input registers before any device work: the shapes must lay out exactly the trace the job states
Shapes checked_job_shapes(const gs_prover_job &job) {
    if (!job.air.ninputs) return Shapes();
    if (!job.air.inputs) fail(GS_ERR_ARG, "invalid job: input registers without their declarations");
    Shapes shapes = job_shapes(job.air);
    const InputLayout L = input_layout(job.air, shapes);
    if (L.length != job.steps) fail(GS_ERR_ARG, "the inputs lay out a trace of %llu steps, the job states %llu", (unsigned long long)L.length, (unsigned long long)job.steps);
    return shapes;
}
// the root of unity of the evaluation domain from the job's (root_of_unity_log2: squared down from a root of higher order)
F domain_root(const gs_prover_job &job, uint64_t N) {
    F w = from16(job.root_of_unity);
    if (!job.root_of_unity_log2) return w;
    if (job.root_of_unity_log2 > 63 || N > (1ull << job.root_of_unity_log2)) fail(GS_ERR_ARG, "the field has no root of unity of order %llu", (unsigned long long)N);
    for (uint64_t order = 1ull << job.root_of_unity_log2; order > N; order >>= 1) w = hf_mul(w, w);
    return w;
}

std::vector<uint64_t> unique_in_order(const std::vector<uint64_t> &v) {
    std::vector<uint64_t> out;
    std::map<uint64_t, bool> seen;
    for (uint64_t e : v)
        if (!seen.count(e)) { seen[e] = true; out.push_back(e); }
    return out;
}

}  // namespace

// What the last prove() on this thread did: wall-clock of the phases (host clock at the phase boundaries; no device
// synchronisation is added, so a phase lasts until its last BLOCKING call returned) and the transform work it launched.
static thread_local gs_prover_stats g_stats;
static thread_local bool g_sync_phases = false;      // gs_prover_sync_phases
static thread_local bool g_member_sequence = false;  // gs_prover_member_sequence
// gs_composition_tail[_coset] takes up to four assertions per register, 64 asserted registers, 96 committed vectors
static bool tail_fits(const gs_prover_job &job, uint32_t vectors) {
    if (g_member_sequence || vectors > 96) return false;
    std::vector<std::pair<uint32_t, uint32_t>> per_reg;
    for (uint32_t i = 0; i < job.nassertions; i++) {
        bool found = false;
        for (auto &e : per_reg)
            if (e.first == job.assertions[i].reg) { found = true; if (++e.second > 4) return false; }
        if (!found) per_reg.push_back({job.assertions[i].reg, 1u});
    }
    return per_reg.size() <= 64;
}

// the two transform entry points, counted: rows * n points per call.  The library serves a transform of fewer than 256 points
// or of a polynomial of at most 8 coefficients with a Horner kernel (ntt.hip: ntt_run), which is not an NTT: counted apart.
static int counted_interpolate_roots(gs_ctx *c, const void *ys, uint32_t rows, const uint8_t *omega, uint64_t n, void *out) {
    if (n >= 256) { g_stats.ntt_points += (uint64_t)rows * n; g_stats.ntt_transforms += rows; }
    else g_stats.horner_points += (uint64_t)rows * n;
    return A.gs_interpolate_roots(c, ys, rows, omega, n, out);
}
static int counted_eval_polys_at_roots(gs_ctx *c, const void *polys, uint32_t rows, uint64_t poly_len, const uint8_t *omega, uint64_t n, void *out) {
    if (n >= 256 && poly_len > 8) { g_stats.ntt_points += (uint64_t)rows * n; g_stats.ntt_transforms += rows; }
    else g_stats.horner_points += (uint64_t)rows * n;
    return A.gs_eval_polys_at_roots(c, polys, rows, poly_len, omega, n, out);
}

extern "C" {

static int bind_api(Api &api, void *dl_handle) {
    if (!dl_handle) return GS_ERR_ARG;
#define X(name)                                                 \
    api.name = (decltype(api.name))dlsym(dl_handle, #name);     \
    if (!api.name) return GS_ERR_UNSUPPORTED;
    GS_API_LIST(X)
#undef X
    // the library must compute in the field this build of the driver does its host-side scalars in
    auto esize = (int (*)())dlsym(dl_handle, "gs_element_size");
    auto modulus = (int (*)(uint8_t *))dlsym(dl_handle, "gs_field_modulus");
    if (!esize || !modulus || (uint64_t)esize() != ELEM) return GS_ERR_UNSUPPORTED;
    uint8_t m[GS_PROVER_ELT_MAX] = {0}, want[GS_PROVER_ELT_MAX] = {0};
    if (modulus(m)) return GS_ERR_UNSUPPORTED;
#if defined(GF_RUNTIME_MODULUS)
    // the runtime-modulus build of the driver computes in whatever field the library it is bound to was given (gs_set_modulus) — one per
    // process, like the library's
    { uint32_t pl[GF_LIMBS]; memcpy(pl, m, sizeof pl); if (gf_rt_configure(pl)) return GS_ERR_UNSUPPORTED; }
#endif
#if defined(GS_WIDE_BITS)
    for (int i = 0; i < GF_LIMBS; i++) { const uint32_t w = gf_p_limb(i); memcpy(want + 4 * i, &w, 4); }
#else
    { const hfe pp = hf_p(); memcpy(want, &pp, 16); }
#endif
    return memcmp(m, want, ELEM) ? GS_ERR_UNSUPPORTED : GS_OK;
}
int gs_prover_bind(void *dl_handle) {
    const int rc = bind_api(g_default_api, dl_handle);
    if (rc == GS_OK) g_bound = true;
    return rc;
}
int gs_prover_open(void *dl_handle, gs_prover_binding **out) {
    if (!out) return GS_ERR_ARG;
    Api *api = new Api();
    const int rc = bind_api(*api, dl_handle);
    if (rc) { delete api; return rc; }
    *out = reinterpret_cast<gs_prover_binding *>(api);
    return GS_OK;
}
void gs_prover_close(gs_prover_binding *b) { delete reinterpret_cast<Api *>(b); }
int gs_prover_element_size(void) { return (int)ELEM; }

static void prove_impl(Ctx &x, const gs_prover_job &job, Bytes &out);
static int prove_entry(gs_ctx *ctx, const struct gs_prover_job *job, uint8_t *out, uint64_t cap, uint64_t *len, char *err, uint64_t errcap);
static bool remainder_is_low_degree(const std::vector<F> &remainder, uint64_t E, uint64_t m, F rou, int method);

// Serialized proof into out[0..cap); *len receives the size (also when cap is too small: GS_ERR_ARG then).
int gs_prover_prove_on(const gs_prover_binding *b, gs_ctx *ctx, const struct gs_prover_job *job, uint8_t *out, uint64_t cap, uint64_t *len, char *err,
                       uint64_t errcap) {
    if (!b) return GS_ERR_ARG;
    UseApi use(reinterpret_cast<const Api *>(b));
    return prove_entry(ctx, job, out, cap, len, err, errcap);
}
int gs_prover_prove(gs_ctx *ctx, const struct gs_prover_job *job, uint8_t *out, uint64_t cap, uint64_t *len, char *err, uint64_t errcap) {
    if (!g_bound) return GS_ERR_UNSUPPORTED;
    UseApi use(&g_default_api);
    return prove_entry(ctx, job, out, cap, len, err, errcap);
}
static int prove_entry(gs_ctx *ctx, const struct gs_prover_job *job, uint8_t *out, uint64_t cap, uint64_t *len, char *err, uint64_t errcap) {
    if (!ctx || !job || !len) return GS_ERR_ARG;
    try {
        Ctx x{ctx};
        Bytes proof;
        prove_impl(x, *job, proof);
        *len = proof.size();
        if (proof.size() > cap || !out) return GS_ERR_ARG;
        memcpy(out, proof.data(), proof.size());
        return GS_OK;
    } catch (const Fail &f) {
        if (err && errcap) snprintf(err, (size_t)errcap, "%s", f.msg.c_str());
        return f.code ? f.code : GS_ERR_ARG;
    } catch (const std::exception &e) {
        if (err && errcap) snprintf(err, (size_t)errcap, "%s", e.what());
        return GS_ERR_OOM;
    }
}

static int remainder_check_entry(const uint8_t *values, uint64_t len, uint32_t extension_factor, uint64_t max_degree_plus1, const uint8_t *root_of_unity, int method);
int gs_prover_remainder_check(const uint8_t *values, uint64_t len, uint32_t extension_factor, uint64_t max_degree_plus1, const uint8_t *root_of_unity,
                              int method) {
    if (!g_bound) return GS_ERR_UNSUPPORTED;
    UseApi use(&g_default_api);
    return remainder_check_entry(values, len, extension_factor, max_degree_plus1, root_of_unity, method);
}
int gs_prover_remainder_check_on(const gs_prover_binding *b, const uint8_t *values, uint64_t len, uint32_t extension_factor, uint64_t max_degree_plus1,
                                 const uint8_t *root_of_unity, int method) {
    if (!b) return GS_ERR_ARG;
    UseApi use(reinterpret_cast<const Api *>(b));
    return remainder_check_entry(values, len, extension_factor, max_degree_plus1, root_of_unity, method);
}
static int remainder_check_entry(const uint8_t *values, uint64_t len, uint32_t extension_factor, uint64_t max_degree_plus1, const uint8_t *root_of_unity, int method) {
    if (!values || !root_of_unity || !len || (method != 0 && method != 1)) return GS_ERR_ARG;
    try {
        std::vector<F> v(len);
        for (uint64_t i = 0; i < len; i++) v[i] = from16(values + ELEM * i);
        return remainder_is_low_degree(v, extension_factor, max_degree_plus1, from16(root_of_unity), method) ? 1 : 0;
    } catch (const Fail &f) {
        return f.code ? f.code : GS_ERR_ARG;
    } catch (const std::exception &) {
        return GS_ERR_OOM;
    }
}

void gs_prover_sync_phases(int on) { g_sync_phases = on != 0; }
void gs_prover_member_sequence(int on) { g_member_sequence = on != 0; }

int gs_prover_abi_version(void) { return GS_PROVER_ABI_VERSION; }

int gs_prover_input_layout(const struct gs_input_register *inputs, uint32_t ninputs, const uint32_t *shapes, uint64_t *length, char *err, uint64_t errcap) {
    if ((ninputs && (!inputs || !shapes)) || !length) return GS_ERR_ARG;
    try {
        gs_prover_air air;
        memset(&air, 0, sizeof air);
        air.inputs = inputs; air.ninputs = ninputs; air.input_shapes = shapes;
        *length = input_layout(air, job_shapes(air)).length;
        return GS_OK;
    } catch (const Fail &f) {
        if (err && errcap) snprintf(err, (size_t)errcap, "%s", f.msg.c_str());
        return f.code ? f.code : GS_ERR_ARG;
    } catch (const std::exception &e) {
        if (err && errcap) snprintf(err, (size_t)errcap, "%s", e.what());
        return GS_ERR_OOM;
    }
}

int gs_prover_last_stats(struct gs_prover_stats *out) {
    if (!out) return GS_ERR_ARG;
    *out = g_stats;
    return GS_OK;
}

}  // extern "C"

namespace {

struct Layer {           // one FRI layer: the tree / rows it queries and the child it produced
    Tree *pTree;         // tree over the rows of transposeVector(column, 4)
    const void *column;  // the layer's values in natural order (4 * rows of them); rows are read strided, never transposed
    uint64_t rows;
    Tree cTree;          // tree over the next layer's rows
    Buf next;            // the folded column: `rows` values
    uint64_t column_length;
};

}  // namespace

// GSTARK_PROVER_TIMING=1: host wall-clock at the phase boundaries on stderr (no device synchronisation is added, so a phase
// shows the time until its last BLOCKING call returned)
struct PhaseClock {
    bool on = getenv("GSTARK_PROVER_TIMING") != nullptr;      // also echo the marks on stderr
    const bool sync = g_sync_phases;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0, last_readme = t0;
    PhaseClock() { memset(&g_stats, 0, sizeof g_stats); }
    // one of the reference's log points (lib/Stark.ts:92-152; README.md:62-73).  Measuring mode only (gs_prover_sync_phases): the
    // device is drained here, so the entry is this phase's own wall-clock
    void readme(Ctx &x, const char *fmt, ...) {
        if (!sync) return;
        x.check(A.gs_sync(x.c), "gs_sync");
        auto now = std::chrono::steady_clock::now();
        if (g_stats.nreadme < GS_PROVER_MAX_PHASES) {
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(g_stats.readme_label[g_stats.nreadme], sizeof g_stats.readme_label[0], fmt, ap);
            va_end(ap);
            g_stats.readme_ms[g_stats.nreadme++] = std::chrono::duration<double, std::milli>(now - last_readme).count();
        }
        last_readme = now;
    }
    void mark(const char *what) {
        auto now = std::chrono::steady_clock::now();
        const double total = std::chrono::duration<double, std::milli>(now - t0).count();
        const double delta = std::chrono::duration<double, std::milli>(now - last).count();
        if (g_stats.nphases < GS_PROVER_MAX_PHASES) {
            snprintf(g_stats.phase_label[g_stats.nphases], sizeof g_stats.phase_label[0], "%s", what);
            g_stats.phase_ms[g_stats.nphases++] = delta;
        }
        g_stats.total_ms = total;
        if (on) fprintf(stderr, "[prover] %-44s %8.3f ms  (+%.3f)\n", what, total, delta);
        last = now;
    }
    // the interval since the last mark as TWO entries: `first_ms` of it under `a` (time the host spent blocked, measured by the caller),
    // the rest under `b`
    void mark_split(const char *a, double first_ms, const char *b) {
        auto now = std::chrono::steady_clock::now();
        const double delta = std::chrono::duration<double, std::milli>(now - last).count();
        const double total = std::chrono::duration<double, std::milli>(now - t0).count();
        const char *labels[2] = {a, b};
        const double parts[2] = {std::min(first_ms, delta), delta - std::min(first_ms, delta)};
        for (int k = 0; k < 2; k++)
            if (g_stats.nphases < GS_PROVER_MAX_PHASES) {
                snprintf(g_stats.phase_label[g_stats.nphases], sizeof g_stats.phase_label[0], "%s", labels[k]);
                g_stats.phase_ms[g_stats.nphases++] = parts[k];
            }
        g_stats.total_ms = total;
        if (on) fprintf(stderr, "[prover] %-44s %8.3f ms  (+%.3f)\n[prover] %-44s %8.3f ms  (+%.3f)\n", a, total - parts[1], parts[0], b, total, parts[1]);
        last = now;
    }
};

// LowDegreeProver.verifyRemainder (:223-252): the remainder — `len` values on the powers of `rou` (order len) — must agree, at
// every position i that is not a multiple of E, with the polynomial of degree < m through the first m of those positions.
//   method 0: as the reference does it (interpolate the first m, evaluate at the rest: ~3 m^2 products);
//   method 1: the same predicate on coefficients.  The excluded points rou^(E j) are the B-th roots of unity, B = len / E, so the
//     checked points are the roots of V(x) = (x^len - 1)/(x^B - 1) = 1 + x^B + x^2B + ...  Let g be the interpolant of ALL len values
//     (an inverse transform of size len).  A polynomial f of degree < m agrees with g on the roots of V  <=>  g - f = V h with
//     deg h < B  <=>  the coefficients g_k, k >= m, depend on k mod B only (V h is h's coefficients repeated E times).  That f is
//     the reference's interpolant (degree < m through m of the points), so both methods accept exactly the same remainders;
//     len log len products instead of 3 m^2.  Used when E divides len (always, for the domains of this prover).
// returns true when the remainder passes
static bool remainder_is_low_degree(const std::vector<F> &remainder, uint64_t E, uint64_t m, F rou, int method) {
    const uint64_t len = remainder.size();
    std::vector<uint64_t> positions;
    for (uint64_t i = 0; i < len; i++) if (!E || i % E) positions.push_back(i);
    if (m > positions.size()) fail(GS_ERR_ARG, "Remainder degree is greater than number of remainder values");
    if (!m || m == positions.size()) return true;
    if (method == 1 && E && len >= E && len % E == 0 && !(len & (len - 1))) {
        const uint64_t B = len / E;
        std::vector<F> g(len), w(len / 2 ? len / 2 : 1);
        int lg = 0;
        while ((1ull << lg) < len) lg++;
        for (uint64_t i = 0; i < len; i++) {               // bit-reversed input, decimation in time
            uint64_t r = 0;
            for (int b = 0; b < lg; b++) r |= ((i >> b) & 1) << (lg - 1 - b);
            g[r] = remainder[i];
        }
        const F inv = hf_pow(rou, (hfe)(len - 1));          // rou^-1
        F cur = 1;
        for (uint64_t i = 0; i < len / 2; i++) { w[i] = cur; cur = hf_mul(cur, inv); }
        for (uint64_t half = 1; half < len; half <<= 1)
            for (uint64_t base = 0; base < len; base += 2 * half)
                for (uint64_t j = 0; j < half; j++) {
                    const F t = hf_mul(g[base + half + j], w[j * (len / (2 * half))]);
                    const F u = g[base + j];
                    g[base + j] = hf_add(u, t);
                    g[base + half + j] = hf_sub(u, t);
                }
        // (unscaled: the predicate compares coefficients with each other)
        for (uint64_t k = m; k + B < len; k++)
            if (g[k] != g[k + B]) return false;
        return true;
    }
    std::vector<F> domain(len);
    F cur = 1;
    for (uint64_t i = 0; i < len; i++) { domain[i] = cur; cur = hf_mul(cur, rou); }
    Bytes xs(m * ELEM), ys(m * ELEM), poly(m * ELEM);
    for (uint64_t i = 0; i < m; i++) { le16(domain[positions[i]], xs.data() + ELEM * i); le16(remainder[positions[i]], ys.data() + ELEM * i); }
    if (A.gs_small_interpolate(xs.data(), ys.data(), (uint32_t)m, poly.data())) fail(GS_ERR_ARG, "gs_small_interpolate failed");
    const uint32_t rest = (uint32_t)(positions.size() - m);
    Bytes rx(rest * ELEM), rv(rest * ELEM);
    for (uint32_t i = 0; i < rest; i++) le16(domain[positions[m + i]], rx.data() + ELEM * i);
    if (A.gs_small_eval_poly(poly.data(), (uint32_t)m, rx.data(), rest, rv.data())) fail(GS_ERR_ARG, "gs_small_eval_poly failed");
    for (uint32_t i = 0; i < rest; i++)
        if (from16(rv.data() + ELEM * i) != remainder[positions[m + i]]) return false;
    return true;
}

static void prove_impl(Ctx &x, const gs_prover_job &job, Bytes &out) {
    PhaseClock clock;
    const gs_prover_air &air = job.air;
    const uint64_t T = job.steps, E = job.extension_factor, N = T * E;
    const uint32_t R = air.registers;
    const int alg = job.hash_alg;
    if (!T || (T & (T - 1)) || !E || (E & (E - 1)) || !R || !air.nconstraints || !job.nassertions) fail(GS_ERR_ARG, "invalid job");
    const Shapes input_shapes = checked_job_shapes(job);          // iShapes of the proof (lib/Stark.ts:161); empty without input registers
    uint32_t max_degree = 1;
    for (uint32_t i = 0; i < air.nconstraints; i++) max_degree = std::max(max_degree, air.degrees[i]);
    uint64_t cf = 1;
    while (cf < max_degree) cf <<= 1;                              // compositionFactor = 2^ceil(log2(max degree))
    const uint64_t Nc = T * cf;
    if (E < 2 * cf) fail(GS_ERR_ARG, "extension factor must be at least 2x the composition factor");
    const F omega = domain_root(job, N);
    const F comp_rou = hf_pow(omega, (hfe)(N / Nc)), exec_rou = hf_pow(omega, (hfe)E);
    uint8_t s16[ELEM], s16b[ELEM];

    // 1 ----- evaluation context (lib/Stark.ts:92-94): the kernels below derive domain points from omega; the evaluation domain is
    // materialised only by the general Z(x) sequence

    // work of CompositionPolynomial.evaluateAll that does not depend on the trace goes first: the device computes it while
    // the host core below runs the trace recurrence (same values, issue order only)
    const uint64_t combination_degree = cf * T;                                        // CompositionPolynomial.ts:196-204
    const uint64_t composition_degree = std::max(combination_degree - T, T);
    // MiMC with up to four assertions: the whole of CompositionPolynomial.evaluateAll is one kernel over the evaluation domain
    // (gs_mimc_composition, below); otherwise the member-by-member sequence, whose trace-independent part is issued here
    uint32_t assertions_on_r0 = 0;
    for (uint32_t i = 0; i < job.nassertions; i++) assertions_on_r0 += job.assertions[i].reg == 0;
    const bool fused = air.kind == 0 && E <= 32 && air.nconstraints == 1 && assertions_on_r0 == job.nassertions && job.nassertions <= 4;
    // the generic sequence's tail in one pass (gs_composition_tail) when the assertions fit its per-register limits: neither 1/Z(x) nor
    // the power series of the degree adjustment is materialised then
    const bool tail = !fused && tail_fits(job, R + air.nsecret);
    const bool tail_makes_z = tail && E <= 32;
    Buf zInverses;
    if (!fused && !tail_makes_z) {
        zInverses = Buf(x, N * ELEM);
        // ZeroPolynomial.ts:36-44 and the division of CompositionPolynomial.ts:117 in one kernel: x^T - 1 takes only E distinct values
        le16(omega, s16);
        le16(hf_pow(omega, (hfe)((T - 1) * E)), s16b);                                               // :21-23
        if (E <= 32) {
            x.check(A.gs_zero_poly_inverses(x.c, s16, N, T, s16b, zInverses.p), "gs_zero_poly_inverses");
        } else {
            Buf evalDomain(x, N * ELEM), xToTheSteps(x, N * ELEM), num(x, N * ELEM), den(x, N * ELEM);
            le16(omega, s16);
            x.check(A.gs_power_series(x.c, s16, N, evalDomain.p), "gs_power_series(evaluation domain)");
            x.check(A.gs_pluck(x.c, evalDomain.p, N, T, N, xToTheSteps.p), "gs_pluck");                // ZeroPolynomial.ts:40
            le16(1, s16);
            x.check(A.gs_vec_sub_scalar(x.c, xToTheSteps.p, s16, N, num.p), "gs_vec_sub_scalar");
            x.check(A.gs_vec_sub_scalar(x.c, evalDomain.p, s16b, N, den.p), "gs_vec_sub_scalar");
            x.check(A.gs_vec_div(x.c, den.p, num.p, N, zInverses.p), "gs_vec_div(1/Z)");               // CompositionPolynomial.ts:117
        }
    }
    Buf psbPowers;                                                 // x^(compositionDegree - T) over the evaluation domain
    const uint64_t b_inc = composition_degree - T;
    const bool lc_folds = fused && R + air.nsecret == 1;           // LinearCombination folded into the composition kernel too
    if (b_inc > 0 && !lc_folds && !tail) {                         // also what LinearCombination.ts:44-52 multiplies by
        psbPowers = Buf(x, N * ELEM);
        le16(hf_pow(omega, (hfe)b_inc), s16);
        x.check(A.gs_power_series(x.c, s16, N, psbPowers.p), "gs_power_series(psb)");
    }

    // MiMC: the cyclic register K over the evaluation domain — its 64 coefficients from the composition-domain table, then its values
    // at the (k_len * N/Nc)-th roots of unity (the constraint is evaluated on all N points from the extension of P)
    Buf kN;
    uint64_t klen_n = 0;
    if (air.kind == 0) {
        klen_n = air.k_len * (N / Nc);
        Buf kPoly(x, air.k_len * ELEM);
        kN = Buf(x, klen_n * ELEM);
        le16(hf_pow(omega, (hfe)(N / air.k_len)), s16);
        x.check(counted_interpolate_roots(x.c, air.k_table, 1, s16, air.k_len, kPoly.p), "gs_interpolate_roots(K)");
        le16(hf_pow(omega, (hfe)(N / klen_n)), s16);
        x.check(counted_eval_polys_at_roots(x.c, kPoly.p, 1, air.k_len, s16, klen_n, kN.p), "gs_eval_polys_at_roots(K)");
    }

    clock.mark("context + trace-independent work issued");
    clock.readme(x, "Set up evaluation context");
    // 2 ----- execution trace (:97) and the assertions it must satisfy (:356-375)
    Buf trace(x, (uint64_t)R * T * ELEM);
    if (air.kind == 0)
        x.check(A.gs_mimc_trace(x.c, air.seed, air.round_constants, air.nrc, T, trace.p), "gs_mimc_trace");
    else if (air.segments)
        x.check(A.gs_air_trace_segments(x.c, air.t_code, air.t_ninstr, air.i_code, air.i_ninstr, air.consts, air.nconsts, air.vm_regs, R,
                                        air.static_values, air.static_periods, air.nstatic, air.first_rows, air.segments, air.segment_len, trace.p),
                "gs_air_trace_segments");
    else
        x.check(A.gs_air_trace(x.c, air.t_code, air.t_ninstr, air.consts, air.nconsts, air.vm_regs, R, air.static_values, air.static_periods,
                               air.nstatic, air.first_rows, T, trace.p), "gs_air_trace");
    // the asserted cells (:356-375) are compared when the evaluation root comes back: one round trip for both
    std::vector<uint64_t> asserted_at;
    for (uint32_t i = 0; i < job.nassertions; i++) {
        const gs_assertion &a = job.assertions[i];
        if (a.reg >= R) fail(GS_ERR_ARG, "Invalid assertion: register %u is outside of register bank", a.reg);
        if (a.step >= T) fail(GS_ERR_ARG, "Invalid assertion: step %llu is outside of execution trace", (unsigned long long)a.step);
        asserted_at.push_back((uint64_t)a.reg * T + a.step);
    }

    clock.mark("execution trace (host recurrence)");
    clock.readme(x, "Generated execution trace");
    // 3 ----- P(x) and its low-degree extension (:106-109)
    Buf pPolys(x, (uint64_t)R * T * ELEM), pEval(x, (uint64_t)R * N * ELEM);
    le16(exec_rou, s16);
    x.check(counted_interpolate_roots(x.c, trace.p, R, s16, T, pPolys.p), "gs_interpolate_roots(trace)");
    clock.readme(x, "Computed execution trace polynomials P(x)");
    le16(omega, s16);
    x.check(counted_eval_polys_at_roots(x.c, pPolys.p, R, T, s16, N, pEval.p), "gs_eval_polys_at_roots(P)");
    clock.readme(x, "Low-degree extended P(x) polynomials over evaluation domain");
    std::vector<const void *> pRows(R);
    for (uint32_t r = 0; r < R; r++) pRows[r] = pEval.at((uint64_t)r * N * ELEM);
    std::vector<const void *> eVectors(pRows);                    // [P_0.., S_0..] (lib/Stark.ts:113-114)
    for (uint32_t i = 0; i < air.nsecret; i++) eVectors.push_back(air.secret_traces[i]);
    const uint32_t V = (uint32_t)eVectors.size();

    // 4 ----- evaluation Merkle tree (:113-118)
    Tree eTree = commit_rows(x, alg, eVectors.data(), V, N, false);

    clock.mark("P(x), extension, evaluation tree issued");
    clock.readme(x, "Serialized evaluations of P(x) and S(x) polynomials + Built evaluation merkle tree (one fused call)");
    // 5 ----- composition polynomial (CompositionPolynomial.ts:29-146)
    // boundary constraints per asserted register, in order of first appearance (BoundaryConstraints.ts:15-45)
    struct RegData { uint32_t reg; std::vector<F> xs, ys; std::vector<uint64_t> at; };   // at: positions of the xs in the evaluation domain
    std::vector<RegData> rdata;
    for (uint32_t i = 0; i < job.nassertions; i++) {
        const gs_assertion &a = job.assertions[i];
        RegData *d = nullptr;
        for (auto &e : rdata) if (e.reg == a.reg) d = &e;
        if (!d) { rdata.push_back(RegData{a.reg, {}, {}, {}}); d = &rdata.back(); }
        d->at.push_back(a.step * E);
        d->xs.push_back(hf_pow(omega, (hfe)(a.step * E)));
        d->ys.push_back(from16(a.value));
    }
    const uint32_t bcount = (uint32_t)rdata.size();
    // constraint groups by degree, in order of first appearance (:206-225)
    std::vector<std::pair<uint64_t, std::vector<uint32_t>>> groups;
    for (uint32_t i = 0; i < air.nconstraints; i++) {
        uint64_t d = (uint64_t)air.degrees[i] * T;
        bool found = false;
        for (auto &g : groups) if (g.first == d) { g.second.push_back(i); found = true; }
        if (!found) groups.push_back({d, {i}});
    }
    uint32_t dcount = air.nconstraints;
    for (auto &g : groups) if (g.first < combination_degree) dcount += (uint32_t)g.second.size();
    uint32_t bcoef = bcount * (composition_degree > T ? 2 : 1);
    // The coefficients come from the evaluation root (:121): it is read where the first of them is needed, with the asserted cells —
    // the work that needs neither (constraint evaluation, degree adjustment) is queued first and covers the round trip
    std::vector<F> coefficients;
    auto read_evaluation_root = [&] {
        Bytes got(asserted_at.size() * ELEM);
        const uint64_t one = 1;                        // record 1 of a node array (32-byte records) is the root
        DeferWindow win(x);
        x.check(A.gs_gather(x.c, trace.p, ELEM, asserted_at.data(), asserted_at.size(), got.data()), "gs_gather(asserted cells)");
        x.check(A.gs_gather(x.c, eTree.nodes.p, DIGEST, &one, 1, eTree.root.data()), "gs_gather(root)");
        win.end();
        for (uint32_t i = 0; i < job.nassertions; i++)
            if (memcmp(got.data() + i * ELEM, job.assertions[i].value, ELEM))
                fail(GS_ERR_ARG, "Assertion at step %llu, register %u conflicts with execution trace", (unsigned long long)job.assertions[i].step,
                     job.assertions[i].reg);
        trace.release();
        coefficients = prng_many(eTree.root, dcount + bcoef);
    };
    auto coeff_bytes = [&](size_t from, size_t count) {
        Bytes b(count * ELEM);
        for (size_t i = 0; i < count; i++) le16(coefficients[from + i], b.data() + ELEM * i);
        return b;
    };

    Buf cEval(x, N * ELEM);
    bool lc_fused = false;                     // cEval already holds L (LinearCombination folded into the composition kernel)
    if (fused) {
        // K over the evaluation domain (issued before the trace), the interpolant through the assertions, then one kernel for :71-146
        // (what needs no coefficient first: the device is idle while the root travels)
        const RegData &d = rdata[0];
        const uint32_t m = (uint32_t)d.xs.size();
        Bytes xs(m * ELEM), ys(m * ELEM), ipoly(m * ELEM);
        for (uint32_t i = 0; i < m; i++) { le16(d.xs[i], xs.data() + ELEM * i); le16(d.ys[i], ys.data() + ELEM * i); }
        if (A.gs_small_interpolate(xs.data(), ys.data(), m, ipoly.data())) fail(GS_ERR_ARG, "gs_small_interpolate failed");    // BoundaryConstraints.ts:42
        read_evaluation_root();
        const bool q_adjusted = groups[0].first < combination_degree;
        Bytes co(4 * ELEM, 0);
        le16(coefficients[0], co.data());
        if (q_adjusted) le16(coefficients[1], co.data() + ELEM);
        le16(coefficients[dcount], co.data() + 2 * ELEM);
        if (b_inc > 0) le16(coefficients[dcount + 1], co.data() + 3 * ELEM);
        // ... and, with one committed vector, LinearCombination.computeMany (:36-64) on top: the same prng stream continues
        const bool with_lc = lc_folds;
        Bytes lc(2 * ELEM, 0);
        if (with_lc) {
            std::vector<F> all = prng_many(eTree.root, dcount + bcoef + (b_inc > 0 ? 2 : 1));
            le16(all[dcount + bcoef], lc.data());
            if (b_inc > 0) le16(all[dcount + bcoef + 1], lc.data() + ELEM);
        }
        lc_fused = with_lc;
        le16(omega, s16);
        x.check(A.gs_mimc_composition(x.c, pRows[0], N, T, s16, kN.p, klen_n, co.data(), q_adjusted ? combination_degree - groups[0].first : 0, b_inc,
                                      ipoly.data(), d.at.data(), m, with_lc ? lc.data() : nullptr, cEval.p), "gs_mimc_composition");
    } else {
        // 5.1-5.3: the combined, degree-adjusted Q has degree < Nc, so its extension to the evaluation domain (:109-110) is what
        // the constraint expression gives there.  MiMC (one cheap constraint): evaluate it on all N points from the extension of P
        // already at hand — no interpolation + extension; AIR programs: on the composition domain as the reference does.
        const bool direct = air.kind == 0;
        const uint64_t Nq = direct ? N : Nc;
        const F q_rou = direct ? omega : comp_rou;
        Buf q(x, (uint64_t)air.nconstraints * Nq * ELEM);
        if (direct) {
            x.check(A.gs_mimc_constraints(x.c, pRows[0], N, N / T, kN.p, klen_n, q.p), "gs_mimc_constraints");
        } else {
            // P over the composition domain is every (N/Nc)-th element of the extension just computed (:76)
            // (the R rows are one contiguous R x N matrix and N / Nc divides N: ONE strided pick over the whole matrix gives R x Nc)
            // — read in place, with that stride: no plucked copy (R x Nc elements written and read again)
            x.check(A.gs_air_constraints_strided(x.c, air.e_code, air.e_ninstr, air.consts, air.nconsts, air.vm_regs, R, air.nconstraints, pEval.p, N, N / Nc,
                                                 Nc, Nc / T, air.static_tables, air.static_lens, air.nstatic, q.p), "gs_air_constraints_strided");
        }
        // 5.2 degree adjustment (:83-101) and 5.3 merge + extension (:103-111): the adjusted vectors q_i o powers are not
        // materialised — gs_combine_adjusted merges sum k_i q_i + powers o sum k'_i q_i in one pass (one further pass per additional
        // group of constraints with a degree of its own: different powers)
        std::vector<const void *> qa;
        for (uint32_t i = 0; i < air.nconstraints; i++) qa.push_back(q.at((uint64_t)i * Nq * ELEM));
        read_evaluation_root();
        Buf qe(x, N * ELEM), qc;
        if (!direct) qc = Buf(x, Nc * ELEM);
        void *merged = direct ? qe.p : qc.p;
        {
            Bytes plain = coeff_bytes(0, air.nconstraints);
            uint32_t next = air.nconstraints;                       // coefficients of the adjusted terms follow, group by group
            bool first = true;
            for (auto &g : groups) {
                if (g.first == combination_degree) continue;
                Buf powers(x, Nq * ELEM);
                le16(hf_pow(q_rou, (hfe)(combination_degree - g.first)), s16);
                x.check(A.gs_power_series(x.c, s16, Nq, powers.p), "gs_power_series(q powers)");
                if (first) {                                         // every constraint's plain term + this group's adjusted terms
                    Bytes adj(air.nconstraints * ELEM, 0);
                    for (uint32_t i : g.second) le16(coefficients[next++], adj.data() + ELEM * i);
                    x.check(A.gs_combine_adjusted(x.c, qa.data(), plain.data(), adj.data(), air.nconstraints, powers.p, nullptr, Nq, merged), "gs_combine_adjusted(Q)");
                } else {
                    std::vector<const void *> members;
                    Bytes adj(g.second.size() * ELEM);
                    for (size_t k = 0; k < g.second.size(); k++) { members.push_back(qa[g.second[k]]); le16(coefficients[next++], adj.data() + ELEM * k); }
                    x.check(A.gs_combine_adjusted(x.c, members.data(), nullptr, adj.data(), (uint32_t)members.size(), powers.p, merged, Nq, merged), "gs_combine_adjusted(Q)");
                }
                first = false;
            }
            if (first) x.check(A.gs_combine_many(x.c, qa.data(), plain.data(), air.nconstraints, Nq, merged), "gs_combine_many(Q)");
        }
        if (!direct) {
            Buf qcPoly(x, Nc * ELEM);
            le16(comp_rou, s16);
            x.check(counted_interpolate_roots(x.c, qc.p, 1, s16, Nc, qcPoly.p), "gs_interpolate_roots(Q)");
            le16(omega, s16);
            x.check(counted_eval_polys_at_roots(x.c, qcPoly.p, 1, Nc, s16, N, qe.p), "gs_eval_polys_at_roots(Q)");
        }
        // 5.4-5.7 and 6 in ONE pass (gs_composition_tail): D = Q / Z, the boundary quotients from the registers' extensions, their
        // degree-adjusted merge, and LinearCombination.computeMany on top — when the assertions fit its per-register limits
        size_t tail_roots = 0;
        for (auto &d : rdata) tail_roots = std::max(tail_roots, d.at.size());
        if (tail) {
            const uint32_t il = (uint32_t)tail_roots;                   // an interpolant through m assertions has m coefficients
            Bytes ip((size_t)bcount * il * ELEM, 0);
            std::vector<uint64_t> at((size_t)bcount * il, 0);
            std::vector<uint32_t> per_row(bcount);
            for (uint32_t r = 0; r < bcount; r++) {
                const RegData &d = rdata[r];
                const uint32_t m = (uint32_t)d.xs.size();
                Bytes xs(m * ELEM), ys(m * ELEM);
                for (uint32_t i = 0; i < m; i++) { le16(d.xs[i], xs.data() + ELEM * i); le16(d.ys[i], ys.data() + ELEM * i); }
                if (A.gs_small_interpolate(xs.data(), ys.data(), m, ip.data() + (size_t)r * il * ELEM)) fail(GS_ERR_ARG, "gs_small_interpolate failed");   // BoundaryConstraints.ts:42
                per_row[r] = m;
                for (uint32_t k = 0; k < m; k++) at[(size_t)r * il + k] = d.at[k];
            }
            std::vector<const void *> pv;
            for (auto &d : rdata) pv.push_back(pRows[d.reg]);
            Bytes bco = coeff_bytes(dcount, bcoef);
            const uint32_t offset = dcount + bcoef, cnt = b_inc > 0 ? 2 * V : V;
            std::vector<F> co = prng_many(eTree.root, offset + cnt);      // LinearCombination.ts:36-64: the same stream continues
            Bytes cb(cnt * ELEM);
            for (uint32_t i = 0; i < cnt; i++) le16(co[offset + i], cb.data() + ELEM * i);
            le16(omega, s16);
            le16(hf_pow(omega, (hfe)((T - 1) * E)), s16b);                                           // ZeroPolynomial.ts:21-23: the last step's point
            x.check(A.gs_composition_tail(x.c, N, s16, qe.p, tail_makes_z ? nullptr : zInverses.p, T, s16b, pv.data(), bcount, ip.data(), il, at.data(),
                                          per_row.data(), il, bco.data(), b_inc > 0 ? bco.data() + ELEM * bcount : nullptr, eVectors.data(), V, cb.data(),
                                          b_inc > 0 ? cb.data() + ELEM * V : nullptr, nullptr, b_inc, nullptr, cEval.p), "gs_composition_tail");
            lc_fused = true;
        } else {
        // 5.4 D(x) = Q(x) / Z(x) (:113-121)
        Buf dEval(x, N * ELEM);
        x.check(A.gs_vec_mul(x.c, qe.p, zInverses.p, N, dEval.p), "gs_vec_mul(D)");
        // 5.5 boundary constraints (BoundaryConstraints.ts:71-95)
        size_t ilen = 0, zlen = 0;
        std::vector<std::vector<F>> ipolys, zpolys;
        for (auto &d : rdata) {
            const uint32_t m = (uint32_t)d.xs.size();
            Bytes xs(m * ELEM), ys(m * ELEM), co(m * ELEM);
            for (uint32_t i = 0; i < m; i++) { le16(d.xs[i], xs.data() + ELEM * i); le16(d.ys[i], ys.data() + ELEM * i); }
            if (A.gs_small_interpolate(xs.data(), ys.data(), m, co.data())) fail(GS_ERR_ARG, "gs_small_interpolate failed");
            std::vector<F> ip(m), zp{1};
            for (uint32_t i = 0; i < m; i++) ip[i] = from16(co.data() + ELEM * i);
            for (uint32_t i = 0; i < m; i++) {          // zPoly *= (x - xs[i]), BoundaryConstraints.ts:24-30
                std::vector<F> nz(zp.size() + 1, 0);
                const F nx = hf_sub(0, d.xs[i]);
                for (size_t k = 0; k < zp.size(); k++) {
                    nz[k] = hf_add(nz[k], hf_mul(zp[k], nx));
                    nz[k + 1] = hf_add(nz[k + 1], zp[k]);
                }
                zp.swap(nz);
            }
            ilen = std::max(ilen, ip.size());
            zlen = std::max(zlen, zp.size());
            ipolys.push_back(ip);
            zpolys.push_back(zp);
        }
        auto upload_rows = [&](const std::vector<std::vector<F>> &rows, size_t len) {
            Bytes host(rows.size() * len * ELEM, 0);                    // shorter rows are zero-extended (newMatrixFromVectors)
            for (size_t r = 0; r < rows.size(); r++)
                for (size_t k = 0; k < rows[r].size(); k++) le16(rows[r][k], host.data() + (r * len + k) * ELEM);
            Buf b(x, host.size());
            x.check(A.gs_upload(x.c, b.p, host.data(), host.size()), "gs_upload(boundary polynomials)");
            return b;
        };
        Buf iPolys = upload_rows(ipolys, ilen);
        Buf iValues(x, (uint64_t)bcount * N * ELEM), pi(x, (uint64_t)bcount * N * ELEM), bEval(x, (uint64_t)bcount * N * ELEM);
        le16(omega, s16);
        x.check(counted_eval_polys_at_roots(x.c, iPolys.p, bcount, ilen, s16, N, iValues.p), "gs_eval_polys_at_roots(I)");
        std::vector<const void *> pv;
        for (auto &d : rdata) pv.push_back(pRows[d.reg]);
        x.check(A.gs_sub_matrix_from_vectors(x.c, pv.data(), iValues.p, bcount, N, pi.p), "gs_sub_matrix_from_vectors");
        size_t max_roots = 0;
        for (auto &d : rdata) max_roots = std::max(max_roots, d.at.size());
        if (max_roots <= 4) {
            // the divisors' roots are domain points: look-ups in the domain's table 1/(omega^j - 1) instead of evaluating Z_r(x)
            // and inverting it (same values: BoundaryConstraints.ts:88,92)
            std::vector<uint64_t> at(bcount * max_roots, 0);
            std::vector<uint32_t> per_row(bcount);
            for (uint32_t r = 0; r < bcount; r++) {
                per_row[r] = (uint32_t)rdata[r].at.size();
                for (size_t k = 0; k < rdata[r].at.size(); k++) at[r * max_roots + k] = rdata[r].at[k];
            }
            x.check(A.gs_div_by_domain_roots(x.c, pi.p, bcount, N, s16, at.data(), per_row.data(), (uint32_t)max_roots, bEval.p), "gs_div_by_domain_roots");
        } else {
            Buf zPolys = upload_rows(zpolys, zlen), zValues(x, (uint64_t)bcount * N * ELEM);
            x.check(counted_eval_polys_at_roots(x.c, zPolys.p, bcount, zlen, s16, N, zValues.p), "gs_eval_polys_at_roots(Zb)");
            x.check(A.gs_vec_div(x.c, pi.p, zValues.p, (uint64_t)bcount * N, bEval.p), "gs_vec_div(B)");
        }
        iValues.release(); pi.release();
        // 5.6 degree adjustment of B (:124-138) and 5.7 merge (:140-146), with D added in the same pass
        std::vector<const void *> ba;
        for (uint32_t i = 0; i < bcount; i++) ba.push_back(bEval.at((uint64_t)i * N * ELEM));
        Bytes bco = coeff_bytes(dcount, bcoef);
        x.check(A.gs_combine_adjusted(x.c, ba.data(), bco.data(), b_inc > 0 ? bco.data() + ELEM * bcount : nullptr, bcount, b_inc > 0 ? psbPowers.p : nullptr,
                                      dEval.p, N, cEval.p), "gs_combine_adjusted(B + D)");
        }
    }
    if (!fused) zInverses.release();

    clock.readme(x, lc_fused ? "Computed composition polynomial C(x) + Combined P(x) and S(x) evaluations with C(x) evaluations (one kernel)"
                            : "Computed composition polynomial C(x)");
    // 6 ----- random linear combination (LinearCombination.ts:36-64)
    Buf lEval;
    if (lc_fused) {
        lEval = std::move(cEval);
    } else {
        lEval = Buf(x, N * ELEM);
        // psIncrementalDegree = compositionDegree - T: the same powers as B's; P_r o powers is not materialised, C is added in the pass
        const uint32_t offset = dcount + bcoef, cnt = b_inc > 0 ? 2 * V : V;
        std::vector<F> co = prng_many(eTree.root, offset + cnt);
        Bytes cb(cnt * ELEM);
        for (uint32_t i = 0; i < cnt; i++) le16(co[offset + i], cb.data() + ELEM * i);
        x.check(A.gs_combine_adjusted(x.c, eVectors.data(), cb.data(), b_inc > 0 ? cb.data() + ELEM * V : nullptr, V, b_inc > 0 ? psbPowers.p : nullptr, cEval.p, N,
                                      lEval.p), "gs_combine_adjusted(L)");
    }
    cEval.release();
    psbPowers.release();

    if (!lc_fused) clock.readme(x, "Combined P(x) and S(x) evaluations with C(x) evaluations");
    clock.mark("composition + LC issued (root read inside)");
    // 7 ----- low-degree proof (LowDegreeProver.ts:39-68, 176-221)
    if (N < 128) fail(GS_ERR_ARG, "Invalid array length");
    // transposeVector(v, 4) is never materialised: row r of it is v[r], v[r + rows], v[r + 2 rows], v[r + 3 rows], which the hashing,
    // folding and gathering below read in place
    Tree pTree0 = commit_rows4(x, alg, lEval.p, N / 4, N > 256);                                      // :45-46

    // layers (:176-221): the loop below is the recursion unrolled.  No root is read back inside it: the point every layer folds at,
    // prng(root of the tree above) (:194), is derived on the device from the root where it lies, so all layers
    // are enqueued without a round trip.  The launch that produces a tree's root also POSTS it to the host and derives that point
    // (gs_merkle_commit_rows_seed): the host picks each root up as soon as its tree exists and derives the layer's query positions and
    // batch-proof plans while the device folds the layers below
    double waited_ms = 0;          // host time blocked on roots that had not arrived yet (the device was the slower side)
    auto await_root = [&](uint64_t ticket, Tree &t) {
        const auto w0 = std::chrono::steady_clock::now();
        x.check(A.gs_readback_wait(x.c, ticket, t.root.data()), "gs_readback_wait(root)");
        waited_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    };
    std::vector<uint64_t> tickets;
    tickets.push_back(pTree0.ticket);
    std::vector<Layer> layers;
    Tree *pTree = &pTree0;
    const void *column_src = lEval.p;   // the current layer's values in natural order (the remainder at the end)
    uint64_t len = N;
    uint64_t max_degree_plus1 = composition_degree;
    uint32_t depth = 0;
    layers.reserve(32);
    // every layer's buffers first, then ONE call for the whole recursion (gs_fri_layers = per layer gs_fri_fold_at :189-198 +
    // gs_merkle_commit_rows_seed :201-202, with as few dependent launches as the sizes allow)
    while (len > 256) {
        const uint64_t rows = len / 4;
        layers.emplace_back();
        Layer &L = layers.back();
        L.pTree = pTree;
        L.column = column_src;
        L.rows = rows;
        L.column_length = rows;
        L.next = Buf(x, rows * ELEM);
        L.cTree.n = rows / 4;
        L.cTree.leaves = Buf(x, rows / 4 * DIGEST);
        L.cTree.nodes = Buf(x, rows / 4 * DIGEST);
        if (rows > 256) L.cTree.point = Buf(x, ELEM);                                                 // (no layer below the last tree)
        L.cTree.root.resize(DIGEST);
        column_src = L.next.p;
        pTree = &L.cTree;
        len = rows;
        max_degree_plus1 /= 4;
        depth++;
    }
    if (!layers.empty()) {
        std::vector<gs_fri_layer> outs(layers.size());
        for (size_t d = 0; d < layers.size(); d++)
            outs[d] = gs_fri_layer{layers[d].next.p, layers[d].cTree.leaves.p, layers[d].cTree.nodes.p, layers[d].cTree.point.p, 0};
        le16(omega, s16);
        x.check(A.gs_fri_layers(x.c, (gs_hash_alg)alg, s16, N, 1, lEval.p, N, pTree0.point.p, (uint32_t)outs.size(), outs.data()), "gs_fri_layers");
        for (size_t d = 0; d < layers.size(); d++) { layers[d].cTree.ticket = outs[d].ticket; tickets.push_back(outs[d].ticket); }
    }
    if (layers.size() > 60) fail(GS_ERR_ARG, "too many FRI components");
    clock.mark("FRI layers issued");
    clock.readme(x, "Computed low-degree proof: %zu FRI layers folded and committed", layers.size());
    // Everything the proof reads back — the queried rows and their batch proofs of every tree, the remainder (:179-187: the natural-
    // order vector the last polyValues came from) — is requested in ONE deferral window; the requests are planned root by root as
    // the roots arrive, the window closes with the single synchronisation of the proof
    struct Component { Bytes columnRoot; MerkleProof columnProof, polyProof; };
    std::vector<Component> components(layers.size());
    Readbacks rb;
    DeferWindow win(x);
    // spot checks of the evaluation tree (lib/Stark.ts:146-152, 274-296) and of the linear combination (LowDegreeProver.ts:52-54,
    // 302-309): positions from the root of the first FRI tree
    await_root(tickets[0], pTree0);
    const uint32_t exe_count = (uint32_t)std::min<uint64_t>(job.exe_query_count, N - N / E);
    const std::vector<uint64_t> exe_positions = query_indexes(pTree0.root, exe_count, N, (uint32_t)E);   // QueryIndexGenerator.ts:28-32
    std::vector<uint64_t> lc_positions;
    for (uint64_t p : exe_positions) lc_positions.push_back(p % (N / 4));
    lc_positions = unique_in_order(lc_positions);
    MerkleProof lcProof;
    rb.prove_batch(x, pTree0, lc_positions, &lcProof);
    rb.gather_rows4(x, lEval.p, N / 4, lc_positions, &lcProof);
    std::vector<uint64_t> aug;
    for (uint64_t p : exe_positions) { aug.push_back(p); aug.push_back((p + E) % N); }
    aug = unique_in_order(aug);
    MerkleProof evProof;
    rb.prove_batch(x, eTree, aug, &evProof);
    // the leaves of the evaluation tree are the committed vectors' elements side by side (lib/Stark.ts:284-296)
    std::vector<MerkleProof> cols(V);
    for (uint32_t r = 0; r < V; r++) rb.gather(x, eVectors[r], ELEM, aug, 1, V == 1 ? &evProof : &cols[r]);
    // queries of every layer (:209-219)
    for (size_t d = 0; d < layers.size(); d++) {
        Layer &L = layers[d];
        await_root(tickets[d + 1], L.cTree);
        if (d + 1 == layers.size()) clock.mark_split("waiting for FRI roots (device busy)", waited_ms, "query plans while the device folds");
        std::vector<uint64_t> positions = query_indexes(L.cTree.root, job.fri_query_count, L.column_length, (uint32_t)E);
        std::vector<uint64_t> rows_wanted;
        for (uint64_t p : positions) rows_wanted.push_back(p % (L.column_length / 4));
        rows_wanted = unique_in_order(rows_wanted);
        Component &c = components[d];
        c.columnRoot = L.cTree.root;
        rb.prove_batch(x, L.cTree, rows_wanted, &c.columnProof);
        rb.gather_rows4(x, L.next.p, L.column_length / 4, rows_wanted, &c.columnProof);
        rb.prove_batch(x, *L.pTree, positions, &c.polyProof);
        rb.gather_rows4(x, L.column, L.rows, positions, &c.polyProof);
    }
    if (layers.empty()) clock.mark_split("waiting for FRI roots (device busy)", waited_ms, "query plans while the device folds");
    clock.mark("last root here: the last layer's plan");
    std::vector<F> remainder(len);
    Bytes remainder_raw(len * ELEM);
    {
        std::vector<uint64_t> all(len);
        for (uint64_t i = 0; i < len; i++) all[i] = i;
        x.check(A.gs_gather(x.c, column_src, ELEM, all.data(), len, remainder_raw.data()), "gs_gather(remainder)");
    }
    win.end();
    clock.mark("remainder + answers fetched (one sync)");
    {
        Bytes &raw = remainder_raw;
        for (uint64_t i = 0; i < len; i++) remainder[i] = from16(raw.data() + ELEM * i);
        // verifyRemainder (:223-252)
        F rou = omega;
        for (uint32_t d = 0; d < depth; d++) { rou = hf_mul(rou, rou); rou = hf_mul(rou, rou); }      // omega^(4^depth)
        if (!remainder_is_low_degree(remainder, E, max_degree_plus1, rou, 1))
            fail(GS_ERR_ARG, "Low degree proof failed: Remainder is not a valid degree %llu polynomial", (unsigned long long)(max_degree_plus1 - 1));
    }
    clock.mark("remainder checked");
    clock.readme(x, "Computed low-degree proof: query answers + Computed %zu evaluation spot checks (one read-back), remainder verified", exe_positions.size());
    if (V > 1) {
        evProof.value_size = (uint64_t)V * ELEM;
        evProof.nvalues = (uint32_t)aug.size();
        evProof.values.resize(aug.size() * V * ELEM);
        for (size_t i = 0; i < aug.size(); i++)
            for (uint32_t r = 0; r < V; r++) memcpy(evProof.values.data() + (i * V + r) * ELEM, cols[r].values.data() + i * ELEM, ELEM);
    }

    clock.mark("query answers unpacked");
    // ----- Serializer.serializeProof (:35-79)
    out.clear();
    out.insert(out.end(), eTree.root.begin(), eTree.root.end());
    write_merkle_proof(out, evProof, (uint64_t)V * ELEM);
    out.insert(out.end(), pTree0.root.begin(), pTree0.root.end());
    write_merkle_proof(out, lcProof, 4 * ELEM);
    if (components.size() > 255) fail(GS_ERR_ARG, "too many FRI components");
    out.push_back((uint8_t)components.size());
    for (auto &c : components) {
        out.insert(out.end(), c.columnRoot.begin(), c.columnRoot.end());
        write_merkle_proof(out, c.columnProof, 4 * ELEM);
        write_merkle_proof(out, c.polyProof, 4 * ELEM);
    }
    if (remainder.size() > MAX_ARRAY) fail(GS_ERR_ARG, "remainder too long");
    out.push_back(remainder.size() == MAX_ARRAY ? 0 : (uint8_t)remainder.size());
    for (F v : remainder) { uint8_t b[ELEM]; le16(v, b); out.insert(out.end(), b, b + ELEM); }
    write_input_shapes(out, input_shapes);
    clock.mark("serialized");
    clock.readme(x, "Proof serialized");
}

// ---- one proof across several GPUs (same helpers, same coefficient streams, same wire format)
#include "prover_dist.h"

// ---- Stark.verify(): the CPU-side half of the acceptance loop, native
#include "verifier.h"
