/*
 * gstark.h — flat C ABI of the MI355X-native prime-field polynomial + Merkle-hash backend
 * that sits behind genSTARK's prove() pipeline.
 *
 * Every entry point replaces one member of the TypeScript surface that genSTARK calls on
 * `@guildofweavers/galois` (FiniteField / Vector / Matrix), `@guildofweavers/merkle`
 * (Hash / MerkleTree) and `@guildofweavers/air-assembly` (ProvingContext) — the packages whose
 * wasm/JS arithmetic this library replaces.  The reference call site (file:line under the genSTARK
 * checkout) is given next to each function.  An N-API addon (napi/) and the Python host mirror
 * (genstark_amd/) bind exactly these symbols; nothing else is exported.
 *
 * Conventions
 *  - A field element is gs_element_size() bytes (16; 32 in the 256- / 224-bit flavours below), little-endian,
 *    canonical (value < p); p = 2^128 - 9*2^32 + 1 in the main library (the 128-bit field galois accelerates;
 *    examples/mimc/mimc128.ts:13).  `gs_elt *` arguments point at ONE such element in host memory.
 *    This is the byte layout `Vector.copyValue` / `Matrix.rowsToBuffers` hand to the proof
 *    (lib/Stark.ts:284-296, lib/utils/serialization.ts:131-146).
 *  - A Vector is n contiguous elements.  A Matrix is rows*cols elements, row-major, contiguous
 *    (`Matrix.toBuffer()` must be row-contiguous: lib/components/LowDegreeProver.ts:45).
 *  - A digest is 32 bytes.  A digest Vector is n contiguous digests.
 *  - `void *` buffer arguments are DEVICE pointers obtained from gs_alloc (or any hipMalloc'd /
 *    torch-owned device memory); `const gs_elt *x`-style arguments and everything documented
 *    "host" are HOST pointers.
 *  - All calls are asynchronous on the context's HIP stream and ordered on it; gs_download,
 *    gs_gather and gs_sync block until the stream is drained.  Like the reference (synchronous,
 *    single-threaded JS) a context must not be used from two threads at once; a thread other than the
 *    one that called gs_ctx_create must make the context's device its current HIP device first
 *    (hipSetDevice is per-thread state; several contexts, one per thread, may share a GPU).  The one exception is
 *    gs_free: it only parks the block in the context's cache (under a lock shared with gs_alloc), so a garbage
 *    collector's finaliser may call it from any thread.
 *  - The library is built once per field: libgstark_hip.so computes in GF(2^128 - 9*2^32 + 1), the flavours
 *    libgstark_hip_q64.so / _q32.so / _q17.so in GF(2^64 - 21*2^30 + 1) / GF(2^32 - 3*2^25 + 1) / GF(96769) with 16-byte elements, the flavours
 *    libgstark_hip_p256.so / _p224.so in GF(2^256 - 351*2^32 + 1) / GF(2^224 - 2^96 + 1) with 32-byte elements
 *    (gs_field_modulus / gs_element_size say which), and libgstark_hip_rt.so in whatever odd modulus below 2^256 gs_set_modulus
 *    names (32-byte elements, generic kernels).
 *  - Every function returns GS_OK (0) or a negative gs_status; gs_last_error(ctx) describes it.
 *    There is NO CPU fallback: without a gfx950 device gs_ctx_create fails.
 */
#ifndef GSTARK_H
#define GSTARK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 1

typedef enum gs_status {
    GS_OK = 0,
    GS_ERR_ARG = -1,          /* bad argument (size not a power of two, null pointer, ...) */
    GS_ERR_DEVICE = -2,       /* HIP runtime error / no gfx950 device */
    GS_ERR_UNSUPPORTED = -3,  /* valid request this build does not implement */
    GS_ERR_OOM = -4
} gs_status;

typedef enum gs_hash_alg {    /* createHash(algorithm, useWasm): lib/Stark.ts:19-20,50 */
    GS_HASH_SHA256 = 0,
    GS_HASH_BLAKE2S256 = 1
} gs_hash_alg;

typedef struct gs_ctx gs_ctx;
typedef uint8_t gs_elt;       /* first byte of one field element in HOST memory: gs_element_size() bytes, little-endian */

#define GS_ELEMENT_BYTES 16   /* of the main (128-bit) library; gs_element_size() is authoritative */
#define GS_DIGEST_BYTES 32
#define GS_MAX_COMBINE 64     /* vectors per kernel launch of gs_combine_many / gs_hash_merge_rows / gs_sub_matrix_from_vectors: longer lists
                                 (AIRs with more than 32 registers) are accepted and split inside the library */

/* ---- context / memory -------------------------------------------------------------------------
 * Replaces: the galois wasm linear memory handed over as `wasmOptions.memory`
 * (lib/Stark.ts:37-40,346-354) and the JS-GC ownership of Vector/Matrix objects. */
int gs_abi_version(void);
const char *gs_backend_name(void);                     /* "hip-gfx950" for the product library */
int gs_ctx_create(int device, void *hip_stream /* NULL: library-owned stream */, gs_ctx **out);
void gs_ctx_destroy(gs_ctx *ctx);
const char *gs_last_error(const gs_ctx *ctx);
int gs_sync(gs_ctx *ctx);
void *gs_stream(gs_ctx *ctx);                          /* the hipStream_t the kernels run on */
int gs_element_size(void);                             /* FiniteField.elementSize: lib/Stark.ts:260,285,299 */
int gs_field_modulus(gs_elt *out_le);                  /* FiniteField.modulus */
/* createPrimeField(modulus) (index.ts:14) for a modulus none of the fixed builds computes in: libgstark_hip_rt.so takes any ODD modulus
 * below 2^256 here — `bytes` little-endian bytes, at most gs_element_size() — ONCE per process and before the first gs_ctx_create (the
 * field constants are process-wide; a second, different modulus is GS_ERR_UNSUPPORTED; 32-byte canonical elements; generic kernels
 * only: AIR programs are interpreted, nothing is tuned).  A fixed build answers GS_OK for its own modulus, GS_ERR_UNSUPPORTED otherwise. */
int gs_set_modulus(const uint8_t *modulus_le, uint32_t bytes);

int gs_alloc(gs_ctx *ctx, uint64_t bytes, void **dptr);
int gs_free(gs_ctx *ctx, void *dptr);                 /* parks the block in the context's cache (no device sync) */
int gs_cache_trim(gs_ctx *ctx);                        /* synchronises and returns every cached block to the driver */
/* newVectorFrom (BoundaryConstraints.ts:24,40).  host_src may be reused as soon as the call returns; the copy itself is ordered on
 * the context's queue like every other entry (payloads up to 4 MiB are staged in pinned memory and do not block the host). */
int gs_upload(gs_ctx *ctx, void *dst, const void *host_src, uint64_t bytes);
int gs_download(gs_ctx *ctx, void *host_dst, const void *src, uint64_t bytes); /* toValues / toBuffer: CompositionPolynomial.ts:58 */
int gs_copy(gs_ctx *ctx, void *dst, const void *src, uint64_t bytes);
/* rowsToBuffers / copyValue / MerkleTree node reads: lib/Stark.ts:290, LowDegreeProver.ts:53,214,217.
 * Copies `count` records of `rec_bytes` at record indices idx[] (host) from src to host_out. */
int gs_gather(gs_ctx *ctx, const void *src, uint64_t rec_bytes, const uint64_t *idx_host,
              uint64_t count, void *host_out);
/* Device-to-device forms used where the result travels on to another device (csrc/prover_dist.cc: query answers of a proof that
 * is spread over several GPUs are packed by their owners and exchanged with one collective):
 *   gs_gather_words:      dst[t] = the 16-byte word at DEVICE ADDRESS addrs[t] (addrs: count 64-bit addresses in device memory).
 *   gs_transpose_records: dst[c*rows + r] = src[r*cols + c] for records of rec_bytes (a multiple of 16) — strided shares of a vector,
 *                         gathered rank after rank, back into natural order; the re-sharding of leaf digests before a Merkle subtree. */
int gs_gather_words(gs_ctx *ctx, const void *addrs, uint64_t count, void *dst);
int gs_transpose_records(gs_ctx *ctx, const void *src, uint64_t rows, uint64_t cols, uint64_t rec_bytes, void *dst);
/* Deferred read-backs: between gs_defer_begin and gs_defer_end, gs_gather and gs_merkle_prove_batch queue their device work and
 * return at once (shapes — ncols_out, col_lens_out — are filled immediately, they are host knowledge); the host output buffers,
 * which must stay valid, are filled by gs_defer_end after ONE synchronisation for the whole window.  The ~40 query answers of a
 * proof (lib/Stark.ts:146-152; LowDegreeProver.ts:209-219) become one round trip.  Any other staging user inside the window
 * delivers what is queued first. */
int gs_defer_begin(gs_ctx *ctx);
int gs_defer_end(gs_ctx *ctx);
/* Posted read-backs: gs_readback_post queues a copy of `bytes` (a multiple of 16, at most 256) from device memory behind the work
 * already issued and returns a ticket at once; gs_readback_wait blocks until THAT copy has landed — not for whatever was queued
 * after it — and delivers the bytes.  The FRI layers are issued without a round trip (LowDegreeProver.ts:176-221 as one queue);
 * with every tree's root posted behind it, the host derives the query positions of a layer (QueryIndexGenerator.ts:39-67) and
 * plans its batch proofs while the device is still folding the layers below.  At most 64 tickets are outstanding; older ones
 * lapse. */
int gs_readback_post(gs_ctx *ctx, const void *src, uint32_t bytes, uint64_t *ticket);
int gs_readback_wait(gs_ctx *ctx, uint64_t ticket, void *host_dst);

/* ---- vector arithmetic (FiniteField vector ops) ------------------------------------------------ */
/* getPowerSeries(base, n): out[i] = base^i.  CompositionPolynomial.ts:94,132; LinearCombination.ts:46;
 * also how air-assembly builds execution/evaluation/composition domains (lib/Stark.ts:90-91). */
int gs_power_series(gs_ctx *ctx, const gs_elt *base, uint64_t n, void *out);
/* addVectorElements(a, b): CompositionPolynomial.ts:145; LinearCombination.ts:63 */
int gs_vec_add(gs_ctx *ctx, const void *a, const void *b, uint64_t n, void *out);
/* subVectorElements(a, b: Vector): ZeroPolynomial.ts:41-42 (Vector form) */
int gs_vec_sub(gs_ctx *ctx, const void *a, const void *b, uint64_t n, void *out);
/* mulVectorElements(a, b: Vector): CompositionPolynomial.ts:98,120,136; LinearCombination.ts:50 */
int gs_vec_mul(gs_ctx *ctx, const void *a, const void *b, uint64_t n, void *out);
/* scalar forms — (a, b: bigint): ZeroPolynomial.ts:41-42; LinearCombination.ts:75 */
int gs_vec_add_scalar(gs_ctx *ctx, const void *a, const gs_elt *s, uint64_t n, void *out);
int gs_vec_sub_scalar(gs_ctx *ctx, const void *a, const gs_elt *s, uint64_t n, void *out);
int gs_vec_mul_scalar(gs_ctx *ctx, const void *a, const gs_elt *s, uint64_t n, void *out);
/* invVectorElements: out[i] = a[i]^-1, with 0^-1 := 0 (batch inversion) */
int gs_vec_inv(gs_ctx *ctx, const void *a, uint64_t n, void *out);
/* divVectorElements(a, b): out[i] = a[i] * b[i]^-1.  CompositionPolynomial.ts:117;
 * divMatrixElements (BoundaryConstraints.ts:92) is the same call over rows*cols elements. */
int gs_vec_div(gs_ctx *ctx, const void *a, const void *b, uint64_t n, void *out);
/* expVectorElements(a, e): out[i] = a[i]^e (examples/poseidon/utils.ts:34) */
int gs_vec_exp(gs_ctx *ctx, const void *a, const gs_elt *e, uint64_t n, void *out);
/* combineManyVectors(v[], k): out[i] = sum_j v_j[i] * k_j.  vecs_host: host array of `count`
 * device pointers; coeffs_host: count*16 bytes.  CompositionPolynomial.ts:105,142; LinearCombination.ts:60 */
int gs_combine_many(gs_ctx *ctx, const void *const *vecs_host, const uint8_t *coeffs_host,
                    uint32_t count, uint64_t n, void *out);
/* A merge together with its degree adjustment (CompositionPolynomial.ts:83-107 for Q, :124-146 for B; LinearCombination.ts:44-63):
 *   out[i] = sum_j coeffs[j] * v_j[i]  +  powers[i] * sum_j adj_coeffs[j] * v_j[i]  (+ plus[i])
 * — the reference materialises every adjusted vector v_j o powers (mulVectorElements) and merges 2*count vectors; the sum is the
 * same field element (sum_j k'_j (v_j[i] powers[i]) = powers[i] sum_j k'_j v_j[i]), the vectors are read once and nothing
 * intermediate is written.  coeffs_host or adj_coeffs_host may be NULL (that sum is absent; powers is only read with
 * adj_coeffs_host); plus may be NULL, and may be `out`.  Called by the native driver. */
int gs_combine_adjusted(gs_ctx *ctx, const void *const *vecs_host, const uint8_t *coeffs_host, const uint8_t *adj_coeffs_host, uint32_t count,
                        const void *powers, const void *plus, uint64_t n, void *out);
/* combineVectors(a, b) -> scalar sum_i a[i]*b[i] (host out).  CompositionPolynomial.ts:168,188 */
int gs_combine(gs_ctx *ctx, const void *a, const void *b, uint64_t n, gs_elt *out_host);
/* pluckVector(v, skip, times): out[i] = v[(i*skip) mod vlen], i < times.  ZeroPolynomial.ts:40 */
int gs_pluck(gs_ctx *ctx, const void *v, uint64_t vlen, uint64_t skip, uint64_t times, void *out);
/* Fused ZeroPolynomial.evaluateAll (ZeroPolynomial.ts:36-44) + the division of CompositionPolynomial.ts:117 (SURVEY 8a row A8,
 * "zpoly_numden"): over the evaluation domain {omega^i}, i < n, of a trace of `steps` steps
 *     out[i] = (omega^i - x_last) / (omega^(i*steps) - 1),   0 where the denominator vanishes
 * — the values pluckVector / subVectorElements / divVectorElements produce, without materialising them (the denominator takes
 * only n/steps distinct values).  n / steps <= 32. */
int gs_zero_poly_inverses(gs_ctx *ctx, const gs_elt *omega, uint64_t n, uint64_t steps, const gs_elt *x_last, void *out);
/* The division of BoundaryConstraints.ts:87-92 when the divisor's roots are domain points (they are: assertions sit on
 * execution-domain steps): num is rows x n, row-major;
 *     out[r][i] = num[r][i] / prod_{a < roots_per_row[r]} (omega^i - omega^(root_index[r*max_roots + a])),   0 where a factor vanishes
 * — the values evalPolysAtRoots(Z polynomials) + divMatrixElements produce.  At most 4 roots per row (GS_ERR_UNSUPPORTED above:
 * use the general members).  The library keeps the table 1/(omega^j - 1) of the domain with its transform plan. */
int gs_div_by_domain_roots(gs_ctx *ctx, const void *num, uint32_t rows, uint64_t n, const gs_elt *omega,
                           const uint64_t *root_index_host, const uint32_t *roots_per_row_host, uint32_t max_roots, void *out);
/* The same two over a COSET of a domain: point i is shift * w^i, i < m, w a primitive m-th root of unity — what ONE RANK of a proof
 * spread over several GPUs holds of the evaluation domain (csrc/prover_dist.h: w = omega^ranks, shift = omega^rank).  At shift = 1
 * they are the two functions above.
 *   gs_zero_poly_inverses_coset:  out[i] = (x_i - x_last) / (x_i^steps - 1),  x_i^steps takes m / steps (<= 32) values: a table again
 *   gs_div_by_domain_roots_coset: out[r][i] = num[r][i] / prod_a (x_i - w^k_a): the roots are powers of w (NOT shifted), so each factor
 *     is w^-k * u_s[(i - k) mod m] with u_s[j] = 1 / (shift * w^j - 1) — one table per (w, m, shift), cached with the transform plan */
int gs_zero_poly_inverses_coset(gs_ctx *ctx, const gs_elt *w, uint64_t m, const gs_elt *shift, uint64_t steps, const gs_elt *x_last, void *out);
int gs_div_by_domain_roots_coset(gs_ctx *ctx, const void *num, uint32_t rows, uint64_t m, const gs_elt *w, const gs_elt *shift,
                                 const uint64_t *root_index_host, const uint32_t *roots_per_row_host, uint32_t max_roots, void *out);
/* transposeVector(v, cols, step): rows = n/(cols*step); out[r*cols+c] = v[(r + c*rows)*step].
 * LowDegreeProver.ts:42,162,190,198 */
int gs_transpose_vector(gs_ctx *ctx, const void *v, uint64_t n, uint32_t cols, uint64_t step, void *out);
/* transposeMatrix(m): out[c*rows+r] = m[r*cols+c].  LowDegreeProver.ts:181 (joinMatrixRows is a no-op view) */
int gs_transpose_matrix(gs_ctx *ctx, const void *m, uint64_t rows, uint64_t cols, void *out);
/* subMatrixElementsFromVectors(vectors[], m): out[r][i] = v_r[i] - m[r][i].  BoundaryConstraints.ts:91 */
int gs_sub_matrix_from_vectors(gs_ctx *ctx, const void *const *vecs_host, const void *m,
                               uint32_t rows, uint64_t cols, void *out);

/* ---- polynomial ops ---------------------------------------------------------------------------- */
/* evalPolysAtRoots(polys: Matrix, roots) / evalPolyAtRoots(poly, roots): forward NTT of each of
 * `rows` polynomials with `poly_len` (<= n) coefficients, zero-extended, at the n-th roots
 * {omega^i}; out is rows*n, natural order (out[r][i] = p_r(omega^i)).
 * lib/Stark.ts:109; CompositionPolynomial.ts:110; BoundaryConstraints.ts:87-88 */
int gs_eval_polys_at_roots(gs_ctx *ctx, const void *polys, uint32_t rows, uint64_t poly_len,
                           const gs_elt *omega, uint64_t n, void *out);
/* interpolateRoots(roots, ys: Vector|Matrix): inverse NTT of each row; out[r] = coefficients of the
 * unique deg<n polynomial with p(omega^i) = ys[r][i].  lib/Stark.ts:106; CompositionPolynomial.ts:109 */
int gs_interpolate_roots(gs_ctx *ctx, const void *ys, uint32_t rows, const gs_elt *omega,
                         uint64_t n, void *out);
/* evalPolyAt(poly, x) -> scalar.  BoundaryConstraints.ts:59-60; LowDegreeProver.ts:248 */
int gs_eval_poly_at(gs_ctx *ctx, const void *poly, uint64_t len, const gs_elt *x, gs_elt *out_host);
/* interpolateQuarticBatch(xs, ys): per row, the cubic through (xs[r][c], ys[r][c]), c<4; out rows*4
 * coefficients (ascending).  LowDegreeProver.ts:137,191 */
int gs_interpolate_quartic_batch(gs_ctx *ctx, const void *xs, const void *ys, uint64_t rows, void *out);
/* Same result as gs_interpolate_quartic_batch when xs = transposeVector(powerSeries(omega, n), 4, step)
 * (the only shape the prover builds: LowDegreeProver.ts:190-191): xs[r][c] = omega^((r + c*rows)*step),
 * rows*4*step == n.  xs is never materialised. */
int gs_interpolate_quartic_domain(gs_ctx *ctx, const gs_elt *omega, uint64_t n, uint64_t step,
                                  const void *ys, uint64_t rows, void *out);
/* evalQuarticBatch(polys, x) -> Vector of rows values.  LowDegreeProver.ts:140,195 */
int gs_eval_quartic_batch(gs_ctx *ctx, const void *polys, uint64_t rows, const gs_elt *x, void *out);
/* One FRI folding step on a column of m values (LowDegreeProver.ts:189-198: transposeVector(column, 4) + interpolateQuarticBatch +
 * evalQuarticBatch fused — the transposed matrix and the cubics are never written): rows = m/4, m * step = n,
 *     out[r] = P_r(x),  P_r the cubic through (omega^((r + c*rows) * step), column[r + c*rows]), c < 4. */
int gs_fri_fold(gs_ctx *ctx, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const gs_elt *x, void *out);
/* The same folding step with x = field.prng(seed) derived ON THE DEVICE from a 32-byte seed in device memory — the root of the tree
 * that commits to `column`'s parent (nodes + 32 of gs_merkle_build), LowDegreeProver.ts:194: prng(seed) = sha256(seed) read as a
 * big-endian integer, mod p.  The host never needs the root to issue the next layer: a driver enqueues every FRI layer without a
 * round trip and reads all the roots back once. */
int gs_fri_fold_seeded(gs_ctx *ctx, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m, const void *seed32_dev, void *out);
/* ... with x = field.prng(seed) * scale: the fold on ONE RANK'S SHARE of a column that is spread over several GPUs sees the domain
 * scaled by shift^-step (csrc/prover_dist.h), and asks for the same point in its own coordinates. */
int gs_fri_fold_seeded_scaled(gs_ctx *ctx, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t m,
                              const void *seed32_dev, const gs_elt *scale, void *out);
/* The same with less latency per layer (a small layer is a handful of dependent launches, each ~4 us of dispatch):
 *   gs_merkle_commit_rows_seed: gs_merkle_commit_rows whose LAST launch — the one that produces the root — also (a) posts the
 *     root to the host exactly like gs_readback_post(nodes + 32, 32) (*ticket for gs_readback_wait; ticket may be NULL) and
 *     (b) derives field.prng(root) into point_out (16 bytes of device memory; may be NULL): LowDegreeProver.ts:194 and :201-202
 *     in the launch that already holds the root;
 *   gs_fri_fold_at: gs_fri_fold at the point stored at x_dev (device memory). */
int gs_merkle_commit_rows_seed(gs_ctx *ctx, gs_hash_alg alg, const void *const *vecs_host, uint32_t count, uint64_t n, void *leaves, void *nodes,
                               void *point_out, uint64_t *ticket);
int gs_fri_fold_at(gs_ctx *ctx, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t len, const void *x_dev, void *out);
/* A RUN of FRI layers in one call (LowDegreeProver.ts:176-221, the recursion from one column down): for i < nlayers, with
 * column_0 = column (len values), len_i = len / 4^i, step_i = step * 4^i and x_0 the point stored at x_dev,
 *     layers[i].next   = gs_fri_fold_at(column_i, len_i, step_i, x_i)                      (len_i / 4 values; column_{i+1})
 *     layers[i].leaves, .nodes, .ticket, .point_out = gs_merkle_commit_rows_seed over the four quarters of layers[i].next
 *                                                                                           (len_i / 16 leaves; :201-202)
 *     x_{i+1} = field.prng(root_i)                                                          (:194; also stored at point_out if non-NULL)
 * — byte for byte what those two calls per layer give, with as few dependent launches as the sizes allow: a layer of at most 2^15
 * leaves is ONE launch (every workgroup folds the rows of its own 256 leaves, hashes them and builds the subtree above them; the
 * workgroup that finishes last builds the tree over the subtree roots, posts the root and derives the next point), and from the
 * first layer that fits one workgroup on, ALL remaining layers run in that same launch.  Every root is posted (layers[i].ticket
 * for gs_readback_wait).  Requirements: len a power of two >= 32, len * step == n, len / 4^nlayers >= 8. */
struct gs_fri_layer {
    void *next;          /* out: len_i / 4 elements */
    void *leaves;        /* out: len_i / 16 digests */
    void *nodes;         /* out: len_i / 16 digests, heap order (nodes + 32 = the root) */
    void *point_out;     /* out, may be NULL: field.prng(root_i), one element */
    uint64_t ticket;     /* out: posted root */
};
int gs_fri_layers(gs_ctx *ctx, gs_hash_alg alg, const gs_elt *omega, uint64_t n, uint64_t step, const void *column, uint64_t len,
                  const void *x_dev, uint32_t nlayers, struct gs_fri_layer *layers);

/* ---- hashing / Merkle (merkle package) --------------------------------------------------------- */
/* Hash.digest(Buffer) on host bytes (verifier side; lib/utils/index.ts:37) — runs on the device
 * like every other hash so that prover and verifier share one implementation. */
int gs_hash_digest(gs_ctx *ctx, gs_hash_alg alg, const uint8_t *msg_host, uint64_t len, uint8_t out_host[32]);
/* Hash.mergeVectorRows(vectors): out[i] = H(v_0[i] || ... || v_{k-1}[i]).  lib/Stark.ts:115 */
int gs_hash_merge_rows(gs_ctx *ctx, gs_hash_alg alg, const void *const *vecs_host, uint32_t count,
                       uint64_t n, void *out_digests);
/* Hash.digestValues(buf, valueSize): out[i] = H(buf[i*valueSize .. (i+1)*valueSize)).
 * LowDegreeProver.ts:45,163,201 */
int gs_hash_digest_values(gs_ctx *ctx, gs_hash_alg alg, const void *buf, uint64_t value_size,
                          uint64_t count, void *out_digests);
/* MerkleTree.create(leaves, hash): nodes is n digests in heap order — nodes[1] is the root,
 * nodes[i] = H(nodes[2i] || nodes[2i+1]) for i < n/2, nodes[n/2 + i] = H(leaf[2i] || leaf[2i+1]);
 * nodes[0] is zero.  n must be a power of two >= 2.  lib/Stark.ts:118; LowDegreeProver.ts:46,164,202.
 * (.root / .proveBatch read nodes through gs_gather.) */
int gs_merkle_build(gs_ctx *ctx, gs_hash_alg alg, const void *leaves, uint64_t n, void *nodes);
/* Hash.mergeVectorRows(vectors) followed by MerkleTree.create(hashedRows, hash) — lib/Stark.ts:115-118, LowDegreeProver.ts:45-46,
 * 201-202 — as ONE entry: leaves[i] = H(v_0[i] || ... || v_{k-1}[i]) (n digests, written: proveBatch reads them) and the tree over
 * them in `nodes` (layout as gs_merkle_build).  Same bytes as the two members called one after the other; the device hashes the
 * leaves and the layers above them in the same launches (csrc/hash.hip).  n a power of two >= 2. */
int gs_merkle_commit_rows(gs_ctx *ctx, gs_hash_alg alg, const void *const *vecs_host, uint32_t count, uint64_t n,
                          void *leaves, void *nodes);

/* MerkleTree.proveBatch(indexes): lib/Stark.ts:150; LowDegreeProver.ts:52,213,216.  Plans the batch proof on the
 * host and fetches every needed digest in one device gather.  Outputs (host): values_out = count digests in
 * REQUEST order; *ncols_out = number of node columns (one per distinct leaf pair, ascending); col_lens_out[i] =
 * digests in column i (capacity `count` entries); nodes_out = the columns concatenated (capacity nodes_cap digests,
 * count * log2(n) always suffices).  Column layout: SURVEY.md appendix A.7 (sibling leaf first when it was not
 * itself requested, then per level the sibling nodes that are not on another requested path). */
int gs_merkle_prove_batch(gs_ctx *ctx, const void *leaves, const void *nodes, uint64_t n, const uint64_t *idx_host,
                          uint32_t count, uint8_t *values_out, uint32_t *ncols_out, uint32_t *col_lens_out,
                          uint8_t *nodes_out, uint64_t nodes_cap);

/* ---- small-polynomial host arithmetic ----------------------------------------------------------------
 * FiniteField.interpolate(xs, ys) (Lagrange; BoundaryConstraints.ts:42, LowDegreeProver.ts:243) and evalPolyAt over a
 * list of points (LowDegreeProver.ts:248) are only ever called with at most a few hundred points (#assertions,
 * <= 256 remainder values).  They are O(n^2) host arithmetic in the reference's JS layer and stay host arithmetic
 * here (native 64-bit limbs); all pointers are HOST pointers, n <= 4096. */
int gs_small_interpolate(const uint8_t *xs_host, const uint8_t *ys_host, uint32_t n, uint8_t *coeffs_out);
int gs_small_eval_poly(const uint8_t *poly_host, uint32_t len, const uint8_t *xs_host, uint32_t m, uint8_t *out_host);

/* ---- host helper: Fiat-Shamir query positions -----------------------------------------------------------------------
 * lib/components/QueryIndexGenerator.ts:39-67 (getPseudorandomIndexes + its sha256-of-a-bigint helper), host only, no
 * context: state = sha256(seed); candidate i = sha256(Buffer.from((state + i).toString(16), 'hex')) mod max — the
 * reference's quirk that an odd number of hex digits loses the last nibble included —, multiples of
 * `exclude_multiples_of` (0 = none) and repeats skipped, until `count` distinct indexes exist or count*1000 candidates were
 * tried (GS_ERR_ARG).  ~700 hashes per proof: a measurable slice of a 20 ms prove() when the host language is Python. */
int gs_pseudorandom_indexes(const uint8_t *seed_host, uint32_t seed_len, uint32_t count, uint64_t max,
                            uint32_t exclude_multiples_of, uint64_t *out_host);

/* ---- AIR (air-assembly ProvingContext, MiMC instance) ------------------------------------------ */
/* context.generateExecutionTrace() for the MiMC AIR of examples/mimc/mimc128Assembly.ts:28-51:
 * trace[0] = seed, trace[i+1] = trace[i]^3 + rc[i mod nrc] (examples/mimc/utils.ts:7-15).
 * Inherently sequential: computed on the host CPU, then copied to `out` (steps elements). */
int gs_mimc_trace(gs_ctx *ctx, const gs_elt *seed, const uint8_t *rc_host, uint32_t nrc,
                  uint64_t steps, void *out);
/* context.evaluateTransitionConstraints(pPolys) body for that AIR over the composition domain
 * (CompositionPolynomial.ts:76): q[j] = p[(j + shift) mod nc] - (p[j]^3 + k[j mod klen]),
 * p = P on the composition domain, shift = nc/steps, k = cyclic static register values. */
int gs_mimc_constraints(gs_ctx *ctx, const void *p_comp, uint64_t nc, uint64_t shift,
                        const void *k_table, uint64_t klen, void *out);
/* CompositionPolynomial.evaluateAll for that AIR in one pass over the evaluation domain (CompositionPolynomial.ts:71-146 with
 * BoundaryConstraints.ts:71-95 and ZeroPolynomial.ts:36-44; the combined Q has degree < the composition domain size, so its
 * extension is the constraint expression on the evaluation domain): for i < n, x = omega^i, p = P over the evaluation domain,
 *     Q      = p[(i + n/steps) mod n] - (p[i]^3 + k[i mod klen])                      k: the cyclic register over this domain
 *     D      = Q * (d0 + d1 * x^q_inc) * (x - x_last) / (x^steps - 1)                 x_last = omega^((steps - 1) * n/steps)
 *     B      = (p[i] - I(x)) / prod_{a < nroots} (x - omega^root_index[a])            I = sum_c ipoly[c] x^c, nroots coefficients
 *     C      = D + B * (b0 + b1 * x^b_inc)                                            0^-1 = 0 in both divisions
 *     out[i] = C                                  when lc_coeffs_host is NULL,
 *            = C + p[i] * (l0 + l1 * x^b_inc)     otherwise: LinearCombination.computeMany for one register, no secret registers
 *                                                 (LinearCombination.ts:36-64; same degree increment, same prng stream)
 * coeffs_host = d0, d1, b0, b1; lc_coeffs_host = l0, l1; q_inc, b_inc multiples of steps; n/steps <= 32; 1 <= nroots <= 4 (GS_ERR_UNSUPPORTED otherwise:
 * the member-by-member sequence gives the same values). */
int gs_mimc_composition(gs_ctx *ctx, const void *p_eval, uint64_t n, uint64_t steps, const gs_elt *omega, const void *k_table,
                        uint64_t klen, const uint8_t *coeffs_host, uint64_t q_inc, uint64_t b_inc, const uint8_t *ipoly_host,
                        const uint64_t *root_index_host, uint32_t nroots, const uint8_t *lc_coeffs_host, void *out);

/* ---- AIR (air-assembly ProvingContext), generic straight-line programs -------------------------------------
 * air-assembly compiles an AIR's transition function and constraint evaluator into generated code over the field
 * object (lib/Stark.ts:97, CompositionPolynomial.ts:76).  Here the same two functions are straight-line programs for a
 * small register machine (no branches, no loops), run by one device thread per composition-domain point
 * (constraints) or by one host core per step (trace: steps are sequentially dependent).  An instruction is four
 * uint32 words {op, dst, a, b} over a scratch file of vm_regs field elements:
 *     0 LOADC  vm[dst] = consts[a]            1 LOADR  vm[dst] = current row, register a
 *     2 LOADN  vm[dst] = next row, register a 3 LOADS  vm[dst] = static register a at this step / point
 *     4 ADD    5 SUB    6 MUL   vm[dst] = vm[a] op vm[b]
 *     7 POW    vm[dst] = vm[a]^b  (b: literal exponent)      8 POWC  vm[dst] = vm[a]^consts[b] (128-bit exponent)
 *     9 OUT    output[dst] = vm[a]   (next-row register for the trace, constraint index for the evaluator)
 * Static registers are cyclic: in the trace, register s at step i is static_values[s][i mod period_s]; in the evaluator
 * it is static_tables[s][j mod len_s], the register's polynomial evaluated over the composition domain. */
#define GS_AIR_MAX_VM_REGS 64
#define GS_AIR_MAX_REGISTERS 64
int gs_air_trace(gs_ctx *ctx, const uint32_t *code_host, uint32_t ninstr, const uint8_t *consts_host, uint32_t nconsts,
                 uint32_t vm_regs, uint32_t registers, const uint8_t *static_values_host, const uint32_t *static_periods_host,
                 uint32_t nstatic, const uint8_t *first_row_host, uint64_t steps, void *out /* registers x steps */);
/* Trace made of `segments` INDEPENDENT runs of `segment_len` steps (AirScript `for each (input) { init {...} for steps [...] }`
 * with several inputs: examples/rescue/hash4x128.ts:60-81, examples/poseidon/hash6x128.ts:62-81): segment s starts from
 * first_rows[s] (registers elements) and fills steps [s*segment_len, (s+1)*segment_len); static registers are indexed by the
 * global step.  The segments do not depend on each other: one DEVICE thread interprets the program for each of them.
 * init_code (optional, init_ninstr = 0 for none) is the `init { ... }` block: a program over the same constant pool that
 * maps first_rows[s] (read with LOADR, e.g. the raw inputs padded with zeros) to the segment's actual first row. */
int gs_air_trace_segments(gs_ctx *ctx, const uint32_t *code_host, uint32_t ninstr, const uint32_t *init_code_host, uint32_t init_ninstr,
                          const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs, uint32_t registers,
                          const uint8_t *static_values_host, const uint32_t *static_periods_host, uint32_t nstatic,
                          const uint8_t *first_rows_host /* segments x registers */, uint64_t segments, uint64_t segment_len,
                          void *out /* registers x (segments*segment_len) */);
/* AIR programs compiled instead of interpreted (the reference's air-assembly generates code for an AIR when it is instantiated):
 * gs_air_trace_segments and gs_air_constraints turn their program into HIP source, compile it with hiprtc (about a second per
 * kernel) and launch that; same values; the interpreter remains the fallback.  Code objects are kept per process and on disk
 * (GSTARK_JIT_CACHE_DIR, default ~/.cache/gstark_jit).  enable = 2 (the default of a new context, GSTARK_AIR_JIT unset or "auto"):
 * compiled whenever the code object already exists, otherwise interpreted NOW and built in the background for later launches and
 * processes; 1 (GSTARK_AIR_JIT=1): compile on first use; 0 (GSTARK_AIR_JIT=0): always interpret. */
int gs_air_jit(gs_ctx *ctx, int enable);
uint64_t gs_air_jit_launches(const gs_ctx *ctx);       /* how many launches ran compiled programs so far (0: everything was interpreted) */
/* Measurement aid (bench.py: roofline.kernels[]; nothing a proof depends on): while enabled, every kernel launch of the hot path adds
 * what it MUST move — the bytes it has to read once and write once given its arguments (SURVEY 8d's "algorithmic bytes": an NTT pass
 * 16 B in + 16 B out per element, a Merkle layer 64 B in + 32 B out per node, ...) — and its work units (hash kernels: compressions;
 * transforms and pointwise kernels: field elements written) to a per-kernel tally of the context.  Joined with a rocprofv3 kernel
 * trace of the same run by kernel name this gives every kernel's achieved GB/s against the HBM roof.  enable(1) also clears the tally. */
struct gs_traffic_entry {
    char kernel[64];             /* as rocprofv3 prints it up to the parameter list, e.g. "k_merkle_fused<1, 2>" */
    uint64_t launches, bytes, units;
};
int gs_traffic_enable(gs_ctx *ctx, int on);
int gs_traffic_read(gs_ctx *ctx, struct gs_traffic_entry *out, uint32_t cap, uint32_t *count);
/* Compile-only check, no context and no device: GS_OK when the source generated for the program builds for gfx950 (kind 0: the
 * trace program with its optional init program, arguments as gs_air_trace_segments; kind 1: the constraint program, arguments as
 * gs_air_constraints).  GS_ERR_UNSUPPORTED otherwise, with the compiler's log (NUL-terminated, truncated) in log_out. */
int gs_air_jit_check(int kind, const uint32_t *code_host, uint32_t ninstr, const uint32_t *init_code_host, uint32_t init_ninstr,
                     const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs, uint32_t registers,
                     const uint64_t *static_lens_host, uint32_t nstatic, char *log_out, uint64_t log_cap);
int gs_air_constraints(gs_ctx *ctx, const uint32_t *code_host, uint32_t ninstr, const uint8_t *consts_host, uint32_t nconsts,
                       uint32_t vm_regs, uint32_t registers, uint32_t constraints, const void *p_comp /* registers x nc */,
                       uint64_t nc, uint64_t shift, const void *static_tables /* device, concatenated */,
                       const uint64_t *static_lens_host, uint32_t nstatic, void *out /* constraints x nc */);
/* The tail of CompositionPolynomial.evaluateAll and LinearCombination.computeMany in ONE pass over the evaluation domain
 * (CompositionPolynomial.ts:113-146, BoundaryConstraints.ts:71-95, LinearCombination.ts:36-64), for every point x_i = omega^i:
 *   c[i] = q[i] * z_inv[i]                                                                          D(x) = Q(x) / Z(x)
 *        + sum_b (b_coeffs[b] + powers[i] * b_adj[b]) * (b_vecs[b][i] - I_b(x_i)) / prod_a (x_i - omega^root[b][a])
 *   l[i] = c[i] + sum_v (l_coeffs[v] + powers[i] * l_adj[v]) * l_vecs[v][i]
 * I_b: row b of ipolys_host (ilen coefficients, zero-extended: the interpolant through register b's assertions); the divisor's roots
 * are domain points, given by their positions (as gs_div_by_domain_roots).  b_adj / l_adj NULL: no degree adjustment of that sum.
 * The two vectors that depend on the domain alone need not exist: z_inv NULL -> 1/Z(x_i) as gs_zero_poly_inverses(omega, n, z_steps,
 * x_last) defines it, computed per point (n / z_steps <= 32); powers NULL (with an adjusted sum) -> powers[i] = omega^(i *
 * powers_exponent), i.e. gs_power_series of omega^powers_exponent, from the domain's power tables.  c_out NULL: C is not stored.
 * lcount = 0: l = c.  Replaces gs_zero_poly_inverses + gs_power_series + gs_vec_mul + gs_eval_polys_at_roots +
 * gs_sub_matrix_from_vectors + gs_div_by_domain_roots + two gs_combine_adjusted: nine passes and seven intermediate vectors / matrices.
 * GS_ERR_UNSUPPORTED beyond 4 roots / 4 coefficients per row (callers then use the separate entries). */
int gs_composition_tail(gs_ctx *ctx, uint64_t n, const gs_elt *omega, const void *q, const void *z_inv /* or NULL: */, uint64_t z_steps,
                        const gs_elt *x_last, const void *const *b_vecs_host, uint32_t bcount, const uint8_t *ipolys_host /* bcount x ilen */,
                        uint32_t ilen, const uint64_t *root_index_host /* bcount x max_roots */, const uint32_t *roots_per_row_host,
                        uint32_t max_roots, const uint8_t *b_coeffs_host, const uint8_t *b_adj_host /* or NULL */,
                        const void *const *l_vecs_host, uint32_t lcount, const uint8_t *l_coeffs_host, const uint8_t *l_adj_host /* or NULL */,
                        const void *powers /* n elements, or NULL: */, uint64_t powers_exponent, void *c_out /* or NULL */, void *l_out);
/* The same over ONE RANK'S COSET of the evaluation domain (a proof spread over several GPUs: point i is shift * omega^i, omega = the
 * n-th root w^ranks, shift = w^rank; q, the vectors and the outputs are the rank's strided shares): I_b is evaluated at the coset's
 * points, the divisors' roots are still given as positions in units of omega (domain points w^(ranks * position)), 1/Z and the
 * powers are those of gs_zero_poly_inverses_coset and of the strided share of the power series (x_i^powers_exponent, full exponent). */
int gs_composition_tail_coset(gs_ctx *ctx, uint64_t n, const gs_elt *omega, const gs_elt *shift, const void *q, const void *z_inv, uint64_t z_steps,
                              const gs_elt *x_last, const void *const *b_vecs_host, uint32_t bcount, const uint8_t *ipolys_host, uint32_t ilen,
                              const uint64_t *root_index_host, const uint32_t *roots_per_row_host, uint32_t max_roots,
                              const uint8_t *b_coeffs_host, const uint8_t *b_adj_host, const void *const *l_vecs_host, uint32_t lcount,
                              const uint8_t *l_coeffs_host, const uint8_t *l_adj_host, const void *powers, uint64_t powers_exponent,
                              void *c_out, void *l_out);
/* gs_air_constraints with the registers read IN PLACE from columns of a larger domain: register r at point j is
 * p[r * prow + j * pstride] (next row: point (j + shift) mod nc).  CompositionPolynomial.ts:76 evaluates the constraints over the
 * composition domain, whose points are every (N / nc)-th point of the evaluation domain the trace polynomials were just extended
 * over (lib/Stark.ts:109): with prow = N and pstride = N / nc the extension is read directly and the plucked copy (registers x nc
 * elements written and read again) is never made.  gs_air_constraints is prow = nc, pstride = 1.  (nc - 1) * pstride < prow. */
int gs_air_constraints_strided(gs_ctx *ctx, const uint32_t *code_host, uint32_t ninstr, const uint8_t *consts_host, uint32_t nconsts,
                               uint32_t vm_regs, uint32_t registers, uint32_t constraints, const void *p /* registers x prow */,
                               uint64_t prow, uint64_t pstride, uint64_t nc, uint64_t shift, const void *static_tables,
                               const uint64_t *static_lens_host, uint32_t nstatic, void *out /* constraints x nc */);

#ifdef __cplusplus
}
#endif
#endif /* GSTARK_H */
