/* gstark_prover.h — C interface of the native prove() driver (genstark_amd/csrc/prover.cc -> libgstark_prover.so).
 *
 * One call = one Stark.prove() of the reference (lib/Stark.ts:81-163) + Serializer.serializeProof (lib/Serializer.ts:35-79):
 * the driver issues the entry points of gstark.h in the reference's order from native host code and returns the serialized
 * proof.  It contains no device code and no arithmetic kernels of its own: gs_prover_bind() takes the dlopen() handle of the
 * implementation of gstark.h the caller loaded (libgstark_hip.so) and resolves every gs_* symbol it needs from it.  A node
 * addon would expose gs_prover_prove as one N-API function next to the member-by-member surface of INTEGRATION.md. */
#ifndef GSTARK_PROVER_H
#define GSTARK_PROVER_H

#include "gstark.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Scalars of a job (assertion values, the MiMC seed, the root of unity) are field elements of gs_element_size() bytes, little-endian,
 * in fields of GS_PROVER_ELT_MAX bytes (the bytes above the element size are ignored): one job layout for every field flavour. */
#define GS_PROVER_ELT_MAX 32

typedef struct gs_assertion gs_assertion;
struct gs_assertion {          /* lib/Stark.ts:356-375: register `reg` holds `value` at `step` */
    uint64_t step;
    uint32_t reg;
    uint8_t value[GS_PROVER_ELT_MAX];
};

/* An input register of an air-assembly component — `(input secret|public vector|scalar (childof n)|(peerof n) (steps n) (shift n))`,
 * assembly/lib128.aa:81-150 — as Stark.prove / Stark.verify need it: initProvingContext(inputs) and initVerificationContext(proof.iShapes,
 * publicInputs) (lib/Stark.ts:90,176) size the trace from the SHAPES of the inputs and lay the values out as trace-length columns
 * (one value held for `steps` steps, or for as long as its children take; the whole column rotated by `shift`). */
struct gs_input_register {
    int32_t parent;             /* (childof n): an EARLIER input register whose every value this one refines into a list; -1 = none */
    int32_t peer;               /* (peerof n): an earlier input register of the same shape; -1 = none */
    uint32_t steps;             /* (steps n): trace steps one value is held; 0 = not stated (follows from its children or its peer) */
    int32_t shift;              /* (shift n): the column rotated right by n steps (negative: left) */
    uint32_t secret;            /* 1 = the prover's alone: its values reach the verifier inside the proof's leaves */
};
/* Where static register s of an AIR with input registers takes its values from (verifier side: the prover's job carries the columns). */
#define GS_STATIC_CYCLE 0       /* cyclic values: the next entry of static_values / static_periods */
#define GS_STATIC_INPUT 1       /* input register `index` (a public one: from public_inputs; a secret one: from the leaves) */
#define GS_STATIC_MASK 2        /* 1 on the first step of every value of input register `index`, 0 elsewhere (`(mask (input n))`) */
struct gs_static_source {
    uint32_t kind, index;
};

/* What the AIR module contributes (lib/Stark.ts:35-58: `air`): counts, degrees and the two device routines. */
struct gs_prover_air {
    uint32_t kind;               /* 0 = MiMC (gs_mimc_trace / gs_mimc_constraints), 1 = register-machine programs (gs_air_*) */
    uint32_t registers;          /* trace registers */
    uint32_t nconstraints;
    const uint32_t *degrees;     /* constraint degrees */
    /* kind 0 */
    uint8_t seed[GS_PROVER_ELT_MAX];
    const uint8_t *round_constants;   /* host, nrc elements */
    uint32_t nrc;
    const void *k_table;         /* device: the cyclic register over the composition domain */
    uint64_t k_len;
    /* kind 1 */
    const uint32_t *t_code; uint32_t t_ninstr;          /* transition program */
    const uint32_t *i_code; uint32_t i_ninstr;          /* init program (segments only, may be 0) */
    const uint32_t *e_code; uint32_t e_ninstr;          /* constraint evaluator */
    const uint8_t *consts; uint32_t nconsts; uint32_t vm_regs;
    const uint8_t *static_values; const uint32_t *static_periods; uint32_t nstatic;   /* host, for the trace */
    const void *static_tables; const uint64_t *static_lens;                           /* device + host lens, for the evaluator */
    const uint8_t *first_rows; uint64_t segments; uint64_t segment_len;               /* segments = 0: one serial trace */
    /* secret registers (lib/Stark.ts:113): their low-degree extensions over the evaluation domain, prepared by the caller; */
    /* static_values / static_tables then hold the public registers followed by the secret ones */
    const void *const *secret_traces; uint32_t nsecret;
    /* input registers (ninputs = 0: an AIR without them — nothing below is read and the proof carries no shapes) */
    const struct gs_input_register *inputs; uint32_t ninputs;
    /* input_shapes: per input register its rank followed by that many dimensions.  prove: written into the proof (iShapes,
     * lib/Stark.ts:161, lib/Serializer.ts:70-78) after the validation the verifier applies — they must lay out a trace of job.steps
     * steps.  verify: ignored (the proof brings them). */
    const uint32_t *input_shapes;
    /* verify: one entry per static register (nstatic of them, the secret ones last); NULL = all cyclic.  static_values / static_periods
     * then list the GS_STATIC_CYCLE registers only, in order. */
    const struct gs_static_source *static_sources;
    /* verify: the values of the PUBLIC input registers in declaration order (Stark.verify's publicInputs, lib/Stark.ts:167), each
     * flattened row-major over its nesting, elements of gs_element_size() bytes back to back; public_input_counts[k] = how many
     * values public register k holds (must equal the product of its shape in the proof) */
    const uint8_t *public_inputs; const uint64_t *public_input_counts; uint32_t npublic_inputs;
};

struct gs_prover_job {
    uint64_t steps;              /* trace length; verify with input registers: 0 = whatever the proof's shapes lay out (otherwise they must agree) */
    uint32_t extension_factor, exe_query_count, fri_query_count;
    int32_t hash_alg;
    uint8_t root_of_unity[GS_PROVER_ELT_MAX];   /* primitive (steps*extension_factor)-th root: galois getRootOfUnity, computed by the caller */
    const gs_assertion *assertions;
    uint32_t nassertions;
    /* 0: root_of_unity has order steps * extension_factor.  k > 0: it has order 2^k and the driver squares it down to the order of
     * the evaluation domain (getRootOfUnity(n) = getRootOfUnity(2n)^2 for the generator galois picks) — what a verifier passes when
     * the trace length comes with the proof */
    uint32_t root_of_unity_log2;
    struct gs_prover_air air;
};


/* What the last gs_prover_prove() on the calling thread did (bench.py reports it; nothing in the proof depends on it). */
#define GS_PROVER_MAX_PHASES 16
struct gs_prover_stats {
    uint64_t ntt_points;         /* points of the transforms launched as NTT passes: sum of rows * n over gs_interpolate_roots / gs_eval_polys_at_roots calls */
    uint64_t ntt_transforms;     /* ... and how many transforms that was */
    uint64_t horner_points;      /* points of the calls the library serves with its Horner kernel instead (fewer than 256 points, or at most 8 coefficients) */
    uint32_t nphases;
    double total_ms;             /* host wall-clock of the whole call */
    double phase_ms[GS_PROVER_MAX_PHASES];       /* host wall-clock between phase boundaries; no device synchronisation is added */
    char phase_label[GS_PROVER_MAX_PHASES][48];
    /* the reference's own phase log (lib/Stark.ts:92-152, lib/utils/Logger.ts; README.md:62-73), filled only by a proof that ran with
     * gs_prover_sync_phases(1): the device is synchronised at each of the reference's log points, so every entry is the wall-clock
     * of that phase alone (the default, asynchronous proof overlaps them: phase_ms above) */
    uint32_t nreadme;
    double readme_ms[GS_PROVER_MAX_PHASES];
    char readme_label[GS_PROVER_MAX_PHASES][160];
};
/* per calling thread: the next proofs synchronise the device at the reference's log points and fill readme_ms (a measuring mode:
 * a proof takes a little longer; its bytes are the same) */
void gs_prover_sync_phases(int on);
/* per calling thread: the next proofs of generic AIRs run the tail of the composition polynomial and the linear combination as the
 * member-by-member sequence of entries (gs_zero_poly_inverses, gs_power_series, gs_vec_mul, gs_eval_polys_at_roots,
 * gs_sub_matrix_from_vectors, gs_div_by_domain_roots, gs_combine_adjusted twice) instead of the one pass of gs_composition_tail —
 * a checking mode (the tests compare the two): same bytes, a few passes more */
void gs_prover_member_sequence(int on);

/* Resolves the gs_* entry points from `dl_handle` (the handle dlopen() returned for the ABI library).  GS_ERR_UNSUPPORTED if one
 * is missing, or if the library computes in another field / element size than this build of the driver (one driver library per
 * field flavour: libgstark_prover.so, libgstark_prover_q64.so, ... — csrc/build.sh).  This is the process-wide DEFAULT binding, used
 * by the entry points that take no binding. */
int gs_prover_bind(void *dl_handle);
/* A binding of its own (several ABI libraries of one flavour in a process: the HIP library and a test double): every entry point
 * below has an `_on` form that takes it.  Bindings are immutable and may be shared between threads. */
typedef struct gs_prover_binding gs_prover_binding;
int gs_prover_open(void *dl_handle, gs_prover_binding **out);
void gs_prover_close(gs_prover_binding *b);
int gs_prover_element_size(void);      /* of this build of the driver */
/* Layout version of the structs of this header and of gstark_comm.h (gs_prover_job, gs_prover_air, gs_comm): a caller checks it once
 * after loading the driver.  2: gs_prover_air gained the input-register fields and gs_prover_job root_of_unity_log2 (round 6), gs_comm
 * solo_below (round 5).  Callers ZERO-INITIALISE every struct with sizeof of the header they were built against: a field added later
 * reads as 0 = "not used" only if the version matches. */
#define GS_PROVER_ABI_VERSION 2
int gs_prover_abi_version(void);
/* The serialized proof into out[0..cap); *len receives its size (GS_ERR_ARG with *len set when cap is too small).  On failure
 * err[0..errcap) holds the reference's message where there is one ("Assertion at step ... conflicts with execution trace"). */
int gs_prover_prove(gs_ctx *ctx, const struct gs_prover_job *job, uint8_t *out, uint64_t cap, uint64_t *len, char *err, uint64_t errcap);
int gs_prover_prove_on(const gs_prover_binding *b, gs_ctx *ctx, const struct gs_prover_job *job, uint8_t *out, uint64_t cap, uint64_t *len, char *err,
                       uint64_t errcap);
/* Stark.verify() (lib/Stark.ts:167-248; LowDegreeProver.ts:70-172) of a serialized proof, native and CPU-only like the reference's:
 * GS_OK = the proof is valid for the statement; otherwise the reference's message in err ("Verification of evaluation Merkle proof
 * failed", "Degree 4 polynomial didn't evaluate to column value at depth 2", ...).  The job is the one prove() takes; the verifier
 * reads of it the sizes, query counts, hash, root of unity, assertions and, of the AIR, kind / registers / nsecret / degrees and
 * (kind 0) round_constants or (kind 1) e_code, consts, vm_regs, static_values / static_periods / nstatic (public registers first;
 * the trailing nsecret entries are the secret ones, whose values come with the proof).  No gs_ctx: nothing touches a device; the
 * binding supplies the library's host-side helpers (index generator, small interpolation).  An AIR with input registers (air.ninputs,
 * air.static_sources, air.public_inputs) is sized from the shapes the proof carries, as the reference does (lib/Stark.ts:176); for any
 * other AIR the job states the trace length and a proof that carries shapes is malformed.  The number of FRI layers and the remainder
 * length must be the ones the domain implies (fold while more than 256 values are left) — anything else is a malformed proof. */
int gs_prover_verify(const struct gs_prover_job *job, const uint8_t *proof, uint64_t len, char *err, uint64_t errcap);
int gs_prover_verify_on(const gs_prover_binding *b, const struct gs_prover_job *job, const uint8_t *proof, uint64_t len, char *err, uint64_t errcap);
int gs_prover_last_stats(struct gs_prover_stats *out);
/* LowDegreeProver.verifyRemainder (LowDegreeProver.ts:223-252) on its own, for tests: `len` values on the powers of root_of_unity
 * (order len); 1 = the values at the positions that are not multiples of extension_factor lie on a polynomial of degree
 * < max_degree_plus1, 0 = they do not, < 0 = error.  method 0: the reference's procedure (interpolate the first max_degree_plus1
 * positions, evaluate at the rest); method 1: the coefficient form the driver uses — same verdict on every input. */
int gs_prover_remainder_check(const uint8_t *values, uint64_t len, uint32_t extension_factor, uint64_t max_degree_plus1, const uint8_t *root_of_unity,
                              int method);
int gs_prover_remainder_check_on(const gs_prover_binding *b, const uint8_t *values, uint64_t len, uint32_t extension_factor, uint64_t max_degree_plus1,
                                 const uint8_t *root_of_unity, int method);

/* The layout rules of input registers on their own, for tests: the trace length the shapes lay out (`shapes`: per register its rank and
 * dimensions, as gs_prover_air.input_shapes), or GS_ERR_ARG with the loader's message in err — the same function prove() and verify() apply
 * (prover.cc: input_layout; genstark_amd/airassembly.py: _Layout is what it restates). */
int gs_prover_input_layout(const struct gs_input_register *inputs, uint32_t ninputs, const uint32_t *shapes, uint64_t *length, char *err, uint64_t errcap);

#ifdef __cplusplus
}
#endif
#endif
