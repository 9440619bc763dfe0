/* gstark_comm.h — the communicator the native driver uses when ONE proof is spread over several GPUs (gs_prover_prove_dist,
 * csrc/prover_dist.h).
 *
 * A communicator is a table of two collectives over DEVICE buffers of the bound ABI library.  They are enqueued on the stream of
 * the gs_ctx they are handed (gs_stream) and return at once: the driver never copies exchanged data through the host.  The product
 * implementation is RCCL over xGMI (csrc/comm_rccl.cc -> libgstark_rccl.so: ncclAllGather, grouped ncclSend / ncclRecv on a
 * communicator the library owns, one rank per process and GPU); the tests inject others through the same table (threads of one
 * process exchanging through gs_copy: ranks that share the box's single GPU, or the oracle's host memory).
 * The exchanged layouts (which strided share goes where) are decided by the driver, not here. */
#ifndef GSTARK_COMM_H
#define GSTARK_COMM_H

#include "gstark_prover.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gs_comm gs_comm;
struct gs_comm {
    void *self;
    int32_t rank, size;              /* size: a power of two */
    /* every rank contributes `bytes` bytes at `send`; `recv` (size * bytes) receives rank r's contribution at r * bytes */
    int (*all_gather)(void *self, gs_ctx *ctx, const void *send, void *recv, uint64_t bytes);
    /* piece h of `send` (bytes each) goes to rank h; piece h of `recv` is what rank h sent here */
    int (*all_to_all)(void *self, gs_ctx *ctx, const void *send, void *recv, uint64_t bytes);
    /* optional (may be NULL): device time of the collectives issued since the last call, in issue order; returns how many were
     * written (<= cap).  Blocks until they have completed. */
    uint32_t (*take_timings)(void *self, double *ms_out, uint32_t cap);
    const char *name;                /* "rccl", "threads", ... (reported by bench.py) */
    /* tuning: a FRI layer (after the first) with fewer values than this is all-gathered once and finished on every rank instead of
     * being folded on shares (two collectives per layer are not worth a latency-bound layer).  0 = the default (2^22 values). */
    uint64_t fri_gather_below;
    /* tuning: a statement with fewer evaluation-domain points PER RANK than this is not sharded at all — rank 0 proves it with the
     * single-device sequence and the serialized proof is delivered to every rank (two small all-gathers): a proof of a millisecond
     * is a chain of dependent launches that more devices cannot shorten, and every collective of the sharded form costs more than
     * the work it would spread.  0 = the default (2^20 points per rank); 1 = always shard (tests). */
    uint64_t solo_below;
    /* optional (both or neither; NULL = every collective stays in stream order): overlap of a collective with computation.
     * fork(): the collectives issued from now until join() run BEHIND everything ctx's stream holds at this moment, but are not ordered
     * with what the caller enqueues on it afterwards (RCCL: they go to the communicator's own stream, which first waits for an event
     * recorded here); join(): ctx's stream continues only after those collectives have completed.  Between the two the caller neither
     * reads what they write nor overwrites or frees what they read.  The driver uses it for the evaluation tree's leaf-digest
     * all-to-all, which runs while the constraint kernels — which need none of it — occupy the device. */
    int (*fork)(void *self, gs_ctx *ctx);
    int (*join)(void *self, gs_ctx *ctx);
};

/* One proof of `job` across the comm->size ranks (every rank calls this with the same job; SPMD).  Every rank receives the same
 * serialized proof — byte for byte what gs_prover_prove gives on one device.  Requirements: steps * extension_factor divisible by
 * 4 * size^2, extension_factor divisible by size.  Secret input registers (job->air.secret_traces: their extensions over the whole
 * evaluation domain, the same on every rank) are committed and combined share by share like the trace registers. */
int gs_prover_prove_dist(gs_ctx *ctx, const struct gs_prover_job *job, const gs_comm *comm, uint8_t *out, uint64_t cap, uint64_t *len,
                         char *err, uint64_t errcap);

int gs_prover_prove_dist_on(const gs_prover_binding *b, gs_ctx *ctx, const struct gs_prover_job *job, const gs_comm *comm, uint8_t *out, uint64_t cap,
                            uint64_t *len, char *err, uint64_t errcap);

/* The collectives the last gs_prover_prove_dist on the calling thread issued, in order. */
struct gs_prover_collective {
    char label[40];
    uint32_t kind;                   /* 0 = all_gather, 1 = all_to_all */
    uint64_t bytes;                  /* contributed by this rank (all_gather) / sent to every peer (all_to_all) */
    double ms;                       /* device time when the communicator measures it (take_timings), else -1 */
};
int gs_prover_last_collectives(struct gs_prover_collective *out, uint32_t cap, uint32_t *count);

#ifdef __cplusplus
}
#endif
#endif
