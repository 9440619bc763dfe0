/* comm_threads.c — TEST DOUBLE of the communicator of include/gstark_comm.h: the ranks are threads of one process.
 *
 * Test infrastructure (tests/ only; the product communicator is csrc/comm_rccl.cc).  Each rank drives its own gs_ctx of whatever ABI
 * library the test loaded — the oracle's (host memory) in the CPU tier, libgstark_hip.so with several contexts on the box's single
 * GPU in the -m gpu tier.  A collective is: drain my stream (gs_sync), meet at a barrier, copy the peers' pieces with gs_copy on my
 * own stream, drain, meet again.  Same layouts as RCCL's ncclAllGather and a grouped ncclSend/ncclRecv exchange. */
#define _POSIX_C_SOURCE 200809L
#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gstark_comm.h"

#define MAX_RANKS 64

typedef int (*copy_fn)(gs_ctx *, void *, const void *, uint64_t);
typedef int (*sync_fn)(gs_ctx *);

struct group {
    int size;
    pthread_barrier_t bar;
    const void *send[MAX_RANKS];
    copy_fn copy;
    sync_fn sync;
    volatile int failed;
    /* measuring mode (GSTARK_COMM_TAKE_TURNS=1; tools/dist_only.py): between two collectives only ONE rank at a time has work on the
     * device — rank r starts its stretch when rank r - 1 has drained its own.  With the ranks sharing one GPU every kernel then runs
     * alone, so a kernel trace shows each rank's work at its uncontended duration and the wall time is the plain sum of the ranks'
     * stretches.  Same bytes, same collectives. */
    int take_turns;
    int turn;
    pthread_mutex_t mu;
    pthread_cond_t cv;
};
#define MAX_DEFERRED 8
struct rank_state {
    struct group *grp;
    int rank;
    /* gs_comm::fork / join: a collective issued between the two is only NOTED, and carried out at join — the latest moment the contract
     * allows.  A driver that reads a forked collective's output before join, or frees its buffers, therefore fails the byte comparisons
     * of the tests (an eager double would hide exactly that mistake). */
    int forked, ndeferred;
    struct { const void *send; void *recv; uint64_t bytes; int all_to_all; } deferred[MAX_DEFERRED];
};

static int meet(struct group *g) {
    pthread_barrier_wait(&g->bar);
    return g->failed ? GS_ERR_DEVICE : GS_OK;
}

static void wait_turn(struct group *g, int rank) {
    if (!g->take_turns) return;
    pthread_mutex_lock(&g->mu);
    while (g->turn != rank && !g->failed) pthread_cond_wait(&g->cv, &g->mu);
    pthread_mutex_unlock(&g->mu);
}
static void pass_turn(struct group *g, int rank) {
    if (!g->take_turns) return;
    pthread_mutex_lock(&g->mu);
    g->turn = rank + 1;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
}

static int exchange(struct rank_state *st, gs_ctx *ctx, const void *send, void *recv, uint64_t bytes, int all_to_all) {
    struct group *g = st->grp;
    int rc = g->sync(ctx);
    if (rc) g->failed = 1;
    pass_turn(g, st->rank);                      /* my stretch of work is done: the next rank may start its own */
    g->send[st->rank] = send;
    if ((rc = meet(g))) return rc;
    if (g->take_turns && st->rank == 0) { pthread_mutex_lock(&g->mu); g->turn = 0; pthread_mutex_unlock(&g->mu); }    /* everyone has passed; nobody waits before the second meeting */
    wait_turn(g, st->rank);                      /* (the copies of a collective take turns as well) */
    for (int h = 0; h < g->size && !rc; h++) {
        const uint8_t *src = (const uint8_t *)g->send[h] + (all_to_all ? (uint64_t)st->rank * bytes : 0);
        rc = g->copy(ctx, (uint8_t *)recv + (uint64_t)h * bytes, src, bytes);
    }
    if (!rc) rc = g->sync(ctx);
    if (rc) { g->failed = 1; if (g->take_turns) { pthread_mutex_lock(&g->mu); pthread_cond_broadcast(&g->cv); pthread_mutex_unlock(&g->mu); } }
    pass_turn(g, st->rank);
    int rc2 = meet(g);
    if (g->take_turns && st->rank == 0) { pthread_mutex_lock(&g->mu); g->turn = 0; pthread_cond_broadcast(&g->cv); pthread_mutex_unlock(&g->mu); }
    if (!rc && !rc2) {
        if (g->take_turns && st->rank != 0) {    /* rank 0 has reset the counter when it reads 0 again after everybody's last pass */
            pthread_mutex_lock(&g->mu);
            while (g->turn > st->rank && !g->failed) pthread_cond_wait(&g->cv, &g->mu);
            pthread_mutex_unlock(&g->mu);
        }
        wait_turn(g, st->rank);
    }
    return rc ? rc : rc2;
}
/* measuring mode: the stretch before a proof's first collective and after its last one take turns too (tools/dist_only.py calls
 * these around every proof; no-ops otherwise) */
int gs_threads_comm_begin(const gs_comm *c) {
    if (!c || !c->self) return GS_ERR_ARG;
    struct rank_state *st = (struct rank_state *)c->self;
    wait_turn(st->grp, st->rank);
    return GS_OK;
}
int gs_threads_comm_end(const gs_comm *c, gs_ctx *ctx) {
    if (!c || !c->self) return GS_ERR_ARG;
    struct rank_state *st = (struct rank_state *)c->self;
    struct group *g = st->grp;
    if (!g->take_turns) return GS_OK;
    int rc = g->sync(ctx);
    pass_turn(g, st->rank);
    meet(g);
    if (st->rank == 0) { pthread_mutex_lock(&g->mu); g->turn = 0; pthread_cond_broadcast(&g->cv); pthread_mutex_unlock(&g->mu); }
    meet(g);
    return rc;
}
static int note_or_exchange(struct rank_state *st, gs_ctx *ctx, const void *send, void *recv, uint64_t bytes, int all_to_all) {
    if (!st->forked) return exchange(st, ctx, send, recv, bytes, all_to_all);
    if (st->ndeferred == MAX_DEFERRED) return GS_ERR_UNSUPPORTED;
    st->deferred[st->ndeferred].send = send;
    st->deferred[st->ndeferred].recv = recv;
    st->deferred[st->ndeferred].bytes = bytes;
    st->deferred[st->ndeferred].all_to_all = all_to_all;
    st->ndeferred++;
    return GS_OK;
}
static int t_all_gather(void *self, gs_ctx *ctx, const void *send, void *recv, uint64_t bytes) {
    return note_or_exchange((struct rank_state *)self, ctx, send, recv, bytes, 0);
}
static int t_all_to_all(void *self, gs_ctx *ctx, const void *send, void *recv, uint64_t bytes) {
    return note_or_exchange((struct rank_state *)self, ctx, send, recv, bytes, 1);
}
static int t_fork(void *self, gs_ctx *ctx) {
    struct rank_state *st = (struct rank_state *)self;
    (void)ctx;
    if (st->forked) return GS_ERR_ARG;           /* forks do not nest */
    st->forked = 1;
    st->ndeferred = 0;
    return GS_OK;
}
static int t_join(void *self, gs_ctx *ctx) {
    struct rank_state *st = (struct rank_state *)self;
    if (!st->forked) return GS_ERR_ARG;
    st->forked = 0;
    int rc = GS_OK;
    for (int i = 0; i < st->ndeferred && !rc; i++)
        rc = exchange(st, ctx, st->deferred[i].send, st->deferred[i].recv, st->deferred[i].bytes, st->deferred[i].all_to_all);
    st->ndeferred = 0;
    return rc;
}

/* `size` communicators (one per rank thread) over the ABI library behind `abi_dl_handle`; out[r] is rank r's. */
int gs_threads_comm_create(int size, void *abi_dl_handle, gs_comm *out) {
    if (size < 1 || size > MAX_RANKS || !abi_dl_handle || !out) return GS_ERR_ARG;
    struct group *g = (struct group *)calloc(1, sizeof *g);
    if (!g) return GS_ERR_OOM;
    g->size = size;
    g->copy = (copy_fn)dlsym(abi_dl_handle, "gs_copy");
    g->sync = (sync_fn)dlsym(abi_dl_handle, "gs_sync");
    if (!g->copy || !g->sync || pthread_barrier_init(&g->bar, NULL, (unsigned)size)) { free(g); return GS_ERR_UNSUPPORTED; }
    const char *tt = getenv("GSTARK_COMM_TAKE_TURNS");
    g->take_turns = tt && tt[0] == '1';
    pthread_mutex_init(&g->mu, NULL);
    pthread_cond_init(&g->cv, NULL);
    for (int r = 0; r < size; r++) {
        struct rank_state *st = (struct rank_state *)calloc(1, sizeof *st);
        if (!st) return GS_ERR_OOM;
        st->grp = g;
        st->rank = r;
        memset(&out[r], 0, sizeof out[r]);
        out[r].self = st;
        out[r].rank = r;
        out[r].size = size;
        out[r].all_gather = t_all_gather;
        out[r].all_to_all = t_all_to_all;
        out[r].take_timings = NULL;
        out[r].name = "threads";
        const char *eager = getenv("GSTARK_COMM_THREADS_NO_FORK");      /* =1: no fork / join in the table (the driver's in-order path) */
        if (!(eager && eager[0] == '1')) { out[r].fork = t_fork; out[r].join = t_join; }
    }
    return GS_OK;
}
