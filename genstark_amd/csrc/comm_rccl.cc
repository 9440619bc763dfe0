// comm_rccl.cc — the communicator of include/gstark_comm.h over RCCL (xGMI inside a node): libgstark_rccl.so.
//
// One rank per process and GPU.  The collectives of a distributed proof (csrc/prover_dist.h) are issued on the HIP stream of the
// gs_ctx they concern (gs_stream), on device buffers the library owns: ncclAllGather for sub-roots, small layers, the remainder and
// the packed query answers; ncclAllToAll (RCCL's pairwise exchange; fallback: grouped ncclSend / ncclRecv) for the re-sharding of leaf
// digests and trace columns (every pair of ranks talks over its own xGMI link: no ring).  Nothing is staged through the host, and no call blocks it; a HIP event pair around every
// collective gives its device time (take_timings).  The unique id is made on rank 0 (gs_rccl_unique_id) and handed to
// gs_rccl_comm_create on every rank by the launcher.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>
#include <vector>

#include "../../include/gstark_comm.h"

namespace {

struct RcclState {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1, device = 0;
    void *(*stream_of)(gs_ctx *) = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed, spare;
    bool no_alltoall = false;     // fixed at creation (GSTARK_RCCL_GROUPED_ALLTOALL=1): grouped ncclSend / ncclRecv instead of ncclAllToAll
    // overlap (gs_comm::fork / join): between the two the collectives run on `side`, ordered behind `forked` (recorded on the context's
    // stream at fork) and ahead of `joined` (which the context's stream waits for at join)
    hipStream_t side = nullptr;
    hipEvent_t forked = nullptr, joined = nullptr;
    bool on_side = false;
    bool timings = true;          // an event pair around every collective (gs_rccl_comm_timings): each record costs the stream a few us
    char err[256] = {0};
};

bool begin(RcclState *s, hipStream_t st) {
    if (!s->timings) return true;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!s->spare.empty()) { ev = s->spare.back(); s->spare.pop_back(); }
    else {
        if (hipEventCreate(&ev.first) != hipSuccess) return false;
        if (hipEventCreate(&ev.second) != hipSuccess) { (void)hipEventDestroy(ev.first); return false; }
    }
    s->timed.push_back(ev);
    return hipEventRecord(ev.first, st) == hipSuccess;
}
bool end(RcclState *s, hipStream_t st) { return !s->timings || hipEventRecord(s->timed.back().second, st) == hipSuccess; }

int r_all_gather(void *self, gs_ctx *ctx, const void *send, void *recv, uint64_t bytes) {
    RcclState *s = (RcclState *)self;
    hipStream_t st = s->on_side ? s->side : (hipStream_t)s->stream_of(ctx);
    if (!begin(s, st)) return GS_ERR_DEVICE;
    if (ncclAllGather(send, recv, bytes, ncclUint8, s->comm, st) != ncclSuccess) return GS_ERR_DEVICE;
    return end(s, st) ? GS_OK : GS_ERR_DEVICE;
}
int r_all_to_all(void *self, gs_ctx *ctx, const void *send, void *recv, uint64_t bytes) {
    RcclState *s = (RcclState *)self;
    hipStream_t st = s->on_side ? s->side : (hipStream_t)s->stream_of(ctx);
    if (!begin(s, st)) return GS_ERR_DEVICE;
    // RCCL's own all-to-all (an extension over NCCL: piece j of rank i's buffer becomes piece i of rank j's — exactly this layout; it is what
    // torch.distributed's all_to_all_single runs on ROCm, i.e. the exercised path).  The grouped send / recv form below (which includes the
    // pair a rank forms with itself) is the other protocol; WHICH of the two a communicator speaks is fixed when it is created
    // (GSTARK_RCCL_GROUPED_ALLTOALL=1 in the launcher's environment, the same on every rank) — a failed call is an error, never a reason
    // to switch protocol in the middle of a run on a communicator that is already in an error state, with ranks that may disagree.
    if (!s->no_alltoall) {
        if (ncclAllToAll(send, recv, bytes, ncclUint8, s->comm, st) != ncclSuccess) return GS_ERR_DEVICE;
        return end(s, st) ? GS_OK : GS_ERR_DEVICE;
    }
    bool ok = ncclGroupStart() == ncclSuccess;
    for (int h = 0; h < s->size && ok; h++) {
        ok = ncclSend((const uint8_t *)send + (uint64_t)h * bytes, bytes, ncclUint8, h, s->comm, st) == ncclSuccess &&
             ncclRecv((uint8_t *)recv + (uint64_t)h * bytes, bytes, ncclUint8, h, s->comm, st) == ncclSuccess;
    }
    ok = (ncclGroupEnd() == ncclSuccess) && ok;
    if (!ok) return GS_ERR_DEVICE;
    return end(s, st) ? GS_OK : GS_ERR_DEVICE;
}
int r_fork(void *self, gs_ctx *ctx) {
    RcclState *s = (RcclState *)self;
    if (s->on_side) return GS_ERR_ARG;                  // forks do not nest
    if (!s->side) {
        if (hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess) return GS_ERR_DEVICE;
        if (hipEventCreateWithFlags(&s->forked, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->joined, hipEventDisableTiming) != hipSuccess) return GS_ERR_DEVICE;
    }
    if (hipEventRecord(s->forked, (hipStream_t)s->stream_of(ctx)) != hipSuccess || hipStreamWaitEvent(s->side, s->forked, 0) != hipSuccess) return GS_ERR_DEVICE;
    s->on_side = true;
    return GS_OK;
}
int r_join(void *self, gs_ctx *ctx) {
    RcclState *s = (RcclState *)self;
    if (!s->on_side) return GS_ERR_ARG;
    s->on_side = false;
    if (hipEventRecord(s->joined, s->side) != hipSuccess || hipStreamWaitEvent((hipStream_t)s->stream_of(ctx), s->joined, 0) != hipSuccess) return GS_ERR_DEVICE;
    return GS_OK;
}
uint32_t r_take_timings(void *self, double *ms_out, uint32_t cap) {
    RcclState *s = (RcclState *)self;
    uint32_t n = 0;
    for (auto &ev : s->timed) {
        float ms = -1.f;
        if (hipEventSynchronize(ev.second) != hipSuccess || hipEventElapsedTime(&ms, ev.first, ev.second) != hipSuccess) ms = -1.f;
        if (n < cap) ms_out[n] = ms;
        n++;
        s->spare.push_back(ev);
    }
    s->timed.clear();
    return n < cap ? n : cap;
}

}  // namespace

extern "C" {

int gs_rccl_unique_id(uint8_t out[128]) {
    static_assert(sizeof(ncclUniqueId) <= 128, "unique id larger than the ABI's 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return GS_ERR_DEVICE;
    memset(out, 0, 128);
    memcpy(out, &id, sizeof id);
    return GS_OK;
}

// abi_dl_handle: the dlopen handle of the ABI library whose contexts will use this communicator (gs_stream is taken from it)
int gs_rccl_comm_create(void *abi_dl_handle, const uint8_t unique_id[128], int rank, int size, int device, gs_comm *out, char *err, uint64_t errcap) {
    if (!abi_dl_handle || !unique_id || !out || size < 1 || rank < 0 || rank >= size) return GS_ERR_ARG;
    RcclState *s = new RcclState();
    { const char *g = getenv("GSTARK_RCCL_GROUPED_ALLTOALL"); s->no_alltoall = g && g[0] == '1'; }
    s->rank = rank; s->size = size; s->device = device;
    s->stream_of = (void *(*)(gs_ctx *))dlsym(abi_dl_handle, "gs_stream");
    auto bad = [&](const char *what, const char *detail) {
        if (err && errcap) snprintf(err, (size_t)errcap, "%s: %s", what, detail);
        delete s;
        return GS_ERR_DEVICE;
    };
    if (!s->stream_of) return bad("gs_stream", "not exported by the ABI library");
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return bad("hipSetDevice", hipGetErrorString(he));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclResult_t nr = ncclCommInitRank(&s->comm, size, id, rank);
    if (nr != ncclSuccess) return bad("ncclCommInitRank", ncclGetErrorString(nr));
    memset(out, 0, sizeof *out);
    out->self = s;
    out->rank = rank;
    out->size = size;
    out->all_gather = r_all_gather;
    out->all_to_all = r_all_to_all;
    out->take_timings = r_take_timings;
    out->name = "rccl";
    { const char *o = getenv("GSTARK_RCCL_NO_OVERLAP"); if (!(o && o[0] == '1')) { out->fork = r_fork; out->join = r_join; } }
    return GS_OK;
}

// on (default): every collective is bracketed by a HIP event pair and take_timings reports its device time; off: no events on the
// stream (an event record delays the kernel behind it by a few microseconds — a latency-bound proof issues ~10 collectives)
void gs_rccl_comm_timings(gs_comm *c, int on) {
    if (c && c->self) ((RcclState *)c->self)->timings = on != 0;
}

void gs_rccl_comm_destroy(gs_comm *c) {
    if (!c || !c->self) return;
    RcclState *s = (RcclState *)c->self;
    for (auto &ev : s->timed) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto &ev : s->spare) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (s->forked) (void)hipEventDestroy(s->forked);
    if (s->joined) (void)hipEventDestroy(s->joined);
    if (s->side) (void)hipStreamDestroy(s->side);
    if (s->comm) ncclCommDestroy(s->comm);
    delete s;
    c->self = nullptr;
}

}  // extern "C"
