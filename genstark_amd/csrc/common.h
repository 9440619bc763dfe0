// common.h — context, error handling and launch helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include <array>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gstark.h"
#if defined(GS_SMALL_Q)
#include "gf_small.h"   // build flavour for a prime below 2^64 (same names, same 16-byte elements)
#elif defined(GS_WIDE_BITS)
#include "gf_wide.h"    // build flavour for the 256- / 224-bit primes (same names, 32-byte elements)
#else
#include "gf128.h"
#endif

#if defined(GF_RUNTIME_MODULUS)
// The runtime-modulus build: every translation unit owns a copy of the field constants in device constant memory (gf_wide.h) and
// registers a function that pushes the host copy into it; gs_ctx_create (ctx.hip) runs them for the context's device.
typedef int (*gs_rt_push_fn)(void);
void gs_rt_register(gs_rt_push_fn fn);
static int gs_rt_push_this_unit(void) {
    return hipMemcpyToSymbol(HIP_SYMBOL(gf_rt_device), &gf_rt_host(), sizeof(GfRuntime)) == hipSuccess ? 0 : -1;
}
namespace { struct GsRtUnit { GsRtUnit() { gs_rt_register(gs_rt_push_this_unit); } } gs_rt_unit; }
#endif

// bytes of one element in memory and on the ABI (gs_element_size()), and the same in 16-byte words
#define GS_ELT ((uint64_t)sizeof(fe))
#define GS_EW ((uint32_t)(sizeof(fe) / 16))
#ifndef GF_LIMBS
#define GF_LIMBS 4
GF_HD uint32_t fe_limb(const fe &a, int i) { return i == 0 ? a.w0 : (i == 1 ? a.w1 : (i == 2 ? a.w2 : a.w3)); }
GF_HD void fe_set_limb(fe &a, int i, uint32_t v) { if (i == 0) a.w0 = v; else if (i == 1) a.w1 = v; else if (i == 2) a.w2 = v; else a.w3 = v; }
#else
GF_HD uint32_t fe_limb(const fe &a, int i) { return a.w[i]; }
GF_HD void fe_set_limb(fe &a, int i, uint32_t v) { a.w[i] = v; }
#endif

struct NttPlan;  // ntt.hip

#define GS_HOST_TRACE_MAX_SEGMENTS 8   // default of gs_ctx::host_trace_segments

struct gs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char err[512] = {0};
    // size-bucketed cache of device blocks: gs_free parks a block here instead of hipFree (which would
    // synchronise the device); every consumer is ordered on `stream`, so immediate reuse is safe.
    // gs_alloc / gs_free / gs_cache_trim take `blocks_mutex`: a host-language finaliser may release a vector from another thread than
    // the one driving the context (everything else of a context stays single-threaded, include/gstark.h)
    std::mutex blocks_mutex;
    std::multimap<uint64_t, void *> free_blocks;
    std::map<void *, uint64_t> live_blocks;
    uint64_t cached_bytes = 0;
    // twiddle tables per (omega, n)
    std::map<std::string, NttPlan *> plans;
    // pinned host staging + small device scratch for scalars / index lists / pointer tables
    void *h_stage = nullptr;      // pinned + mapped: kernels may read/write it directly (zero-copy for tiny results)
    void *h_stage_dev = nullptr;  // the device-side address of h_stage
    void *d_stage = nullptr;
    uint64_t stage_bytes = 0;
    void *fri_x = nullptr;        // the evaluation point gs_fri_fold_seeded derives on the device
    void *fri_points = nullptr;   // gs_fri_layers: points between layers the caller did not ask for + the arrival counter of its multi-workgroup launches
    // host-side scalar tables that depend on the domain alone (1 / (g^j - 1), powers of omega by fixed exponents): a proof server asks
    // for the same ones with every proof, and an inversion is 4 us of host time on the critical path of a 0.5 ms proof (gs_memo)
    std::map<std::string, std::vector<fe>> scalar_memo;
    uint64_t jit_launches = 0;    // compiled-program launches so far (gs_air_jit_launches)
    // gs_traffic_enable / gs_traffic_read: per kernel name {launches, algorithmic bytes, work units} of the launches issued while on
    bool traffic_on = false;
    std::map<std::string, std::array<uint64_t, 3>> traffic;
    uint64_t host_trace_segments = GS_HOST_TRACE_MAX_SEGMENTS;   // traces of at most this many segments run on a host core (GSTARK_HOST_TRACE_SEGMENTS; air_vm.hip)
    int air_jit = 2;              // AIR programs: 0 interpreted, 1 compiled on first use (hiprtc), 2 auto = compiled when the code object already exists (gs_air_jit / GSTARK_AIR_JIT)
    // deferred read-backs (gs_defer_begin / gs_defer_end): gathers only record the device addresses of the 16-byte words they want;
    // gs_defer_end fetches all of them with ONE kernel and one synchronisation
    struct DeferredCopy { void *dst; uint64_t first_word, bytes; };
    bool defer = false;
    std::vector<uint64_t> defer_addrs;
    std::vector<DeferredCopy> defer_copies;
    // pinned buffer the host-side trace generators write into; it is uploaded asynchronously, `trace_done` marks the end
    // of the last upload so the next trace waits for THAT copy only, not for whatever else is queued on the stream
    void *h_trace = nullptr;
    uint64_t trace_bytes = 0;
    hipEvent_t trace_done = nullptr;
    bool trace_pending = false;
    // posted read-backs (gs_readback_post / gs_readback_wait): a ring of small slots in coherent mapped pinned memory, [RB_SLOTS] x 256
    // bytes of data followed by [RB_SLOTS] 64-bit flags.  The copying kernel writes the data, then (system-scope release) the slot's
    // flag = ticket + 1; the host spins on that flag — no event packet on the stream (an event record costs the NEXT kernel ~6 us)
    static constexpr uint32_t RB_SLOTS = 64, RB_SLOT_BYTES = 256;
    void *h_rb = nullptr, *h_rb_dev = nullptr;
    uint32_t rb_bytes[RB_SLOTS] = {};
    uint64_t rb_next = 0;
    // the landing area of deferred fetches (gs_defer_end): coherent mapped pinned memory, [fl_bytes of data][one 64-bit flag].  The
    // gathering kernel's last workgroup to finish (arrival counter d_fl_count) stores the flag = fl_seq with system-scope release and the
    // host spins on it: the Fiat-Shamir round trip of a proof (root -> coefficients) no longer waits for a stream synchronisation
    void *h_fl = nullptr, *h_fl_dev = nullptr, *d_fl_count = nullptr;
    uint64_t fl_bytes = 0, fl_seq = 0;
    // upload ring (gs_push_reserve / gs_push_commit): small host payloads (programs, coefficient tables, first rows) are assembled
    // in pinned memory and copied asynchronously — no synchronisation per upload; two halves, a half is reused only after the
    // copies issued from it have completed (one event per half)
    static constexpr uint64_t UP_HALF = 8ull << 20;
    void *h_up = nullptr;
    uint64_t up_off = 0;
    int up_half = 0;
    hipEvent_t up_done[2] = {};
    bool up_pending[2] = {false, false};
};

int gs_fail(gs_ctx *c, int code, const char *fmt, ...);

// one launch's entry in the traffic tally (see gs_traffic_enable in gstark.h); `name` may carry printf-style template arguments
static inline void gs_traffic(gs_ctx *c, uint64_t bytes, uint64_t units, const char *fmt, ...) __attribute__((format(printf, 4, 5)));
static inline void gs_traffic(gs_ctx *c, uint64_t bytes, uint64_t units, const char *fmt, ...) {
    if (!c->traffic_on) return;
    char name[64];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    auto &e = c->traffic[name];
    e[0] += 1; e[1] += bytes; e[2] += units;
}

// key: a tag + the bytes of whatever the table depends on
struct gs_memo_key {
    std::string k;
    explicit gs_memo_key(const char *tag) : k(tag) {}
    gs_memo_key &add(const void *p, size_t n) { k.append((const char *)p, n); return *this; }
    gs_memo_key &add(uint64_t v) { return add(&v, sizeof v); }
    gs_memo_key &add(const fe &v) { return add(&v, sizeof v); }
};
template <class Build>
static inline const std::vector<fe> &gs_memo(gs_ctx *c, const gs_memo_key &key, Build build) {
    auto it = c->scalar_memo.find(key.k);
    if (it == c->scalar_memo.end()) {
        if (c->scalar_memo.size() > 1024) c->scalar_memo.clear();
        it = c->scalar_memo.emplace(key.k, std::vector<fe>()).first;
        build(it->second);
    }
    return it->second;
}
// omega^e for an exponent that recurs from proof to proof
static inline fe gs_memo_pow(gs_ctx *c, const fe &w, uint64_t e) {
    return gs_memo(c, gs_memo_key("pow").add(w).add(e), [&](std::vector<fe> &t) { t.push_back(fe_pow_u64(w, e)); })[0];
}

#define GS_HIP(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return gs_fail((c), GS_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define GS_LAUNCH_CHECK(c) GS_HIP(c, hipGetLastError())

static inline bool gs_is_pow2(uint64_t x) { return x && !(x & (x - 1)); }
static inline int gs_log2(uint64_t x) { int l = 0; while ((1ull << l) < x) l++; return l; }

static inline fe fe_from_bytes(const uint8_t *b) { fe r; memcpy(&r, b, sizeof(fe)); return r; }
static inline void fe_to_bytes(uint8_t *b, const fe &v) { memcpy(b, &v, sizeof(fe)); }
static inline fe fe_from_u64(uint64_t v) { return fe_make((uint32_t)v, (uint32_t)(v >> 32), 0, 0); }

// staging helpers (ctx.hip)
int gs_stage_reserve(gs_ctx *c, uint64_t bytes);
int gs_defer_flush(gs_ctx *c);                                    // fetch and deliver every queued read-back
// `bytes` of pinned staging to fill (GS_ERR_UNSUPPORTED: too large for the ring, use a blocking copy), then one asynchronous copy
// of it to `dst`; the caller's own buffers are free again as soon as it has filled the staging
int gs_push_reserve(gs_ctx *c, uint64_t bytes, void **host);
int gs_push_commit(gs_ctx *c, void *dst, const void *host, uint64_t bytes);
int gs_push(gs_ctx *c, void *dst, const void *host_src, uint64_t bytes);     // both steps for one source; blocking copy when too large
// a slot of the posted-read-back ring for a kernel that delivers the bytes ITSELF (writes `bytes` to *slot_dev, fences, then stores
// `value` to *flag_dev at system scope — what k_post_words does for gs_readback_post); *ticket is what gs_readback_wait takes
int gs_readback_reserve(gs_ctx *c, uint32_t bytes, void **slot_dev, unsigned long long **flag_dev, unsigned long long *value, uint64_t *ticket);
int gs_trace_begin(gs_ctx *c, uint64_t bytes);   // h_trace has >= bytes and no upload of it is in flight
int gs_trace_end(gs_ctx *c);                     // call after the last hipMemcpyAsync out of h_trace
// temp device block from the cache (same lifetime rules as gs_alloc/gs_free)
int gs_tmp_alloc(gs_ctx *c, uint64_t bytes, void **p);
void gs_tmp_free(gs_ctx *c, void *p);

// grid sizing for streaming kernels: enough 256-thread blocks to fill 256 CUs x 8, grid-stride the rest
static inline unsigned gs_grid(uint64_t work_items, unsigned block = 256, unsigned cap = 256 * 8) {
    uint64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// air_jit.hip: GS_OK = the compiled kernel was launched, GS_ERR_UNSUPPORTED = interpret instead
int gs_jit_trace_segments(gs_ctx *c, const uint32_t *code, uint32_t ninstr, const uint32_t *icode, uint32_t init_ninstr, const uint8_t *consts_host,
                          uint32_t nconsts, uint32_t vm_regs, uint32_t registers, const uint64_t *soff, const uint64_t *slen, const uint8_t *statics_host,
                          uint32_t nstatic, const fe *dconst, const fe *dstat, const fe *drows, uint64_t segments, uint64_t seglen, fe *out);
int gs_jit_constraints(gs_ctx *c, const uint32_t *code, uint32_t ninstr, const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs,
                       uint32_t registers, const uint64_t *soff, const uint64_t *slen, const fe *dconst, const fe *p, uint64_t nc, uint64_t shift,
                       const fe *statics, fe *out, uint64_t prow, uint64_t pstride);

// NTT entry points implemented in ntt.hip and used by other units
void gs_plans_destroy(gs_ctx *c);
int gs_plan_pow_tables(gs_ctx *c, const fe &omega, uint64_t n, const fe **tw_lo, const fe **tw_hi, int *log_lo);
int gs_plan_inverse_table(gs_ctx *c, const fe &omega, uint64_t n, const fe **u);   // 1 / (omega^j - 1), cached with the plan
int gs_plan_inverse_table_shifted(gs_ctx *c, const fe &omega, uint64_t n, const fe &shift, const fe **u);   // 1 / (shift * omega^j - 1)
