// ctx.hip — context, device memory cache, host<->device copies, gather.  C ABI: include/gstark.h.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"
#include <chrono>
#include <sched.h>
#include <time.h>

int gs_fail(gs_ctx *c, int code, const char *fmt, ...) {
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#if defined(GF_RUNTIME_MODULUS)
static std::vector<gs_rt_push_fn> &gs_rt_units() { static std::vector<gs_rt_push_fn> v; return v; }
void gs_rt_register(gs_rt_push_fn fn) { gs_rt_units().push_back(fn); }
#endif

extern "C" {

int gs_abi_version(void) { return GS_ABI_VERSION; }
const char *gs_backend_name(void) { return "hip-gfx950"; }

int gs_element_size(void) { return (int)sizeof(fe); }

int gs_field_modulus(uint8_t *out_le) {
#ifdef GS_WIDE_BITS
    fe p;
    for (int i = 0; i < GF_LIMBS; i++) p.w[i] = gf_p_limb(i);
#else
    fe p = fe_make(GF_P0, GF_P1, GF_P2, GF_P3);
#endif
    memcpy(out_le, &p, sizeof(fe));
    return GS_OK;
}

// createPrimeField(modulus) of index.ts:14 for a modulus no fixed build knows: the runtime-modulus build takes it here, ONCE per process
// (the constants are process-wide: every context, every thread).  A fixed build accepts its own modulus and nothing else.
int gs_set_modulus(const uint8_t *modulus_le, uint32_t bytes) {
    if (!modulus_le || !bytes || bytes > sizeof(fe)) return GS_ERR_ARG;
    fe want = fe_zero();
    memcpy(&want, modulus_le, bytes);
#if defined(GF_RUNTIME_MODULUS)
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    const int rc = gf_rt_configure(want.w);
    return rc == 0 ? GS_OK : (rc == -2 ? GS_ERR_UNSUPPORTED : GS_ERR_ARG);        // -2: one modulus per process
#else
    uint8_t mine[sizeof(fe)];
    gs_field_modulus(mine);
    return memcmp(mine, &want, sizeof(fe)) ? GS_ERR_UNSUPPORTED : GS_OK;
#endif
}

int gs_ctx_create(int device, void *hip_stream, gs_ctx **out) {
    if (!out) return GS_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return GS_ERR_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GS_ERR_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return GS_ERR_DEVICE;  // kernels are built for gfx950 only
    if (hipSetDevice(device) != hipSuccess) return GS_ERR_DEVICE;
#if defined(GF_RUNTIME_MODULUS)
    if (!gf_rt_host().set) return GS_ERR_UNSUPPORTED;                  // gs_set_modulus first
    for (gs_rt_push_fn push : gs_rt_units()) if (push()) return GS_ERR_DEVICE;       // this device's copies of the field constants, unit by unit
#endif
    gs_ctx *c = new gs_ctx();
    c->device = device;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return GS_ERR_DEVICE; }
        c->own_stream = true;
    }
    const char *jit = getenv("GSTARK_AIR_JIT");     // 0: interpret, 1: compile on first use, unset / "auto": compiled when already built (air_jit.hip)
    c->air_jit = !jit || !jit[0] || jit[0] == 'a' ? 2 : (jit[0] != '0' ? 1 : 0);
#if defined(GF_RUNTIME_MODULUS)
    c->air_jit = 0;         // AIR programs are interpreted in this flavour (a compiled program would bake a modulus in: generic kernels only)
#endif
    c->host_trace_segments = GS_HOST_TRACE_MAX_SEGMENTS;
    if (const char *hs = getenv("GSTARK_HOST_TRACE_SEGMENTS")) c->host_trace_segments = strtoull(hs, nullptr, 10);
    *out = c;
    return GS_OK;
}

int gs_air_jit(gs_ctx *c, int enable) {
    if (!c) return GS_ERR_ARG;
    c->air_jit = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
#if defined(GF_RUNTIME_MODULUS)
    c->air_jit = 0;
#endif
    return GS_OK;
}
uint64_t gs_air_jit_launches(const gs_ctx *c) { return c ? c->jit_launches : 0; }
int gs_traffic_enable(gs_ctx *c, int on) {
    if (!c) return GS_ERR_ARG;
    c->traffic_on = on != 0;
    if (on) c->traffic.clear();
    return GS_OK;
}
int gs_traffic_read(gs_ctx *c, struct gs_traffic_entry *out, uint32_t cap, uint32_t *count) {
    if (!c || !count) return GS_ERR_ARG;
    *count = (uint32_t)c->traffic.size();
    uint32_t i = 0;
    for (auto &kv : c->traffic) {
        if (i >= cap || !out) break;
        memset(&out[i], 0, sizeof out[i]);
        snprintf(out[i].kernel, sizeof out[i].kernel, "%s", kv.first.c_str());
        out[i].launches = kv.second[0]; out[i].bytes = kv.second[1]; out[i].units = kv.second[2];
        i++;
    }
    return GS_OK;
}

void gs_ctx_destroy(gs_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    gs_plans_destroy(c);
    for (auto &kv : c->free_blocks) hipFree(kv.second);
    for (auto &kv : c->live_blocks) hipFree(kv.first);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->d_stage) hipFree(c->d_stage);
    if (c->h_trace) hipHostFree(c->h_trace);
    if (c->trace_done) hipEventDestroy(c->trace_done);
    if (c->h_up) hipHostFree(c->h_up);
    for (hipEvent_t e : c->up_done) if (e) hipEventDestroy(e);
    if (c->h_rb) hipHostFree(c->h_rb);
    if (c->h_fl) hipHostFree(c->h_fl);
    if (c->d_fl_count) hipFree(c->d_fl_count);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *gs_last_error(const gs_ctx *c) { return c ? c->err : "null context"; }
void *gs_stream(gs_ctx *c) { return c ? (void *)c->stream : nullptr; }

int gs_sync(gs_ctx *c) {
    if (!c) return GS_ERR_ARG;
    GS_HIP(c, hipStreamSynchronize(c->stream));
    return GS_OK;
}

static uint64_t round_block(uint64_t bytes) {
    if (bytes < 256) return 256;
    return (bytes + 255) & ~(uint64_t)255;
}

int gs_alloc(gs_ctx *c, uint64_t bytes, void **dptr) {
    if (!c || !dptr) return GS_ERR_ARG;
    uint64_t sz = round_block(bytes);
    std::lock_guard<std::mutex> lock(c->blocks_mutex);
    auto it = c->free_blocks.find(sz);
    void *p = nullptr;
    if (it != c->free_blocks.end()) {
        p = it->second;
        c->free_blocks.erase(it);
        c->cached_bytes -= sz;
    } else {
        GS_HIP(c, hipSetDevice(c->device));
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess) {
            // release the cache and retry once
            hipStreamSynchronize(c->stream);
            for (auto &kv : c->free_blocks) hipFree(kv.second);
            c->free_blocks.clear();
            c->cached_bytes = 0;
            e = hipMalloc(&p, sz);
            if (e != hipSuccess) return gs_fail(c, GS_ERR_OOM, "hipMalloc(%llu): %s", (unsigned long long)sz, hipGetErrorString(e));
        }
    }
    c->live_blocks[p] = sz;
    *dptr = p;
    return GS_OK;
}

int gs_free(gs_ctx *c, void *dptr) {
    if (!c) return GS_ERR_ARG;
    if (!dptr) return GS_OK;
    std::lock_guard<std::mutex> lock(c->blocks_mutex);
    auto it = c->live_blocks.find(dptr);
    if (it == c->live_blocks.end()) return gs_fail(c, GS_ERR_ARG, "gs_free: pointer was not allocated by gs_alloc");
    c->free_blocks.insert({it->second, dptr});
    c->cached_bytes += it->second;
    c->live_blocks.erase(it);
    return GS_OK;
}

int gs_cache_trim(gs_ctx *c) {
    if (!c) return GS_ERR_ARG;
    GS_HIP(c, hipStreamSynchronize(c->stream));
    std::lock_guard<std::mutex> lock(c->blocks_mutex);
    for (auto &kv : c->free_blocks) hipFree(kv.second);
    c->free_blocks.clear();
    c->cached_bytes = 0;
    return GS_OK;
}

int gs_upload(gs_ctx *c, void *dst, const void *host_src, uint64_t bytes) {
    if (!c || (!dst && bytes) || (!host_src && bytes)) return GS_ERR_ARG;
    if (!bytes) return GS_OK;
    return gs_push(c, dst, host_src, bytes);      // the caller may reuse host_src at once either way
}

int gs_download(gs_ctx *c, void *host_dst, const void *src, uint64_t bytes) {
    if (!c || (!host_dst && bytes) || (!src && bytes)) return GS_ERR_ARG;
    if (!bytes) return GS_OK;
    GS_HIP(c, hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    GS_HIP(c, hipStreamSynchronize(c->stream));
    return GS_OK;
}

int gs_copy(gs_ctx *c, void *dst, const void *src, uint64_t bytes) {
    if (!c || (!dst && bytes) || (!src && bytes)) return GS_ERR_ARG;
    if (!bytes) return GS_OK;
    GS_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return GS_OK;
}

}  // extern "C"

// out[t] = the 16-byte word at device address addr[t] (addresses and results in mapped pinned host memory)
__global__ void k_gather_words(const uint64_t *__restrict__ addr, uint64_t total, uint4 *__restrict__ out) {
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x)
        out[t] = *reinterpret_cast<const uint4 *>(addr[t]);
}

// the same gather into coherent mapped pinned host memory, announced to a polling host: every workgroup makes its words visible at
// system scope, then arrives at a device-scope counter; the last one to arrive resets the counter and stores the flag (release, system)
__global__ void k_gather_words_flag(const uint64_t *__restrict__ addr, uint64_t total, uint4 *__restrict__ out, unsigned int *__restrict__ arrived,
                                    unsigned long long *flag, unsigned long long value) {
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x)
        out[t] = *reinterpret_cast<const uint4 *>(addr[t]);
    __syncthreads();                     // (every wave's stores have left the CU; ONE thread fences for the workgroup — hash.hip: k_merkle_subtree)
    if (threadIdx.x == 0) {
        __threadfence_system();
        const unsigned int before = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1 == gridDim.x) {
            __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int gs_push_reserve(gs_ctx *c, uint64_t bytes, void **host) {
    const uint64_t H = gs_ctx::UP_HALF;
    bytes = (bytes + 255) & ~(uint64_t)255;
    if (bytes > H / 2) return GS_ERR_UNSUPPORTED;
    if (!c->h_up) {
        GS_HIP(c, hipHostMalloc(&c->h_up, 2 * H, hipHostMallocDefault));
        for (hipEvent_t &e : c->up_done) GS_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (c->up_off + bytes > H) {                          // this half is full: mark its copies, move to the other one
        GS_HIP(c, hipEventRecord(c->up_done[c->up_half], c->stream));
        c->up_pending[c->up_half] = true;
        c->up_half ^= 1;
        c->up_off = 0;
        if (c->up_pending[c->up_half]) {
            GS_HIP(c, hipEventSynchronize(c->up_done[c->up_half]));
            c->up_pending[c->up_half] = false;
        }
    }
    *host = (uint8_t *)c->h_up + (uint64_t)c->up_half * H + c->up_off;
    c->up_off += bytes;
    return GS_OK;
}

int gs_push_commit(gs_ctx *c, void *dst, const void *host, uint64_t bytes) {
    GS_HIP(c, hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, c->stream));
    return GS_OK;
}

int gs_push(gs_ctx *c, void *dst, const void *host_src, uint64_t bytes) {
    void *h = nullptr;
    int rc = gs_push_reserve(c, bytes, &h);
    if (rc == GS_OK) {
        memcpy(h, host_src, bytes);
        return gs_push_commit(c, dst, h, bytes);
    }
    if (rc != GS_ERR_UNSUPPORTED) return rc;
    // large and pageable: the runtime stages the copy; synchronise so the caller may reuse host_src at once
    GS_HIP(c, hipMemcpyAsync(dst, host_src, bytes, hipMemcpyHostToDevice, c->stream));
    GS_HIP(c, hipStreamSynchronize(c->stream));
    return GS_OK;
}

// spin on a flag in coherent pinned memory until it holds `value` (posted read-backs and deferred fetches alike): a short spin, then
// yields, naps only after 2 ms, and a look at the queue every 200 us so that a lost copy is an error, not a hang
static int gs_poll_flag(gs_ctx *c, const volatile uint64_t *flag, uint64_t value, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    auto checked = t0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != value) {
        const auto now = std::chrono::steady_clock::now();
        if (now - t0 < std::chrono::microseconds(50)) {
            __builtin_ia32_pause();                            // data that is about to land
        } else if (now - t0 < std::chrono::milliseconds(2)) {
            sched_yield();                                     // the device is still working towards it: let another lane's host thread run if
                                                               // one is runnable (returns at once otherwise — a nanosleep here costs >= 50 us of
                                                               // timer slack on top of the wait: measured, +45 us on a 0.5 ms proof)
        } else {
            struct timespec ts = {0, 20000};
            nanosleep(&ts, nullptr);
        }
        if (now - checked > std::chrono::microseconds(200)) {
            checked = now;
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) return gs_fail(c, GS_ERR_DEVICE, "%s: %s", what, hipGetErrorString(q));
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != value) return gs_fail(c, GS_ERR_DEVICE, "%s: the queue drained without delivering the data", what);
        }
    }
    return GS_OK;
}

// `words` 16-byte words from src into a slot of mapped pinned host memory, then the slot's flag (one wave; the release fence orders
// the flag behind the data for the polling host)
__global__ void k_post_words(const uint4 *__restrict__ src, uint32_t words, uint4 *__restrict__ out, unsigned long long *flag, unsigned long long value) {
    if (threadIdx.x < words) out[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int gs_readback_reserve(gs_ctx *c, uint32_t bytes, void **slot_dev, unsigned long long **flag_dev, unsigned long long *value, uint64_t *ticket) {
    if (!bytes || bytes > gs_ctx::RB_SLOT_BYTES || (bytes & 15)) return gs_fail(c, GS_ERR_ARG, "readback_post: 16..%u bytes, a multiple of 16", gs_ctx::RB_SLOT_BYTES);
    const size_t data_bytes = (size_t)gs_ctx::RB_SLOTS * gs_ctx::RB_SLOT_BYTES;
    if (!c->h_rb) {
        GS_HIP(c, hipHostMalloc(&c->h_rb, data_bytes + gs_ctx::RB_SLOTS * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->h_rb, 0, data_bytes + gs_ctx::RB_SLOTS * sizeof(uint64_t));
        GS_HIP(c, hipHostGetDevicePointer(&c->h_rb_dev, c->h_rb, 0));
    }
    // a slot is reused 64 posts later; its older copy precedes this one on the stream, and flags are ticket numbers: no wait here
    const uint32_t slot = (uint32_t)(c->rb_next % gs_ctx::RB_SLOTS);
    *slot_dev = (uint8_t *)c->h_rb_dev + (size_t)slot * gs_ctx::RB_SLOT_BYTES;
    *flag_dev = (unsigned long long *)((uint8_t *)c->h_rb_dev + data_bytes) + slot;
    *value = (unsigned long long)(c->rb_next + 1);
    c->rb_bytes[slot] = bytes;
    *ticket = c->rb_next++;
    return GS_OK;
}

extern "C" int gs_readback_post(gs_ctx *c, const void *src, uint32_t bytes, uint64_t *ticket) {
    if (!c || !src || !ticket) return GS_ERR_ARG;
    void *slot;
    unsigned long long *flag, value;
    int rc = gs_readback_reserve(c, bytes, &slot, &flag, &value, ticket);
    if (rc) return rc;
    hipLaunchKernelGGL(k_post_words, dim3(1), dim3(64), 0, c->stream, (const uint4 *)src, bytes / 16, (uint4 *)slot, flag, value);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

extern "C" int gs_readback_wait(gs_ctx *c, uint64_t ticket, void *host_dst) {
    if (!c || !host_dst) return GS_ERR_ARG;
    if (ticket >= c->rb_next || c->rb_next - ticket > gs_ctx::RB_SLOTS) return gs_fail(c, GS_ERR_ARG, "readback_wait: ticket %llu is not outstanding", (unsigned long long)ticket);
    const uint32_t slot = (uint32_t)(ticket % gs_ctx::RB_SLOTS);
    const size_t data_bytes = (size_t)gs_ctx::RB_SLOTS * gs_ctx::RB_SLOT_BYTES;
    const volatile uint64_t *flag = (const volatile uint64_t *)((const uint8_t *)c->h_rb + data_bytes) + slot;
    const int rc = gs_poll_flag(c, flag, ticket + 1, "readback_wait");
    if (rc) return rc;
    memcpy(host_dst, (const uint8_t *)c->h_rb + (size_t)slot * gs_ctx::RB_SLOT_BYTES, c->rb_bytes[slot]);
    return GS_OK;
}

int gs_defer_flush(gs_ctx *c) {
    const uint64_t total = c->defer_addrs.size();
    if (!total) { c->defer_copies.clear(); return GS_OK; }
    const bool was = c->defer;
    c->defer = false;                                     // the staging buffer is ours now
    const uint64_t addr_bytes = (total * 8 + 255) & ~(uint64_t)255;
    int rc = gs_stage_reserve(c, addr_bytes + total * 16);
    c->defer = was;
    if (rc) return rc;
    memcpy(c->h_stage, c->defer_addrs.data(), total * 8);
    // the words land in coherent pinned memory followed by a flag the host polls (no stream synchronisation: its wake-up costs more than
    // the whole gather when the fetch is a root and two cells)
    if (total * 16 > c->fl_bytes) {
        uint64_t nb = 1 << 16;
        while (nb < total * 16) nb <<= 1;
        GS_HIP(c, hipStreamSynchronize(c->stream));
        if (c->h_fl) hipHostFree(c->h_fl);
        c->h_fl = c->h_fl_dev = nullptr;
        c->fl_bytes = 0;
        GS_HIP(c, hipHostMalloc(&c->h_fl, nb + 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset((uint8_t *)c->h_fl + nb, 0, 64);
        GS_HIP(c, hipHostGetDevicePointer(&c->h_fl_dev, c->h_fl, 0));
        c->fl_bytes = nb;
        c->fl_seq = 0;
    }
    if (!c->d_fl_count) {
        GS_HIP(c, hipMalloc(&c->d_fl_count, 256));
        GS_HIP(c, hipMemsetAsync(c->d_fl_count, 0, 256, c->stream));
    }
    const uint64_t seq = ++c->fl_seq;
    hipLaunchKernelGGL(k_gather_words_flag, dim3(gs_grid(total)), dim3(256), 0, c->stream, (const uint64_t *)c->h_stage_dev, total, (uint4 *)c->h_fl_dev,
                       (unsigned int *)c->d_fl_count, (unsigned long long *)((uint8_t *)c->h_fl_dev + c->fl_bytes), (unsigned long long)seq);
    GS_LAUNCH_CHECK(c);
    rc = gs_poll_flag(c, (const volatile uint64_t *)((const uint8_t *)c->h_fl + c->fl_bytes), seq, "defer_end");
    if (rc) return rc;
    const uint8_t *words = (const uint8_t *)c->h_fl;
    for (auto &d : c->defer_copies) memcpy(d.dst, words + d.first_word * 16, d.bytes);
    c->defer_addrs.clear();
    c->defer_copies.clear();
    return GS_OK;
}

int gs_stage_reserve(gs_ctx *c, uint64_t bytes) {
    if (c->defer && !c->defer_addrs.empty()) {            // another user of the staging buffer inside a deferral window
        int rc = gs_defer_flush(c);
        if (rc) return rc;
    }
    if (bytes <= c->stage_bytes) return GS_OK;
    uint64_t nb = 1 << 16;
    while (nb < bytes) nb <<= 1;
    GS_HIP(c, hipStreamSynchronize(c->stream));
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->d_stage) hipFree(c->d_stage);
    c->h_stage = c->d_stage = c->h_stage_dev = nullptr;
    c->stage_bytes = 0;
    GS_HIP(c, hipHostMalloc(&c->h_stage, nb, hipHostMallocMapped));
    GS_HIP(c, hipHostGetDevicePointer(&c->h_stage_dev, c->h_stage, 0));
    GS_HIP(c, hipMalloc(&c->d_stage, nb));
    c->stage_bytes = nb;
    return GS_OK;
}

int gs_trace_begin(gs_ctx *c, uint64_t bytes) {
    if (c->trace_pending) {
        GS_HIP(c, hipEventSynchronize(c->trace_done));
        c->trace_pending = false;
    }
    if (!c->trace_done) GS_HIP(c, hipEventCreateWithFlags(&c->trace_done, hipEventDisableTiming));
    if (bytes <= c->trace_bytes) return GS_OK;
    uint64_t nb = 1 << 16;
    while (nb < bytes) nb <<= 1;
    if (c->h_trace) hipHostFree(c->h_trace);
    c->h_trace = nullptr;
    c->trace_bytes = 0;
    GS_HIP(c, hipHostMalloc(&c->h_trace, nb, hipHostMallocDefault));
    c->trace_bytes = nb;
    return GS_OK;
}

int gs_trace_end(gs_ctx *c) {
    GS_HIP(c, hipEventRecord(c->trace_done, c->stream));
    c->trace_pending = true;
    return GS_OK;
}

int gs_tmp_alloc(gs_ctx *c, uint64_t bytes, void **p) { return gs_alloc(c, bytes, p); }
void gs_tmp_free(gs_ctx *c, void *p) { gs_free(c, p); }

// out[i] = src[idx[i]] for records of rec16*16 bytes (rec_bytes is a multiple of 16 on every call site;
// other sizes take the byte path)
__global__ void k_gather16(const uint4 *__restrict__ src, const uint64_t *__restrict__ idx, uint64_t count, uint32_t rec16,
                           uint4 *__restrict__ out) {
    uint64_t total = count * rec16;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = t / rec16, o = t % rec16;
        out[t] = src[idx[r] * rec16 + o];
    }
}
__global__ void k_gather_bytes(const uint8_t *__restrict__ src, const uint64_t *__restrict__ idx, uint64_t count, uint64_t rec,
                               uint8_t *__restrict__ out) {
    uint64_t total = count * rec;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = t / rec, o = t % rec;
        out[t] = src[idx[r] * rec + o];
    }
}

extern "C" int gs_gather(gs_ctx *c, const void *src, uint64_t rec_bytes, const uint64_t *idx_host, uint64_t count,
                         void *host_out) {
    if (!c || !src || !rec_bytes || (!idx_host && count) || (!host_out && count)) return GS_ERR_ARG;
    if (!count) return GS_OK;
    uint64_t idx_bytes = (count * 8 + 255) & ~(uint64_t)255, data_bytes = count * rec_bytes;
    if (c->defer && rec_bytes % 16 == 0 && ((uintptr_t)src % 16) == 0) {
        // inside a deferral window: only note which words are wanted; gs_defer_end fetches them all at once
        c->defer_copies.push_back({host_out, (uint64_t)c->defer_addrs.size(), data_bytes});
        for (uint64_t i = 0; i < count; i++)
            for (uint64_t w = 0; w < rec_bytes; w += 16) c->defer_addrs.push_back((uint64_t)(uintptr_t)src + idx_host[i] * rec_bytes + w);
        return GS_OK;
    }
    int rc = gs_stage_reserve(c, idx_bytes + data_bytes);
    if (rc) return rc;
    // zero-copy: the kernel reads the index list from, and writes the (tiny) result into, mapped pinned host memory:
    // one launch + one stream synchronisation per call, no staging copies
    memcpy(c->h_stage, idx_host, count * 8);
    const uint64_t *d_idx = (const uint64_t *)c->h_stage_dev;
    uint8_t *d_out = (uint8_t *)c->h_stage_dev + idx_bytes;
    if (rec_bytes % 16 == 0 && ((uintptr_t)src % 16) == 0) {
        uint32_t rec16 = (uint32_t)(rec_bytes / 16);
        hipLaunchKernelGGL(k_gather16, dim3(gs_grid(count * rec16)), dim3(256), 0, c->stream, (const uint4 *)src, d_idx, count, rec16,
                           (uint4 *)d_out);
    } else {
        hipLaunchKernelGGL(k_gather_bytes, dim3(gs_grid(data_bytes)), dim3(256), 0, c->stream, (const uint8_t *)src, d_idx, count,
                           rec_bytes, d_out);
    }
    GS_LAUNCH_CHECK(c);
    uint8_t *h_out = (uint8_t *)c->h_stage + idx_bytes;
    GS_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(host_out, h_out, data_bytes);
    return GS_OK;
}

// dst[t] = the record word at (r, c) of a rows x cols matrix of rec16-word records, written transposed
__global__ void k_transpose_records(const uint4 *__restrict__ src, uint64_t rows, uint64_t cols, uint32_t rec16, uint4 *__restrict__ dst) {
    const uint64_t total = rows * cols * rec16;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o = t % rec16, e = t / rec16;        // destination record e = c * rows + r: consecutive lanes write consecutive words
        const uint64_t r = e % rows, cc = e / rows;
        dst[t] = src[(r * cols + cc) * rec16 + o];
    }
}
extern "C" int gs_gather_words(gs_ctx *c, const void *addrs, uint64_t count, void *dst) {
    if (!c || (!addrs && count) || (!dst && count)) return GS_ERR_ARG;
    if (!count) return GS_OK;
    if (((uintptr_t)addrs & 7) || ((uintptr_t)dst & 15)) return gs_fail(c, GS_ERR_ARG, "gather_words: misaligned buffer");
    hipLaunchKernelGGL(k_gather_words, dim3(gs_grid(count)), dim3(256), 0, c->stream, (const uint64_t *)addrs, count, (uint4 *)dst);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}
extern "C" int gs_transpose_records(gs_ctx *c, const void *src, uint64_t rows, uint64_t cols, uint64_t rec_bytes, void *dst) {
    if (!c || !src || !dst) return GS_ERR_ARG;
    if (!rec_bytes || rec_bytes % 16) return gs_fail(c, GS_ERR_ARG, "transpose_records: record size must be a multiple of 16");
    if (!rows || !cols) return GS_OK;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return gs_fail(c, GS_ERR_ARG, "transpose_records: misaligned buffer");
    gs_traffic(c, 2 * rows * cols * rec_bytes, rows * cols, "k_transpose_records");
    hipLaunchKernelGGL(k_transpose_records, dim3(gs_grid(rows * cols * (rec_bytes / 16))), dim3(256), 0, c->stream, (const uint4 *)src, rows, cols,
                       (uint32_t)(rec_bytes / 16), (uint4 *)dst);
    GS_LAUNCH_CHECK(c);
    return GS_OK;
}

extern "C" int gs_defer_begin(gs_ctx *c) {
    if (!c) return GS_ERR_ARG;
    if (c->defer) return gs_fail(c, GS_ERR_ARG, "defer_begin: already deferring");
    c->defer = true;
    c->defer_addrs.clear();
    c->defer_copies.clear();
    return GS_OK;
}

extern "C" int gs_defer_end(gs_ctx *c) {
    if (!c) return GS_ERR_ARG;
    if (!c->defer) return gs_fail(c, GS_ERR_ARG, "defer_end: not deferring");
    int rc = gs_defer_flush(c);
    c->defer = false;
    return rc;
}
