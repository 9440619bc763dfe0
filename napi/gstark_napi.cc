// napi/gstark_napi.cc — thin N-API shim over the C ABI of include/gstark.h.
//
// This is the binding a genSTARK maintainer adds in place of the wasm loaders of @guildofweavers/galois and
// @guildofweavers/merkle: it dlopen()s libgstark_hip.so and forwards calls 1:1.  No arithmetic happens here.
// JS-side conventions (used by js/galois.js and js/merkle.js, which rebuild the FiniteField / Hash / MerkleTree
// objects lib/Stark.ts consumes):
//   * a context is an External; a device pointer is a BigInt; sizes/counts are Numbers (or BigInts);
//   * host byte arguments are Buffers (inputs are read, outputs are written in place);
//   * arrays of device pointers / of 64-bit indexes are JS arrays of BigInt / Number.
// Every gs_* function returns gs_status; a non-zero status is thrown as an Error carrying gs_last_error().
//
// build (no node-gyp needed):  g++ -O2 -shared -fPIC -I/usr/include/node napi/gstark_napi.cc -o napi/gstark_napi.node -ldl
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <numeric>
#include <string>
#include <vector>

#include "../include/gstark.h"
#include "../include/gstark_prover.h"

namespace {

void *g_lib = nullptr;

#define NAPI_OK(env, call)                                                \
    do {                                                                  \
        if ((call) != napi_ok) {                                          \
            napi_throw_error((env), nullptr, "N-API call failed: " #call); \
            return nullptr;                                               \
        }                                                                 \
    } while (0)

// One descriptor per exported function: argument kinds after the optional leading ctx.
//   c ctx | p device pointer | u uint64 | i int/uint32 | b host bytes in | o host bytes out |
//   a array of device pointers | x array of uint64 | w array of uint32
struct FnDesc {
    const char *name;
    const char *sig;
};
const FnDesc kFns[] = {
    {"gs_sync", "c"},
    {"gs_free", "cp"},
    {"gs_cache_trim", "c"},
    {"gs_upload", "cpbu"},
    {"gs_download", "copu"},
    {"gs_copy", "cppu"},
    {"gs_gather", "cpuxuo"},
    {"gs_power_series", "cbup"},
    {"gs_vec_add", "cppup"},
    {"gs_vec_sub", "cppup"},
    {"gs_vec_mul", "cppup"},
    {"gs_vec_add_scalar", "cpbup"},
    {"gs_vec_sub_scalar", "cpbup"},
    {"gs_vec_mul_scalar", "cpbup"},
    {"gs_vec_inv", "cpup"},
    {"gs_vec_div", "cppup"},
    {"gs_vec_exp", "cpbup"},
    {"gs_combine_many", "cabiup"},
    {"gs_combine", "cppuo"},
    {"gs_air_jit", "ci"},
    {"gs_air_jit_launches", "c"},
    {"gs_defer_begin", "c"},
    {"gs_defer_end", "c"},
    {"gs_pluck", "cpuuup"},
    {"gs_zero_poly_inverses", "cbuubp"},
    {"gs_div_by_domain_roots", "cpiubxwip"},
    {"gs_transpose_vector", "cpuiup"},
    {"gs_transpose_matrix", "cpuup"},
    {"gs_sub_matrix_from_vectors", "capiup"},
    {"gs_eval_polys_at_roots", "cpiubup"},
    {"gs_interpolate_roots", "cpibup"},
    {"gs_eval_poly_at", "cpubo"},
    {"gs_interpolate_quartic_batch", "cppup"},
    {"gs_interpolate_quartic_domain", "cbuupup"},
    {"gs_eval_quartic_batch", "cpubp"},
    {"gs_fri_fold", "cbuupubp"},
    {"gs_fri_fold_seeded", "cbuupupp"},
    {"gs_hash_digest", "cibuo"},
    {"gs_hash_merge_rows", "ciaiup"},
    {"gs_hash_digest_values", "cipuup"},
    {"gs_merkle_build", "cipup"},
    {"gs_mimc_trace", "cbbiup"},
    {"gs_mimc_constraints", "cpuupup"},
    {"gs_mimc_composition", "cpuubpubuubxibp"},
    {"gs_air_trace", "cwibiiibwibup"},
    {"gs_air_trace_segments", "cwiwibiiibwibuup"},
    {"gs_air_constraints", "cwibiiiipuupxip"},
    {"gs_air_jit_check", "iwiwibiiixiou"},
    {"gs_set_modulus", "bi"},
    {"gs_small_interpolate", "bbio"},
    {"gs_pseudorandom_indexes", "biiuio"},
    {"gs_small_eval_poly", "bibio"},
};

// napi_get_buffer_info on a value that is not a Buffer ABORTS the process in node 12 (an assertion inside node::Buffer::Data, found by
// tests/addon_validation.js): ask first
bool buffer_info(napi_env env, napi_value v, void **data, size_t *len) {
    bool is = false;
    if (napi_is_buffer(env, v, &is) != napi_ok || !is) return false;
    return napi_get_buffer_info(env, v, data, len) == napi_ok;
}

bool get_u64(napi_env env, napi_value v, uint64_t *out) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return false;
    if (t == napi_bigint) {
        bool lossless;
        return napi_get_value_bigint_uint64(env, v, out, &lossless) == napi_ok;
    }
    if (t == napi_number) {
        double d;
        if (napi_get_value_double(env, v, &d) != napi_ok || d < 0) return false;
        *out = (uint64_t)d;
        return true;
    }
    return false;
}

typedef int (*fn16)(uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t,
                    uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t);

// call(name, ...args): generic forwarder.  All ABI parameters are integers or pointers, which the x86-64 SysV
// calling convention passes in 8-byte slots; narrower parameters read the low bytes of their slot.
napi_value Call(napi_env env, napi_callback_info info) {
    size_t argc = 20;
    napi_value argv[20];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    if (argc < 1) { napi_throw_type_error(env, nullptr, "call(name, ...args)"); return nullptr; }
    if (!g_lib) { napi_throw_error(env, nullptr, "call load(path) first"); return nullptr; }
    char name[64];
    size_t len;
    if (napi_get_value_string_utf8(env, argv[0], name, sizeof name, &len) != napi_ok) { napi_throw_type_error(env, nullptr, "call(name, ...args): the name is a string"); return nullptr; }
    const FnDesc *d = nullptr;
    for (const FnDesc &f : kFns)
        if (!strcmp(f.name, name)) d = &f;
    if (!d) { napi_throw_error(env, nullptr, (std::string("unknown gstark function ") + name).c_str()); return nullptr; }
    const size_t n = strlen(d->sig);
    if (argc - 1 != n) { napi_throw_type_error(env, nullptr, (std::string(name) + ": wrong number of arguments").c_str()); return nullptr; }
    void *sym = dlsym(g_lib, name);
    if (!sym) { napi_throw_error(env, nullptr, (std::string("symbol not found: ") + name).c_str()); return nullptr; }
    uintptr_t a[16] = {0};
    std::vector<std::vector<uint64_t>> arrays;
    std::vector<std::vector<uint32_t>> arrays32;
    arrays.reserve(4);
    arrays32.reserve(4);
    gs_ctx *ctx = nullptr;
    for (size_t i = 0; i < n; i++) {
        napi_value v = argv[i + 1];
        switch (d->sig[i]) {
            case 'c': {
                void *p;
                if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_type_error(env, nullptr, (std::string(name) + ": expected a context").c_str()); return nullptr; }
                ctx = (gs_ctx *)p;
                a[i] = (uintptr_t)p;
                break;
            }
            case 'p': case 'u': case 'i': {
                uint64_t x;
                if (!get_u64(env, v, &x)) { napi_throw_type_error(env, nullptr, (std::string(name) + ": expected a number/BigInt").c_str()); return nullptr; }
                a[i] = (uintptr_t)x;
                break;
            }
            case 'b': case 'o': {
                void *data;
                size_t blen;
                if (!buffer_info(env, v, &data, &blen)) { napi_throw_type_error(env, nullptr, (std::string(name) + ": expected a Buffer").c_str()); return nullptr; }
                a[i] = (uintptr_t)data;
                break;
            }
            case 'a': case 'x': {
                uint32_t alen;
                NAPI_OK(env, napi_get_array_length(env, v, &alen));
                arrays.emplace_back(alen);
                for (uint32_t k = 0; k < alen; k++) {
                    napi_value e;
                    NAPI_OK(env, napi_get_element(env, v, k, &e));
                    if (!get_u64(env, e, &arrays.back()[k])) { napi_throw_type_error(env, nullptr, (std::string(name) + ": bad array element").c_str()); return nullptr; }
                }
                a[i] = (uintptr_t)arrays.back().data();
                break;
            }
            case 'w': {
                uint32_t alen;
                NAPI_OK(env, napi_get_array_length(env, v, &alen));
                arrays32.emplace_back(alen);
                for (uint32_t k = 0; k < alen; k++) {
                    napi_value e;
                    uint64_t x;
                    NAPI_OK(env, napi_get_element(env, v, k, &e));
                    if (!get_u64(env, e, &x)) { napi_throw_type_error(env, nullptr, (std::string(name) + ": bad array element").c_str()); return nullptr; }
                    arrays32.back()[k] = (uint32_t)x;
                }
                a[i] = (uintptr_t)arrays32.back().data();
                break;
            }
            default:
                napi_throw_error(env, nullptr, "bad descriptor");
                return nullptr;
        }
    }
    int rc = ((fn16)sym)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
    if (rc != GS_OK) {
        typedef const char *(*errfn)(const gs_ctx *);
        errfn le = (errfn)dlsym(g_lib, "gs_last_error");
        std::string msg = std::string(name) + " failed (" + std::to_string(rc) + ")";
        if (ctx && le) msg += std::string(": ") + le(ctx);
        napi_throw_error(env, nullptr, msg.c_str());
        return nullptr;
    }
    napi_value undef;
    napi_get_undefined(env, &undef);
    return undef;
}

// load(path): dlopen libgstark_hip.so; returns the backend name
napi_value Load(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    char path[1024];
    size_t len;
    if (argc < 1 || napi_get_value_string_utf8(env, argv[0], path, sizeof path, &len) != napi_ok) { napi_throw_type_error(env, nullptr, "load(path)"); return nullptr; }
    void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { napi_throw_error(env, nullptr, (std::string("cannot load ") + path + ": " + dlerror() + " (there is no CPU fallback)").c_str()); return nullptr; }
    typedef const char *(*namefn)(void);
    namefn nf = (namefn)dlsym(lib, "gs_backend_name");
    if (!nf) { napi_throw_error(env, nullptr, "not a gstark library"); return nullptr; }
    g_lib = lib;
    napi_value out;
    NAPI_OK(env, napi_create_string_utf8(env, nf(), NAPI_AUTO_LENGTH, &out));
    return out;
}

// fieldInfo(): { elementSize, modulus: Buffer } of the loaded library (gs_element_size / gs_field_modulus)
napi_value FieldInfo(napi_env env, napi_callback_info info) {
    if (!g_lib) { napi_throw_error(env, nullptr, "no library loaded"); return nullptr; }
    typedef int (*sizefn)(void);
    typedef int (*modfn)(uint8_t *);
    sizefn sf = (sizefn)dlsym(g_lib, "gs_element_size");
    modfn mf = (modfn)dlsym(g_lib, "gs_field_modulus");
    if (!sf || !mf) { napi_throw_error(env, nullptr, "not a gstark library"); return nullptr; }
    const int es = sf();
    uint8_t mod[64] = {0};
    if (es <= 0 || es > 64 || mf(mod) != GS_OK) { napi_throw_error(env, nullptr, "gs_field_modulus failed"); return nullptr; }
    napi_value out, v;
    NAPI_OK(env, napi_create_object(env, &out));
    NAPI_OK(env, napi_create_uint32(env, (uint32_t)es, &v));
    NAPI_OK(env, napi_set_named_property(env, out, "elementSize", v));
    void *copy;
    NAPI_OK(env, napi_create_buffer_copy(env, (size_t)es, mod, &copy, &v));
    NAPI_OK(env, napi_set_named_property(env, out, "modulus", v));
    return out;
}

napi_value CtxCreate(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    int32_t device = 0;
    if (argc >= 1) napi_get_value_int32(env, argv[0], &device);
    if (!g_lib) { napi_throw_error(env, nullptr, "call load(path) first"); return nullptr; }
    typedef int (*createfn)(int, void *, gs_ctx **);
    createfn cf = (createfn)dlsym(g_lib, "gs_ctx_create");
    gs_ctx *ctx = nullptr;
    int rc = cf(device, nullptr, &ctx);
    if (rc != GS_OK || !ctx) {
        napi_throw_error(env, nullptr, ("gs_ctx_create failed (" + std::to_string(rc) + "): no gfx950 device; there is no CPU fallback").c_str());
        return nullptr;
    }
    napi_value out;
    NAPI_OK(env, napi_create_external(env, ctx, nullptr, nullptr, &out));
    return out;
}

napi_value CtxDestroy(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    void *p;
    if (!g_lib) { napi_throw_error(env, nullptr, "call load(path) first"); return nullptr; }
    if (argc < 1 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p) { napi_throw_type_error(env, nullptr, "ctxDestroy(ctx)"); return nullptr; }
    typedef void (*dfn)(gs_ctx *);
    ((dfn)dlsym(g_lib, "gs_ctx_destroy"))((gs_ctx *)p);
    napi_value undef;
    napi_get_undefined(env, &undef);
    return undef;
}

// alloc(ctx, bytes) -> BigInt device pointer
napi_value Alloc(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    void *p;
    if (!g_lib) { napi_throw_error(env, nullptr, "call load(path) first"); return nullptr; }
    uint64_t bytes;
    if (argc < 2 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p || !get_u64(env, argv[1], &bytes)) { napi_throw_type_error(env, nullptr, "alloc(ctx, bytes)"); return nullptr; }
    typedef int (*afn)(gs_ctx *, uint64_t, void **);
    void *d = nullptr;
    int rc = ((afn)dlsym(g_lib, "gs_alloc"))((gs_ctx *)p, bytes, &d);
    if (rc != GS_OK) { napi_throw_error(env, nullptr, "gs_alloc failed"); return nullptr; }
    napi_value out;
    NAPI_OK(env, napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)d, &out));
    return out;
}

// merkleProveBatch(ctx, leavesPtr, nodesPtr, n, indexes[]) -> { values: Buffer, colLens: number[], nodes: Buffer }
napi_value MerkleProveBatch(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    void *p;
    if (!g_lib) { napi_throw_error(env, nullptr, "call load(path) first"); return nullptr; }
    uint64_t leaves, nodes, n;
    if (argc < 5 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p || !get_u64(env, argv[1], &leaves) || !get_u64(env, argv[2], &nodes) ||
        !get_u64(env, argv[3], &n) || n < 2 || (n & (n - 1)) || n > (1ull << 40)) { napi_throw_type_error(env, nullptr, "merkleProveBatch: bad arguments"); return nullptr; }
    uint32_t count;
    if (napi_get_array_length(env, argv[4], &count) != napi_ok || count > (1u << 20)) { napi_throw_type_error(env, nullptr, "merkleProveBatch: bad index list"); return nullptr; }
    std::vector<uint64_t> idx(count);
    for (uint32_t k = 0; k < count; k++) {
        napi_value e;
        NAPI_OK(env, napi_get_element(env, argv[4], k, &e));
        if (!get_u64(env, e, &idx[k]) || idx[k] >= n) { napi_throw_type_error(env, nullptr, "merkleProveBatch: bad index"); return nullptr; }
    }
    int depth = 0;
    while ((1ull << depth) < n) depth++;
    const uint64_t cap = (uint64_t)count * (depth ? depth : 1);
    std::vector<uint8_t> values((size_t)count * 32), nd((size_t)cap * 32);
    std::vector<uint32_t> lens(count ? count : 1);
    uint32_t ncols = 0;
    typedef int (*pfn)(gs_ctx *, const void *, const void *, uint64_t, const uint64_t *, uint32_t, uint8_t *, uint32_t *, uint32_t *, uint8_t *, uint64_t);
    int rc = ((pfn)dlsym(g_lib, "gs_merkle_prove_batch"))((gs_ctx *)p, (const void *)(uintptr_t)leaves, (const void *)(uintptr_t)nodes, n, idx.data(),
                                                          count, values.data(), &ncols, lens.data(), nd.data(), cap);
    if (rc != GS_OK) {
        typedef const char *(*errfn)(const gs_ctx *);
        napi_throw_error(env, nullptr, ((errfn)dlsym(g_lib, "gs_last_error"))((gs_ctx *)p));
        return nullptr;
    }
    uint64_t total = 0;
    for (uint32_t i = 0; i < ncols; i++) total += lens[i];
    napi_value out, v, nb, arr;
    NAPI_OK(env, napi_create_object(env, &out));
    NAPI_OK(env, napi_create_buffer_copy(env, values.size(), values.data(), nullptr, &v));
    NAPI_OK(env, napi_create_buffer_copy(env, (size_t)total * 32, nd.data(), nullptr, &nb));
    NAPI_OK(env, napi_create_array_with_length(env, ncols, &arr));
    for (uint32_t i = 0; i < ncols; i++) {
        napi_value e;
        NAPI_OK(env, napi_create_uint32(env, lens[i], &e));
        NAPI_OK(env, napi_set_element(env, arr, i, e));
    }
    NAPI_OK(env, napi_set_named_property(env, out, "values", v));
    NAPI_OK(env, napi_set_named_property(env, out, "nodes", nb));
    NAPI_OK(env, napi_set_named_property(env, out, "colLens", arr));
    return out;
}

// proveMimcSerialized(ctx, proverLibPath, job) -> Buffer: ONE call = Stark.prove() + Serializer.serializeProof() of the MiMC AIR through
// the native driver (include/gstark_prover.h; the driver is bound to the ABI library load() opened).
//   job = { steps, extensionFactor, exeQueryCount, friQueryCount, hashAlg, rootOfUnity: Buffer(es), seed: Buffer(es),
//           roundConstants: Buffer(es*n), kTable: BigInt (device pointer), kLen, assertions: [{step, register, value: Buffer(es)}] }   (es = gs_element_size() of the loaded library)
// The driver library is the build for the loaded ABI library's field (js/prover.js picks libgstark_prover*.so by modulus); the addon
// holds ONE binding of it to that ABI library (gs_prover_open: nothing process-wide is written).  Scalars of a job are gs_element_size()
// bytes each.
void *g_prover = nullptr;
gs_prover_binding *g_binding = nullptr;
size_t g_es = 16;
typedef int (*prove_on_fn)(const gs_prover_binding *, gs_ctx *, const gs_prover_job *, uint8_t *, uint64_t, uint64_t *, char *, uint64_t);
std::string g_prover_path;          // the driver build g_binding belongs to ...
void *g_prover_for_lib = nullptr;   // ... and the ABI library it is bound to: load() may have replaced g_lib by another field's since
bool open_driver(napi_env env, napi_value path_value) {
    char path[1024];
    size_t len;
    if (napi_get_value_string_utf8(env, path_value, path, sizeof path, &len) != napi_ok) { napi_throw_type_error(env, nullptr, "driver library path expected"); return false; }
    if (g_prover && g_prover_for_lib == g_lib && g_prover_path == path) return true;
    if (g_prover) {      // a binding of another driver flavour, or to a library that is no longer the loaded one: never reused
        typedef void (*closefn)(gs_prover_binding *);
        closefn cf = (closefn)dlsym(g_prover, "gs_prover_close");
        if (cf && g_binding) cf(g_binding);
        dlclose(g_prover);
        g_prover = nullptr; g_binding = nullptr; g_prover_for_lib = nullptr; g_prover_path.clear();
    }
    void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { napi_throw_error(env, nullptr, (std::string("cannot load ") + path + ": " + dlerror()).c_str()); return false; }
    typedef int (*openfn)(void *, gs_prover_binding **);
    typedef int (*sizefn)();
    openfn of = (openfn)dlsym(lib, "gs_prover_open");
    sizefn sf = (sizefn)dlsym(g_lib, "gs_element_size");
    if (!of || !sf || of(g_lib, &g_binding) != GS_OK) {
        napi_throw_error(env, nullptr, "gs_prover_open failed: the driver library is not the build for the loaded library's field");
        dlclose(lib);
        return false;
    }
    g_es = (size_t)sf();
    g_prover = lib;
    g_prover_path = path;
    g_prover_for_lib = g_lib;
    return true;
}
typedef int (*verify_on_fn)(const gs_prover_binding *, const gs_prover_job *, const uint8_t *, uint64_t, char *, uint64_t);
// a 4th argument (a Buffer holding a serialized proof) turns either call below into Stark.verify() of that proof for the statement the job
// describes (include/gstark_prover.h: gs_prover_verify_on — native, CPU only): returns true or throws the reference's message
napi_value verify_instead(napi_env env, napi_value proof_value, const gs_prover_job &job) {
    void *d;
    size_t len;
    if (!buffer_info(env, proof_value, &d, &len)) { napi_throw_type_error(env, nullptr, "the proof must be a Buffer"); return nullptr; }
    char err[512] = {0};
    const int rc = ((verify_on_fn)dlsym(g_prover, "gs_prover_verify_on"))(g_binding, &job, (const uint8_t *)d, len, err, sizeof err);
    if (rc != GS_OK) { napi_throw_error(env, nullptr, err[0] ? err : "verification failed"); return nullptr; }
    napi_value t;
    NAPI_OK(env, napi_get_boolean(env, true, &t));
    return t;
}
napi_value ProveMimcSerialized(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    void *ctx;
    if (!g_lib) { napi_throw_error(env, nullptr, "call load(path) first"); return nullptr; }
    napi_valuetype job_type;
    if (argc < 3 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || !ctx || napi_typeof(env, argv[2], &job_type) != napi_ok || job_type != napi_object) {
        napi_throw_type_error(env, nullptr, "(ctx, proverLibPath, job[, proof])");
        return nullptr;
    }
    if (!open_driver(env, argv[1])) return nullptr;
    const size_t es = g_es;
    auto prop = [&](const char *name) { napi_value v; napi_get_named_property(env, argv[2], name, &v); return v; };
    auto u64 = [&](const char *name, uint64_t *out) { return get_u64(env, prop(name), out); };
    auto bytes = [&](napi_value v, const uint8_t **data, size_t *len) { void *d; bool ok = buffer_info(env, v, &d, len); *data = (const uint8_t *)d; return ok; };
    gs_prover_job job;
    memset(&job, 0, sizeof job);
    uint64_t t, ef, exe, fri, alg, klen, ktab;
    const uint8_t *rou, *seed, *rc;
    size_t nrou, nseed, nrc;
    if (!u64("steps", &t) || !u64("extensionFactor", &ef) || !u64("exeQueryCount", &exe) || !u64("friQueryCount", &fri) || !u64("hashAlg", &alg) ||
        !u64("kLen", &klen) || !u64("kTable", &ktab) || !bytes(prop("rootOfUnity"), &rou, &nrou) || nrou != es || !bytes(prop("seed"), &seed, &nseed) ||
        nseed != es || !bytes(prop("roundConstants"), &rc, &nrc) || nrc % es) {
        napi_throw_type_error(env, nullptr, "proveMimcSerialized: malformed job");
        return nullptr;
    }
    job.steps = t; job.extension_factor = (uint32_t)ef; job.exe_query_count = (uint32_t)exe; job.fri_query_count = (uint32_t)fri; job.hash_alg = (int32_t)alg;
    memcpy(job.root_of_unity, rou, es);
    const uint32_t degree = 3;
    job.air.kind = 0; job.air.registers = 1; job.air.nconstraints = 1; job.air.degrees = &degree;
    memcpy(job.air.seed, seed, es);
    job.air.round_constants = rc; job.air.nrc = (uint32_t)(nrc / es);
    job.air.k_table = (const void *)(uintptr_t)ktab; job.air.k_len = klen;
    napi_value arr = prop("assertions");
    uint32_t na = 0;
    NAPI_OK(env, napi_get_array_length(env, arr, &na));
    std::vector<gs_assertion> as(na);
    for (uint32_t i = 0; i < na; i++) {
        napi_value e, v;
        NAPI_OK(env, napi_get_element(env, arr, i, &e));
        uint64_t step, reg;
        const uint8_t *val; size_t nval;
        napi_get_named_property(env, e, "step", &v);
        bool ok = get_u64(env, v, &step);
        napi_get_named_property(env, e, "register", &v);
        ok = ok && get_u64(env, v, &reg);
        napi_get_named_property(env, e, "value", &v);
        ok = ok && bytes(v, &val, &nval) && nval == es;
        if (!ok) { napi_throw_type_error(env, nullptr, "proveMimcSerialized: malformed assertion"); return nullptr; }
        as[i].step = step; as[i].reg = (uint32_t)reg; memcpy(as[i].value, val, es);
    }
    job.assertions = as.data(); job.nassertions = na;
    if (argc >= 4) return verify_instead(env, argv[3], job);
    static thread_local std::vector<uint8_t> out;      // (kept between calls: a fresh vector of this size is 4 MB of zeroing per proof)
    if (out.size() < (1u << 22)) out.resize(1u << 22);
    uint64_t n = 0;
    char err[512] = {0};
    int rcode = ((prove_on_fn)dlsym(g_prover, "gs_prover_prove_on"))(g_binding, (gs_ctx *)ctx, &job, out.data(), out.size(), &n, err, sizeof err);
    if (rcode != GS_OK) { napi_throw_error(env, nullptr, (std::string("native prove() failed: ") + err).c_str()); return nullptr; }
    napi_value buf;
    NAPI_OK(env, napi_create_buffer_copy(env, (size_t)n, out.data(), nullptr, &buf));
    return buf;
}

// proveGenericSerialized(ctx, proverLibPath, job) -> Buffer: the same ONE call for an AIR given as register-machine programs (kind 1 of
// gs_prover_air: what js/air_generic.js holds for the reference's Rescue / Poseidon examples).
//   job = { steps, extensionFactor, exeQueryCount, friQueryCount, hashAlg, rootOfUnity: Buffer(16), assertions: [...], registers, degrees: number[],
//           tCode / iCode / eCode: number[] (4 words per instruction), consts: Buffer(16 * nconsts), vmRegs, staticValues: Buffer, staticPeriods: number[],
//           staticTables: BigInt (device pointer), staticLens: number[], firstRows: Buffer, segments, segmentLen }
napi_value ProveGenericSerialized(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    void *ctx;
    if (!g_lib) { napi_throw_error(env, nullptr, "call load(path) first"); return nullptr; }
    napi_valuetype job_type;
    if (argc < 3 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || !ctx || napi_typeof(env, argv[2], &job_type) != napi_ok || job_type != napi_object) {
        napi_throw_type_error(env, nullptr, "(ctx, proverLibPath, job[, proof])");
        return nullptr;
    }
    if (!open_driver(env, argv[1])) return nullptr;
    const size_t es = g_es;
    auto prop = [&](const char *name) { napi_value v; napi_get_named_property(env, argv[2], name, &v); return v; };
    auto u64 = [&](const char *name, uint64_t *out) { return get_u64(env, prop(name), out); };
    auto bytes = [&](napi_value v, const uint8_t **data, size_t *len) { void *d; bool ok = buffer_info(env, v, &d, len); *data = (const uint8_t *)d; return ok; };
    auto words = [&](const char *name, std::vector<uint32_t> &out) {
        napi_value arr = prop(name);
        uint32_t n = 0;
        if (napi_get_array_length(env, arr, &n) != napi_ok) return false;
        out.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            napi_value e;
            uint64_t x;
            if (napi_get_element(env, arr, i, &e) != napi_ok || !get_u64(env, e, &x)) return false;
            out[i] = (uint32_t)x;
        }
        return true;
    };
    gs_prover_job job;
    memset(&job, 0, sizeof job);
    uint64_t t, ef, exe, fri, alg, regs, vmregs, tables, segments, seglen;
    const uint8_t *rou, *consts, *svals, *first;
    size_t nrou, nconsts, nsvals, nfirst;
    std::vector<uint32_t> degrees, tcode, icode, ecode, periods, lens32;
    if (!u64("steps", &t) || !u64("extensionFactor", &ef) || !u64("exeQueryCount", &exe) || !u64("friQueryCount", &fri) || !u64("hashAlg", &alg) ||
        !u64("registers", &regs) || !u64("vmRegs", &vmregs) || !u64("staticTables", &tables) || !u64("segments", &segments) || !u64("segmentLen", &seglen) ||
        !bytes(prop("rootOfUnity"), &rou, &nrou) || nrou != es || !bytes(prop("consts"), &consts, &nconsts) || nconsts % es ||
        !bytes(prop("staticValues"), &svals, &nsvals) || !bytes(prop("firstRows"), &first, &nfirst) || !words("degrees", degrees) || !words("tCode", tcode) ||
        !words("iCode", icode) || !words("eCode", ecode) || !words("staticPeriods", periods) || !words("staticLens", lens32) || tcode.size() % 4 || icode.size() % 4 ||
        ecode.size() % 4 || lens32.size() != periods.size() || nfirst != (segments ? segments : 1) * regs * es ||
        nsvals < es * std::accumulate(periods.begin(), periods.end(), (uint64_t)0)) {
        napi_throw_type_error(env, nullptr, "proveGenericSerialized: malformed job");
        return nullptr;
    }
    std::vector<uint64_t> lens(lens32.begin(), lens32.end());
    // optional: secret registers and input registers (include/gstark_prover.h, round 6).  secretTraces: BigInt[] (device pointers: the secret
    // registers' extensions; prove) or nsecret (verify); inputRegisters: 5 numbers per register {parent + 1, peer + 1 (0 = none), steps, shift
    // as uint32, secret}; inputShapes: per register rank, dimensions...; staticSources: {kind, index} pairs; publicInputs: Buffer +
    // publicInputCounts; rootOfUnityLog2
    auto has = [&](const char *name) { bool h = false; return napi_has_named_property(env, argv[2], name, &h) == napi_ok && h; };
    std::vector<uint32_t> in_regs, in_shapes, sources, pub_counts32;
    std::vector<uint64_t> secret_ptrs, pub_counts;
    std::vector<const void *> secret_traces;
    std::vector<gs_input_register> decls;
    std::vector<gs_static_source> srcs;
    const uint8_t *pub = nullptr;
    size_t npub = 0;
    uint64_t nsecret = 0, root_log2 = 0;
    bool extra_ok = true;
    if (has("secretTraces")) {
        napi_value arr = prop("secretTraces");
        uint32_t n = 0;
        extra_ok = napi_get_array_length(env, arr, &n) == napi_ok && n <= 64;
        secret_ptrs.resize(extra_ok ? n : 0);
        for (uint32_t i = 0; i < n && extra_ok; i++) { napi_value e; extra_ok = napi_get_element(env, arr, i, &e) == napi_ok && get_u64(env, e, &secret_ptrs[i]) && secret_ptrs[i]; }
        for (uint64_t p : secret_ptrs) secret_traces.push_back((const void *)(uintptr_t)p);
        nsecret = secret_ptrs.size();
    } else if (has("nsecret")) extra_ok = u64("nsecret", &nsecret) && nsecret <= 64;
    if (extra_ok && has("inputRegisters")) {
        extra_ok = words("inputRegisters", in_regs) && in_regs.size() % 5 == 0 && in_regs.size() / 5 <= 255;
        for (size_t j = 0; extra_ok && j < in_regs.size() / 5; j++) {
            gs_input_register d;
            d.parent = (int32_t)in_regs[5 * j] - 1; d.peer = (int32_t)in_regs[5 * j + 1] - 1; d.steps = in_regs[5 * j + 2];
            d.shift = (int32_t)in_regs[5 * j + 3]; d.secret = in_regs[5 * j + 4] ? 1u : 0u;
            extra_ok = d.parent < (int32_t)j && d.peer < (int32_t)j;
            decls.push_back(d);
        }
    }
    if (extra_ok && has("inputShapes")) {
        extra_ok = words("inputShapes", in_shapes);
        size_t at = 0, regs_seen = 0;                       // the flat list must hold exactly one (rank, dims...) group per declared register
        while (extra_ok && at < in_shapes.size()) { extra_ok = in_shapes[at] <= 255 && at + 1 + in_shapes[at] <= in_shapes.size(); at += 1 + (extra_ok ? in_shapes[at] : 0); regs_seen++; }
        extra_ok = extra_ok && regs_seen == decls.size();
    }
    if (extra_ok && has("staticSources")) {
        extra_ok = words("staticSources", sources) && sources.size() % 2 == 0;
        for (size_t i = 0; extra_ok && i < sources.size(); i += 2) srcs.push_back(gs_static_source{sources[i], sources[i + 1]});
    }
    if (extra_ok && has("publicInputs")) {
        extra_ok = bytes(prop("publicInputs"), &pub, &npub) && words("publicInputCounts", pub_counts32);
        uint64_t total = 0;
        for (uint32_t c : pub_counts32) { pub_counts.push_back(c); total += c; }
        extra_ok = extra_ok && npub >= total * es;
    }
    if (extra_ok && has("rootOfUnityLog2")) extra_ok = u64("rootOfUnityLog2", &root_log2) && root_log2 < 64;
    if (!extra_ok) { napi_throw_type_error(env, nullptr, "proveGenericSerialized: malformed job (secret / input registers)"); return nullptr; }
    job.steps = t; job.extension_factor = (uint32_t)ef; job.exe_query_count = (uint32_t)exe; job.fri_query_count = (uint32_t)fri; job.hash_alg = (int32_t)alg;
    memcpy(job.root_of_unity, rou, es);
    gs_prover_air &a = job.air;
    a.kind = 1; a.registers = (uint32_t)regs; a.nconstraints = (uint32_t)degrees.size(); a.degrees = degrees.data();
    a.t_code = tcode.data(); a.t_ninstr = (uint32_t)(tcode.size() / 4);
    a.i_code = icode.empty() ? nullptr : icode.data(); a.i_ninstr = (uint32_t)(icode.size() / 4);
    a.e_code = ecode.data(); a.e_ninstr = (uint32_t)(ecode.size() / 4);
    a.consts = consts; a.nconsts = (uint32_t)(nconsts / es); a.vm_regs = (uint32_t)vmregs;
    a.static_values = svals; a.static_periods = periods.data(); a.nstatic = (uint32_t)periods.size();
    a.static_tables = (const void *)(uintptr_t)tables; a.static_lens = lens.data();
    a.first_rows = first; a.segments = segments; a.segment_len = seglen;
    a.nsecret = (uint32_t)nsecret;
    a.secret_traces = secret_traces.empty() ? nullptr : secret_traces.data();
    if (!decls.empty()) {
        a.inputs = decls.data(); a.ninputs = (uint32_t)decls.size();
        a.input_shapes = in_shapes.empty() ? nullptr : in_shapes.data();
        if (!srcs.empty()) {      // verify: one source per static register; staticPeriods / staticValues then list the cyclic ones only
            uint32_t cyclic = 0;
            for (const gs_static_source &q : srcs) cyclic += q.kind == GS_STATIC_CYCLE;
            if (cyclic != periods.size()) { napi_throw_type_error(env, nullptr, "proveGenericSerialized: staticPeriods must list exactly the cyclic static registers"); return nullptr; }
            a.static_sources = srcs.data();
            a.nstatic = (uint32_t)srcs.size();
        }
        a.public_inputs = pub; a.public_input_counts = pub_counts.empty() ? nullptr : pub_counts.data(); a.npublic_inputs = (uint32_t)pub_counts.size();
    }
    job.root_of_unity_log2 = (uint32_t)root_log2;
    napi_value arr = prop("assertions");
    uint32_t na = 0;
    NAPI_OK(env, napi_get_array_length(env, arr, &na));
    std::vector<gs_assertion> as(na);
    for (uint32_t i = 0; i < na; i++) {
        napi_value e, v;
        NAPI_OK(env, napi_get_element(env, arr, i, &e));
        uint64_t step, reg;
        const uint8_t *val; size_t nval;
        napi_get_named_property(env, e, "step", &v);
        bool ok = get_u64(env, v, &step);
        napi_get_named_property(env, e, "register", &v);
        ok = ok && get_u64(env, v, &reg);
        napi_get_named_property(env, e, "value", &v);
        ok = ok && bytes(v, &val, &nval) && nval == es;
        if (!ok) { napi_throw_type_error(env, nullptr, "proveGenericSerialized: malformed assertion"); return nullptr; }
        as[i].step = step; as[i].reg = (uint32_t)reg; memcpy(as[i].value, val, es);
    }
    job.assertions = as.data(); job.nassertions = na;
    if (argc >= 4) return verify_instead(env, argv[3], job);
    static thread_local std::vector<uint8_t> out;      // (kept between calls: a fresh vector of this size is 4 MB of zeroing per proof)
    if (out.size() < (1u << 22)) out.resize(1u << 22);
    uint64_t n = 0;
    char err[512] = {0};
    int rcode = ((prove_on_fn)dlsym(g_prover, "gs_prover_prove_on"))(g_binding, (gs_ctx *)ctx, &job, out.data(), out.size(), &n, err, sizeof err);
    if (rcode != GS_OK) { napi_throw_error(env, nullptr, (std::string("native prove() failed: ") + err).c_str()); return nullptr; }
    napi_value buf;
    NAPI_OK(env, napi_create_buffer_copy(env, (size_t)n, out.data(), nullptr, &buf));
    return buf;
}

// packElements(values: BigInt[], elementSize) -> Buffer of their little-endian elements (values must be non-negative and fit: the caller
// reduces them), unpackElements(buffer, elementSize) -> BigInt[].  A column of an input register is 10^4..10^5 elements per proof; in
// JavaScript a BigInt costs 0.5-2 us to take apart or put together, here a copy of its words.
napi_value PackElements(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    bool is_array = false;
    uint32_t es = 0, n = 0;
    if (argc < 2 || napi_is_array(env, argv[0], &is_array) != napi_ok || !is_array || napi_get_value_uint32(env, argv[1], &es) != napi_ok ||
        (es != 16 && es != 32)) {
        napi_throw_type_error(env, nullptr, "packElements(values: BigInt[], elementSize: 16 | 32)");
        return nullptr;
    }
    NAPI_OK(env, napi_get_array_length(env, argv[0], &n));
    void *data = nullptr;
    napi_value buf;
    NAPI_OK(env, napi_create_buffer(env, (size_t)n * es, &data, &buf));
    uint8_t *out = (uint8_t *)data;
    const size_t max_words = es / 8;
    for (uint32_t i = 0; i < n; i++) {
        napi_value v;
        NAPI_OK(env, napi_get_element(env, argv[0], i, &v));
        uint64_t words[4] = {0, 0, 0, 0};
        size_t count = max_words;
        int sign = 0;
        if (napi_get_value_bigint_words(env, v, &sign, &count, words) != napi_ok) {
            napi_throw_type_error(env, nullptr, "packElements: every value must be a BigInt");
            return nullptr;
        }
        // (count comes back as the number of words the value NEEDS: more than the element holds means it does not fit)
        if (sign || count > max_words) {
            napi_throw_range_error(env, nullptr, "packElements: a value is negative or wider than the element");
            return nullptr;
        }
        memcpy(out + (size_t)i * es, words, es);
    }
    return buf;
}
napi_value UnpackElements(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    bool is_buffer = false;
    uint32_t es = 0;
    if (argc < 2 || napi_is_buffer(env, argv[0], &is_buffer) != napi_ok || !is_buffer || napi_get_value_uint32(env, argv[1], &es) != napi_ok ||
        (es != 16 && es != 32)) {
        napi_throw_type_error(env, nullptr, "unpackElements(buffer: Buffer, elementSize: 16 | 32)");
        return nullptr;
    }
    void *data = nullptr;
    size_t len = 0;
    NAPI_OK(env, napi_get_buffer_info(env, argv[0], &data, &len));
    if (len % es) { napi_throw_range_error(env, nullptr, "unpackElements: the buffer is not a whole number of elements"); return nullptr; }
    const size_t n = len / es;
    napi_value arr;
    NAPI_OK(env, napi_create_array_with_length(env, n, &arr));
    for (size_t i = 0; i < n; i++) {
        uint64_t words[4];
        memcpy(words, (const uint8_t *)data + i * es, es);
        napi_value v;
        NAPI_OK(env, napi_create_bigint_words(env, 0, es / 8, words, &v));
        NAPI_OK(env, napi_set_element(env, arr, (uint32_t)i, v));
    }
    return arr;
}

napi_value Init(napi_env env, napi_value exports) {
    const struct { const char *name; napi_callback cb; } fns[] = {
        {"load", Load}, {"fieldInfo", FieldInfo}, {"ctxCreate", CtxCreate}, {"ctxDestroy", CtxDestroy}, {"alloc", Alloc}, {"call", Call},
        {"merkleProveBatch", MerkleProveBatch}, {"proveMimcSerialized", ProveMimcSerialized}, {"proveGenericSerialized", ProveGenericSerialized},
        {"packElements", PackElements}, {"unpackElements", UnpackElements},
    };
    for (auto &f : fns) {
        napi_value fn;
        if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.cb, nullptr, &fn) != napi_ok) return nullptr;
        if (napi_set_named_property(env, exports, f.name, fn) != napi_ok) return nullptr;
    }
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
