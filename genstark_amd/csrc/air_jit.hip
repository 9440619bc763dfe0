// air_jit.hip — AIR programs compiled instead of interpreted (opt-in: gs_air_jit(ctx, 1) or GSTARK_AIR_JIT=1).
//
// The reference's air-assembly GENERATES code for an AIR's transition function and constraint evaluator when a module is
// instantiated ("generated JS over BigInt").  The register machine of air_vm.hip is the portable form of the same
// programs; here they are turned into straight-line HIP source — VM registers become variables, static-register offsets and
// periods become literals — compiled once per program with hiprtc for gfx950 and cached for the process.  The compiler then keeps
// the state in VGPRs and schedules independent products side by side, which an interpreter stepping through LDS cannot: the
// interpreter spends ~2 400 cycles per VM instruction with one wave per SIMD, a compiled S-box + MDS round is bound by its
// dependency chains only.  Same arithmetic (the field header the library itself is built from is embedded in the source), same
// values; any failure to compile falls back to the interpreter.
#include <hip/hiprtc.h>

#include <stdlib.h>

#include <mutex>

#include "common.h"

enum { J_LOADC = 0, J_LOADR = 1, J_LOADN = 2, J_LOADS = 3, J_ADDV = 4, J_SUBV = 5, J_MULV = 6, J_POW = 7, J_POWC = 8, J_OUT = 9 };

static const char *kFieldHeader =
#if defined(GS_SMALL_Q)
#include "jit_gf_small.inc"
#elif defined(GS_WIDE_BITS)
#include "jit_gf_wide.inc"
#else
#include "jit_gf128.inc"
#endif
    ;

#define GS_STR2(x) #x
#define GS_STR(x) GS_STR2(x)

struct JitKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    bool failed = false;
};
static std::mutex g_jit_mutex;
static std::map<std::string, JitKernel> g_jit_cache;   // per process: the lanes of a pool share the compiled programs

static std::string jit_preamble() {
    std::string s;
#if defined(GS_SMALL_Q)
    s += "#define GS_SMALL_Q " GS_STR(GS_SMALL_Q) "\n";
#elif defined(GS_WIDE_BITS)
    s += "#define GS_WIDE_BITS " GS_STR(GS_WIDE_BITS) "\n";
#endif
    s += "#include \"gs_field.cuh\"\n";
    // g independent square-and-multiply chains with one exponent (see pow_group in air_vm.hip)
    s += "template <int G> __device__ __forceinline__ void gs_pow_group(fe (&x)[G], const fe &e) {\n"
         "    fe r[G];\n"
         "    const unsigned int *ev = reinterpret_cast<const unsigned int *>(&e);\n"
         "    int top = (int)(sizeof(fe) / 4) - 1;\n"
         "    while (top > 0 && ev[top] == 0) top--;\n"
         "    if (top == 0 && ev[0] == 0) { for (int i = 0; i < G; i++) x[i] = fe_one(); return; }\n"
         "    unsigned int bits = ev[0];\n"
         "    const bool odd = bits & 1u;\n"
         "    for (int i = 0; i < G; i++) r[i] = odd ? x[i] : fe_one();\n"
         "    bits >>= 1;\n"
         "    int k = 1;\n"
         "    for (int w = 0; w <= top; w++) {\n"
         "        const int nb = (w == top) ? 32 - __clz(ev[top] | 1u) : 32;\n"
         "        for (; k < nb; k++) {\n"
         "#pragma unroll\n"
         "            for (int i = 0; i < G; i++) x[i] = fe_sqr(x[i]);\n"
         "            if (bits & 1u) {\n"
         "#pragma unroll\n"
         "                for (int i = 0; i < G; i++) r[i] = fe_mul(r[i], x[i]);\n"
         "            }\n"
         "            bits >>= 1;\n"
         "        }\n"
         "        k = 0;\n"
         "        if (w < top) bits = ev[w + 1];\n"
         "    }\n"
         "    for (int i = 0; i < G; i++) x[i] = r[i];\n"
         "}\n";
    return s;
}

// straight-line statements for one program; `index` names the loop variable static tables are indexed by
static bool jit_body(std::string &s, const uint32_t *code, uint32_t ninstr, const uint64_t *soff, const uint64_t *slen, bool allow_statics,
                     const char *cur, const char *nxt, const char *index, const char *sink) {
    char buf[256];
    for (uint32_t pc = 0; pc < ninstr; pc++) {
        const uint32_t op = code[4 * pc], d = code[4 * pc + 1], a = code[4 * pc + 2], b = code[4 * pc + 3];
        switch (op) {
            case J_LOADC: snprintf(buf, sizeof buf, "        t%u = consts[%u];\n", d, a); break;
            case J_LOADR: snprintf(buf, sizeof buf, "        t%u = %s%u;\n", d, cur, a); break;
            case J_LOADN:
                if (!nxt) return false;
                snprintf(buf, sizeof buf, "        t%u = %s%u;\n", d, nxt, a);
                break;
            case J_LOADS:
                if (!allow_statics) return false;
                if (slen[a] & (slen[a] - 1)) snprintf(buf, sizeof buf, "        t%u = statics[%lluull + %s %% %lluull];\n", d, (unsigned long long)soff[a], index, (unsigned long long)slen[a]);
                else snprintf(buf, sizeof buf, "        t%u = statics[%lluull + (%s & %lluull)];\n", d, (unsigned long long)soff[a], index, (unsigned long long)(slen[a] - 1));
                break;
            case J_ADDV: snprintf(buf, sizeof buf, "        t%u = fe_add(t%u, t%u);\n", d, a, b); break;
            case J_SUBV: snprintf(buf, sizeof buf, "        t%u = fe_sub(t%u, t%u);\n", d, a, b); break;
            case J_MULV: snprintf(buf, sizeof buf, "        t%u = fe_mul(t%u, t%u);\n", d, a, b); break;
            case J_POW:
            case J_POWC: {
                // adjacent exponentiations with the same exponent whose results do not feed each other: interleaved chains
                uint32_t g = 1;
                while (g < 4 && pc + g < ninstr) {
                    const uint32_t *nx = code + 4 * (pc + g);
                    bool ok = nx[0] == op && nx[3] == b;
                    for (uint32_t i = 0; ok && i < g; i++) ok = nx[2] != code[4 * (pc + i) + 1];
                    if (!ok) break;
                    g++;
                }
                if (op == J_POW && b <= 8 && g == 1) {
                    snprintf(buf, sizeof buf, "        t%u = fe_pow_u64(t%u, %uull);\n", d, a, b);
                    break;
                }
                s += "        {\n";
                snprintf(buf, sizeof buf, "            fe x[%u] = {", g);
                s += buf;
                for (uint32_t i = 0; i < g; i++) { snprintf(buf, sizeof buf, "%st%u", i ? ", " : "", code[4 * (pc + i) + 2]); s += buf; }
                s += "};\n";
                if (op == J_POWC) snprintf(buf, sizeof buf, "            const fe e = consts[%u];\n", b);
                else snprintf(buf, sizeof buf, "            const fe e = fe_make(%uu, 0u, 0u, 0u);\n", b);
                s += buf;
                snprintf(buf, sizeof buf, "            gs_pow_group<%u>(x, e);\n", g);
                s += buf;
                for (uint32_t i = 0; i < g; i++) { snprintf(buf, sizeof buf, "            t%u = x[%u];\n", code[4 * (pc + i) + 1], i); s += buf; }
                snprintf(buf, sizeof buf, "        }\n");
                pc += g - 1;
                break;
            }
            case J_OUT: snprintf(buf, sizeof buf, "        %s%u = t%u;\n", sink, d, a); break;
            default: return false;
        }
        s += buf;
    }
    return true;
}

static JitKernel *jit_get(gs_ctx *c, const std::string &source, const char *entry) {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    auto it = g_jit_cache.find(source);
    if (it != g_jit_cache.end()) return it->second.failed ? nullptr : &it->second;
    JitKernel &k = g_jit_cache[source];
    k.failed = true;
    hiprtcProgram prog;
    const char *header_names[] = {"gs_field.cuh"};
    const char *headers[] = {kFieldHeader};
    if (hiprtcCreateProgram(&prog, source.c_str(), "gs_air_jit.hip", 1, headers, header_names) != HIPRTC_SUCCESS) return nullptr;
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    const char *verbose = getenv("GSTARK_AIR_JIT_VERBOSE");
    if (verbose) fprintf(stderr, "[gstark] compiling an AIR program (%zu bytes of source, entry %s)\n", source.size(), entry);
    hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, 0);
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        gs_fail(c, GS_ERR_DEVICE, "air jit: %.400s", log.c_str());
        if (verbose) fprintf(stderr, "[gstark] hiprtc failed, interpreting instead:\n%s\n%.3000s\n", log.c_str(), verbose[0] == '2' ? source.c_str() : "");
        hiprtcDestroyProgram(&prog);
        return nullptr;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    if (hipModuleLoadData(&k.module, code.data()) != hipSuccess) return nullptr;
    if (hipModuleGetFunction(&k.fn, k.module, entry) != hipSuccess) return nullptr;
    k.failed = false;
    return &k;
}

// ---- trace segments --------------------------------------------------------------------------------------------------------------
// returns GS_OK when the compiled kernel was launched, GS_ERR_UNSUPPORTED when the caller should interpret instead
int gs_jit_trace_segments(gs_ctx *c, const uint32_t *code, uint32_t ninstr, const uint32_t *icode, uint32_t init_ninstr, uint32_t vm_regs,
                          uint32_t registers, const uint64_t *soff, const uint64_t *slen, const fe *dconst, const fe *dstat, const fe *drows,
                          uint64_t segments, uint64_t seglen, fe *out) {
    std::string s = jit_preamble();
    char buf[256];
    s += "extern \"C\" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void gs_jit_trace(const fe *__restrict__ consts, const fe *__restrict__ statics, const fe *__restrict__ first_rows,\n"
         "                                         unsigned long long segments, unsigned long long seglen, fe *__restrict__ out) {\n"
         "    const unsigned long long g = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;\n"
         "    if (g >= segments) return;\n"
         "    const unsigned long long steps = segments * seglen;\n";
    for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "    fe r%u = first_rows[g * %uull + %uull], n%u;\n", r, registers, r, r); s += buf; }
    for (uint32_t t = 0; t < vm_regs; t++) { snprintf(buf, sizeof buf, "    fe t%u;\n", t); s += buf; }
    if (init_ninstr) {
        s += "    {\n";
        for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        n%u = r%u;\n", r, r); s += buf; }
        if (!jit_body(s, icode, init_ninstr, soff, slen, false, "r", nullptr, "0ull", "n")) return GS_ERR_UNSUPPORTED;
        for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        r%u = n%u;\n", r, r); s += buf; }
        s += "    }\n";
    }
    s += "    for (unsigned long long k = 0; k < seglen; k++) {\n"
         "        const unsigned long long i = g * seglen + k;\n";
    for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        out[%uull * steps + i] = r%u;\n", r, r); s += buf; }
    s += "        if (k + 1 == seglen) break;\n";
    for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        n%u = r%u;\n", r, r); s += buf; }
    if (!jit_body(s, code, ninstr, soff, slen, true, "r", nullptr, "i", "n")) return GS_ERR_UNSUPPORTED;
    for (uint32_t r = 0; r < registers; r++) { snprintf(buf, sizeof buf, "        r%u = n%u;\n", r, r); s += buf; }
    s += "    }\n}\n";
    JitKernel *k = jit_get(c, s, "gs_jit_trace");
    if (!k) return GS_ERR_UNSUPPORTED;
    unsigned long long a_segments = segments, a_seglen = seglen;
    void *args[] = {(void *)&dconst, (void *)&dstat, (void *)&drows, (void *)&a_segments, (void *)&a_seglen, (void *)&out};
    const unsigned block = 64, grid = (unsigned)((segments + block - 1) / block);
    if (hipModuleLaunchKernel(k->fn, grid, 1, 1, block, 1, 1, 0, c->stream, args, nullptr) != hipSuccess) return GS_ERR_UNSUPPORTED;
    c->jit_launches++;
    return GS_OK;
}

// ---- constraints -----------------------------------------------------------------------------------------------------------------
int gs_jit_constraints(gs_ctx *c, const uint32_t *code, uint32_t ninstr, uint32_t vm_regs, uint32_t registers, const uint64_t *soff,
                       const uint64_t *slen, const fe *dconst, const fe *p, uint64_t nc, uint64_t shift, const fe *statics, fe *out) {
    (void)registers;
    std::string s = jit_preamble();
    char buf[256];
    s += "extern \"C\" __global__ __launch_bounds__(128) void gs_jit_constraints(const fe *__restrict__ consts, const fe *__restrict__ p, unsigned long long nc,\n"
         "                                               unsigned long long shift, const fe *__restrict__ statics, fe *__restrict__ out) {\n";
    for (uint32_t t = 0; t < vm_regs; t++) { snprintf(buf, sizeof buf, "    fe t%u;\n", t); s += buf; }
    s += "    for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < nc; j += (unsigned long long)gridDim.x * blockDim.x) {\n"
         "        unsigned long long jn = j + shift;\n"
         "        if (jn >= nc) jn -= nc;\n";
    // loads of trace registers and outputs are memory operations here: rewrite them on the fly
    std::vector<uint32_t> tmp(code, code + 4 * (size_t)ninstr);
    std::string body;
    for (uint32_t pc = 0; pc < ninstr; pc++) {
        const uint32_t op = tmp[4 * pc], d = tmp[4 * pc + 1], a = tmp[4 * pc + 2];
        if (op == J_LOADR) { snprintf(buf, sizeof buf, "        t%u = p[%uull * nc + j];\n", d, a); body += buf; }
        else if (op == J_LOADN) { snprintf(buf, sizeof buf, "        t%u = p[%uull * nc + jn];\n", d, a); body += buf; }
        else if (op == J_OUT) { snprintf(buf, sizeof buf, "        out[%uull * nc + j] = t%u;\n", d, a); body += buf; }
        else {
            // runs of arithmetic go through the common generator (so that exponentiation groups are found)
            uint32_t end = pc;
            while (end < ninstr && tmp[4 * end] != J_LOADR && tmp[4 * end] != J_LOADN && tmp[4 * end] != J_OUT) end++;
            if (!jit_body(body, tmp.data() + 4 * pc, end - pc, soff, slen, true, "r", nullptr, "j", "n")) return GS_ERR_UNSUPPORTED;
            pc = end - 1;
        }
    }
    s += body;
    s += "    }\n}\n";
    JitKernel *k = jit_get(c, s, "gs_jit_constraints");
    if (!k) return GS_ERR_UNSUPPORTED;
    unsigned long long a_nc = nc, a_shift = shift;
    void *args[] = {(void *)&dconst, (void *)&p, (void *)&a_nc, (void *)&a_shift, (void *)&statics, (void *)&out};
    const unsigned block = 128, grid = gs_grid(nc, block, 256 * 16);
    if (hipModuleLaunchKernel(k->fn, grid, 1, 1, block, 1, 1, 0, c->stream, args, nullptr) != hipSuccess) return GS_ERR_UNSUPPORTED;
    c->jit_launches++;
    return GS_OK;
}
