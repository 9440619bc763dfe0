// air_vm.hip — generic AIR: transition functions and constraint evaluators as straight-line register-machine programs
// (include/gstark.h, "AIR ... generic straight-line programs").  Replaces the code air-assembly generates for
// ProvingContext.generateExecutionTrace / evaluateTransitionConstraints (lib/Stark.ts:97; CompositionPolynomial.ts:76).
//
// Constraint evaluation: one thread per composition-domain point interprets the program; the instruction stream and the
// constant pool are wave-uniform (scalar loads), the scratch file lives in private memory.  Trace generation: steps are
// sequentially dependent, one host core interprets the program on native 64-bit limbs.
#include "common.h"
#if !defined(GS_SMALL_Q) && !defined(GS_WIDE_BITS)
#define GS_VM_LAZY 1       // long exponentiations of the interpreter in the lazy five-limb form (pow_group)
#include "gf128_lazy.h"
#endif
#include "host_field.h"
#include "host_pow.h"

enum { OP_LOADC = 0, OP_LOADR = 1, OP_LOADN = 2, OP_LOADS = 3, OP_ADDV = 4, OP_SUBV = 5, OP_MULV = 6, OP_POW = 7, OP_POWC = 8, OP_OUT = 9 };


struct StaticDesc {
    uint64_t offset[GS_AIR_MAX_REGISTERS];  // element offset into the concatenated table
    uint64_t len[GS_AIR_MAX_REGISTERS];
};

template <int NREG>
__global__ void k_air_constraints(const uint4 *__restrict__ code, uint32_t ninstr, const fe *__restrict__ consts,
                                  const fe *__restrict__ p, uint64_t nc, uint64_t shift, const fe *__restrict__ statics, StaticDesc sd,
                                  fe *__restrict__ out, uint64_t prow, uint64_t pstride) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < nc; j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t jn = j + shift;
        if (jn >= nc) jn -= nc;
        const uint64_t jp = j * pstride, jnp = jn * pstride;      // register a at point j: p[a * prow + j * pstride]
        fe vm[NREG];
        for (uint32_t pc = 0; pc < ninstr; pc++) {
            const uint4 ins = code[pc];
            const uint32_t dst = ins.y, a = ins.z, b = ins.w;
            switch (ins.x) {
                case OP_LOADC: vm[dst] = consts[a]; break;
                case OP_LOADR: vm[dst] = p[(uint64_t)a * prow + jp]; break;
                case OP_LOADN: vm[dst] = p[(uint64_t)a * prow + jnp]; break;
                case OP_LOADS: vm[dst] = statics[sd.offset[a] + j % sd.len[a]]; break;
                case OP_ADDV: vm[dst] = fe_add(vm[a], vm[b]); break;
                case OP_SUBV: vm[dst] = fe_sub(vm[a], vm[b]); break;
                case OP_MULV: vm[dst] = fe_mul(vm[a], vm[b]); break;
                case OP_POW: vm[dst] = fe_pow_u64(vm[a], b); break;
                case OP_POWC: vm[dst] = fe_pow(vm[a], consts[b]); break;
                default: out[(uint64_t)dst * nc + j] = vm[a]; break;  // OP_OUT
            }
        }
    }
}

// g (<= 4) independent square-and-multiply chains with the SAME exponent, interleaved: one thread per trace segment is bound by
// the latency of a dependent fe_mul chain (~780 cycles per product against ~360 of issue), four chains fill the gaps
template <int G>
__device__ __forceinline__ void pow_group(fe (&x)[4], const fe &e) {
    fe r[4];
    uint32_t ev[GF_LIMBS];
#pragma unroll
    for (int l = 0; l < GF_LIMBS; l++) ev[l] = fe_limb(e, l);
    int top = GF_LIMBS - 1;
    while (top > 0 && ev[top] == 0) top--;
    if (top == 0 && ev[0] == 0) {
#pragma unroll
        for (int i = 0; i < G; i++) x[i] = fe_one();
        return;
    }
    // no product by one at the start, no squaring after the top bit: the squaring comes BEFORE every bit but the first.  Odd exponents
    // (every S-box exponent and its inverse) start with r = x and keep the inner loop free of the "first product" test (wave-uniform).
    uint32_t bits = ev[0];
    const bool odd = bits & 1u;
#ifdef GS_VM_LAZY
    // exponents longer than a word (Rescue's inverse S-box: 127 squarings + ~64 products): the whole chain in the lazy five-limb form
    // of gf128_lazy.h — lz_sqr is ~53 instructions against the 84 of a canonical product, and a chain of dependent products on one
    // wave per SIMD costs its instruction count (interpreted trace of 2 048 Rescue hashes: 3.6 -> 2.4 ms)
    if (top >= 1) {
        const lzk K = lzk_make();
        lz lx[G], lr[G];
#pragma unroll
        for (int i = 0; i < G; i++) { lx[i] = lz_unpack(x[i]); lr[i] = odd ? lx[i] : lz_unpack(fe_one()); }
        bits >>= 1;
        int kk = 1;
        for (int w = 0; w <= top; w++) {
            const int nb = (w == top) ? 32 - __clz(ev[top] | 1u) : 32;
            for (; kk < nb; kk++) {
#pragma unroll
                for (int i = 0; i < G; i++) lx[i] = lz_sqr(lx[i], K);
                if (bits & 1u) {
#pragma unroll
                    for (int i = 0; i < G; i++) lr[i] = lz_mul_v(lr[i], lx[i], K);
                }
                bits >>= 1;
            }
            kk = 0;
            if (w < top) bits = ev[w + 1];
        }
#pragma unroll
        for (int i = 0; i < G; i++) x[i] = lz_pack(lr[i]);
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < G; i++) r[i] = odd ? x[i] : fe_one();
    bits >>= 1;
    int k = 1;
    for (int w = 0; w <= top; w++) {
        const int nb = (w == top) ? 32 - __clz(ev[top] | 1u) : 32;
        for (; k < nb; k++) {
#pragma unroll
            for (int i = 0; i < G; i++) x[i] = fe_sqr(x[i]);
            if (bits & 1u) {
#pragma unroll
                for (int i = 0; i < G; i++) r[i] = fe_mul(r[i], x[i]);
            }
            bits >>= 1;
        }
        k = 0;
        if (w < top) bits = ev[w + 1];
    }
#pragma unroll
    for (int i = 0; i < G; i++) x[i] = r[i];
}

// x <- x^e for one chain and a long exponent: left-to-right sliding windows of 3 bits over x, x^3, x^5, x^7 (the exponent is the
// same for the whole wave: the window values are scalar, the table look-up is a uniform select).  Rescue's inverse S-box: 127
// squarings + 36 products instead of 127 + 64; a Fermat inversion in the 224-bit field 224 + 79 instead of 224 + 222.
__device__ __noinline__ void pow_window(fe &x, const fe &e) {
    uint32_t ev[GF_LIMBS];
#pragma unroll
    for (int l = 0; l < GF_LIMBS; l++) ev[l] = fe_limb(e, l);
    int top = GF_LIMBS - 1;
    while (top > 0 && ev[top] == 0) top--;
    if (top == 0 && ev[0] == 0) { x = fe_one(); return; }
    const int nbits = 32 * top + 32 - __clz(ev[top]);
    auto bit = [&](int i) -> uint32_t {
        uint32_t w = ev[0];
#pragma unroll
        for (int l = 1; l < GF_LIMBS; l++) w = (i >> 5) == l ? ev[l] : w;
        return (w >> (i & 31)) & 1u;
    };
    const fe x2 = fe_sqr(x), x3 = fe_mul(x, x2), x5 = fe_mul(x3, x2), x7 = fe_mul(x5, x2);
    fe acc = x;
    bool first = true;
    int i = nbits - 1;
    while (i >= 0) {
        if (!bit(i)) { acc = fe_sqr(acc); i--; continue; }
        int j = i - 2 < 0 ? 0 : i - 2;
        while (!bit(j)) j++;
        uint32_t val = 0;
        for (int k = i; k >= j; k--) val = 2 * val + bit(k);
        const fe m = val == 1 ? x : (val == 3 ? x3 : (val == 5 ? x5 : x7));
        if (first) acc = m;
        else {
            for (int k = i; k >= j; k--) acc = fe_sqr(acc);
            acc = fe_mul(acc, m);
        }
        first = false;
        i = j - 1;
    }
    x = acc;
}

// One chain, long exponent.  In the multi-limb fields the windows' fewer products win (point multiplication 198 -> 167 ms); in the
// 128-bit field the bookkeeping of the generic window walk costs more than the 15% of products it saves (Rescue trace 3.34 against
// 2.75 ms with plain square-and-multiply) — the code generator, which unrolls the walk when it writes the source, does get them.
__device__ __forceinline__ void pow_chain(fe (&x)[4], const fe &e) {
#if defined(GS_WIDE_BITS)
    pow_window(x[0], e);
#else
    pow_group<1>(x, e);
#endif
}

__device__ __forceinline__ fe fe_from_lane(const fe &v, int src, int width) {
    fe r = v;
#pragma unroll
    for (int l = 0; l < GF_LIMBS; l++) fe_set_limb(r, l, (uint32_t)__shfl((int)fe_limb(v, l), src, width));
    return r;
}

// `lanes` (1 or 4) consecutive threads per independent trace segment: the same register machine, next-row outputs go to a private
// row buffer.  With 4 lanes every lane interprets the whole program, but a group of adjacent long exponentiations (an S-box layer)
// is one member per lane, results swapped with shuffles: a segment's time is the latency of its dependent products, and the other
// lanes of the wave are idle anyway (see air_jit.hip, which does the same in generated code).
template <int NREG, bool LDS>
__global__ void k_air_trace_segments(const uint4 *__restrict__ code, uint32_t ninstr, const uint4 *__restrict__ icode, uint32_t init_ninstr,
                                     const fe *__restrict__ consts,
                                     const fe *__restrict__ statics, StaticDesc sd, const fe *__restrict__ first_rows, uint32_t registers,
                                     uint64_t segments, uint64_t seglen, uint32_t nvm, uint32_t lanes, fe *__restrict__ out) {
    // The interpreter's register file (vm), the current row and the row being produced are indexed by the instruction stream:
    // as private arrays they live in scratch memory, and with one wave per SIMD nothing hides a scratch round trip (~670 cycles
    // per VM instruction measured).  LDS variant: slot s of lane l at lds[s * 64 + l] (16-byte words, conflict-free).
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    fe *const lds = reinterpret_cast<fe *>(vm_smem) + threadIdx.x;
    fe p_vm[LDS ? 1 : NREG], p_row[LDS ? 1 : GS_AIR_MAX_REGISTERS], p_next[LDS ? 1 : GS_AIR_MAX_REGISTERS];
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t g = tid / lanes;
    const uint32_t sub = (uint32_t)(tid % lanes);
    if (g >= segments) return;
    const uint64_t steps = segments * seglen;
#define vm(i) (*(LDS ? &lds[(uint32_t)(i) * 64] : &p_vm[LDS ? 0 : (i)]))
#define row(i) (*(LDS ? &lds[(nvm + (uint32_t)(i)) * 64] : &p_row[LDS ? 0 : (i)]))
#define next(i) (*(LDS ? &lds[(nvm + registers + (uint32_t)(i)) * 64] : &p_next[LDS ? 0 : (i)]))
    for (uint32_t r = 0; r < registers; r++) row(r) = next(r) = first_rows[g * registers + r];
    if (init_ninstr) {   // the `init { ... }` block: inputs -> first row (no static registers, no next row)
        for (uint32_t pc = 0; pc < init_ninstr; pc++) {
            const uint4 ins = icode[pc];
            const uint32_t dst = ins.y, a = ins.z, b = ins.w;
            switch (ins.x) {
                case OP_LOADC: vm(dst) = consts[a]; break;
                case OP_LOADR: vm(dst) = row(a); break;
                case OP_ADDV: vm(dst) = fe_add(vm(a), vm(b)); break;
                case OP_SUBV: vm(dst) = fe_sub(vm(a), vm(b)); break;
                case OP_MULV: vm(dst) = fe_mul(vm(a), vm(b)); break;
                case OP_POW: vm(dst) = fe_pow_u64(vm(a), b); break;
                case OP_POWC: vm(dst) = fe_pow(vm(a), consts[b]); break;
                default: next(dst) = vm(a); break;
            }
        }
        for (uint32_t r = 0; r < registers; r++) row(r) = next(r);
    }
    for (uint64_t k = 0; k < seglen; k++) {
        const uint64_t i = g * seglen + k;
        for (uint32_t r = 0; r < registers; r++) {
            if (r % lanes == sub) out[(uint64_t)r * steps + i] = row(r);
            next(r) = row(r);
        }
        if (k + 1 == seglen) break;
        for (uint32_t pc = 0; pc < ninstr; pc++) {
            const uint4 ins = code[pc];
            const uint32_t dst = ins.y, a = ins.z, b = ins.w;
            switch (ins.x) {
                case OP_LOADC: vm(dst) = consts[a]; break;
                case OP_LOADR: vm(dst) = row(a); break;
                case OP_LOADS: vm(dst) = statics[sd.offset[a] + i % sd.len[a]]; break;
                case OP_ADDV: vm(dst) = fe_add(vm(a), vm(b)); break;
                case OP_SUBV: vm(dst) = fe_sub(vm(a), vm(b)); break;
                case OP_MULV: vm(dst) = fe_mul(vm(a), vm(b)); break;
                case OP_POW:
                case OP_POWC: {
                    // adjacent exponentiations with the same exponent whose results do not feed each other (the compiler puts
                    // the S-boxes of a round next to each other): up to four at a time as interleaved chains
                    uint32_t g = 1;
                    while (g < 4 && pc + g < ninstr) {
                        const uint4 nx = code[pc + g];
                        bool ok = nx.x == ins.x && nx.w == b;
                        for (uint32_t i = 0; ok && i < g; i++) ok = nx.z != code[pc + i].y;      // a source must not be an earlier result
                        if (!ok) break;
                        g++;
                    }
                    fe x[4];
                    const fe e = ins.x == OP_POWC ? consts[b] : fe_make(b, 0, 0, 0);
                    bool long_e = false;
#pragma unroll
                    for (int l = 1; l < GF_LIMBS; l++) long_e |= fe_limb(e, l) != 0;
                    if (lanes > 1 && g > 1 && long_e) {          // one member per lane (g <= 4 = lanes)
                        x[0] = vm(code[pc + (sub < g ? sub : 0)].z);
                        pow_chain(x, e);
                        for (uint32_t i = 0; i < g; i++) vm(code[pc + i].y) = fe_from_lane(x[0], (int)i, (int)lanes);
                        pc += g - 1;
                        break;
                    }
                    for (uint32_t i = 0; i < g; i++) x[i] = vm(code[pc + i].z);
                    if (g == 1 && long_e) pow_chain(x, e);
                    else if (g == 4) pow_group<4>(x, e);
                    else if (g == 3) pow_group<3>(x, e);
                    else if (g == 2) pow_group<2>(x, e);
                    else pow_group<1>(x, e);
                    for (uint32_t i = 0; i < g; i++) vm(code[pc + i].y) = x[i];
                    pc += g - 1;
                    break;
                }
                default: next(dst) = vm(a); break;  // OP_OUT
            }
        }
        for (uint32_t r = 0; r < registers; r++) row(r) = next(r);
    }
}
#undef vm
#undef row
#undef next

static int check_program(gs_ctx *c, const uint32_t *code, uint32_t ninstr, uint32_t nconsts, uint32_t vm_regs, uint32_t registers,
                         uint32_t nstatic, uint32_t nout, bool allow_next) {
    if (!code || !ninstr) return gs_fail(c, GS_ERR_ARG, "air program: empty");
    if (vm_regs == 0 || vm_regs > GS_AIR_MAX_VM_REGS) return gs_fail(c, GS_ERR_ARG, "air program: vm_regs must be in 1..%d", GS_AIR_MAX_VM_REGS);
    if (registers == 0 || registers > GS_AIR_MAX_REGISTERS || nstatic > GS_AIR_MAX_REGISTERS) return gs_fail(c, GS_ERR_ARG, "air program: too many registers");
    for (uint32_t pc = 0; pc < ninstr; pc++) {
        const uint32_t op = code[4 * pc], dst = code[4 * pc + 1], a = code[4 * pc + 2], b = code[4 * pc + 3];
        bool ok = true;
        switch (op) {
            case OP_LOADC: ok = dst < vm_regs && a < nconsts; break;
            case OP_LOADR: ok = dst < vm_regs && a < registers; break;
            case OP_LOADN: ok = allow_next && dst < vm_regs && a < registers; break;
            case OP_LOADS: ok = dst < vm_regs && a < nstatic; break;
            case OP_ADDV: case OP_SUBV: case OP_MULV: ok = dst < vm_regs && a < vm_regs && b < vm_regs; break;
            case OP_POW: ok = dst < vm_regs && a < vm_regs; break;
            case OP_POWC: ok = dst < vm_regs && a < vm_regs && b < nconsts; break;
            case OP_OUT: ok = dst < nout && a < vm_regs; break;
            default: ok = false;
        }
        if (!ok) return gs_fail(c, GS_ERR_ARG, "air program: invalid instruction %u (op %u)", pc, op);
    }
    return GS_OK;
}

// One host core interprets the program on native limbs (host_field.h): the whole trace of an unsegmented AIR (steps are
// sequentially dependent), and segmented AIRs with only a few segments — one device thread per segment is an order of magnitude
// slower than a host core per step, which only pays off when many segments run side by side.
static void host_run(const uint32_t *code, uint32_t ninstr, const std::vector<hfe> &consts, std::vector<hfe> &vm, const std::vector<hfe> &row,
                     std::vector<hfe> &next, const std::vector<std::vector<hfe>> *statics, uint64_t i) {
    for (uint32_t pc = 0; pc < ninstr; pc++) {
        const uint32_t op = code[4 * pc], dst = code[4 * pc + 1], a = code[4 * pc + 2], b = code[4 * pc + 3];
        switch (op) {
            case OP_LOADC: vm[dst] = consts[a]; break;
            case OP_LOADR: vm[dst] = row[a]; break;
            case OP_LOADS: vm[dst] = (*statics)[a][i % (*statics)[a].size()]; break;
            case OP_ADDV: vm[dst] = hf_add(vm[a], vm[b]); break;
            case OP_SUBV: vm[dst] = hf_sub(vm[a], vm[b]); break;
            case OP_MULV: vm[dst] = hf_mul(vm[a], vm[b]); break;
            case OP_POW: vm[dst] = hf_pow(vm[a], (hfe)b); break;
            case OP_POWC: {
                // adjacent exponentiations with this exponent whose results do not feed each other (an S-box layer): in lock step
                int g = 1;
                while (g < 4 && pc + g < ninstr) {
                    const uint32_t *nx = code + 4 * (pc + g);
                    bool ok = nx[0] == OP_POWC && nx[3] == b;
                    for (int k = 0; ok && k < g; k++) ok = nx[2] != code[4 * (pc + k) + 1];
                    if (!ok) break;
                    g++;
                }
                hfe x[4];
                for (int k = 0; k < g; k++) x[k] = vm[code[4 * (pc + k) + 2]];
                host_pow_group(x, g, consts[b]);
                for (int k = 0; k < g; k++) vm[code[4 * (pc + k) + 1]] = x[k];
                pc += g - 1;
                break;
            }
            default: next[dst] = vm[a]; break;
        }
    }
}

static int host_trace(gs_ctx *c, const uint32_t *code_host, uint32_t ninstr, const uint32_t *init_code_host, uint32_t init_ninstr,
                      const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs, uint32_t registers, const uint8_t *static_values_host,
                      const uint32_t *static_periods_host, uint32_t nstatic, const uint8_t *first_rows_host, uint64_t segments,
                      uint64_t segment_len, void *out) {
    std::vector<hfe> consts(nconsts ? nconsts : 1), vm(vm_regs), row(registers), next(registers);
    for (uint32_t i = 0; i < nconsts; i++) consts[i] = hf_load(consts_host + GS_ELT * i);
    std::vector<std::vector<hfe>> statics(nstatic);
    {
        const uint8_t *p = static_values_host;
        for (uint32_t s = 0; s < nstatic; s++) {
            if (!static_periods_host[s]) return gs_fail(c, GS_ERR_ARG, "air_trace: empty static register");
            statics[s].resize(static_periods_host[s]);
            for (uint32_t i = 0; i < static_periods_host[s]; i++, p += GS_ELT) statics[s][i] = hf_load(p);
        }
    }
    const uint64_t steps = segments * segment_len;
    int rc;
    if ((rc = gs_trace_begin(c, (uint64_t)registers * steps * GS_ELT))) return rc;
    hfe *t = (hfe *)c->h_trace;  // registers x steps, row-major like the Matrix the caller gets
    for (uint64_t g = 0; g < segments; g++) {
        for (uint32_t r = 0; r < registers; r++) row[r] = hf_load(first_rows_host + GS_ELT * (g * registers + r));
        if (init_ninstr) {   // the `init { ... }` block: inputs -> first row
            next = row;
            host_run(init_code_host, init_ninstr, consts, vm, row, next, nullptr, 0);
            row = next;
        }
        for (uint64_t k = 0; k < segment_len; k++) {
            const uint64_t i = g * segment_len + k;
            for (uint32_t r = 0; r < registers; r++) t[(uint64_t)r * steps + i] = row[r];
            if (k + 1 == segment_len) break;
            next = row;
            host_run(code_host, ninstr, consts, vm, row, next, &statics, i);
            row = next;
        }
    }
    GS_HIP(c, hipMemcpyAsync(out, c->h_trace, (size_t)registers * steps * GS_ELT, hipMemcpyHostToDevice, c->stream));
    return gs_trace_end(c);
}

extern "C" {

int gs_air_constraints(gs_ctx *c, const uint32_t *code_host, uint32_t ninstr, const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs,
                       uint32_t registers, uint32_t constraints, const void *p_comp, uint64_t nc, uint64_t shift, const void *static_tables,
                       const uint64_t *static_lens_host, uint32_t nstatic, void *out) {
    return gs_air_constraints_strided(c, code_host, ninstr, consts_host, nconsts, vm_regs, registers, constraints, p_comp, nc, 1, nc, shift, static_tables,
                                      static_lens_host, nstatic, out);
}

int gs_air_constraints_strided(gs_ctx *c, const uint32_t *code_host, uint32_t ninstr, const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs,
                               uint32_t registers, uint32_t constraints, const void *p_comp, uint64_t prow, uint64_t pstride, uint64_t nc, uint64_t shift,
                               const void *static_tables, const uint64_t *static_lens_host, uint32_t nstatic, void *out) {
    if (!c || !p_comp || !out || (!consts_host && nconsts) || (nstatic && (!static_tables || !static_lens_host))) return GS_ERR_ARG;
    int rc = check_program(c, code_host, ninstr, nconsts, vm_regs, registers, nstatic, constraints, true);
    if (rc) return rc;
    if (!nc) return gs_fail(c, GS_ERR_ARG, "air_constraints: empty domain");
    if (!pstride || (nc - 1) > (UINT64_MAX / pstride) || (nc - 1) * pstride >= prow)
        return gs_fail(c, GS_ERR_ARG, "air_constraints: %llu points at stride %llu do not fit rows of %llu elements", (unsigned long long)nc,
                       (unsigned long long)pstride, (unsigned long long)prow);
    StaticDesc sd;
    uint64_t off = 0;
    for (uint32_t s = 0; s < GS_AIR_MAX_REGISTERS; s++) {
        sd.offset[s] = off;
        sd.len[s] = s < nstatic ? static_lens_host[s] : 1;
        if (s < nstatic) {
            if (!static_lens_host[s]) return gs_fail(c, GS_ERR_ARG, "air_constraints: empty static table");
            off += static_lens_host[s];
        }
    }
    // program + constants to device scratch (small; one sync so the staging buffer can be reused)
    const uint64_t code_bytes = ((uint64_t)ninstr * 16 + 255) & ~(uint64_t)255, const_bytes = (uint64_t)(nconsts ? nconsts : 1) * GS_ELT;
    void *dprog;
    if ((rc = gs_tmp_alloc(c, code_bytes + const_bytes, &dprog))) return rc;
    {   // assembled in the upload ring, one asynchronous copy (no round trip; the caller's buffers are free at once)
        void *h = nullptr;
        if ((rc = gs_push_reserve(c, code_bytes + const_bytes, &h)) == GS_OK) {
            memcpy(h, code_host, (size_t)ninstr * 16);
            if (nconsts) memcpy((uint8_t *)h + code_bytes, consts_host, (size_t)nconsts * GS_ELT);
            rc = gs_push_commit(c, dprog, h, code_bytes + const_bytes);
        } else if (rc == GS_ERR_UNSUPPORTED) {
            rc = gs_push(c, dprog, code_host, (uint64_t)ninstr * 16);
            if (rc == GS_OK && nconsts) rc = gs_push(c, (uint8_t *)dprog + code_bytes, consts_host, (uint64_t)nconsts * GS_ELT);
        }
        if (rc) { gs_tmp_free(c, dprog); return rc; }
    }
    if (c->air_jit) {   // compiled form of the program (air_jit.hip); the interpreter below is the fallback
        int jrc = gs_jit_constraints(c, code_host, ninstr, consts_host, nconsts, vm_regs, registers, sd.offset, sd.len, (const fe *)((uint8_t *)dprog + code_bytes),
                                     (const fe *)p_comp, nc, shift % nc, (const fe *)static_tables, (fe *)out, prow, pstride);
        if (jrc == GS_OK) { gs_tmp_free(c, dprog); return GS_OK; }
    }
    const uint4 *dcode = (const uint4 *)dprog;
    const fe *dconst = (const fe *)((uint8_t *)dprog + code_bytes);
    dim3 grid(gs_grid(nc)), block(256);
    if (vm_regs <= 16)
        hipLaunchKernelGGL(k_air_constraints<16>, grid, block, 0, c->stream, dcode, ninstr, dconst, (const fe *)p_comp, nc, shift % nc, (const fe *)static_tables, sd, (fe *)out, prow, pstride);
    else if (vm_regs <= 32)
        hipLaunchKernelGGL(k_air_constraints<32>, grid, block, 0, c->stream, dcode, ninstr, dconst, (const fe *)p_comp, nc, shift % nc, (const fe *)static_tables, sd, (fe *)out, prow, pstride);
    else
        hipLaunchKernelGGL(k_air_constraints<64>, grid, block, 0, c->stream, dcode, ninstr, dconst, (const fe *)p_comp, nc, shift % nc, (const fe *)static_tables, sd, (fe *)out, prow, pstride);
    hipError_t e = hipGetLastError();
    gs_tmp_free(c, dprog);  // stream-ordered reuse: later users of the block are queued behind this kernel
    if (e != hipSuccess) return gs_fail(c, GS_ERR_DEVICE, "air_constraints launch: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_air_trace_segments(gs_ctx *c, const uint32_t *code_host, uint32_t ninstr, const uint32_t *init_code_host, uint32_t init_ninstr,
                          const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs, uint32_t registers, const uint8_t *static_values_host,
                          const uint32_t *static_periods_host, uint32_t nstatic, const uint8_t *first_rows_host, uint64_t segments,
                          uint64_t segment_len, void *out) {
    if (!c || !first_rows_host || !out || (!consts_host && nconsts) || (nstatic && (!static_values_host || !static_periods_host))) return GS_ERR_ARG;
    int rc = check_program(c, code_host, ninstr, nconsts, vm_regs, registers, nstatic, registers, false);
    if (rc) return rc;
    if (init_ninstr && (rc = check_program(c, init_code_host, init_ninstr, nconsts, vm_regs, registers, 0, registers, false))) return rc;
    if (!segments || !segment_len) return gs_fail(c, GS_ERR_ARG, "air_trace_segments: empty");
    StaticDesc sd;
    uint64_t nstat = 0;
    for (uint32_t s = 0; s < GS_AIR_MAX_REGISTERS; s++) {
        sd.offset[s] = nstat;
        sd.len[s] = s < nstatic ? static_periods_host[s] : 1;
        if (s < nstatic) {
            if (!static_periods_host[s]) return gs_fail(c, GS_ERR_ARG, "air_trace_segments: empty static register");
            nstat += static_periods_host[s];
        }
    }
    if (segments <= c->host_trace_segments)
        return host_trace(c, code_host, ninstr, init_code_host, init_ninstr, consts_host, nconsts, vm_regs, registers, static_values_host,
                          static_periods_host, nstatic, first_rows_host, segments, segment_len, out);
    // program, constants, static values and first rows -> one device block (pageable caller memory: one sync)
    const uint64_t main_b = ((uint64_t)ninstr * 16 + 255) & ~(uint64_t)255, code_b = main_b + (((uint64_t)init_ninstr * 16 + 255) & ~(uint64_t)255);
    const uint64_t const_b = ((uint64_t)(nconsts ? nconsts : 1) * GS_ELT + 255) & ~(uint64_t)255;
    const uint64_t stat_b = ((nstat ? nstat : 1) * GS_ELT + 255) & ~(uint64_t)255, rows_b = segments * registers * GS_ELT;
    void *d;
    if ((rc = gs_tmp_alloc(c, code_b + const_b + stat_b + rows_b, &d))) return rc;
    uint8_t *p = (uint8_t *)d;
    {   // assembled in the upload ring, one asynchronous copy (no round trip; the caller's buffers are free at once)
        void *h = nullptr;
        const uint64_t total = code_b + const_b + stat_b + rows_b;
        if ((rc = gs_push_reserve(c, total, &h)) == GS_OK) {
            uint8_t *hp = (uint8_t *)h;
            memcpy(hp, code_host, (size_t)ninstr * 16);
            if (init_ninstr) memcpy(hp + main_b, init_code_host, (size_t)init_ninstr * 16);
            if (nconsts) memcpy(hp + code_b, consts_host, (size_t)nconsts * GS_ELT);
            if (nstat) memcpy(hp + code_b + const_b, static_values_host, (size_t)nstat * GS_ELT);
            memcpy(hp + code_b + const_b + stat_b, first_rows_host, (size_t)rows_b);
            rc = gs_push_commit(c, p, h, total);
        } else if (rc == GS_ERR_UNSUPPORTED) {      // a very long list of first rows: part by part
            rc = gs_push(c, p, code_host, (uint64_t)ninstr * 16);
            if (rc == GS_OK && init_ninstr) rc = gs_push(c, p + main_b, init_code_host, (uint64_t)init_ninstr * 16);
            if (rc == GS_OK && nconsts) rc = gs_push(c, p + code_b, consts_host, (uint64_t)nconsts * GS_ELT);
            if (rc == GS_OK && nstat) rc = gs_push(c, p + code_b + const_b, static_values_host, nstat * GS_ELT);
            if (rc == GS_OK) rc = gs_push(c, p + code_b + const_b + stat_b, first_rows_host, rows_b);
        }
        if (rc) { gs_tmp_free(c, d); return rc; }
    }
    const uint4 *dcode = (const uint4 *)p, *dinit = (const uint4 *)(p + main_b);
    const fe *dconst = (const fe *)(p + code_b), *dstat = (const fe *)(p + code_b + const_b), *drows = (const fe *)(p + code_b + const_b + stat_b);
    if (c->air_jit) {   // compiled form of the program (air_jit.hip); the interpreter below is the fallback
        int jrc = gs_jit_trace_segments(c, code_host, ninstr, init_code_host, init_ninstr, consts_host, nconsts, vm_regs, registers, sd.offset, sd.len, nstat ? static_values_host : nullptr, nstatic, dconst, dstat, drows,
                                        segments, segment_len, (fe *)out);
        if (jrc == GS_OK) { gs_tmp_free(c, d); return GS_OK; }
    }
    // 4 lanes per segment when the program has a layer of long exponentiations to split (same rule as the code generator's)
    uint32_t lanes = 1;
    for (uint32_t pc = 0; pc + 1 < ninstr && lanes == 1; pc++) {
        const uint32_t *in = code_host + 4 * pc, *nx = in + 4;
        if (in[0] != OP_POWC || nx[0] != OP_POWC || nx[3] != in[3] || nx[2] == in[1]) continue;
        const uint8_t *ex = consts_host + (size_t)in[3] * GS_ELT;
        for (int i = 4; i < GS_ELT; i++)
            if (ex[i]) lanes = 4;
    }
    dim3 block(64), grid((unsigned)((segments * lanes + 63) / 64));   // one wave per 64 / lanes segments: the few long-running threads spread over the CUs
    const uint64_t lds_bytes = ((uint64_t)vm_regs + 2ull * registers) * 64 * GS_ELT;
#define GS_LAUNCH_TRACE(N, L, SH)                                                                                                        \
    hipLaunchKernelGGL((k_air_trace_segments<N, L>), grid, block, SH, c->stream, dcode, ninstr, dinit, init_ninstr, dconst, dstat, sd, drows, \
                       registers, segments, segment_len, vm_regs, lanes, (fe *)out)
    if (lds_bytes <= 160 * 1024) {   // gfx950: 160 KB of LDS per workgroup
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_air_trace_segments<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        GS_LAUNCH_TRACE(1, true, (size_t)lds_bytes);
    }
    else if (vm_regs <= 16) GS_LAUNCH_TRACE(16, false, 0);
    else if (vm_regs <= 32) GS_LAUNCH_TRACE(32, false, 0);
    else GS_LAUNCH_TRACE(64, false, 0);
#undef GS_LAUNCH_TRACE
    hipError_t e = hipGetLastError();
    gs_tmp_free(c, d);
    if (e != hipSuccess) return gs_fail(c, GS_ERR_DEVICE, "air_trace_segments launch: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_air_trace(gs_ctx *c, const uint32_t *code_host, uint32_t ninstr, const uint8_t *consts_host, uint32_t nconsts, uint32_t vm_regs,
                 uint32_t registers, const uint8_t *static_values_host, const uint32_t *static_periods_host, uint32_t nstatic,
                 const uint8_t *first_row_host, uint64_t steps, void *out) {
    if (!c || !first_row_host || !out || (!consts_host && nconsts) || (nstatic && (!static_values_host || !static_periods_host))) return GS_ERR_ARG;
    int rc = check_program(c, code_host, ninstr, nconsts, vm_regs, registers, nstatic, registers, false);
    if (rc) return rc;
    if (!steps) return gs_fail(c, GS_ERR_ARG, "air_trace: empty");
    return host_trace(c, code_host, ninstr, nullptr, 0, consts_host, nconsts, vm_regs, registers, static_values_host, static_periods_host, nstatic,
                      first_row_host, 1, steps, out);
}

}  // extern "C"
