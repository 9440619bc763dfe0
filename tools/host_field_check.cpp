// host_field_check.cpp — the host-side field arithmetic of the library (csrc/host_field*.h, csrc/host_pow.h: what the host trace
// interpreter of air_vm.hip computes with) as a filter, so that tests/test_host_field.py can compare it with Python integers on
// machines without a GPU.  Build: g++ -O2 [-DGS_WIDE_BITS=224|256 | -DGS_SMALL_Q=<q>ull] tools/host_field_check.cpp
// stdin: lines "a b e" (hex, big endian); stdout: "a*b  a^-1  a^e  b^e  m1  m2" computed by hf_mul, hf_inv and host_pow_group (g = 2);
// m1 = two steps of the MiMC recurrence as gs_mimc_trace runs it (weak chain, canonical value taken beside it): ((a^3 + b)^3 + b);
// m2 (128-bit flavour: the same from the NON-canonical 128-bit value ~a — the chain accepts any representative; others: = m1).
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#if defined(GS_WIDE_BITS)
#define GS_ELT 32
#include "../genstark_amd/csrc/gf_wide.h"
#else
#define GS_ELT 16
#endif
#include "../genstark_amd/csrc/host_field.h"
#include "../genstark_amd/csrc/host_pow.h"

static hfe parse(const char *h) {
    uint8_t b[GS_ELT];
    const size_t n = strlen(h);
    for (int i = 0; i < GS_ELT; i++) {
        unsigned v = 0;
        if (n >= 2 * (size_t)(i + 1)) sscanf(h + n - 2 * (i + 1), "%2x", &v);
        b[i] = (uint8_t)v;
    }
    return hf_load(b);
}
static void show(hfe x) {
    uint8_t b[GS_ELT];
    hf_store(b, x);
    for (int i = GS_ELT - 1; i >= 0; i--) printf("%02x", b[i]);
}
#if defined(HF_HAVE_CUBE_ADD_ROWS)
// (the row form is compiled for BMI2 cores: it inlines only into a caller built for them, like the chain's own build in air_mimc.hip)
__attribute__((target("bmi2"), noinline)) static hfe rows_step(hfe x, hfe k) { return hf_cube_add_rows(x, k); }
#endif
int main() {
    static char a[80], b[80], e[80];
    while (scanf("%64s %64s %64s", a, b, e) == 3) {
        hfe x = parse(a), y = parse(b), ex = parse(e), g[2] = {x, y};
        host_pow_group(g, 2, ex);
        show(hf_mul(x, y)); printf(" ");
        show(hf_inv(x)); printf(" ");
        show(g[0]); printf(" ");
        show(g[1]); printf(" ");
        show(hf_mimc_out(hf_mimc_step_weak(hf_mimc_step_weak(x, y), y))); printf(" ");
#if !defined(GS_WIDE_BITS) && !defined(GS_SMALL_Q)
#if defined(HF_HAVE_CUBE_ADD_ROWS)
        // the row form of the step (BMI2 cores: what gs_mimc_trace runs on the GPU box's host) must agree with the portable one
        if (__builtin_cpu_supports("bmi2")) {
            const hfe m = rows_step(rows_step(~x, y), y), w = hf_mimc_step_weak(hf_mimc_step_weak(~x, y), y);
            if (hf_mimc_out(m) != hf_mimc_out(w) || hf_mimc_out(rows_step(x, y)) != hf_mimc_out(hf_mimc_step_weak(x, y))) { printf("ROW FORM DIFFERS\n"); continue; }
            show(hf_mimc_out(m)); printf("\n");
            continue;
        }
#endif
        show(hf_mimc_out(hf_mimc_step_weak(hf_mimc_step_weak(~x, y), y))); printf("\n");
#else
        show(hf_mimc_out(hf_mimc_step_weak(hf_mimc_step_weak(x, y), y))); printf("\n");
#endif
    }
    return 0;
}
