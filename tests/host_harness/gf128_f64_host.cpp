// host build of tools/gf128_f64.h (the fp64-FMA product experiment) for tests/test_lazy_field.py: IEEE fma on the host is the same
// function as v_fma_f64, so the limb values are what the device computes.  Compile with -ffp-contract=off.
#include "../../tools/gf128_f64.h"
extern "C" {
void fz_mul_host(const double *x, const double *w, double *out) {
    fz a, b;
    for (int i = 0; i < 3; i++) { a.l[i] = x[i]; b.l[i] = w[i]; }
    const fz y = fz_mul(a, b);
    for (int i = 0; i < 3; i++) out[i] = y.l[i];
}
void fz_unpack_host(const uint32_t *words, double *out) {
    const fz y = fz_unpack(fe_make(words[0], words[1], words[2], words[3]));
    for (int i = 0; i < 3; i++) out[i] = y.l[i];
}
}
