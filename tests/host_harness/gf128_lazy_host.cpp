// Host-side harness for tests/test_lazy_field.py: genstark_amd/csrc/gf128_lazy.h compiled with g++ / clang++ so the exact
// limb arithmetic of the NTT kernels is checked against Python integers on a machine without a GPU.  Not part of the product.
#include "../../genstark_amd/csrc/gf128_lazy.h"
#include <string.h>
extern "C" {
static const lzk K = lzk_make();
static fe ld(const uint8_t *p) { fe r; memcpy(&r, p, 16); return r; }
static void st(uint8_t *p, fe v) { memcpy(p, &v, 16); }
static lz ldl(const int32_t *p) { lz r; for (int i = 0; i < 5; i++) r.l[i] = p[i]; return r; }
static void stl(int32_t *p, const lz &v) { for (int i = 0; i < 5; i++) p[i] = v.l[i]; }
void z_unpack(const uint8_t *a, int32_t *o, int n) { for (int i = 0; i < n; i++) stl(o + 5 * i, lz_unpack(ld(a + 16 * i))); }
void z_pack(const int32_t *x, uint8_t *o, int n) { for (int i = 0; i < n; i++) st(o + 16 * i, lz_pack(ldl(x + 5 * i))); }
void z_pack_weak(const int32_t *x, uint8_t *o, int n) { for (int i = 0; i < n; i++) st(o + 16 * i, lz_pack_weak(ldl(x + 5 * i))); }
void z_norm(const int32_t *x, int32_t *o, int n) { for (int i = 0; i < n; i++) stl(o + 5 * i, lz_norm(ldl(x + 5 * i))); }
void z_shift(const int32_t *x, int32_t *o, int n) { for (int i = 0; i < n; i++) stl(o + 5 * i, lz_shift_limb(ldl(x + 5 * i), K)); }
void z_sqr(const int32_t *x, int32_t *o, int n) { for (int i = 0; i < n; i++) stl(o + 5 * i, lz_sqr(ldl(x + 5 * i), K)); }
// x^(2^k) * x as the exponentiation chains of the generated AIR kernels run it: k squarings and one product, no packing in between
void z_sqr_chain(const int32_t *x, int k, int32_t *o, int n) {
    for (int i = 0; i < n; i++) { const lz a = ldl(x + 5 * i); lz c = a; for (int q = 0; q < k; q++) c = lz_sqr(c, K); stl(o + 5 * i, lz_mul_v(c, a, K)); }
}
void z_mul_vm(const int32_t *x, const int32_t *w, int32_t *o, int n) { for (int i = 0; i < n; i++) stl(o + 5 * i, lz_mul_vm(ldl(x + 5 * i), ldl(w + 5 * i), K)); }
void z_mont_consts(uint8_t *o) { st(o, lz_mont_r()); st(o + 16, lz_mont_r2()); }
void z_mul_v(const int32_t *x, const int32_t *w, int32_t *o, int n) { for (int i = 0; i < n; i++) stl(o + 5 * i, lz_mul_v(ldl(x + 5 * i), ldl(w + 5 * i), K)); }
// multiplier given as a canonical element: its W-form is built the way the plan builds the radix-16 twiddles
void z_mul_u(const int32_t *x, const uint8_t *w, int32_t *o, int n) {
    for (int i = 0; i < n; i++) { lzw W; lz_wform(ld(w + 16 * i), W); stl(o + 5 * i, lz_mul_u(ldl(x + 5 * i), W, K)); }
}
// multiplier given as NN limbs: its W-form is built on the fly with lz_shift_limb (the running-product twiddles)
void z_mul_u_rows(const int32_t *x, const int32_t *w, int32_t *o, int n) {
    for (int i = 0; i < n; i++) {
        lzw W; lz cur = ldl(w + 5 * i);
        for (int r = 0; r < 5; r++) { for (int j = 0; j < 5; j++) W.w[r][j] = cur.l[j]; if (r < 4) cur = lz_shift_limb(cur, K); }
        stl(o + 5 * i, lz_mul_u(ldl(x + 5 * i), W, K));
    }
}
}
