// Host-side harness for tests/: compiles genstark_amd/csrc/gf128.h and hash_dev.h with g++ so the
// exact device arithmetic can be unit-tested on a machine without a GPU.  Not part of the product.
#include "../../genstark_amd/csrc/gf128.h"
#include <string.h>
extern "C" {
static fe ld(const uint8_t *p) { fe r; memcpy(&r, p, 16); return r; }
static void st(uint8_t *p, fe v) { memcpy(p, &v, 16); }
void h_mul(const uint8_t *a, const uint8_t *b, uint8_t *o, int n) { for (int i = 0; i < n; i++) st(o + 16 * i, fe_mul(ld(a + 16 * i), ld(b + 16 * i))); }
void h_add(const uint8_t *a, const uint8_t *b, uint8_t *o, int n) { for (int i = 0; i < n; i++) st(o + 16 * i, fe_add(ld(a + 16 * i), ld(b + 16 * i))); }
void h_sub(const uint8_t *a, const uint8_t *b, uint8_t *o, int n) { for (int i = 0; i < n; i++) st(o + 16 * i, fe_sub(ld(a + 16 * i), ld(b + 16 * i))); }
void h_inv(const uint8_t *a, uint8_t *o, int n) { for (int i = 0; i < n; i++) st(o + 16 * i, fe_inv(ld(a + 16 * i))); }
void h_pow(const uint8_t *a, const uint8_t *e, uint8_t *o, int n) { for (int i = 0; i < n; i++) st(o + 16 * i, fe_pow(ld(a + 16 * i), ld(e + 16 * i))); }
}
