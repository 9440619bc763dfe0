// host_pow.h — exponentiation by element-sized exponents for the host interpreter of air_vm.hip (include after host_field.h).
#pragma once
// x^e for element-sized exponents: left-to-right windows of 4 bits over a table of the odd powers x, x^3 .. x^15 (short exponents:
// plain square-and-multiply).  A 128-bit exponent is 127 squarings + ~34 products instead of ~64; p - 2 of the 224-bit field, almost
// all ones, 224 + 53 instead of 224 + 222.  Same element: the chain does not change the value.
#ifndef HF_CHAIN_MUL
#define HF_CHAIN_MUL hf_mul
#define HF_CHAIN_END(x) (x)
#endif
#ifndef HF_CHAIN_ADD
#define HF_CHAIN_ADD hf_add
#endif
// x[k] <- x[k]^e for g <= 4 independent bases in lock step: a host core overlaps the products of the g chains (one chain alone is
// bound by the latency of a product, ~2.5x its issue cost).
static void host_pow_group(hfe *x, int g, hfe e) {
    uint8_t eb[GS_ELT];
    hf_store(eb, e);
    int nbits = 0;
    for (int i = GS_ELT * 8 - 1; i >= 0 && !nbits; i--)
        if ((eb[i / 8] >> (i % 8)) & 1) nbits = i + 1;
    if (nbits <= 16) {
        for (int k = 0; k < g; k++) x[k] = hf_pow(x[k], e);
        return;
    }
    auto bit = [&](int i) { return (eb[i / 8] >> (i % 8)) & 1; };
    hfe tab[4][8], acc[4];                       // tab[k][m] = x[k]^(2m+1)
    for (int k = 0; k < g; k++) {
        const hfe x2 = HF_CHAIN_MUL(x[k], x[k]);
        tab[k][0] = x[k];
        for (int m = 1; m < 8; m++) tab[k][m] = HF_CHAIN_MUL(tab[k][m - 1], x2);
    }
    bool first = true;
    int i = nbits - 1;
    while (i >= 0) {
        if (!bit(i)) {
            for (int k = 0; k < g; k++) acc[k] = HF_CHAIN_MUL(acc[k], acc[k]);
            i--;
            continue;
        }
        int j = i - 3 < 0 ? 0 : i - 3;
        while (!bit(j)) j++;
        int val = 0;
        for (int q = i; q >= j; q--) val = 2 * val + bit(q);
        if (first) {
            for (int k = 0; k < g; k++) acc[k] = tab[k][val >> 1];
        } else {
            for (int q = i; q >= j; q--)
                for (int k = 0; k < g; k++) acc[k] = HF_CHAIN_MUL(acc[k], acc[k]);
            for (int k = 0; k < g; k++) acc[k] = HF_CHAIN_MUL(acc[k], tab[k][val >> 1]);
        }
        first = false;
        i = j - 1;
    }
    for (int k = 0; k < g; k++) x[k] = HF_CHAIN_END(acc[k]);
}

