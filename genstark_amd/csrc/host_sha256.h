// host_sha256.h — SHA-256 on the HOST for the few Fiat-Shamir hashes the host side computes itself (query positions,
// random coefficients: QueryIndexGenerator.ts:39-67, galois prng), plus the reference's bigint -> bytes quirk.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
// one block with the SHA extensions (x86 SHA-NI: two rounds per sha256rnds2, the message schedule by sha256msg1 / sha256msg2):
//   W[i] = msg2(msg1(W[i-4], W[i-3]) + alignr(W[i-1], W[i-2], 4), W[i-1])   for the 4-word groups i >= 4
__attribute__((target("sha,sse4.1,ssse3"))) static inline void host_sha256_block_ni(uint32_t h[8], const uint8_t blk[64], const uint32_t K[64]) {
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i tmp = _mm_loadu_si128((const __m128i *)&h[0]), s1 = _mm_loadu_si128((const __m128i *)&h[4]);
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                    // CDAB
    s1 = _mm_shuffle_epi32(s1, 0x1B);                      // EFGH
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);              // ABEF
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                   // CDGH
    const __m128i save0 = s0, save1 = s1;
    __m128i w[16];
    for (int i = 0; i < 16; i++) {
        if (i < 4) {
            w[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(blk + 16 * i)), bswap);
        } else {
            __m128i x = _mm_sha256msg1_epu32(w[i - 4], w[i - 3]);
            x = _mm_add_epi32(x, _mm_alignr_epi8(w[i - 1], w[i - 2], 4));
            w[i] = _mm_sha256msg2_epu32(x, w[i - 1]);
        }
        __m128i m = _mm_add_epi32(w[i], _mm_loadu_si128((const __m128i *)&K[4 * i]));
        s1 = _mm_sha256rnds2_epu32(s1, s0, m);
        m = _mm_shuffle_epi32(m, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, m);
    }
    s0 = _mm_add_epi32(s0, save0);
    s1 = _mm_add_epi32(s1, save1);
    tmp = _mm_shuffle_epi32(s0, 0x1B);                     // FEBA
    s1 = _mm_shuffle_epi32(s1, 0xB1);                      // DCHG
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);                   // DCBA
    s1 = _mm_alignr_epi8(s1, tmp, 8);                      // HGFE
    _mm_storeu_si128((__m128i *)&h[0], s0);
    _mm_storeu_si128((__m128i *)&h[4], s1);
}
static inline bool host_sha256_has_ni() {
    static const bool ok = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
    return ok;
}
#else
static inline bool host_sha256_has_ni() { return false; }
#endif

static inline void host_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    auto rotr = [](uint32_t x, int r) { return (x >> r) | (x << (32 - r)); };
    const size_t total = ((len + 9 + 63) / 64) * 64;
    for (size_t off = 0; off < total; off += 64) {
        uint8_t blk[64];
        for (size_t k = 0; k < 64; k++) {
            size_t p = off + k;
            blk[k] = p < len ? msg[p] : (p == len ? 0x80 : 0);
        }
        if (off + 64 == total) {
            uint64_t bits = (uint64_t)len * 8;
            for (int k = 0; k < 8; k++) blk[63 - k] = (uint8_t)(bits >> (8 * k));
        }
#if defined(__x86_64__)
        if (host_sha256_has_ni()) { host_sha256_block_ni(h, blk, K); continue; }
#endif
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)blk[4 * i] << 24 | (uint32_t)blk[4 * i + 1] << 16 | (uint32_t)blk[4 * i + 2] << 8 | blk[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}


// Buffer.from(value.toString(16), 'hex') for a big-endian value of n bytes: no leading zeros, an odd number of hex digits
// loses the LAST nibble (QueryIndexGenerator.ts:61-67).  Returns the byte count written to out (<= n).
static inline int host_bigint_bytes(const uint8_t *be, int n, uint8_t *out) {
    int lead = 0;
    while (lead < n && be[lead] == 0) lead++;
    if (lead == n) return 0;
    const int nhex = (n - lead) * 2 - ((be[lead] >> 4) == 0 ? 1 : 0);
    const int nbytes = nhex / 2;
    if (nhex & 1) {
        const int first = lead * 2 + 1;   // the string starts at the low nibble of be[lead]
        for (int k = 0; k < nbytes; k++) {
            const int a = first + 2 * k, b = a + 1;
            const uint8_t na = (a & 1) ? (be[a >> 1] & 15) : (be[a >> 1] >> 4), nb = (b & 1) ? (be[b >> 1] & 15) : (be[b >> 1] >> 4);
            out[k] = (uint8_t)((na << 4) | nb);
        }
    } else {
        memcpy(out, be + lead, (size_t)nbytes);
    }
    return nbytes;
}
