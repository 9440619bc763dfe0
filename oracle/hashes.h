/*
 * oracle/hashes.h — TEST INFRASTRUCTURE (CPU oracle).  SHA-256 (FIPS 180-4) and unkeyed
 * BLAKE2s-256 (RFC 7693, digest 32 B, no salt/personalisation), the two algorithms
 * `createHash` accepts (lib/Stark.ts:19-20,50).  The reference computes them inside
 * @guildofweavers/merkle@0.3.12 (absent) and node `crypto`; both are standard functions, pinned in
 * tests/ against Python hashlib.
 */
#ifndef ORACLE_HASHES_H
#define ORACLE_HASHES_H
#include <stddef.h>
#include <stdint.h>

void orc_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
void orc_blake2s256(const uint8_t *msg, size_t len, uint8_t out[32]);
/* alg: 0 = sha256, 1 = blake2s256 (gs_hash_alg) */
void orc_hash(int alg, const uint8_t *msg, size_t len, uint8_t out[32]);

#endif
