"""Mirror of lib/components/QueryIndexGenerator.ts:20-67 — Fiat-Shamir query positions (host, sha256).

Fully specified in the reference tree, so this must be bit-exact, including the quirk that a bigint is
hashed as Buffer.from(value.toString(16), 'hex') (odd trailing nibble dropped; :61-67).
"""
import ctypes as C

from ...field import sha256_bigint


class QueryIndexGenerator:
    def __init__(self, options, backend=None):
        """backend: when given, the ~700 chained hashes per proof run in the library's host helper gs_pseudorandom_indexes
        (same algorithm, pinned to the reference's own module by tests/test_reference_vectors.py) instead of a Python loop."""
        self.extensionFactor = options['extensionFactor']
        self.exeQueryCount = options['exeQueryCount']
        self.friQueryCount = options['friQueryCount']
        self._native = getattr(backend.lib, 'gs_pseudorandom_indexes', None) if backend is not None else None

    def _indexes(self, seed, count, max_):
        if self._native is None or not isinstance(seed, (bytes, bytearray)) or max_ >= 1 << 63:
            return getPseudorandomIndexes(seed, count, max_, self.extensionFactor)
        maxCount = max_ - max_ // self.extensionFactor if self.extensionFactor else max_
        if maxCount < count:
            raise ValueError(f'Cannot select {count} unique pseudorandom indexes from {max_} values')
        out = (C.c_uint64 * max(count, 1))()
        if self._native(bytes(seed), len(seed), count, max_, self.extensionFactor, out):
            raise ValueError(f'Could not generate {count} pseudorandom indexes')
        return list(out[:count])

    def getExeIndexes(self, seed, domainSize):
        queryCount = min(self.exeQueryCount, domainSize - domainSize // self.extensionFactor)
        return self._indexes(seed, queryCount, domainSize)

    def getFriIndexes(self, seed, columnLength):
        return self._indexes(seed, self.friQueryCount, columnLength)


def getPseudorandomIndexes(seed, count, max_, excludeMultiplesOf=0):
    maxCount = max_ - max_ // excludeMultiplesOf if excludeMultiplesOf else max_
    if maxCount < count:
        raise ValueError(f'Cannot select {count} unique pseudorandom indexes from {max_} values')
    indexes = {}  # insertion-ordered, like the reference's Set
    state = sha256_bigint(seed)
    for i in range(count * 1000):
        index = sha256_bigint(state + i) % max_
        if excludeMultiplesOf and index % excludeMultiplesOf == 0:
            continue
        if index in indexes:
            continue
        indexes[index] = None
        if len(indexes) >= count:
            break
    if len(indexes) < count:
        raise ValueError(f'Could not generate {count} pseudorandom indexes')
    return list(indexes)
