"""Mirror of lib/components/QueryIndexGenerator.ts:20-67 — Fiat-Shamir query positions (host, sha256).

Fully specified in the reference tree, so this must be bit-exact, including the quirk that a bigint is
hashed as Buffer.from(value.toString(16), 'hex') (odd trailing nibble dropped; :61-67).
"""
from ..field import sha256_bigint


class QueryIndexGenerator:
    def __init__(self, options):
        self.extensionFactor = options['extensionFactor']
        self.exeQueryCount = options['exeQueryCount']
        self.friQueryCount = options['friQueryCount']

    def getExeIndexes(self, seed, domainSize):
        queryCount = min(self.exeQueryCount, domainSize - domainSize // self.extensionFactor)
        return getPseudorandomIndexes(seed, queryCount, domainSize, self.extensionFactor)

    def getFriIndexes(self, seed, columnLength):
        return getPseudorandomIndexes(seed, self.friQueryCount, columnLength, self.extensionFactor)


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
