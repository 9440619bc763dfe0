"""Mirror of lib/components/LowDegreeProver.ts — FRI with 4-to-1 folding.

prove()/fri() (:39-68, :176-221) run on the device: transpose-4, row hashing, Merkle build, cubic
interpolation + evaluation per row; verify() (:70-172) touches a few hundred elements on the host."""
from ...errors import StarkError
from ...merkle import MerkleTree
from ...utils import readBigInt, rehashMerkleProofValues

MAX_REMAINDER_LENGTH = 256  # :12


def getAugmentedPositions(positions, columnLength):  # :302-309
    rowLength = columnLength // 4
    return list(dict.fromkeys(p % rowLength for p in positions))


def getRootOfUnityDegree(rootOfUnity, field):  # :293-300
    result = 1
    while rootOfUnity != 1:
        result *= 2
        rootOfUnity = field.mul(rootOfUnity, rootOfUnity)
    return result


class LowDegreeProver:
    def __init__(self, idxGenerator, hash_, context, logger=None):
        self.field = context.field
        self.polyRowSize = self.field.elementSize * 4
        self.rootOfUnity = context.rootOfUnity
        self.hash = hash_
        self.idxGenerator = idxGenerator
        self.log = logger or (lambda msg: None)

    # ---- prover
    def prove(self, cEvaluations, domain, maxDegreePlus1):  # :39-68
        f = self.field
        if cEvaluations.length < 2 ** 7:
            # getComponentCount (:287-291) yields a negative array length below 2^7 values
            raise StarkError('Invalid array length')
        polyValues = f.transposeVector(cEvaluations, 4)
        polyHashes = self.hash.digestValues(polyValues, self.polyRowSize)
        pTree = MerkleTree.create(polyHashes, self.hash)
        self.log('Built liner combination merkle tree')
        exeQueryPositions = self.idxGenerator.getExeIndexes(pTree.root, domain.length)
        lcPositions = getAugmentedPositions(exeQueryPositions, cEvaluations.length)
        lcProof = pTree.proveBatch(lcPositions)
        lcProof['values'] = polyValues.rowsToBuffers(lcPositions)
        self.log(f'Computed {len(lcPositions)} linear combination spot checks')
        proof = {'lcRoot': pTree.root, 'lcProof': lcProof, 'components': [], 'remainder': []}
        self.fri(pTree, polyValues, maxDegreePlus1, 0, domain, proof)
        return proof

    def fri(self, pTree, polyValues, maxDegreePlus1, depth, domain, result):  # :176-221
        f = self.field
        if polyValues.rowCount * polyValues.colCount <= MAX_REMAINDER_LENGTH:
            rootOfUnity = f.exp(f._omega_of(domain), 4 ** depth)
            remainder = f.joinMatrixRows(f.transposeMatrix(polyValues))
            values = remainder.toValues()
            self.verifyRemainder(values, maxDegreePlus1, rootOfUnity)
            result['remainder'] = values
            self.log(f'Computed FRI remainder of {remainder.length} values')
            return
        xs = f.transposeVector(domain, 4, 4 ** depth)
        polys = f.interpolateQuarticBatch(xs, polyValues)
        specialX = f.prng(pTree.root)
        column = f.evalQuarticBatch(polys, specialX)
        newPolyValues = f.transposeVector(column, 4)
        rowHashes = self.hash.digestValues(newPolyValues, self.polyRowSize)
        cTree = MerkleTree.create(rowHashes, self.hash)
        self.log(f'Computed FRI layer at depth {depth}')
        self.fri(cTree, newPolyValues, maxDegreePlus1 // 4, depth + 1, domain, result)
        # this layer's queries are seeded by the CHILD tree's root (:209)
        positions = self.idxGenerator.getFriIndexes(cTree.root, column.length)
        augmentedPositions = getAugmentedPositions(positions, column.length)
        columnProof = cTree.proveBatch(augmentedPositions)
        columnProof['values'] = newPolyValues.rowsToBuffers(augmentedPositions)
        polyProof = pTree.proveBatch(positions)
        polyProof['values'] = polyValues.rowsToBuffers(positions)
        while len(result['components']) <= depth:
            result['components'].append(None)
        result['components'][depth] = {'columnRoot': cTree.root, 'columnProof': columnProof, 'polyProof': polyProof}

    # ---- verifier
    def verify(self, proof, lcValues, exeQueryPositions, maxDegreePlus1):  # :70-172
        f = self.field
        rootOfUnity = self.rootOfUnity
        columnLength = getRootOfUnityDegree(rootOfUnity, f)
        quarticRootsOfUnity = [1, f.exp(rootOfUnity, columnLength // 4), f.exp(rootOfUnity, columnLength // 2),
                               f.exp(rootOfUnity, columnLength * 3 // 4)]
        # 1 ----- linear combination
        lcProof = proof['lcProof']
        lcPositions = getAugmentedPositions(exeQueryPositions, columnLength)
        lcChecks = self.parseColumnValues(lcProof['values'], exeQueryPositions, lcPositions, columnLength)
        lcProof = rehashMerkleProofValues(lcProof, self.hash)
        if not MerkleTree.verifyBatch(proof['lcRoot'], lcPositions, lcProof, self.hash):
            raise StarkError('Verification of linear combination Merkle proof failed')
        for a, b in zip(lcValues, lcChecks):
            if a != b:
                raise StarkError('Verification of linear combination correctness failed')
        # 2 ----- recursive components
        pRoot = proof['lcRoot']
        columnLength //= 4
        for depth, component in enumerate(proof['components']):
            columnRoot, columnProof, polyProof = component['columnRoot'], component['columnProof'], component['polyProof']
            positions = self.idxGenerator.getFriIndexes(columnRoot, columnLength)
            augmentedPositions = getAugmentedPositions(positions, columnLength)
            columnValues = self.parseColumnValues(columnProof['values'], positions, augmentedPositions, columnLength)
            if not MerkleTree.verifyBatch(columnRoot, augmentedPositions, rehashMerkleProofValues(columnProof, self.hash), self.hash):
                raise StarkError(f'Verification of column Merkle proof failed at depth {depth}')
            polyValues = self.parsePolyValues(polyProof['values'])
            if not MerkleTree.verifyBatch(pRoot, positions, rehashMerkleProofValues(polyProof, self.hash), self.hash):
                raise StarkError(f'Verification of polynomial Merkle proof failed at depth {depth}')
            xs = []
            for p in positions:
                xe = f.exp(rootOfUnity, p)
                xs.append([f.mul(q, xe) for q in quarticRootsOfUnity])
            specialX = f.prng(pRoot)
            polys = f.interpolateQuarticBatch(f.newMatrixFrom(xs), f.newMatrixFrom(polyValues))
            pEvaluations = f.evalQuarticBatch(polys, specialX).toValues()
            for i in range(len(positions)):
                if pEvaluations[i] != columnValues[i]:
                    raise StarkError(f"Degree 4 polynomial didn't evaluate to column value at depth {depth}")
            pRoot = columnRoot
            rootOfUnity = f.exp(rootOfUnity, 4)
            maxDegreePlus1 //= 4
            columnLength //= 4
        # 3 ----- remainder
        if maxDegreePlus1 > len(proof['remainder']):
            raise StarkError('Remainder degree is greater than number of remainder values')
        remainder = f.newVectorFrom(proof['remainder'])
        polyValues = f.transposeVector(remainder, 4)
        polyHashes = self.hash.digestValues(polyValues, self.polyRowSize)
        cTree = MerkleTree.create(polyHashes, self.hash)
        if cTree.root != bytes(pRoot):
            raise StarkError('Remainder values do not match Merkle root of the last column')
        self.verifyRemainder(proof['remainder'], maxDegreePlus1, rootOfUnity)
        return True

    def verifyRemainder(self, remainder, maxDegreePlus1, rootOfUnity):  # :223-252 (<= 256 values)
        f = self.field
        ef = self.idxGenerator.extensionFactor
        positions = [i for i in range(len(remainder)) if not ef or i % ef]
        domain, x = [], 1
        for _ in range(len(remainder)):
            domain.append(x)
            x = f.mul(x, rootOfUnity)
        xs = [domain[positions[i]] for i in range(maxDegreePlus1)]
        ys = [remainder[positions[i]] for i in range(maxDegreePlus1)]
        poly = f.interpolateValues(xs, ys)
        rest = positions[maxDegreePlus1:]
        values = f.evalPolyAtMany(poly, [domain[p] for p in rest])
        for p, v in zip(rest, values):
            if v != remainder[p]:
                raise StarkError(f'Remainder is not a valid degree {maxDegreePlus1 - 1} polynomial')

    # ---- parsers (:256-282)
    def parsePolyValues(self, buffers):
        es = self.field.elementSize
        return [[readBigInt(b, i * es, es) for i in range(4)] for b in buffers]

    def parseColumnValues(self, buffers, positions, augmentedPositions, columnLength):
        rowLength = columnLength // 4
        es = self.field.elementSize
        out = []
        for position in positions:
            idx = augmentedPositions.index(position % rowLength)
            out.append(readBigInt(buffers[idx], (position // rowLength) * es, es))
        return out
