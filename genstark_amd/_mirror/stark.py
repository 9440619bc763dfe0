"""Mirror of lib/Stark.ts — the caller of the hot path.  prove() issues the same sequence of
FiniteField / Hash / MerkleTree calls as lib/Stark.ts:81-163; each of those calls lands in one HIP
kernel launch (or a short chain) behind include/gstark.h.  verify() is lib/Stark.ts:167-248."""
import math
import os

from .components import CompositionPolynomial, LinearCombination, LowDegreeProver, QueryIndexGenerator
from ..errors import StarkError
from ..merkle import MerkleTree, createHash
from ..serializer import Serializer
from ..utils import NoopLogger, powLog2, readBigInt, rehashMerkleProofValues, sizeOf

DEFAULT_EXE_QUERY_COUNT = 80   # :13-17
DEFAULT_FRI_QUERY_COUNT = 40
MAX_EXE_QUERY_COUNT = 128
MAX_FRI_QUERY_COUNT = 64
HASH_ALGORITHMS = ['sha256', 'blake2s256']
DEFAULT_HASH_ALGORITHM = 'sha256'


def buildSecurityOptions(options, extensionFactor):  # :318-344
    options = options or {}
    exeQueryCount = options.get('exeQueryCount') or DEFAULT_EXE_QUERY_COUNT
    if not isinstance(exeQueryCount, int) or exeQueryCount < 1 or exeQueryCount > MAX_EXE_QUERY_COUNT:
        raise TypeError(f'Execution sample size must be an integer between 1 and {MAX_EXE_QUERY_COUNT}')
    friQueryCount = options.get('friQueryCount') or DEFAULT_FRI_QUERY_COUNT
    if not isinstance(friQueryCount, int) or friQueryCount < 1 or friQueryCount > MAX_FRI_QUERY_COUNT:
        raise TypeError(f'FRI sample size must be an integer between 1 and {MAX_FRI_QUERY_COUNT}')
    hashAlgorithm = options.get('hashAlgorithm') or DEFAULT_HASH_ALGORITHM
    if hashAlgorithm not in HASH_ALGORITHMS:
        raise TypeError(f'Hash algorithm {hashAlgorithm} is not supported')
    if not extensionFactor:
        raise TypeError('Extension factor is undefined')
    return {'extensionFactor': extensionFactor, 'exeQueryCount': exeQueryCount, 'friQueryCount': friQueryCount,
            'hashAlgorithm': hashAlgorithm}


def validateAssertions(trace, assertions):  # :356-375
    registers, steps = trace.rowCount, trace.colCount
    for a in assertions:
        if a['register'] < 0 or a['register'] >= registers:
            raise ValueError(f"Invalid assertion: register {a['register']} is outside of register bank")
        if a['step'] < 0 or a['step'] >= steps:
            raise ValueError(f"Invalid assertion: step {a['step']} is outside of execution trace")
        if trace.getValue(a['register'], a['step']) != a['value']:
            raise StarkError(f"Assertion at step {a['step']}, register {a['register']} conflicts with execution trace")


class Stark:
    def __init__(self, air, options=None, logger=None):  # :35-58 (the AIR module arrives instantiated)
        self.air = air
        sOptions = buildSecurityOptions(options, air.extensionFactor)
        make_hash = getattr(air.field, 'createHash', None)      # a distributed field brings its own Hash (distributed.py)
        self.hash = make_hash(sOptions['hashAlgorithm']) if make_hash else createHash(sOptions['hashAlgorithm'], air.field.backend)
        self.indexGenerator = QueryIndexGenerator(sOptions, air.field.backend)
        self.serializer = Serializer(air, self.hash.digestSize)
        self.logger = logger or NoopLogger()
        # issue trace-independent device work before the host-side trace recurrence (same proof bytes); GSTARK_PREFETCH=0 disables
        self.prefetch = os.environ.get('GSTARK_PREFETCH', '1') != '0'

    @property
    def securityLevel(self):  # :62-77
        ef = self.air.extensionFactor
        es = powLog2(ef / self.air.maxConstraintDegree, self.indexGenerator.exeQueryCount)
        fs = math.log2(ef) * self.indexGenerator.friQueryCount
        hs = self.hash.digestSize * 4
        return math.floor(min(es, fs, hs))

    # ---- prover (:81-163)
    def prove(self, assertions, inputs=None, seed=None):
        log = self.logger.start('Starting STARK computation')
        if not isinstance(assertions, list):
            raise TypeError('Assertions parameter must be an array')
        if len(assertions) == 0:
            raise TypeError('At least one assertion must be provided')
        # 1 ----- evaluation context
        context = self.air.initProvingContext(inputs, seed)
        field = context.field
        evaluationDomainSize = context.evaluationDomain.length
        if self.prefetch:
            CompositionPolynomial.prefetch(context)      # issue order only: device work that does not need the trace
        log('Set up evaluation context')
        # 2 ----- execution trace
        try:
            executionTrace = context.generateExecutionTrace()
            validateAssertions(executionTrace, assertions)
        except Exception as error:
            raise StarkError('Failed to generate the execution trace', error)
        log('Generated execution trace')
        # 3 ----- P(x) and its low-degree extension
        pPolys = field.interpolateRoots(context.executionDomain, executionTrace)
        log('Computed execution trace polynomials P(x)')
        pEvaluations = field.evalPolysAtRoots(pPolys, context.evaluationDomain)
        log('Low-degree extended P(x) polynomials over evaluation domain')
        # 4 ----- evaluation merkle tree
        sEvaluations = context.secretRegisterTraces
        eVectors = [*field.matrixRowsToVectors(pEvaluations), *sEvaluations]
        hashedEvaluations = self.hash.mergeVectorRows(eVectors)
        log('Serialized evaluations of P(x) and S(x) polynomials')
        eTree = MerkleTree.create(hashedEvaluations, self.hash)
        log('Built evaluation merkle tree')
        # 5 ----- composition polynomial
        cLogger = self.logger.sub('Computing composition polynomial')
        cPoly = CompositionPolynomial(assertions, eTree.root, context, cLogger)
        cEvaluations = cPoly.evaluateAll(pPolys, pEvaluations, context)
        self.logger.done(cLogger)
        log('Computed composition polynomial C(x)')
        # 6 ----- random linear combination
        lCombination = LinearCombination(eTree.root, cPoly.compositionDegree, cPoly.coefficientCount, context)
        lEvaluations = lCombination.computeMany(cEvaluations, pEvaluations, sEvaluations)
        log('Combined P(x) and S(x) evaluations with C(x) evaluations')
        # 7 ----- low-degree proof
        try:
            ldLogger = self.logger.sub('Computing low degree proof')
            ldProver = LowDegreeProver(self.indexGenerator, self.hash, context, ldLogger)
            ldProof = ldProver.prove(lEvaluations, context.evaluationDomain, cPoly.compositionDegree)
            self.logger.done(ldLogger)
            log('Computed low-degree proof')
        except Exception as error:
            raise StarkError('Low degree proof failed', error)
        # 8 ----- spot checks of the evaluation tree
        positions = self.indexGenerator.getExeIndexes(ldProof['lcRoot'], evaluationDomainSize)
        augmentedPositions = self.getAugmentedPositions(positions, evaluationDomainSize)
        eValues = self.mergeValues(eVectors, augmentedPositions)
        eProof = eTree.proveBatch(augmentedPositions)
        eProof['values'] = eValues
        log(f'Computed {len(positions)} evaluation spot checks')
        self.logger.done(log, 'STARK computed')
        return {'evRoot': eTree.root, 'evProof': eProof, 'ldProof': ldProof, 'iShapes': context.inputShapes}

    # ---- verifier (:167-248)
    def verify(self, assertions, proof, publicInputs=None):
        log = self.logger.start('Starting STARK verification')
        if len(assertions) < 1:
            raise TypeError('At least one assertion must be provided')
        eRoot = proof['evRoot']
        extensionFactor = self.air.extensionFactor
        context = self.air.initVerificationContext(proof['iShapes'], publicInputs)
        evaluationDomainSize = context.traceLength * extensionFactor
        cPoly = CompositionPolynomial(assertions, eRoot, context)
        lCombination = LinearCombination(eRoot, cPoly.compositionDegree, cPoly.coefficientCount, context)
        log('Set up evaluation context')
        positions = self.indexGenerator.getExeIndexes(proof['ldProof']['lcRoot'], evaluationDomainSize)
        augmentedPositions = self.getAugmentedPositions(positions, evaluationDomainSize)
        log('Computed positions for evaluation spot checks')
        pEvaluations, sEvaluations = {}, {}
        if len(proof['evProof']['values']) != len(augmentedPositions):
            raise StarkError('malformed proof: the evaluation proof does not hold one leaf per queried position')
        for i, mergedEvaluations in enumerate(proof['evProof']['values']):
            p, s = self.parseValues(mergedEvaluations)
            pEvaluations[augmentedPositions[i]] = p
            sEvaluations[augmentedPositions[i]] = s
        log('Decoded evaluation spot checks')
        try:
            evProof = rehashMerkleProofValues(proof['evProof'], self.hash)
            if not MerkleTree.verifyBatch(eRoot, augmentedPositions, evProof, self.hash):
                raise StarkError('Verification of evaluation Merkle proof failed')
        except StarkError:
            raise
        except Exception as error:
            raise StarkError('Verification of evaluation Merkle proof failed', error)
        log('Verified evaluation merkle proof')
        lcValues = []
        try:
            for step in positions:
                x = context.field.exp(context.rootOfUnity, step)
                pValues = pEvaluations[step]
                nValues = pEvaluations[(step + extensionFactor) % evaluationDomainSize]
                sValues = sEvaluations[step]
                cValue = cPoly.evaluateAt(x, pValues, nValues, sValues, context)
                lcValues.append(lCombination.computeOne(x, cValue, pValues, sValues))
        except StarkError:
            raise
        except (IndexError, KeyError, ValueError) as error:       # a proof whose shape does not fit the AIR
            raise StarkError('malformed proof', error)
        log('Verified transition and boundary constraints')
        try:
            ldProver = LowDegreeProver(self.indexGenerator, self.hash, context)
            ldProver.verify(proof['ldProof'], lcValues, positions, cPoly.compositionDegree)
        except Exception as error:
            raise StarkError('Verification of low degree failed', error)
        log('Verified low-degree proof')
        self.logger.done(log, 'STARK verified')
        return True

    # ---- utilities (:252-313)
    def generateExecutionTrace(self, inputs=None, seed=None):
        context = self.air.initProvingContext(inputs, seed)
        return {'dTrace': context.generateExecutionTrace(), 'sTrace': context.generateStaticTrace()}

    def sizeOf(self, proof):
        return sizeOf(proof, self.air.field.elementSize, self.hash.digestSize)['total']

    def serialize(self, proof):
        return self.serializer.serializeProof(proof)

    def parse(self, buffer):
        return self.serializer.parseProof(buffer)

    def getAugmentedPositions(self, positions, evaluationDomainSize):  # :274-282
        skip = self.air.extensionFactor
        out = {}
        for p in positions:
            out[p] = None
            out[(p + skip) % evaluationDomainSize] = None
        return list(out)

    def mergeValues(self, values, positions):  # :284-296 (one device gather per vector instead of per element)
        columns = [v.valuesAt(positions) for v in values]
        return [b''.join(col[i] for col in columns) for i in range(len(positions))]

    def parseValues(self, buffer):  # :298-313
        es = self.air.field.elementSize
        offset = 0
        pValues = []
        for _ in range(self.air.traceRegisterCount):
            pValues.append(readBigInt(buffer, offset, es))
            offset += es
        sValues = []
        for _ in range(self.air.secretInputCount):
            sValues.append(readBigInt(buffer, offset, es))
            offset += es
        return pValues, sValues
