"""Mirror of lib/components/CompositionPolynomial.ts — C(x) = D(x) + B(x), D = Q / Z.

evaluateAll (:71-146) is the prover's heaviest phase: every step below is one device call through
the galois-shaped field object (NTT / pointwise / batch inversion kernels)."""
import math

from .boundary_constraints import BoundaryConstraints
from .zero_polynomial import ZeroPolynomial


def getCombinationDegree(constraints, traceLength):  # :196-204
    maxConstraintDegree = max([1] + [c['degree'] for c in constraints])
    return 2 ** math.ceil(math.log2(maxConstraintDegree)) * traceLength


def groupTransitionConstraints(constraints, traceLength):  # :206-225
    groups = {}
    for i, c in enumerate(constraints):
        groups.setdefault(c['degree'] * traceLength, []).append(i)
    return [{'degree': d, 'indexes': ix} for d, ix in groups.items()]


class CompositionPolynomial:
    def __init__(self, assertions, seed, context, logger=None):  # :29-61
        f = self.field = context.field
        self.bPoly = BoundaryConstraints(assertions, context)
        self.zPoly = ZeroPolynomial(context)
        self.log = logger or (lambda msg: None)
        self.combinationDegree = getCombinationDegree(context.constraints, context.traceLength)
        self.compositionDegree = max(self.combinationDegree - context.traceLength, context.traceLength)
        self.constraintGroups = groupTransitionConstraints(context.constraints, context.traceLength)
        dCoefficientCount = len(context.constraints)
        for g in self.constraintGroups:
            if g['degree'] < self.combinationDegree:
                dCoefficientCount += len(g['indexes'])
        bCoefficientCount = self.bPoly.count
        if self.compositionDegree > context.traceLength:
            bCoefficientCount *= 2
        coefficients = f.prng(seed, dCoefficientCount + bCoefficientCount).toValues()
        self.dCoefficients = coefficients[:dCoefficientCount]
        self.bCoefficients = coefficients[dCoefficientCount:]

    @staticmethod
    def prefetch(context):
        """ISSUE ORDER ONLY (no counterpart in the reference, same values): enqueue the device work of evaluateAll that
        depends on nothing but the context — Z(x) and its inverses, the degree-adjustment power series — before the caller
        starts the host-side trace recurrence, so the GPU computes it while that one core is busy (lib/Stark.ts:97 blocks for
        9 ms at 2^20 MiMC steps).  evaluateAll picks the vectors up from context.prefetched and logs the same phases."""
        f = context.field
        pre = {}
        combinationDegree = getCombinationDegree(context.constraints, context.traceLength)
        compositionDegree = max(combinationDegree - context.traceLength, context.traceLength)
        pre['zInverses'] = ZeroPolynomial(context).inverseOverDomain(context)
        compositionFactor = context.evaluationDomain.length // context.compositionDomain.length
        compositionRou = f.exp(context.rootOfUnity, compositionFactor)
        direct = CompositionPolynomial.directRoute(context)
        for g in groupTransitionConstraints(context.constraints, context.traceLength):
            if g['degree'] != combinationDegree:
                seed = f.exp(context.rootOfUnity if direct else compositionRou, combinationDegree - g['degree'])
                pre['qPowers', g['degree']] = f.getPowerSeries(seed, (context.evaluationDomain if direct else context.compositionDomain).length)
        if compositionDegree > context.traceLength:
            seed = f.exp(context.rootOfUnity, compositionDegree - context.traceLength)
            pre['psbPowers'] = f.getPowerSeries(seed, context.evaluationDomain.length)
        context.prefetched = pre

    @staticmethod
    def directRoute(context):
        """True when Q can be evaluated on the evaluation domain directly (the context offers it, the field is not distributed):
        evaluateAll then skips the interpolation + extension of CompositionPolynomial.ts:109-110 — same values."""
        return hasattr(context, 'evaluateTransitionConstraintsOverEvaluationDomain') and getattr(context.field, 'fusedDomainDivisions', False)

    @property
    def coefficientCount(self):
        return len(self.dCoefficients) + len(self.bCoefficients)

    def evaluateAll(self, pPolys, pEvaluations, context):  # :71-146
        f = self.field
        pre = getattr(context, 'prefetched', None) or {}
        # 1 ----- transition constraints over the composition domain (or, directRoute, over the evaluation domain)
        direct = self.directRoute(context)
        qEvaluations = context.evaluateTransitionConstraintsOverEvaluationDomain(pEvaluations) if direct else context.evaluateTransitionConstraints(pPolys)
        self.log('Computed transition constraint polynomials Q(x)')
        # 2 ----- adjust degrees
        compositionFactor = 1 if direct else context.evaluationDomain.length // context.compositionDomain.length
        compositionRou = f.exp(context.rootOfUnity, compositionFactor)
        qDomainLength = (context.evaluationDomain if direct else context.compositionDomain).length
        qaEvaluations = f.matrixRowsToVectors(qEvaluations)
        for g in self.constraintGroups:
            if g['degree'] == self.combinationDegree:
                continue
            powerSeed = f.exp(compositionRou, self.combinationDegree - g['degree'])
            powers = pre.get(('qPowers', g['degree']))
            if powers is None:
                powers = f.getPowerSeries(powerSeed, qDomainLength)
            for i in g['indexes']:
                qaEvaluations.append(f.mulVectorElements(qaEvaluations[i], powers))
        self.log('Adjusted degrees of Q(x) polynomials')
        # 3 ----- merge into one polynomial and extend to the evaluation domain
        qcEvaluations = f.combineManyVectors(qaEvaluations, self.dCoefficients)
        self.log('Computed linear combination of Q(x) polynomials')
        if direct:
            qeEvaluations = qcEvaluations
        else:
            qcPoly = f.interpolateRoots(context.compositionDomain, qcEvaluations)
            qeEvaluations = f.evalPolyAtRoots(qcPoly, context.evaluationDomain)
        self.log('Performed low degree extensions of Q(x) polynomial')
        # 4 ----- D(x) = Q(x) / Z(x)
        zInverses = pre.get('zInverses')
        self.log('Computed Z(x) polynomial')
        if zInverses is None:
            zInverses = self.zPoly.inverseOverDomain(context)                                       # 1/Z = den/num (:111-117)
        self.log('Computed Z(x) inverses')
        dEvaluations = f.mulVectorElements(qeEvaluations, zInverses)
        self.log('Computed D(x) polynomial')
        # 5 ----- boundary constraints
        bEvaluations = self.bPoly.evaluateAll(pEvaluations, context.evaluationDomain)
        self.log('Computed boundary constraint polynomials B(x)')
        # 6 ----- adjust degrees of boundary constraints
        baEvaluations = f.matrixRowsToVectors(bEvaluations)
        bIncrementalDegree = self.compositionDegree - context.traceLength
        if bIncrementalDegree > 0:
            powerSeed = f.exp(context.rootOfUnity, bIncrementalDegree)
            psbPowers = pre.get('psbPowers')
            if psbPowers is None:
                psbPowers = f.getPowerSeries(powerSeed, context.evaluationDomain.length)
            for i in range(self.bPoly.count):
                baEvaluations.append(f.mulVectorElements(baEvaluations[i], psbPowers))
        self.log('Adjusted degrees of B(x) polynomials')
        # 7 ----- merge
        bcEvaluations = f.combineManyVectors(baEvaluations, self.bCoefficients)
        self.log('Computed linear combination of B(x) polynomials')
        return f.addVectorElements(dEvaluations, bcEvaluations)

    def evaluateAt(self, x, pValues, nValues, hValues, context):  # :150-191 (verifier, scalars)
        f = self.field
        qValues = context.evaluateConstraintsAt(x, pValues, nValues, hValues)
        for g in self.constraintGroups:
            if g['degree'] == self.combinationDegree:
                continue
            power = f.exp(x, self.combinationDegree - g['degree'])
            for i in g['indexes']:
                qValues.append(f.mul(qValues[i], power))
        qcValue = 0
        for v, k in zip(qValues, self.dCoefficients):
            qcValue = f.add(qcValue, f.mul(v, k))
        dValue = f.div(qcValue, self.zPoly.evaluateAt(x))
        bValues = self.bPoly.evaluateAt(pValues, x)
        bIncrementalDegree = self.compositionDegree - context.traceLength
        if bIncrementalDegree > 0:
            power = f.exp(x, bIncrementalDegree)
            for i in range(self.bPoly.count):
                bValues.append(f.mul(bValues[i], power))
        bValue = 0
        for v, k in zip(bValues, self.bCoefficients):
            bValue = f.add(bValue, f.mul(v, k))
        return f.add(dValue, bValue)
