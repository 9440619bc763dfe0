"""Mirror of lib/components/LinearCombination.ts — L(x) = C(x) + sum_j k_j * {P, S, P*x^d, S*x^d}."""


class LinearCombination:
    def __init__(self, seed, compositionDegree, coefficientOffset, context):  # :21-32
        self.field = context.field
        self.seed = seed
        self.rootOfUnity = context.rootOfUnity
        self.domainSize = context.traceLength * context.extensionFactor
        self.coefficientOffset = coefficientOffset
        self.psIncrementalDegree = compositionDegree - context.traceLength
        self.coefficients = None

    def computeMany(self, cEvaluations, pEvaluations, sEvaluations):  # :36-64
        f = self.field
        psEvaluations = [*f.matrixRowsToVectors(pEvaluations), *sEvaluations]
        psEvaluations2 = []
        if self.psIncrementalDegree > 0:
            powerSeed = f.exp(self.rootOfUnity, self.psIncrementalDegree)
            psPowers = f.getPowerSeries(powerSeed, self.domainSize)
            for v in psEvaluations:
                psEvaluations2.append(f.mulVectorElements(v, psPowers))
        allEvaluations = [*psEvaluations, *psEvaluations2]
        coefficients = f.prng(self.seed, self.coefficientOffset + len(allEvaluations)).toValues()
        self.coefficients = coefficients[self.coefficientOffset:]
        psCombination = f.combineManyVectors(allEvaluations, self.coefficients)
        return f.addVectorElements(cEvaluations, psCombination)

    def computeOne(self, x, dValue, pValues, sValues):  # :66-88 (verifier, scalars)
        f = self.field
        psValues = [*pValues, *sValues]
        psValues2 = []
        if self.psIncrementalDegree > 0:
            power = f.exp(x, self.psIncrementalDegree)
            psValues2 = [f.mul(v, power) for v in psValues]
        allValues = [*psValues, *psValues2]
        if self.coefficients is None:
            coefficients = f.prng(self.seed, self.coefficientOffset + len(allValues)).toValues()
            self.coefficients = coefficients[self.coefficientOffset:]
        psCombination = 0
        for v, k in zip(allValues, self.coefficients):
            psCombination = f.add(psCombination, f.mul(v, k))
        return f.add(dValue, psCombination)
