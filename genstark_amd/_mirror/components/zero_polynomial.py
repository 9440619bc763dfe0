"""Mirror of lib/components/ZeroPolynomial.ts — Z(x) = (x^T - 1) / (x - x_last)."""


class ZeroPolynomial:
    def __init__(self, context):
        self.field = context.field
        self.traceLength = context.traceLength
        position = (self.traceLength - 1) * context.extensionFactor
        self.xAtLastStep = self.field.exp(context.rootOfUnity, position)  # :21-23

    def evaluateAt(self, x):  # :28-34
        f = self.field
        numValue = f.sub(f.exp(x, self.traceLength), f.one)
        denValue = f.sub(x, self.xAtLastStep)
        return f.div(numValue, denValue)

    def inverseOverDomain(self, context):
        """1/Z(x) over the evaluation domain (what CompositionPolynomial.ts:117 divides by).  With a field that has the fused member
        (one kernel, nothing materialised: x^T - 1 takes only extensionFactor distinct values) use it; else the reference's sequence."""
        f = self.field
        domain = context.evaluationDomain
        if getattr(f, 'fusedDomainDivisions', False) and domain.length // self.traceLength <= 32:
            return f.zeroPolyInverses(context.rootOfUnity, domain.length, self.traceLength, self.xAtLastStep)
        z = self.evaluateAll(domain)
        return f.divVectorElements(z['denominators'], z['numerators'])

    def evaluateAll(self, domain):  # :36-44
        f = self.field
        xToTheSteps = f.pluckVector(domain, self.traceLength, domain.length)
        numEvaluations = f.subVectorElements(xToTheSteps, f.one)
        denEvaluations = f.subVectorElements(domain, self.xAtLastStep)
        return {'numerators': numEvaluations, 'denominators': denEvaluations}
