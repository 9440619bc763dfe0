"""Mirror of lib/components/BoundaryConstraints.ts — B_r(x) = (P_r(x) - I_r(x)) / Z_r(x)."""


class BoundaryConstraints:
    def __init__(self, assertions, context):  # :15-45
        f = self.field = context.field
        rData = {}
        self.rootOfUnity = context.rootOfUnity
        self.rootIndexes = {}             # register -> positions of its divisor's roots in the evaluation domain
        for c in assertions:
            self.rootIndexes.setdefault(c['register'], []).append(c['step'] * context.extensionFactor)
            x = f.exp(context.rootOfUnity, c['step'] * context.extensionFactor)
            zPoly = f.newVectorFrom([f.neg(x), f.one])
            data = rData.get(c['register'])
            if data:
                data['xs'].append(x)
                data['ys'].append(c['value'])
                data['zPoly'] = f.mulPolys(data['zPoly'], zPoly)
            else:
                rData[c['register']] = {'xs': [x], 'ys': [c['value']], 'zPoly': zPoly}
        self.polys = {}
        for register, data in rData.items():
            iPoly = f.interpolate(f.newVectorFrom(data['xs']), f.newVectorFrom(data['ys']))
            self.polys[register] = {'iPoly': iPoly, 'zPoly': data['zPoly']}

    @property
    def count(self):
        return len(self.polys)

    def evaluateAt(self, pEvaluations, x):  # :55-69
        f = self.field
        out = []
        for register, c in self.polys.items():
            z = f.evalPolyAt(c['zPoly'], x)
            i = f.evalPolyAt(c['iPoly'], x)
            out.append(f.div(f.sub(pEvaluations[register], i), z))
        return out

    def evaluateAll(self, pEvaluations, domain):  # :71-95
        f = self.field
        pVectors = f.matrixRowsToVectors(pEvaluations)
        pValues, iPolys, zPolys = [], [], []
        for register, c in self.polys.items():
            pValues.append(pVectors[register])
            iPolys.append(c['iPoly'])
            zPolys.append(c['zPoly'])
        iValues = f.evalPolysAtRoots(f.newMatrixFromVectors(iPolys), domain)
        piValues = f.subMatrixElementsFromVectors(pValues, iValues)
        roots = [self.rootIndexes[register] for register in self.polys]
        if getattr(f, 'fusedDomainDivisions', False) and max(len(r) for r in roots) <= f.MAX_DOMAIN_ROOTS and getattr(domain, 'series_base', None) == self.rootOfUnity:
            # the divisors' roots are domain points: their inverses are look-ups in the domain's table 1/(omega^j - 1) — the same
            # values as the two lines below, without the batch inversion
            return f.divByDomainRoots(piValues, self.rootOfUnity, roots)
        zValues = f.evalPolysAtRoots(f.newMatrixFromVectors(zPolys), domain)
        return f.divMatrixElements(piValues, zValues)
