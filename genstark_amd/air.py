"""Host mirror of the `@guildofweavers/air-assembly` AirModule / ProvingContext / VerificationContext
surface genSTARK calls, instantiated for the MiMC AIR of
examples/mimc/mimc128Assembly.ts:28-51:

    (registers 1) (constraints 1) (steps T)
    (static (cycle (prng sha256 0x4d694d43 64)))
    transition:  r0' = r0^3 + k          evaluation:  n0 - (r0^3 + k)

The AirAssembly compiler itself is out of scope; the generic constraint evaluator is a later row
(SURVEY.md section 8f-2).  Members used by the callers: lib/Stark.ts:40,63,67,90-113,161,177,255,302,307;
lib/components/CompositionPolynomial.ts:37,76,84-85,94.
"""
import ctypes as C
import hashlib

from ._abi import GstarkError
from .field import Matrix, PrimeField, Vector

MIMC_SEED = bytes.fromhex('4d694d43')


def sha256_prng(seed, count, field):
    """air-assembly `prng.sha256(seed, count, field)` (examples/mimc/mimc128.ts:15): value_i =
    sha256(uint16_be(i + 1) || seed) mod p.  Restated from memory of the absent package — UNVERIFIED;
    a different generator changes the round constants, not the work."""
    return [int.from_bytes(hashlib.sha256((i + 1).to_bytes(2, 'big') + seed).digest(), 'big') % field.modulus
            for i in range(count)]


class _Context:
    def __init__(self, air, trace_length):
        f = air.field
        self.air, self.field = air, f
        self.traceLength = trace_length
        self.extensionFactor = air.extensionFactor
        self.constraints = [{'degree': 3}]
        self.inputShapes = []
        n = trace_length * air.extensionFactor
        if air._rootOfUnity is None:
            air._rootOfUnity = f.getRootOfUnity(n)
        self.rootOfUnity = air._rootOfUnity
        self.compositionFactor = 4  # 2^ceil(log2(max constraint degree))
        self.roundConstants = air.roundConstants
        nk = len(self.roundConstants)
        if trace_length % nk:
            raise GstarkError(f'trace length must be a multiple of the static register cycle ({nk})')
        self.cycleCount = trace_length // nk
        # cyclic register polynomial K (degree < nk) with K(g^i) = k_i, g of order nk: the register's
        # value on any x is K(x^cycleCount)
        if air._kPoly is None:
            g = f.exp(self.rootOfUnity, air.extensionFactor * self.cycleCount)
            ginv, ninv = f.inv(g), f.inv(nk)
            pw = [1] * nk
            for i in range(1, nk):
                pw[i] = pw[i - 1] * ginv % f.modulus
            air._kPoly = [sum(self.roundConstants[i] * pw[(i * j) % nk] for i in range(nk)) * ninv % f.modulus
                          for j in range(nk)]
        self.kPoly = air._kPoly


class VerificationContext(_Context):
    def evaluateConstraintsAt(self, x, rValues, nValues, hValues):
        """CompositionPolynomial.ts:153 — Q(x) for the single MiMC constraint (scalar, host)."""
        f = self.field
        k, xc = 0, f.exp(x, self.cycleCount)
        for c in reversed(self.kPoly):
            k = (k * xc + c) % f.modulus
        return [f.sub(nValues[0], f.add(f.exp(rValues[0], 3), k))]


class ProvingContext(_Context):
    def __init__(self, air, trace_length, seed):
        super().__init__(air, trace_length)
        f = self.field
        self.seed = seed
        n = trace_length * air.extensionFactor
        nc = trace_length * self.compositionFactor
        self.evaluationDomain = f.getPowerSeries(self.rootOfUnity, n)
        self.compositionDomain = f.getPowerSeries(f.exp(self.rootOfUnity, n // nc), nc)
        self.executionDomain = f.getPowerSeries(f.exp(self.rootOfUnity, air.extensionFactor), trace_length)
        self.secretRegisterTraces = []
        # values of the cyclic register over the composition domain: K at the (nk*4)-th roots of unity
        nk = len(self.roundConstants)
        klen = nk * self.compositionFactor
        wk = f.exp(self.compositionDomain.series_base, self.cycleCount)
        kp = f.newVectorFrom(self.kPoly)
        self._kTable = f.evalPolyAtRoots(kp, f.getPowerSeries(wk, klen))
        self._kTableN = None

    def evaluateTransitionConstraintsOverEvaluationDomain(self, pEvaluations):
        """Q over the EVALUATION domain straight from the extension of P that prove() already holds (lib/Stark.ts:109): the
        reference evaluates Q on the smaller composition domain, interpolates the combination and extends it
        (CompositionPolynomial.ts:76-111) — the combined, degree-adjusted polynomial has degree < |composition domain|, so its
        extension IS what the constraint expression gives on every evaluation-domain point.  Same values, no iNTT + NTT."""
        f = self.field
        n = self.evaluationDomain.length
        if self._kTableN is None:
            klen = len(self.roundConstants) * self.extensionFactor
            wk = f.exp(self.rootOfUnity, self.cycleCount)                       # order 64 * extensionFactor
            self._kTableN = f.evalPolyAtRoots(f.newVectorFrom(self.kPoly), f.getPowerSeries(wk, klen))
        q = Matrix(f.backend, 1, n)
        f.backend.call('gs_mimc_constraints', C.c_void_p(pEvaluations.ptr), n, n // self.traceLength,
                       C.c_void_p(self._kTableN.ptr), self._kTableN.length, C.c_void_p(q.ptr))
        return q

    def generateExecutionTrace(self):
        """lib/Stark.ts:97 — 1 x T matrix; sequential on the host CPU inside the library."""
        f = self.field
        m = Matrix(f.backend, 1, self.traceLength)
        rc = b''.join(f.le(k) for k in self.roundConstants)
        f.backend.call('gs_mimc_trace', f.le(self.seed), rc, len(self.roundConstants), self.traceLength, C.c_void_p(m.ptr))
        return m

    def generateStaticTrace(self):
        f = self.field
        nk = len(self.roundConstants)
        return f.newMatrixFrom([[self.roundConstants[i % nk] for i in range(self.traceLength)]])

    def evaluateTransitionConstraints(self, pPolys):
        """CompositionPolynomial.ts:76 — evaluate P over the composition domain, then the constraint
        expression at every point: Q[j] = P(g*x_j) - (P(x_j)^3 + k(x_j))."""
        f = self.field
        nc = self.compositionDomain.length
        p_comp = f.evalPolysAtRoots(pPolys, self.compositionDomain)
        q = Matrix(f.backend, 1, nc)
        f.backend.call('gs_mimc_constraints', C.c_void_p(p_comp.ptr), nc, nc // self.traceLength,
                       C.c_void_p(self._kTable.ptr), self._kTable.length, C.c_void_p(q.ptr))
        return q


class MimcAir:
    """AirModule for `(export mimc (registers 1) (constraints 1) (steps T) ...)`."""

    def __init__(self, steps, extensionFactor=None, field=None, constantCount=64):
        if steps < constantCount or steps & (steps - 1):
            raise GstarkError('steps must be a power of 2 not smaller than the static register cycle')
        self.field = field or PrimeField()
        self.steps = steps
        self.maxConstraintDegree = 3
        self.traceRegisterCount = 1
        self.secretInputCount = 0
        # README.md:112 — default is the smallest power of 2 greater than 2 * constraint degree
        self.extensionFactor = extensionFactor or 8
        ef = self.extensionFactor
        if ef & (ef - 1) or ef < 2 * 2 ** 2 or ef > 32:
            raise GstarkError('Extension factor must be a power of 2 at least 2x the constraint degree and at most 32')
        self.roundConstants = sha256_prng(MIMC_SEED, constantCount, self.field)
        self._rootOfUnity = None   # constants of the instantiated module, computed on first use
        self._kPoly = None

    def initProvingContext(self, inputs=None, seed=None):
        if not seed:
            raise GstarkError('MiMC AIR requires a seed vector [startValue]')
        return ProvingContext(self, self.steps, seed[0] % self.field.modulus)

    def initVerificationContext(self, inputShapes=None, publicInputs=None):
        return VerificationContext(self, self.steps)


def runMimc(field, steps, roundConstants, seed):
    """examples/mimc/utils.ts:7-15 — the independent control computation the example scripts use."""
    out = [seed]
    for i in range(steps - 1):
        out.append(field.add(field.exp(out[i], 3), roundConstants[i % len(roundConstants)]))
    return out
