"""Generic AIR support: the AirModule / ProvingContext / VerificationContext surface of `@guildofweavers/air-assembly`
 for AIRs whose transition function and constraint evaluator are given as
arithmetic expressions over trace registers, next-row registers and cyclic static registers — the subset of AirAssembly
(`add sub mul exp prod vector get load.trace load.static load.const`) the reference's Rescue and Poseidon examples use
(examples/rescue/hash4x128.ts:83-108, assembly/lib128.aa:15-37).

Expressions are compiled to straight-line programs for the register machine of include/gstark.h; the device evaluates the
constraint program at every composition-domain point (`gs_air_constraints`), one host core runs the transition program
step by step (`gs_air_trace`), and the verifier interprets the same program on Python integers.
"""
import ctypes as C

from ._abi import GstarkError
from .field import Matrix, PrimeField, Vector

OP_LOADC, OP_LOADR, OP_LOADN, OP_LOADS, OP_ADD, OP_SUB, OP_MUL, OP_POW, OP_POWC, OP_OUT = range(10)
MAX_VM_REGS = 64


class Expr:
    """A node of an arithmetic expression DAG.  Build with reg(i), nxt(i), static(i), const(v) and + - * **."""
    __slots__ = ('kind', 'args')

    def __init__(self, kind, *args):
        self.kind, self.args = kind, args

    @staticmethod
    def wrap(v):
        return v if isinstance(v, Expr) else Expr('const', int(v))

    def __add__(self, o): return Expr('add', self, Expr.wrap(o))
    def __radd__(self, o): return Expr('add', Expr.wrap(o), self)
    def __sub__(self, o): return Expr('sub', self, Expr.wrap(o))
    def __rsub__(self, o): return Expr('sub', Expr.wrap(o), self)
    def __mul__(self, o): return Expr('mul', self, Expr.wrap(o))
    def __rmul__(self, o): return Expr('mul', Expr.wrap(o), self)

    def __pow__(self, e):
        e = int(e)
        if e < 0:
            raise GstarkError('negative exponents must be rewritten as e mod (p - 1)')
        return Expr('pow', self, e)


def reg(i): return Expr('reg', i)
def nxt(i): return Expr('next', i)
def static(i): return Expr('static', i)
def const(v): return Expr('const', int(v))


def mat_vec(m, v):
    """`mds # vector` of AirScript (examples/rescue/hash4x128.ts:96-97)."""
    out = []
    for row in m:
        acc = None
        for a, b in zip(row, v):
            t = Expr.wrap(a) * b
            acc = t if acc is None else acc + t
        out.append(acc)
    return out


class Program:
    """Straight-line code {op, dst, a, b} + constant pool, with scratch registers reused after their last use."""

    def __init__(self, outputs, modulus, share_consts_with=None):
        """share_consts_with: another Program whose constant pool this one extends (indexes of the other stay valid)."""
        self.modulus = modulus
        order, index = [], {}

        def visit(e):
            key = id(e)
            if key in index:
                return index[key]
            ins = [visit(a) for a in e.args if isinstance(a, Expr)]
            index[key] = len(order)
            order.append((e, ins))
            return index[key]

        outs = [visit(Expr.wrap(o)) for o in outputs]
        # list scheduling: exponentiations are held back until nothing else is ready, so independent ones (the S-box applied to
        # every register of a round) end up next to each other, grouped by exponent — the device trace generator runs such a group
        # as interleaved square-and-multiply chains (k_air_trace_segments: one thread per segment is latency-bound, four
        # independent chains issue in the time of two)
        level = [0] * len(order)             # exponentiations on the longest dependency path ending at (and including) a node
        for n, (e, ins) in enumerate(order):
            level[n] = max([level[i] for i in ins], default=0) + (1 if e.kind == 'pow' else 0)
        groups = {}
        for n, (e, ins) in enumerate(order):
            if e.kind == 'pow':
                groups.setdefault((level[n], e.args[1]), []).append(n)
        schedule, done = [], set()

        def emit(n):                         # n with everything it needs, in the original relative order
            if n in done:
                return
            for i in sorted(set(order[n][1])):
                emit(i)
            done.add(n)
            schedule.append(n)
        for n, (e, ins) in enumerate(order):
            if n in done:
                continue
            if e.kind != 'pow':
                emit(n)
                continue
            members = [m for m in groups[(level[n], e.args[1])] if m not in done][:8]    # at most 8 at a time: register pressure
            for mbr in members:              # first everything the whole group needs ...
                for i in sorted(set(order[mbr][1])):
                    emit(i)
            for mbr in members:              # ... then its members back to back
                done.add(mbr)
                schedule.append(mbr)
        assert len(schedule) == len(order)
        position = {old: new for new, old in enumerate(schedule)}
        order = [(order[old][0], [position[i] for i in order[old][1]]) for old in schedule]
        outs = [position[o] for o in outs]
        last_use = {}
        for n, (_, ins) in enumerate(order):
            for i in ins:
                last_use[i] = n
        for o in outs:
            last_use[o] = len(order)
        if share_consts_with is not None:
            self.consts, const_ix = share_consts_with.consts, share_consts_with._const_ix
        else:
            self.consts, const_ix = [], {}
        self._const_ix = const_ix

        def cidx(v):
            v %= modulus
            if v not in const_ix:
                const_ix[v] = len(self.consts)
                self.consts.append(v)
            return const_ix[v]

        free, where, nregs, code = [], {}, 0, []
        for n, (e, ins) in enumerate(order):
            srcs = [where[i] for i in ins]
            for i in ins:                      # operands dying here free their register for the destination
                if last_use[i] == n and where[i] not in free:
                    free.append(where[i])
            if free:
                dst = free.pop()
            else:
                dst, nregs = nregs, nregs + 1
            where[n] = dst
            k = e.kind
            if k == 'const':
                code.append((OP_LOADC, dst, cidx(e.args[0] % modulus), 0))
            elif k == 'reg':
                code.append((OP_LOADR, dst, e.args[0], 0))
            elif k == 'next':
                code.append((OP_LOADN, dst, e.args[0], 0))
            elif k == 'static':
                code.append((OP_LOADS, dst, e.args[0], 0))
            elif k in ('add', 'sub', 'mul'):
                code.append(({'add': OP_ADD, 'sub': OP_SUB, 'mul': OP_MUL}[k], dst, srcs[0], srcs[1]))
            elif k == 'pow':
                ex = e.args[1]
                if ex < (1 << 32):
                    code.append((OP_POW, dst, srcs[0], ex))
                else:
                    code.append((OP_POWC, dst, srcs[0], cidx_raw(self, const_ix, ex)))
            else:
                raise GstarkError(f'unknown expression kind {k}')
        for k, o in enumerate(outs):
            code.append((OP_OUT, k, where[o], 0))
        if nregs > MAX_VM_REGS:
            raise GstarkError(f'program needs {nregs} scratch registers (max {MAX_VM_REGS})')
        self.code, self.nregs, self.nout = code, max(nregs, 1), len(outs)

    # ---- marshalling for the C ABI
    def abi_args(self, element_size=16):
        # marshalled once per (element size, state of the program): a prover asks for these per proof
        key = (element_size, len(self.code), len(self.consts))
        hit = getattr(self, '_abi', None)
        if hit is None or hit[0] != key:
            flat = (C.c_uint32 * (4 * len(self.code)))(*[w for ins in self.code for w in ins])
            consts = b''.join(int(v).to_bytes(element_size, 'little') for v in self.consts) or bytes(element_size)
            hit = self._abi = (key, flat, consts)
        return hit[1], len(self.code), hit[2], len(self.consts), self.nregs

    # ---- host interpreter (verifier side, Python integers)
    def run(self, cur, nxt_row, statics):
        p = self.modulus
        vm, out = [0] * self.nregs, [0] * self.nout
        for op, d, a, b in self.code:
            if op == OP_LOADC: vm[d] = self.consts[a]
            elif op == OP_LOADR: vm[d] = cur[a]
            elif op == OP_LOADN: vm[d] = nxt_row[a]
            elif op == OP_LOADS: vm[d] = statics[a]
            elif op == OP_ADD: vm[d] = (vm[a] + vm[b]) % p
            elif op == OP_SUB: vm[d] = (vm[a] - vm[b]) % p
            elif op == OP_MUL: vm[d] = (vm[a] * vm[b]) % p
            elif op == OP_POW: vm[d] = pow(vm[a], b, p)
            elif op == OP_POWC: vm[d] = pow(vm[a], self.consts[b], p)
            else: out[d] = vm[a]
        return out


def cidx_raw(prog, const_ix, v):
    key = ('raw', v)
    if key not in const_ix:
        const_ix[key] = len(prog.consts)
        prog.consts.append(v)
    return const_ix[key]


class _Context:
    def __init__(self, air):
        f = air.field
        self.air, self.field = air, f
        self.traceLength, self.extensionFactor = air.steps, air.extensionFactor
        self.constraints = [{'degree': d} for d in air.constraintDegrees]
        self.inputShapes = []
        self.rootOfUnity = air.rootOfUnity
        self.compositionFactor = air.compositionFactor

    def _static_polys(self):
        """Host coefficients of each cyclic register's polynomial K_s (degree < period): value at x is K_s(x^(T/period))."""
        air, f = self.air, self.field
        if air._staticPolys is None:
            polys = []
            for values in air.staticRegisters:
                m = len(values)
                g = f.exp(self.rootOfUnity, self.extensionFactor * (self.traceLength // m))   # order m
                polys.append(f.interpolateRoots(f.getPowerSeries(g, m), f.newVectorFrom(values)).toValues())
            air._staticPolys = polys
        return air._staticPolys


class GenericVerificationContext(_Context):
    def evaluateConstraintsAt(self, x, rValues, nValues, hValues):   # CompositionPolynomial.ts:153
        f = self.field
        statics = []
        for values, poly in zip(self.air.staticRegisters, self._static_polys()):
            xc, k = f.exp(x, self.traceLength // len(values)), 0
            for c in reversed(poly):
                k = (k * xc + c) % f.modulus
            statics.append(k)
        if len(hValues) != self.air.secretInputCount:
            raise GstarkError('wrong number of secret register values')
        return self.air.evaluationProgram.run(rValues, nValues, statics + [v % f.modulus for v in hValues])


class PackedColumn:
    """A column of field elements as packed little-endian words of the field's elementSize (16 bytes unless said otherwise): what
    a secret register of 2^16 values should travel as (packing 262 144 Python integers per proof costs more than the proof)."""

    def __init__(self, data, element_size=16):
        if len(data) % element_size:
            raise GstarkError(f'packed column: length must be a multiple of {element_size} bytes')
        self.data, self.elementSize = bytes(data), element_size

    def __len__(self):
        return len(self.data) // self.elementSize

    def ints(self):
        es = self.elementSize
        return [int.from_bytes(self.data[i:i + es], 'little') for i in range(0, len(self.data), es)]

    @staticmethod
    def of(values, modulus, element_size=16):
        if isinstance(values, PackedColumn):
            if values.elementSize != element_size:
                raise GstarkError(f'packed column of {values.elementSize}-byte elements in a field of {element_size}-byte elements')
            return values
        return PackedColumn(b''.join(int(v % modulus).to_bytes(element_size, 'little') for v in values), element_size)


class GenericProvingContext(_Context):
    def __init__(self, air, first_rows, secret_values=None):
        super().__init__(air)
        f = self.field
        secret_values = list(secret_values or [])
        if len(secret_values) != air.secretInputCount:
            raise GstarkError(f'the AIR has {air.secretInputCount} secret registers: `inputs` must hold one list of values for each')
        for values in secret_values:
            if not len(values) or len(values) & (len(values) - 1) or self.traceLength % len(values):
                raise GstarkError('a secret register holds a power-of-2 number of values dividing the trace length (it repeats cyclically)')
        self.secretValues = [PackedColumn.of(values, f.modulus, f.elementSize) for values in secret_values]
        n, nc = self.traceLength * self.extensionFactor, self.traceLength * self.compositionFactor
        self.firstRows = [[v % f.modulus for v in row] for row in first_rows]
        self.firstRow = self.firstRows[0]
        self.evaluationDomain = f.getPowerSeries(self.rootOfUnity, n)
        self.compositionDomain = f.getPowerSeries(f.exp(self.rootOfUnity, n // nc), nc)
        self.executionDomain = f.getPowerSeries(f.exp(self.rootOfUnity, self.extensionFactor), self.traceLength)
        # secret registers (lib/Stark.ts:113): cyclic like the static ones, but known to the prover only — their low-degree
        # extension is committed next to P(x) and their values reach the verifier inside the proof's leaves
        secret_polys = []                                        # device vectors of K_s' coefficients
        for col in self.secretValues:
            m = len(col)
            g = f.exp(self.rootOfUnity, self.extensionFactor * (self.traceLength // m))
            values = Vector(f.backend, m)
            f.backend.upload(values.ptr, col.data)
            secret_polys.append(f.interpolateRoots(f.getPowerSeries(g, m), values))
        self.secretRegisterTraces = []
        for col, poly in zip(self.secretValues, secret_polys):
            stride = self.traceLength // len(col)                # S(x) = K(x^stride): K's coefficients at multiples of stride
            if stride > 1:
                coeffs = [0] * self.traceLength
                coeffs[::stride] = poly.toValues()
                poly = f.newVectorFrom(coeffs)
            self.secretRegisterTraces.append(f.evalPolyAtRoots(poly, self.evaluationDomain))
        # static registers over the composition domain: K_s at the (period * compositionFactor)-th roots of unity.  The PUBLIC registers'
        # tables are constants of the AIR (computed once, kept on the device with it); the secret ones are this proof's
        def table(m, poly):
            ln = m * self.compositionFactor
            wk = f.exp(self.compositionDomain.series_base, self.traceLength // m)
            return ln, f.evalPolyAtRoots(poly, f.getPowerSeries(wk, ln))
        if getattr(air, '_publicTables', None) is None or air._publicTables[0] is not f.backend:
            air._publicTables = (f.backend, [table(len(v), f.newVectorFrom(p)) for v, p in zip(air.staticRegisters, self._static_polys())])
        tables = air._publicTables[1] + [table(len(c), p) for c, p in zip(self.secretValues, secret_polys)]
        self._staticLens = [ln for ln, _ in tables]
        self._staticTables = Vector(f.backend, max(sum(self._staticLens), 1))
        off = 0
        for ln, tab in tables:
            f.backend.call('gs_copy', C.c_void_p(self._staticTables.ptr + off * f.elementSize), C.c_void_p(tab.ptr), ln * f.elementSize)
            off += ln

    def staticValuesPacked(self):
        """(bytes, periods) of every static register's values for the trace generators: public ones, then the secret columns."""
        air, f = self.air, self.field
        if getattr(air, '_publicPacked', None) is None:      # the public registers' values are constants of the AIR
            air._publicPacked = b''.join([f.le(v % f.modulus) for values in air.staticRegisters for v in values])
        packed = air._publicPacked + b''.join(c.data for c in self.secretValues)
        periods = [len(v) for v in air.staticRegisters] + [len(c) for c in self.secretValues]
        return packed or bytes(f.elementSize), periods

    def generateExecutionTrace(self):   # lib/Stark.ts:97
        air, f = self.air, self.field
        code, ninstr, consts, nconsts, nregs = air.transitionProgram.abi_args(f.elementSize)
        m = Matrix(f.backend, air.traceRegisterCount, self.traceLength)
        svals, plist = self.staticValuesPacked()
        statics = plist
        periods = (C.c_uint32 * max(len(plist), 1))(*plist)
        if air.segmentLength is None:
            f.backend.call('gs_air_trace', code, ninstr, consts, nconsts, nregs, air.traceRegisterCount, svals, periods,
                           len(statics), b''.join(f.le(v) for v in self.firstRow), self.traceLength, C.c_void_p(m.ptr))
        else:
            icode, ininstr = None, 0
            if air.initProgram is not None:
                icode, ininstr = air.initProgram.abi_args(f.elementSize)[:2]
            f.backend.call('gs_air_trace_segments', code, ninstr, icode, ininstr, consts, nconsts, nregs, air.traceRegisterCount, svals, periods,
                           len(statics), b''.join(f.le(v) for row in self.firstRows for v in row), len(self.firstRows),
                           air.segmentLength, C.c_void_p(m.ptr))
        return m

    def generateStaticTrace(self):
        f = self.field
        cols = list(self.air.staticRegisters) + [c.ints() for c in self.secretValues]
        return f.newMatrixFrom([[v[i % len(v)] for i in range(self.traceLength)] for v in cols])

    def evaluateTransitionConstraints(self, pPolys):   # CompositionPolynomial.ts:76
        air, f = self.air, self.field
        nc = self.compositionDomain.length
        p_comp = f.evalPolysAtRoots(pPolys, self.compositionDomain)
        code, ninstr, consts, nconsts, nregs = air.evaluationProgram.abi_args(f.elementSize)
        q = Matrix(f.backend, len(air.constraintDegrees), nc)
        lens = (C.c_uint64 * max(len(self._staticLens), 1))(*self._staticLens)
        f.backend.call('gs_air_constraints', code, ninstr, consts, nconsts, nregs, air.traceRegisterCount, len(air.constraintDegrees),
                       C.c_void_p(p_comp.ptr), nc, nc // self.traceLength, C.c_void_p(self._staticTables.ptr), lens,
                       len(self._staticLens), C.c_void_p(q.ptr))
        return q


class GenericAir:
    """AirModule built from expression-level transition / evaluation functions.

    transition(r, k)    -> list of `registers` Expr: next row from current row r[i] and static registers k[j]
    evaluation(r, n, k) -> list of Expr, one per constraint: must vanish on every step but the last
    init(seed)          -> first row (list of ints)"""

    def __init__(self, steps, registers, constraintDegrees, staticRegisters, transition, evaluation, init, extensionFactor=None,
                 field=None, segmentLength=None, initExpr=None, secretRegisters=0, maskSegments=True):
        """segmentLength = L splits the trace into steps/L independent runs (AirScript's `for each (input)` loop over several
        inputs): `seed` is then a list of steps/L per-segment seeds, segment s starts from init(seed[s]) at step s*L, the
        transition constraints are switched off on the last step of every segment by one more cyclic static register
        (degree + 1), and the trace is generated on the device, one thread per segment (gs_air_trace_segments).
        initExpr(x) -> list of `registers` Expr over the raw inputs x[i] = reg(i): the `init { ... }` block as expressions, run
        by the same device thread before the segment's first step (init(seed) then only returns the raw inputs, zero-padded).
        secretRegisters = S: the last S entries of the static-register list `k` seen by transition / evaluation are SECRET
        registers (AirScript `secret input`, examples/mimc/mimc128.ts:38): cyclic columns supplied per proof through
        `prove(assertions, inputs, seed)` (inputs = S lists of values), committed with the trace and read back by the verifier
        from the proof (lib/Stark.ts:113-114, 284-313)."""
        self.field = field or PrimeField()
        f = self.field
        self.segmentLength = segmentLength
        if segmentLength is not None and (segmentLength < 2 or segmentLength & (segmentLength - 1) or steps % segmentLength):
            raise GstarkError('segment length must be a power of 2 dividing the trace length')
        # maskSegments=False: the AIR's own transition restarts every segment from init(segment seed) (AirAssembly's mask /
        # input registers, e.g. assembly/lib128.aa:99-108), segmentLength then only tells the trace generator that the
        # segments can be computed independently, one device thread each
        if segmentLength is not None and maskSegments:
            if secretRegisters:
                raise GstarkError('secret registers and masked trace segments cannot be combined')
            mask_index = len(staticRegisters)
            staticRegisters = list(staticRegisters) + [[0] * (segmentLength - 1) + [1]]
            constraintDegrees = [d + 1 for d in constraintDegrees]
            inner = evaluation

            def evaluation(r, n, k, inner=inner, mask_index=mask_index):
                live = 1 - k[mask_index]
                return [live * e for e in inner(r, n, k[:mask_index])]
            inner_t = transition
            transition = lambda r, k, inner_t=inner_t, mask_index=mask_index: inner_t(r, k[:mask_index])
        if steps & (steps - 1) or steps < 2:
            raise GstarkError('steps must be a power of 2')
        for values in staticRegisters:
            if len(values) & (len(values) - 1) or steps % len(values):
                raise GstarkError('static register cycles must be powers of 2 dividing the trace length')
        self.steps, self.traceRegisterCount, self.secretInputCount = steps, registers, int(secretRegisters)
        self.constraintDegrees = list(constraintDegrees)
        self.maxConstraintDegree = max(self.constraintDegrees)
        self.compositionFactor = 1 << (self.maxConstraintDegree - 1).bit_length()
        # README.md:112 — a power of 2 at least 2x the constraint degree, at most 32; default: the smallest power of 2 GREATER
        # than 2 * degree (degree 1 -> 4, degree 3 -> 8)
        self.extensionFactor = extensionFactor or 1 << (2 * self.maxConstraintDegree).bit_length()
        ef = self.extensionFactor
        if ef & (ef - 1) or ef < 2 * self.compositionFactor or ef > 32:
            raise GstarkError('Extension factor must be a power of 2 at least 2x the constraint degree and at most 32')
        self.staticRegisters = [[v % f.modulus for v in values] for values in staticRegisters]
        r = [reg(i) for i in range(registers)]
        n = [nxt(i) for i in range(registers)]
        k = [static(j) for j in range(len(staticRegisters) + self.secretInputCount)]
        self.transitionProgram = Program(transition(r, k), f.modulus)
        self.evaluationProgram = Program(evaluation(r, n, k), f.modulus)
        if self.transitionProgram.nout != registers or self.evaluationProgram.nout != len(self.constraintDegrees):
            raise GstarkError('transition must yield one value per register, evaluation one per constraint')
        self.init = init
        self.initProgram = None
        if initExpr is not None:
            if segmentLength is None:
                raise GstarkError('initExpr is for segmented AIRs')
            self.initProgram = Program(initExpr(r), f.modulus, share_consts_with=self.transitionProgram)
            if self.initProgram.nout != registers:
                raise GstarkError('initExpr must yield one value per register')
            if self.initProgram.nregs > self.transitionProgram.nregs:
                self.transitionProgram.nregs = self.initProgram.nregs
        self.rootOfUnity = f.getRootOfUnity(steps * ef)
        self._staticPolys = None

    def firstRows(self, seed):
        if self.segmentLength is None:
            return [self.init(seed or [])]
        segments = self.steps // self.segmentLength
        if seed is None or len(seed) != segments:
            raise GstarkError(f'a segmented AIR needs one seed per segment ({segments})')
        return [self.init(s) for s in seed]

    def _hostFirstRows(self, seed):
        rows = [[v % self.field.modulus for v in row] for row in self.firstRows(seed)]
        if self.initProgram is not None:
            rows = [self.initProgram.run(row, None, []) for row in rows]
        return rows

    def initProvingContext(self, inputs=None, seed=None):
        return GenericProvingContext(self, self.firstRows(seed), inputs)

    def descriptor(self, seed=None):
        """The AIR as plain JSON-able data for the node-side twin (js/air_generic.js, reached by the reference's Stark.js through
        `instantiate({generic: ...})` of js/shims/@guildofweavers/air-assembly): field, trace shape, static registers and the three
        programs in the {op, dst, a, b} encoding of include/gstark.h; integers that may exceed 2^53 travel as decimal strings.
        The node side builds first rows by zero-padding prove()'s seed; an AIR whose init() does anything else must pass the
        `seed` it will be proved with, and the rows init() gives for it are pinned in the descriptor instead."""
        prog = lambda pr: None if pr is None else {'code': [w for ins in pr.code for w in ins], 'consts': [str(v) for v in pr.consts],
                                                   'nregs': pr.nregs, 'nout': pr.nout}
        d = {'modulus': str(self.field.modulus), 'steps': self.steps, 'registers': self.traceRegisterCount,
             'constraintDegrees': list(self.constraintDegrees), 'extensionFactor': self.extensionFactor, 'secretInputCount': self.secretInputCount,
             'staticRegisters': [[str(v) for v in values] for values in self.staticRegisters],
             'transition': prog(self.transitionProgram), 'evaluation': prog(self.evaluationProgram), 'init': prog(self.initProgram)}
        if self.segmentLength is not None:
            d['segmentLength'] = self.segmentLength
        if seed is not None:
            d['firstRows'] = [[str(v % self.field.modulus) for v in row] for row in self.firstRows(seed)]
            return d
        for width in range(self.traceRegisterCount + 1):          # which seed width does init() zero-pad?
            probe = list(range(3, 3 + width))
            try:
                row = [v % self.field.modulus for v in self.init(probe)]
            except Exception:
                continue
            if row == probe + [0] * (self.traceRegisterCount - width):
                d['seedWidth'] = width
                return d
        raise GstarkError('init() is not a zero-padding of the seed: pass the seed to pin the first rows in the descriptor')

    def initVerificationContext(self, inputShapes=None, publicInputs=None):
        return GenericVerificationContext(self)

    def hostTrace(self, seed, steps=None, inputs=None):
        """Independent control computation on Python integers (the role of examples/rescue/utils.ts for the examples)."""
        p, out = self.field.modulus, []
        statics = list(self.staticRegisters) + [values.ints() if isinstance(values, PackedColumn) else [v % p for v in values]
                                                for values in (inputs or [])]
        firsts = self._hostFirstRows(seed)
        seg = self.segmentLength or self.steps
        for i in range(steps or self.steps):
            if i % seg == 0:
                row = [v % p for v in firsts[i // seg]]
            out.append(row)
            row = self.transitionProgram.run(row, None, [v[i % len(v)] for v in statics])
        return out

    def compileCheck(self, lib=None):
        """Does the code generated for this AIR's programs build for gfx950 (gs_air_jit_check: compiler only, no device)?  Returns
        [(name, ok, compiler log)] for the trace program (segmented AIRs: those are the ones the device traces) and the constraint
        program.  `lib`: a loaded flavour of the library; default: the one built for the AIR's field."""
        from ._abi import GS_OK, HIP_LIB_PATHS, load_library
        lib = lib or getattr(getattr(self.field, 'backend', None), 'lib', None) or load_library(HIP_LIB_PATHS[self.field.modulus])
        periods = [len(v) for v in self.staticRegisters] + [self.steps] * self.secretInputCount
        lens = (C.c_uint64 * max(len(periods), 1))(*periods)
        log, out = C.create_string_buffer(4096), []

        def check(name, kind, prog, init):
            code, ninstr, consts, nconsts, nregs = prog.abi_args(self.field.elementSize)
            icode, ininstr = init.abi_args(self.field.elementSize)[:2] if init is not None else (None, 0)
            hook = getattr(lib, 'gs_air_jit_check_statics', None) if kind == 0 and not self.secretInputCount else None
            if hook is not None:
                # the trace program's source depends on the static tables (0/1 selectors become selects): check what a prover compiles
                hook.restype = C.c_int
                hook.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                 C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint64]
                packed = b''.join(self.field.le(v % self.field.modulus) for values in self.staticRegisters for v in values)
                rc = hook(kind, code, ninstr, icode, ininstr, consts, nconsts, nregs, self.traceRegisterCount, lens, len(periods), packed or None, log, len(log))
            else:
                rc = lib.gs_air_jit_check(kind, code, ninstr, icode, ininstr, consts, nconsts, nregs, self.traceRegisterCount, lens, len(periods), log, len(log))
            out.append((name, rc == GS_OK, log.value.decode(errors='replace')))
        if self.segmentLength is not None:
            check('trace', 0, self.transitionProgram, self.initProgram)
        check('constraints', 1, self.evaluationProgram, None)
        return out
