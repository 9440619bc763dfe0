"""AirAssembly source -> AirModule: `instantiate(source, component, options)` of index.ts:18-33 for the subset of the language the
reference's own sources use (assembly/lib128.aa, assembly/lib224.aa, examples/elliptic/pointmul.aa, the inline module of
examples/mimc/mimc128Assembly.ts:28-51):

    (module (field prime P) (const $c scalar|vector|matrix ...) (function $f (result T) (param $x T)* (local $y T)* stmt* expr)
            (export name (registers R) (constraints C) (steps S)
                (static (input secret|public [(childof i)|(peerof i)] [(steps n)] [(shift -1)])* (mask (input i))*
                        (cycle v ... | (prng sha256 0xSEED n) | (power b n))*)
                (init [(param $seed vector R)] expr) (transition (local ..)* stmt* expr) (evaluation (local ..)* stmt* expr)))
    expr: scalar vector get slice add sub mul div exp prod call load.const load.param load.local load.trace load.static
    stmt: (store.local name expr)

The compiler itself lives in the absent `@guildofweavers/air-assembly` package, so the meaning of the static-register
declarations is a restatement — the same one genstark_amd/lib128.py spells out and the reference's examples confirm through their
own control computations (Merkle roots, k*G, a valid Schnorr signature):
  * input registers take nested lists: a register without a parent gets one value per run of the computation, a `(childof i)`
    register a list of values per value of register i, `(peerof i)` the shape of register i;
  * a value of a register with `(steps n)` is held for n steps, a value without it for as long as its children take; the trace
    length is the total (a power of 2);
  * `(shift -1)` rotates the column one step earlier; `(mask (input i))` is 1 on the first step of every value of register i, rotated
    like register i (so with shift -1: on the last step before a new value shows);
  * `(init expr)` is evaluated on the static registers of step 0 (and of the first step of every further run);
  * constraint degrees: trace and static registers count 1, + and - take the maximum, * adds, exp multiplies.
An expression becomes a GenericAir program (air_generic.py): the transition function and the constraints run on the device
register machine, public inputs / masks / cycles are public static registers, secret inputs are committed secret registers.
"""
import re

from ._abi import GstarkError
from .air import sha256_prng
from .air_generic import Expr, GenericAir, PackedColumn
from .field import PrimeField


# ---- reading -----------------------------------------------------------------------------------------------------------------------
def parse(text):
    """S-expressions -> nested lists of atoms (strings); `#` starts a comment."""
    text = re.sub(r'#[^\n]*', '', text)
    tokens = re.findall(r'[()]|[^\s()]+', text)
    pos = 0

    def node():
        nonlocal pos
        if tokens[pos] != '(':
            tok = tokens[pos]
            pos += 1
            return tok
        pos += 1
        out = []
        while tokens[pos] != ')':
            out.append(node())
        pos += 1
        return out
    out = []
    while pos < len(tokens):
        out.append(node())
    return out


def _int(tok):
    return int(tok, 16) if tok.lower().startswith('0x') else int(tok)


class _Function:
    def __init__(self, tree):
        self.name, self.params, self.locals, self.body = tree[1], [], [], []
        for item in tree[2:]:
            if isinstance(item, list) and item[0] == 'result':
                continue
            if isinstance(item, list) and item[0] == 'param':
                self.params.append(item[1] if item[1].startswith('$') else None)
            elif isinstance(item, list) and item[0] == 'local':
                self.locals.append(item[1] if item[1].startswith('$') else None)
            else:
                self.body.append(item)


class _Export:
    def __init__(self, tree):
        self.name = tree[1]
        self.registers = self.constraints = self.steps = None
        self.statics, self.init, self.transition, self.evaluation = [], None, None, None
        for item in tree[2:]:
            kind = item[0]
            if kind == 'registers': self.registers = int(item[1])
            elif kind == 'constraints': self.constraints = int(item[1])
            elif kind == 'steps': self.steps = int(item[1])
            elif kind == 'static': self.statics = [self._static(s) for s in item[1:]]
            elif kind == 'init':
                body = [x for x in item[1:] if not (isinstance(x, list) and x[0] == 'param')]
                params = [x for x in item[1:] if isinstance(x, list) and x[0] == 'param']
                self.init = {'param': params[0][1] if params else None, 'body': body}
            elif kind in ('transition', 'evaluation'):
                body = [x for x in item[1:] if not (isinstance(x, list) and x[0] == 'local')]
                setattr(self, kind, body)
            else:
                raise GstarkError(f'export {self.name}: unknown section {kind}')

    @staticmethod
    def _static(s):
        if s[0] == 'input':
            d = {'kind': 'input', 'secret': s[1] == 'secret', 'parent': None, 'peer': None, 'steps': None, 'shift': 0}
            if s[1] not in ('secret', 'public'):
                raise GstarkError('input register: secret or public expected')
            for opt in s[2:]:
                if opt[0] == 'childof': d['parent'] = int(opt[1])
                elif opt[0] == 'peerof': d['peer'] = int(opt[1])
                elif opt[0] == 'steps': d['steps'] = int(opt[1])
                elif opt[0] == 'shift': d['shift'] = int(opt[1])
                else: raise GstarkError(f'input register: unknown option {opt[0]}')
            return d
        if s[0] == 'mask':
            if not (isinstance(s[1], list) and s[1][0] == 'input'):
                raise GstarkError('mask register: (mask (input i)) expected')
            return {'kind': 'mask', 'input': int(s[1][1])}
        if s[0] == 'cycle':
            return {'kind': 'cycle', 'values': s[1:]}
        raise GstarkError(f'unknown static register kind {s[0]}')


# ---- evaluation over an algebra ------------------------------------------------------------------------------------------------------
class _Ints:
    """Field elements as Python integers."""
    def __init__(self, p): self.p = p
    def const(self, v): return v % self.p
    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def mul(self, a, b): return a * b % self.p
    def pow(self, a, e): return pow(a, e, self.p)
    def inv(self, a): return pow(a, self.p - 2, self.p)


class _Exprs(_Ints):
    """Expression DAG nodes (air_generic.Expr); constants stay integers and fold."""
    def _both(self, a, b): return not isinstance(a, Expr) and not isinstance(b, Expr)
    def add(self, a, b): return _Ints.add(self, a, b) if self._both(a, b) else a + b
    def sub(self, a, b): return _Ints.sub(self, a, b) if self._both(a, b) else a - b
    def mul(self, a, b): return _Ints.mul(self, a, b) if self._both(a, b) else a * b
    def pow(self, a, e): return _Ints.pow(self, a, e) if not isinstance(a, Expr) else a ** e
    def inv(self, a): return _Ints.inv(self, a) if not isinstance(a, Expr) else a ** (self.p - 2)


class _Degrees:
    """Degree of an expression in units of the trace length: registers 1, constants 0."""
    def const(self, v): return 0
    def add(self, a, b): return max(a, b)
    sub = add
    def mul(self, a, b): return a + b
    def pow(self, a, e): return a * e
    def inv(self, a):
        if a:
            raise GstarkError('division by a register inside a constraint has no degree: check the slope by cross-multiplication')
        return 0


class _Evaluator:
    def __init__(self, module, algebra):
        self.m, self.a = module, algebra
        self.ints = _Ints(module.modulus)

    # values: scalar = algebra element, vector = list, matrix = list of lists
    def _zip(self, fn, x, y):
        vx, vy = isinstance(x, list), isinstance(y, list)
        if vx and vy:
            if len(x) != len(y):
                raise GstarkError(f'vector lengths differ: {len(x)} and {len(y)}')
            return [self._zip(fn, a, b) for a, b in zip(x, y)]
        if vx:
            return [self._zip(fn, a, y) for a in x]
        if vy:
            return [self._zip(fn, x, b) for b in y]
        return fn(x, y)

    def constant(self, node):
        """A compile-time integer (exponents)."""
        v = _Evaluator(self.m, self.ints).eval(node, {'params': {}, 'locals': {}})
        if isinstance(v, list):
            if len(v) != 1:
                raise GstarkError('a scalar constant was expected')
            v = v[0]
        return v

    def eval(self, node, env):
        a = self.a
        if not isinstance(node, list):
            raise GstarkError(f'unexpected atom {node}')
        op = node[0]
        if op == 'scalar': return a.const(_int(node[1]))
        if op == 'vector':
            out = []
            for item in node[1:]:
                v = self.eval(item, env)
                out.extend(v) if isinstance(v, list) else out.append(v)
            return out
        if op == 'get':
            v = self.eval(node[1], env)
            return v[int(node[2])]
        if op == 'slice':
            v = self.eval(node[1], env)
            return v[int(node[2]):int(node[3]) + 1]
        if op in ('add', 'sub', 'mul'):
            return self._zip(getattr(a, op), self.eval(node[1], env), self.eval(node[2], env))
        if op == 'div':
            return self._zip(lambda x, y: a.mul(x, a.inv(y)), self.eval(node[1], env), self.eval(node[2], env))
        if op == 'exp':
            e = self.constant_in(node[2], env)
            base = self.eval(node[1], env)
            return [a.pow(b, e) for b in base] if isinstance(base, list) else a.pow(base, e)
        if op == 'prod':
            m, v = self.eval(node[1], env), self.eval(node[2], env)
            if not (isinstance(m, list) and m and isinstance(m[0], list)) or not isinstance(v, list):
                raise GstarkError('prod: matrix x vector expected')
            out = []
            for row in m:
                acc = None
                for c, x in zip(row, v):
                    t = a.mul(c, x)
                    acc = t if acc is None else a.add(acc, t)
                out.append(acc)
            return out
        if op == 'load.const': return self._named(self.m.const_values(a), self.m.const_names, node[1], 'constant')
        if op == 'load.param': return self._named(env['params']['values'], env['params']['names'], node[1], 'parameter')
        if op == 'load.local':
            key = node[1]
            if key not in env['locals']:
                raise GstarkError(f'local {key} read before it is stored')
            return env['locals'][key]
        if op == 'load.trace': return env['trace'][int(node[1])]
        if op == 'load.static': return env['static']
        if op == 'call':
            fn = self.m.functions.get(node[1])
            if fn is None:
                raise GstarkError(f'unknown function {node[1]}')
            args = [self.eval(x, env) for x in node[2:]]
            if len(args) != len(fn.params):
                raise GstarkError(f'{fn.name}: {len(fn.params)} arguments expected')
            inner = dict(env, params={'values': args, 'names': fn.params}, locals={})
            return self.run(fn.body, inner)
        raise GstarkError(f'unknown operation {op}')

    def constant_in(self, node, env):
        # an exponent: a literal, a constant, or a parameter bound to one
        if node[0] == 'load.param' and env['params'].get('values'):
            v = self._named(env['params']['values'], env['params']['names'], node[1], 'parameter')
            if isinstance(v, int) and not isinstance(self.a, _Degrees):
                return v
        return self.constant(node)

    @staticmethod
    def _named(values, names, key, what):
        if key.startswith('$'):
            if key not in names:
                raise GstarkError(f'unknown {what} {key}')
            return values[names.index(key)]
        return values[int(key)]

    def run(self, body, env):
        """stmt* expr"""
        for stmt in body[:-1]:
            if stmt[0] != 'store.local':
                raise GstarkError(f'statement expected, got {stmt[0]}')
            env['locals'][stmt[1]] = self.eval(stmt[2], env)
        return self.eval(body[-1], env)


class Module:
    """One parsed (module ...)."""

    def __init__(self, text):
        tree = parse(text if isinstance(text, str) else bytes(text).decode())
        if len(tree) != 1 or tree[0][0] != 'module':
            raise GstarkError('AirAssembly source: one (module ...) expected')
        self.modulus, self.const_names, self._consts, self.functions, self.exports = None, [], [], {}, {}
        for item in tree[0][1:]:
            kind = item[0]
            if kind == 'field':
                if item[1] != 'prime':
                    raise GstarkError('only prime fields are supported')
                self.modulus = _int(item[2])
            elif kind == 'const':
                named = item[1].startswith('$')
                self.const_names.append(item[1] if named else None)
                self._consts.append(item[2 if named else 1:])
            elif kind == 'function':
                fn = _Function(item)
                self.functions[fn.name] = fn
            elif kind == 'export':
                ex = _Export(item)
                self.exports[ex.name] = ex
            else:
                raise GstarkError(f'module: unknown section {kind}')
        if self.modulus is None:
            raise GstarkError('module: no field')

    def const_values(self, algebra):
        out = []
        for spec in self._consts:
            if spec[0] == 'scalar': out.append(algebra.const(_int(spec[1])))
            elif spec[0] == 'vector': out.append([algebra.const(_int(v)) for v in spec[1:]])
            elif spec[0] == 'matrix': out.append([[algebra.const(_int(v)) for v in row] for row in spec[1:]])
            else: raise GstarkError(f'constant of unknown type {spec[0]}')
        return out


# ---- input registers -> columns ------------------------------------------------------------------------------------------------------
def _shape_of(value):
    shape = []
    while isinstance(value, (list, tuple)):
        shape.append(len(value))
        if not value:
            break
        if any(isinstance(v, (list, tuple)) != isinstance(value[0], (list, tuple)) or
               (isinstance(v, (list, tuple)) and len(v) != len(value[0])) for v in value):
            raise GstarkError('input register: ragged values')
        value = value[0]
    return shape


class _Layout:
    """Where every value of every input register sits in the trace, from the registers' shapes alone."""

    def __init__(self, statics, shapes):
        inputs = [s for s in statics if s['kind'] == 'input']
        if len(shapes) != len(inputs):
            raise GstarkError(f'{len(inputs)} input registers: one entry (shape) for each is needed, got {len(shapes)}')
        self.inputs, self.shapes = inputs, [list(s) for s in shapes]
        depth = []
        for j, d in enumerate(inputs):
            ref = d['parent'] if d['parent'] is not None else d['peer']
            if ref is not None and not (0 <= ref < j):
                raise GstarkError('input register: childof / peerof must name an earlier input register')
            depth.append(0 if ref is None else depth[ref] + (1 if d['parent'] is not None else 0))
            if len(self.shapes[j]) != depth[j] + 1:
                raise GstarkError(f'input register {j}: values nested {depth[j] + 1} deep expected')
            if d['peer'] is not None and self.shapes[j] != self.shapes[d['peer']]:
                raise GstarkError(f'input register {j}: the shape of its peer {d["peer"]} expected')
            if d['parent'] is not None and self.shapes[j][:-1] != self.shapes[d['parent']]:
                raise GstarkError(f'input register {j}: one list per value of register {d["parent"]} expected')
        self.depth = depth
        # steps one value of a register is held: its own (steps n), or what its children take
        span = [None] * len(inputs)
        for j in reversed(range(len(inputs))):
            if inputs[j]['steps'] is not None:
                span[j] = inputs[j]['steps']
        changed = True
        while changed:
            changed = False
            for j, d in enumerate(inputs):
                if span[j] is not None and d['parent'] is not None:
                    want = span[j] * self.shapes[j][-1]
                    root = d['parent']
                    if span[root] is None:
                        span[root], changed = want, True
                    elif span[root] != want and inputs[root]['steps'] is None:
                        raise GstarkError('input registers: the children of one register take different numbers of steps')
                if span[j] is None and d['peer'] is not None and span[d['peer']] is not None:
                    span[j], changed = span[d['peer']], True
                if span[j] is not None and d['peer'] is not None and span[d['peer']] is None:
                    span[d['peer']], changed = span[j], True
        if inputs and any(s is None for s in span):
            raise GstarkError('input registers: cannot tell how many steps a value is held (no (steps n) below it)')
        self.span = span
        self.length = 0
        for j, d in enumerate(inputs):
            count = 1
            for n in self.shapes[j]:
                count *= n
            if self.length and count * span[j] != self.length:
                raise GstarkError('input registers imply different trace lengths')
            self.length = count * span[j]
        # (a register whose values are held for 0 steps, or that has no values, is skipped by the rule above when it comes first: every
        #  register must fill the trace the others lay out)
        for j in range(len(inputs)):
            count = 1
            for n in self.shapes[j]:
                count *= n
            if count * span[j] != self.length:
                raise GstarkError('input registers imply different trace lengths')
        if inputs and (self.length < 2 or self.length & (self.length - 1)):
            raise GstarkError(f'the inputs make a trace of {self.length} steps: a power of 2 is required')

    def column(self, j, values):
        flat = values
        for _ in range(self.depth[j]):
            flat = [v for group in flat for v in group]
        col = [v for v in flat for _ in range(self.span[j])]
        return _rotate(col, self.inputs[j]['shift'])

    def mask(self, j):
        col = ([1] + [0] * (self.span[j] - 1)) * (self.length // self.span[j])
        return _rotate(col, self.inputs[j]['shift'])


def _rotate(col, shift):
    k = (-shift) % len(col) if col else 0
    return col[k:] + col[:k]


def _shrink(col):
    """The shortest power-of-2 period of a column (a cyclic static register of that length denotes the same polynomial)."""
    while len(col) > 1 and len(col) % 2 == 0 and col[:len(col) // 2] == col[len(col) // 2:]:
        col = col[:len(col) // 2]
    return col


def _shrink_packed(data, es):
    """_shrink on a column of packed elements (canonical values: equal elements have equal bytes)."""
    n = len(data) // es
    while n > 1 and n % 2 == 0 and data[:n // 2 * es] == data[n // 2 * es:]:
        n //= 2
        data = data[:n * es]
    return data


class _Col:
    """A static column in closed form: value i of `flat` is held `span` steps, the whole rotated `k` steps earlier and repeated —
    col[t] = flat[((t + k) mod period) div span].  What plan() works on instead of trace-length lists of Python integers: an input
    register is (its values, its span, its shift), a mask (1 0 .. 0, 1, its register's shift), a cycle (its values, 1, 0)."""
    __slots__ = ('flat', 'span', 'k', 'period')

    def __init__(self, flat, span, k):
        self.flat, self.span, self.period = flat, span, len(flat) * span
        self.k = k % self.period if self.period else 0

    def at(self, t):
        return self.flat[((t + self.k) % self.period) // self.span]

    def packed(self, es):
        """One period as packed little-endian elements, shrunk to its shortest power-of-2 period."""
        vb = [v.to_bytes(es, 'little') for v in self.flat]
        body = b''.join(vb) if self.span == 1 else b''.join([x * self.span for x in vb])
        k = self.k * es
        return _shrink_packed(body[k:] + body[:k] if k else body, es)


class _Lane:
    """One scalar per run of the computation (the init block evaluated for all runs at once)."""
    __slots__ = ('v',)

    def __init__(self, v): self.v = v


class _Lanes(_Ints):
    """Integers, or a _Lane of integers: element-wise, integers broadcast."""
    def _op(self, fn, a, b):
        la, lb = isinstance(a, _Lane), isinstance(b, _Lane)
        if la and lb: return _Lane([fn(x, y) for x, y in zip(a.v, b.v)])
        if la: return _Lane([fn(x, b) for x in a.v])
        if lb: return _Lane([fn(a, y) for y in b.v])
        return fn(a, b)
    def add(self, a, b): return self._op(lambda x, y: (x + y) % self.p, a, b)
    def sub(self, a, b): return self._op(lambda x, y: (x - y) % self.p, a, b)
    def mul(self, a, b): return self._op(lambda x, y: x * y % self.p, a, b)
    def pow(self, a, e): return _Lane([pow(x, e, self.p) for x in a.v]) if isinstance(a, _Lane) else pow(a, e, self.p)
    def inv(self, a): return self.pow(a, self.p - 2)


# ---- the AirModule -------------------------------------------------------------------------------------------------------------------
class AssemblyAir:
    """The AirModule `instantiate(source, component, options)` hands to Stark (index.ts:18-33; lib/Stark.ts:40): shape-agnostic —
    initProvingContext(inputs, seed) / initVerificationContext(inputShapes, publicInputs) size the trace from the inputs."""

    def __init__(self, source, component='default', extensionFactor=None, field=None):
        self.module = source if isinstance(source, Module) else Module(source)
        if component not in self.module.exports:
            raise GstarkError(f'component {component} is not exported (exports: {sorted(self.module.exports)})')
        self.export = ex = self.module.exports[component]
        self.field = field or PrimeField(self.module.modulus)
        if self.field.modulus != self.module.modulus:
            raise GstarkError(f'the module is over the field of {self.module.modulus} elements, the field object over {self.field.modulus}')
        self.traceRegisterCount = ex.registers
        self.inputRegisters = [s for s in ex.statics if s['kind'] == 'input']
        self.secretInputCount = sum(1 for s in self.inputRegisters if s['secret'])
        # lib order -> (public index | secret index)
        self._where, npub, nsec = [], 0, 0
        for s in ex.statics:
            if s['kind'] == 'input' and s['secret']:
                self._where.append(('secret', nsec))
                nsec += 1
            else:
                self._where.append(('public', npub))
                npub += 1
        self._publicCount = npub
        degrees = self._run(_Degrees(), [1] * ex.registers, [1] * ex.registers, [1] * len(ex.statics), ex.evaluation)
        if len(degrees) != ex.constraints:
            raise GstarkError(f'{component}: the evaluation yields {len(degrees)} values, {ex.constraints} constraints declared')
        self.constraintDegrees = [max(d, 1) for d in degrees]
        self.constraints = [{'degree': d} for d in self.constraintDegrees]
        self.maxConstraintDegree = max(self.constraintDegrees)
        cf = 1 << (self.maxConstraintDegree - 1).bit_length()
        self.extensionFactor = extensionFactor or 1 << (2 * self.maxConstraintDegree).bit_length()
        if self.extensionFactor < 2 * cf:
            raise GstarkError('Extension factor must be a power of 2 at least 2x the constraint degree and at most 32')
        self._cache = {}
        self._evaluationProgram = None

    # -- expressions
    def _run(self, algebra, r, n, k, body):
        ev = _Evaluator(self.module, algebra)
        out = ev.run(body, {'params': {}, 'locals': {}, 'trace': [list(r), list(n) if n is not None else None], 'static': list(k)})
        return out if isinstance(out, list) else [out]

    def _lib_order(self, k):
        return [k[i] if kind == 'public' else k[self._publicCount + i] for kind, i in self._where]

    def _first_row(self, statics_at_step, seed):
        ex = self.export
        ev = _Evaluator(self.module, _Ints(self.module.modulus))
        env = {'params': {}, 'locals': {}, 'static': list(statics_at_step)}
        if ex.init['param'] is not None:
            if seed is None or len(seed) != ex.registers:
                raise GstarkError(f'{ex.name}: the init block takes a seed vector of {ex.registers} values')
            env['params'] = {'values': [[v % self.module.modulus for v in seed]], 'names': [ex.init['param']]}
        row = ev.run(ex.init['body'], env)
        row = row if isinstance(row, list) else [row]
        if len(row) != ex.registers:
            raise GstarkError(f'{ex.name}: the init block yields {len(row)} values for {ex.registers} registers')
        return row

    # -- static registers for given shapes
    def _columns(self, layout, inputs):
        """Full-length integer columns in lib order (None where `inputs` does not provide one: secret registers on the verifier side)."""
        ex, p, cols, j = self.export, self.module.modulus, [], 0
        for s in ex.statics:
            if s['kind'] == 'input':
                values = inputs[j] if inputs is not None and j < len(inputs) else None
                cols.append(None if values is None else [v % p for v in layout.column(j, values)])
                j += 1
            elif s['kind'] == 'mask':
                cols.append(layout.mask(s['input']))
            else:
                cols.append(self._cycle(s['values']))
        return cols

    def _cycle(self, spec):
        p = self.module.modulus
        if len(spec) == 1 and isinstance(spec[0], list):
            g = spec[0]
            if g[0] == 'prng':
                if g[1] != 'sha256':
                    raise GstarkError('cycle: only (prng sha256 seed n) is known')
                seed = g[2][2:] if g[2].lower().startswith('0x') else g[2]
                return sha256_prng(bytes.fromhex(seed), int(g[3]), self.field)
            if g[0] == 'power':
                return [pow(_int(g[1]), i, p) for i in range(int(g[2]))]
            raise GstarkError(f'cycle: unknown generator {g[0]}')
        return [_int(v) % p for v in spec]

    def _inner(self, length, public_cols, segment):
        """The GenericAir of one (trace length, public static columns, segmentation).  public_cols: lists of integers, or the same
        columns as packed elements (bytes) — the form plan() has them in; the key is the packed form either way."""
        es = self.field.elementSize
        packed = tuple(c if isinstance(c, bytes) else b''.join([int(v).to_bytes(es, 'little') for v in c]) for c in public_cols)
        key = (length, packed, segment)
        air = self._cache.get(key)
        if air is None:
            public_cols = [[int.from_bytes(c[i:i + es], 'little') for i in range(0, len(c), es)] if isinstance(c, bytes) else c for c in public_cols]
            transition = lambda r, k: self._run(_Exprs(self.module.modulus), r, None, self._lib_order(k), self.export.transition)
            evaluation = lambda r, n, k: self._run(_Exprs(self.module.modulus), r, n, self._lib_order(k), self.export.evaluation)
            air = GenericAir(length, self.traceRegisterCount, self.constraintDegrees, public_cols, transition, evaluation, lambda seed: list(seed),
                             self.extensionFactor, self.field, secretRegisters=self.secretInputCount, segmentLength=segment, maskSegments=False)
            if len(self._cache) > 8:
                self._cache.clear()
            self._cache[key] = air
        return air

    @property
    def evaluationProgram(self):
        """The constraint evaluator as a register-machine program (include/gstark.h opcodes).  It does not depend on the inputs' shapes:
        static register k of the program is the k-th PUBLIC static register in declaration order, the secret ones follow — the order the
        inner GenericAir of every shape uses.  What the native verifier runs for a proof whose shapes it reads itself (csrc/verifier.h)."""
        if self._evaluationProgram is None:
            from .air_generic import Program, nxt, reg, static
            n = self.traceRegisterCount
            k = [static(j) for j in range(len(self.export.statics))]
            self._evaluationProgram = Program(self._run(_Exprs(self.module.modulus), [reg(i) for i in range(n)], [nxt(i) for i in range(n)],
                                                        self._lib_order(k), self.export.evaluation), self.module.modulus)
        return self._evaluationProgram

    def staticSources(self):
        """[(kind, index)] per static register in the programs' order (public ones in declaration order, then the secret inputs):
        kind 0 = cyclic values, 1 = input register `index` (among the input registers), 2 = mask of input register `index`
        (struct gs_static_source, include/gstark_prover.h); and the cyclic registers' values in the same order."""
        sources, secret, cycles, j = [], [], [], 0
        for s in self.export.statics:
            if s['kind'] == 'input':
                (secret if s['secret'] else sources).append((1, j))
                j += 1
            elif s['kind'] == 'mask':
                sources.append((2, s['input']))
            else:
                sources.append((0, 0))
                cycles.append(_shrink(self._cycle(s['values'])))
        return sources + secret, cycles

    def _public_split(self, cols):
        return [_shrink(c) for c, (kind, _) in zip(cols, self._where) if kind == 'public']

    def _length_without_inputs(self):
        ex = self.export
        cycles = [len(self._cycle(s['values'])) for s in ex.statics if s['kind'] == 'cycle']
        return max([ex.steps or 1] + cycles)

    # -- AirModule surface (lib/Stark.ts:90,176)
    def initProvingContext(self, inputs=None, seed=None):
        air, packed, firsts, shapes = self.plan(inputs, seed)
        context = air.initProvingContext(packed, firsts)
        context.inputShapes = shapes
        return context

    def _closed_columns(self, layout, inputs, length):
        """The static registers in lib order as _Col (plan(): every input register comes with its values)."""
        ex, p, cols, j = self.export, self.module.modulus, [], 0
        for s in ex.statics:
            if s['kind'] == 'input':
                flat = inputs[j]
                for _ in range(layout.depth[j]):
                    flat = [v for group in flat for v in group]
                cols.append(_Col([v % p for v in flat], layout.span[j], -layout.inputs[j]['shift']))
                j += 1
            elif s['kind'] == 'mask':
                i = s['input']
                cols.append(_Col([1] + [0] * (layout.span[i] - 1), 1, -layout.inputs[i]['shift']))
            else:
                cols.append(_Col(self._cycle(s['values']), 1, 0))
        return cols

    def _first_rows(self, cols, starts):
        """The init block on the static registers of every step in `starts` at once (no seed parameter): one row per step."""
        ex = self.export
        ev = _Evaluator(self.module, _Lanes(self.module.modulus))
        row = ev.run(ex.init['body'], {'params': {}, 'locals': {}, 'static': [_Lane([c.at(t) for t in starts]) for c in cols]})
        row = row if isinstance(row, list) else [row]
        if len(row) != ex.registers:
            raise GstarkError(f'{ex.name}: the init block yields {len(row)} values for {ex.registers} registers')
        return [[x.v[i] if isinstance(x, _Lane) else x for x in row] for i in range(len(starts))]

    def plan(self, inputs=None, seed=None):
        """What initProvingContext decides before any device work: (the inner GenericAir for these input shapes, the secret registers'
        packed columns, the first row — or the first row of every independent run —, the input shapes).  Needs no backend: the node-side
        compile() of js/shims/@guildofweavers/air-assembly asks for exactly this (genstark_amd/aa_json.py).  Columns stay in closed form
        (_Col) or packed: nothing here costs a Python-integer operation per trace step."""
        ex, p = self.export, self.module.modulus
        inputs = list(inputs or [])
        layout = _Layout(ex.statics, [_shape_of(v) for v in inputs])
        length = layout.length or self._length_without_inputs()
        es = self.field.elementSize
        cols = self._closed_columns(layout, inputs, length)
        public = [c.packed(es) for c, (kind, _) in zip(cols, self._where) if kind == 'public']
        secret = [c for c, (kind, _) in zip(cols, self._where) if kind == 'secret']
        # independent runs: a mask on a top-level input means the transition restarts from init on its last step; the segments are
        # then generated side by side (one device thread each when there are many) — checked against the transition below
        tops = [j for j, d in enumerate(layout.inputs) if layout.depth[j] == 0]
        runs = layout.shapes[tops[0]][0] if tops else 1
        masked = any(s['kind'] == 'mask' and layout.depth[s['input']] == 0 for s in ex.statics)
        segment = length // runs if (runs > 1 and masked and ex.init['param'] is None) else None
        air = self._inner(length, public, segment)
        packed = [PackedColumn(c.packed(es), es) for c in secret]
        if segment is None:
            firsts = self._first_row([c.at(0) for c in cols], seed)
        else:
            firsts = self._first_rows(cols, range(0, runs * segment, segment))
            # the restart the segmentation relies on: the row the transition produces on the last step of run 0 is run 1's first row
            statics = list(air.staticRegisters)
            row = [v % p for v in firsts[0]]
            for i in range(segment):
                row = air.transitionProgram.run(row, None, [v[i % len(v)] for v in statics] + [c.at(i) for c in secret])
            if row != [v % p for v in firsts[1]]:
                air = self._inner(length, public, None)
                firsts = firsts[0]
        return air, packed, firsts, [list(s) for s in layout.shapes]

    def initVerificationContext(self, inputShapes=None, publicInputs=None):
        ex = self.export
        shapes = [list(s) for s in (inputShapes or [])]
        layout = _Layout(ex.statics, shapes)
        length = layout.length or self._length_without_inputs()
        # publicInputs: the values of the PUBLIC input registers, in declaration order (lib/Stark.ts:167)
        public_values, given, j = [], list(publicInputs or []), 0
        for d in layout.inputs:
            if d['secret']:
                public_values.append(None)
            else:
                if j >= len(given):
                    raise GstarkError(f'{ex.name}: the values of {sum(1 for x in layout.inputs if not x["secret"])} public input registers are needed')
                public_values.append(given[j])
                j += 1
        for jj, v in enumerate(public_values):
            if v is not None and _shape_of(v) != shapes[jj]:
                raise GstarkError(f'public input register {jj}: shape {shapes[jj]} expected')
        cols = self._columns(layout, public_values)
        context = self._inner(length, self._public_split(cols), None).initVerificationContext()
        context.inputShapes = shapes
        return context


def instantiate(source, component='default', options=None, logger=None, field=None):
    """index.ts:18-33: AirAssembly source (text, bytes or a path to an .aa file) + export name + StarkOptions -> Stark."""
    from ._mirror.stark import Stark      # index.ts:18-33 returns a Stark object: the restated caller (prove / verify / serialize)
    if isinstance(source, str) and '(' not in source:
        with open(source) as fh:
            source = fh.read()
    options = dict(options or {})
    air = AssemblyAir(source, component, options.get('extensionFactor'), field)
    return Stark(air, options, logger)
