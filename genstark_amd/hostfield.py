"""A verifier that needs no GPU: the galois / merkle members `Stark.verify()` touches (lib/Stark.ts:167-248 and the scalar
halves of lib/components/*.ts), on host integers and the language runtime's hashes.

The reference verifies on the CPU by construction; here the PROVER's field object is bound to the HIP library, so a machine
without an MI355X could not even construct it.  `HostField` is the verifier-side field: a few hundred field operations, some
4-point interpolations and <= 256-point Lagrange interpolation per proof, a few hundred hashes.  It implements only what the
verification path calls; every vector-sized prover member raises.

    stark = Stark(MimcAir(steps, extensionFactor, HostField()), options)       # no Backend, no device
    stark.verify(assertions, stark.parse(proof_bytes))
"""
import hashlib

from ._abi import GstarkError
from .field import ELEMENT_SIZE, MODULUS, sha256_bigint

DIGEST_SIZE = 32


class HostVector:
    host = True

    def __init__(self, values, element_size=ELEMENT_SIZE):
        self.values, self.length, self.elementSize = list(values), len(values), element_size
        self.series_base = None

    def toValues(self):
        return list(self.values)

    def getValue(self, i):
        return self.values[i]


class HostMatrix:
    host = True

    def __init__(self, rows):
        self.rows = [list(r) for r in rows]
        self.rowCount, self.colCount = len(self.rows), len(self.rows[0]) if self.rows else 0
        self.quartic_domain = None

    def toValues(self):
        return [list(r) for r in self.rows]


class HostField:
    def __init__(self, modulus=MODULUS, elementSize=None):
        # the prover's libraries store an element in 16 bytes, or in 32 for the 256- / 224-bit fields
        self.modulus, self.elementSize = modulus, elementSize or (ELEMENT_SIZE if modulus < (1 << 128) else 32)
        self.zero, self.one, self.isOptimized = 0, 1, False
        self.backend = None

    def createHash(self, algorithm):
        return HostHash(algorithm, self.elementSize)

    # ---- scalars
    def add(self, a, b): return (a + b) % self.modulus
    def sub(self, a, b): return (a - b) % self.modulus
    def mul(self, a, b): return a * b % self.modulus
    def neg(self, a): return -a % self.modulus
    def inv(self, a): return pow(a, -1, self.modulus) if a % self.modulus else 0      # extended Euclid: ~10x cheaper than Fermat
    def div(self, a, b): return a * self.inv(b) % self.modulus

    def exp(self, base, exponent):
        if exponent < 0:
            return pow(self.inv(base), -exponent, self.modulus)
        return pow(base, exponent, self.modulus)

    def prng(self, seed, length=None):
        if length is None:
            return sha256_bigint(seed) % self.modulus
        out, state = [], sha256_bigint(seed)
        for _ in range(length):
            out.append(state % self.modulus)
            state = sha256_bigint(state)
        return HostVector(out)

    def getRootOfUnity(self, order):
        if order <= 0 or order & (order - 1) or (self.modulus - 1) % order:
            raise GstarkError(f'Order {order} of root of unity is invalid')
        for i in range(2, 1 << 16):
            g = pow(i, (self.modulus - 1) // order, self.modulus)
            if pow(g, order, self.modulus) == 1 and (order == 1 or pow(g, order // 2, self.modulus) != 1):
                return g
        raise GstarkError(f'Root of unity for order {order} was not found')

    # ---- the small vector / polynomial members of the verification path
    def newVectorFrom(self, values):
        return HostVector([v % self.modulus for v in values], self.elementSize)

    def newMatrixFrom(self, values):
        return HostMatrix([[v % self.modulus for v in row] for row in values])

    def getPowerSeries(self, base, length):
        out, x = [], 1
        for _ in range(length):
            out.append(x)
            x = x * base % self.modulus
        v = HostVector(out, self.elementSize)
        v.series_base = base % self.modulus
        return v

    def interpolateValues(self, xs, ys):
        """Lagrange interpolation, O(n^2) (BoundaryConstraints.ts:42; LowDegreeProver.ts:243)."""
        p, n = self.modulus, len(xs)
        master = [1] + [0] * n
        for i, x in enumerate(xs):
            for d in range(i + 1, 0, -1):
                master[d] = (master[d - 1] - master[d] * x) % p
            master[0] = -master[0] * x % p
        out = [0] * n
        for j in range(n):
            q, carry = [0] * n, 0
            for d in range(n, 0, -1):
                carry = (master[d] + carry * xs[j]) % p
                q[d - 1] = carry
            den = 0
            for d in range(n - 1, -1, -1):
                den = (den * xs[j] + q[d]) % p
            s = ys[j] * pow(den, -1, p) % p
            for d in range(n):
                out[d] = (out[d] + q[d] * s) % p
        return out

    def interpolate(self, xs, ys):
        xv = xs if isinstance(xs, (list, tuple)) else xs.toValues()
        yv = ys if isinstance(ys, (list, tuple)) else ys.toValues()
        if len(xv) != len(yv):
            raise GstarkError('Number of x coordinates must be the same as number of y coordinates')
        return HostVector(self.interpolateValues(xv, yv))

    def interpolateRoots(self, roots, ys):
        """Only the short cyclic registers of an AIR reach the verifier through this member."""
        return HostVector(self.interpolateValues(roots.toValues(), ys.toValues()))

    def evalPolyAt(self, poly, x):
        s = 0
        for c in reversed(poly.toValues()):
            s = (s * x + c) % self.modulus
        return s

    def evalPolyAtMany(self, poly_values, xs):
        return [self.evalPolyAt(HostVector(poly_values), x) for x in xs]

    def mulPolys(self, a, b):
        av, bv = a.toValues(), b.toValues()
        out = [0] * (len(av) + len(bv) - 1)
        for i, x in enumerate(av):
            for j, y in enumerate(bv):
                out[i + j] = (out[i + j] + x * y) % self.modulus
        return HostVector(out)

    def transposeVector(self, v, columns, step=1):
        vals = v.toValues()
        rows = len(vals) // (columns * step)
        return HostMatrix([[vals[(r + c * rows) * step] for c in range(columns)] for r in range(rows)])

    def _interpolate4(self, x, y):
        """The cubic through four points: Lagrange basis written out, ONE modular inversion (of the product of the four
        denominators) instead of four."""
        p = self.modulus
        x0, x1, x2, x3 = x
        d = [(x0 - x1) * (x0 - x2) * (x0 - x3) % p, (x1 - x0) * (x1 - x2) * (x1 - x3) % p,
             (x2 - x0) * (x2 - x1) * (x2 - x3) % p, (x3 - x0) * (x3 - x1) * (x3 - x2) % p]
        p01, p23 = d[0] * d[1] % p, d[2] * d[3] % p
        inv_all = pow(p01 * p23 % p, -1, p)
        i01, i23 = inv_all * p23 % p, inv_all * p01 % p
        w = [y[0] * (i01 * d[1] % p) % p, y[1] * (i01 * d[0] % p) % p, y[2] * (i23 * d[3] % p) % p, y[3] * (i23 * d[2] % p) % p]
        c = [0, 0, 0, 0]
        for j, (a, b, e) in enumerate(((x1, x2, x3), (x0, x2, x3), (x0, x1, x3), (x0, x1, x2))):
            ab = a * b % p
            c[0] -= w[j] * (ab * e % p)
            c[1] += w[j] * ((ab + (a + b) * e) % p)
            c[2] -= w[j] * ((a + b + e) % p)
            c[3] += w[j]
        return [v % p for v in c]

    def interpolateQuarticBatch(self, xs, ys):
        return HostMatrix([self._interpolate4(x, y) for x, y in zip(xs.toValues(), ys.toValues())])

    def evalQuarticBatch(self, polys, x):
        return HostVector([self.evalPolyAt(HostVector(row), x) for row in polys.toValues()])

    def __getattr__(self, name):
        raise AttributeError(f'HostField is the verifier-side field: FiniteField.{name} is a prover member (use PrimeField on the HIP backend)')


class HostHash:
    def __init__(self, algorithm, element_size=ELEMENT_SIZE):
        self.elementSize = element_size
        if algorithm not in ('sha256', 'blake2s256'):
            raise TypeError(f'Hash algorithm {algorithm} is not supported')
        self.algorithm, self.digestSize, self.isOptimized = algorithm, DIGEST_SIZE, False
        self._host = hashlib.sha256 if algorithm == 'sha256' else (lambda data: hashlib.blake2s(data, digest_size=32))

    def digest(self, value):
        return self._host(bytes(value)).digest()

    def merge(self, a, b):
        return self.digest(bytes(a) + bytes(b))

    def digestMany(self, messages):
        return [self._host(bytes(m)).digest() for m in messages]

    def digestValues(self, values, valueSize):
        """LowDegreeProver.ts:163 — the rows of the remainder matrix."""
        if isinstance(values, (bytes, bytearray)):
            raw = bytes(values)
        else:
            raw = b''.join(int(v).to_bytes(self.elementSize, 'little') for row in values.toValues() for v in row)
        if len(raw) % valueSize:
            raise GstarkError('Values buffer cannot contain partial number of elements')
        out = HostVector([self._host(raw[i:i + valueSize]).digest() for i in range(0, len(raw), valueSize)], DIGEST_SIZE)
        return out

    def createTree(self, leaves):
        return HostMerkleTree(leaves.toValues(), self)


class HostMerkleTree:
    """MerkleTree.create over host digests (LowDegreeProver.ts:164: the tree over the remainder, at most 64 leaves)."""

    def __init__(self, leaves, hash_):
        n = len(leaves)
        if n < 2 or n & (n - 1):
            raise GstarkError('Number of leaves must be a power of 2')
        nodes = [None] * n
        level = list(leaves)
        width = n
        while width > 1:
            level = hash_.digestMany([level[2 * i] + level[2 * i + 1] for i in range(width // 2)])
            width //= 2
            nodes[width:2 * width] = level
        self.root, self.depth = nodes[1], n.bit_length() - 1
