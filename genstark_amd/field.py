"""Host mirror of the `@guildofweavers/galois` surface genSTARK calls (FiniteField / Vector / Matrix).

Member names and argument meaning follow the TypeScript interface so that the callers
(genstark_amd/stark.py, genstark_amd/components/*.py — mirrors of lib/Stark.ts and
lib/components/*.ts) read like the reference.  Vectors and matrices are device-resident; every
vector/matrix/polynomial operation is one call into the C ABI (include/gstark.h -> HIP kernels).
Scalar (single bigint) operations and Lagrange interpolation of a handful of points are host
integer arithmetic, exactly as in the reference's JS layer.

Call sites of every member: SURVEY.md section 8(b).
"""
import ctypes as C
import hashlib
import weakref
from hashlib import sha256 as _sha256

from ._abi import Backend, GstarkError

ELEMENT_SIZE = 16
MODULUS = 2**128 - 9 * 2**32 + 1


def _le(v):
    return int(v).to_bytes(ELEMENT_SIZE, 'little')


def sha256_bigint(value):
    """galois utils.sha256 (same helper as lib/components/QueryIndexGenerator.ts:61-67): a bigint is
    hashed as Buffer.from(value.toString(16), 'hex') — no leading zeros, odd trailing nibble dropped."""
    if isinstance(value, int):
        nhex = (value.bit_length() + 3) >> 2
        if nhex & 1:
            value >>= 4                  # odd number of hex digits: Buffer.from(..., 'hex') drops the last nibble
        value = value.to_bytes(nhex >> 1, 'big')
    return int.from_bytes(_sha256(value).digest(), 'big')


class _DeviceBuffer:
    """Owns one gs_alloc'd range; freed when the last Vector/Matrix viewing it is collected."""

    def __init__(self, backend, nbytes):
        self.backend, self.nbytes = backend, nbytes
        self.ptr = backend.alloc(max(nbytes, 16))

    def __del__(self):
        try:
            if self.ptr:
                self.backend.free(self.ptr)
                self.ptr = 0
        except Exception:
            pass


class Vector:
    """galois Vector: `length` elements of `elementSize` bytes, little-endian."""

    def __init__(self, backend, length, owner=None, offset=0, element_size=None):
        element_size = element_size or backend.element_size          # 16; 32 in the 256- / 224-bit library flavours
        self.backend, self.length, self.elementSize = backend, int(length), element_size
        self._owner = owner or _DeviceBuffer(backend, self.length * element_size)
        self._offset = offset
        self.series_base = None  # set when the vector is known to be {base^i}: lets NTT calls skip a readback

    @property
    def ptr(self):
        return self._owner.ptr + self._offset

    @property
    def byteLength(self):
        return self.length * self.elementSize

    def getValue(self, index):
        return int.from_bytes(self.backend.download(self.ptr, self.elementSize, index * self.elementSize), 'little')

    def toValues(self):
        raw = self.toBuffer()
        es = self.elementSize
        return [int.from_bytes(raw[i * es:(i + 1) * es], 'little') for i in range(self.length)]

    def toBuffer(self, start=0, count=None):
        count = self.length - start if count is None else count
        return self.backend.download(self.ptr, count * self.elementSize, start * self.elementSize)

    def copyValue(self, index, destination, offset):
        """lib/Stark.ts:290 — writes element `index` into `destination` (bytearray), returns bytes written."""
        destination[offset:offset + self.elementSize] = self.backend.download(
            self.ptr, self.elementSize, index * self.elementSize)
        return self.elementSize

    def valuesAt(self, indexes):
        """Batched copyValue: the raw bytes of the elements at `indexes` (one device gather)."""
        return self.backend.gather(self.ptr, self.elementSize, list(indexes))


class Matrix:
    """galois Matrix: rowCount x colCount elements, row-major contiguous (LowDegreeProver.ts:45)."""

    def __init__(self, backend, rows, cols, owner=None, offset=0):
        self.backend, self.rowCount, self.colCount, self.elementSize = backend, int(rows), int(cols), backend.element_size
        self._owner = owner or _DeviceBuffer(backend, self.rowCount * self.colCount * self.elementSize)
        self._offset = offset
        self.quartic_domain = None  # (omega, n, step) when built by transposeVector(power series, 4, step)

    @property
    def ptr(self):
        return self._owner.ptr + self._offset

    def getValue(self, row, col):
        off = (row * self.colCount + col) * self.elementSize
        return int.from_bytes(self.backend.download(self.ptr, self.elementSize, off), 'little')

    def toValues(self):
        raw = self.toBuffer()
        c, es = self.colCount, self.elementSize
        return [[int.from_bytes(raw[(r * c + k) * es:(r * c + k + 1) * es], 'little') for k in range(c)]
                for r in range(self.rowCount)]

    def toBuffer(self):
        return self.backend.download(self.ptr, self.rowCount * self.colCount * self.elementSize)

    def rowsToBuffers(self, indexes):
        """LowDegreeProver.ts:53,214,217 — one Buffer (colCount*elementSize bytes) per requested row."""
        return self.backend.gather(self.ptr, self.colCount * self.elementSize, list(indexes))

    def row(self, r):
        return Vector(self.backend, self.colCount, owner=self._owner, offset=self._offset + r * self.colCount * self.elementSize)


class PrimeField:
    """createPrimeField(modulus[, wasmOptions]) — index.ts:14; lib/Stark.ts:40."""

    def __init__(self, modulus=None, backend=None):
        """modulus None: the field the backend's library is built for (default library: the 128-bit field)."""
        self.backend = backend or Backend(modulus=modulus)
        if modulus is None:
            modulus = self.backend.modulus
        if modulus != self.backend.modulus:
            raise GstarkError(f'the backend computes in the field of {self.backend.modulus} elements, not {modulus}: '
                              'construct Backend(modulus=...) for one of the built fields')
        self._lastEvaluation = None     # (polys, ptr, omega, n, result) of the latest evalPolysAtRoots, weakly held
        self.modulus = modulus
        self.elementSize = self.backend.element_size   # 16 bytes; 32 in the 256- / 224-bit library flavours
        self.zero, self.one = 0, 1

    def le(self, v):
        return int(v).to_bytes(self.elementSize, 'little')

    # ---- scalar arithmetic (bigint in, bigint out)
    def mod(self, v): return v % self.modulus
    def add(self, a, b): return (a + b) % self.modulus
    def sub(self, a, b): return (a - b) % self.modulus
    def mul(self, a, b): return (a * b) % self.modulus
    def neg(self, a): return (-a) % self.modulus
    def inv(self, a): return pow(a, -1, self.modulus) if a % self.modulus else 0
    def div(self, a, b): return a * self.inv(b) % self.modulus

    def exp(self, base, exponent):
        if exponent < 0:  # examples/rescue/hash4x128.ts:14 passes a negative exponent
            return pow(self.inv(base), -exponent, self.modulus)
        return pow(base, exponent, self.modulus)

    def prng(self, seed, length=None):
        """FiniteField.prng (CompositionPolynomial.ts:58; LinearCombination.ts:58; LowDegreeProver.ts:194).
        Construction restated from galois' documented behaviour (UNVERIFIED)."""
        if length is None:
            return sha256_bigint(seed) % self.modulus
        out, state = [], sha256_bigint(seed)
        for _ in range(length):
            out.append(state % self.modulus)
            state = sha256_bigint(state)
        return self.newVectorFrom(out)

    def getRootOfUnity(self, order):
        """First i^((p-1)/order), i = 2,3,..., of exact order `order` (UNVERIFIED)."""
        if order <= 0 or order & (order - 1) or (self.modulus - 1) % order:
            raise GstarkError(f'Order {order} of root of unity is invalid')
        for i in range(2, 1 << 16):
            g = pow(i, (self.modulus - 1) // order, self.modulus)
            if pow(g, order, self.modulus) == 1 and (order == 1 or pow(g, order // 2, self.modulus) != 1):
                return g
        raise GstarkError(f'Root of unity for order {order} was not found')

    # ---- construction
    def newVector(self, length):
        return Vector(self.backend, length)

    def newVectorFrom(self, values):
        v = Vector(self.backend, len(values))
        if values:
            self.backend.upload(v.ptr, b''.join(self.le(x % self.modulus) for x in values))
        return v

    def newMatrix(self, rows, cols):
        return Matrix(self.backend, rows, cols)

    def newMatrixFrom(self, values):
        rows, cols = len(values), len(values[0]) if values else 0
        m = Matrix(self.backend, rows, cols)
        if rows * cols:
            self.backend.upload(m.ptr, b''.join(self.le(x % self.modulus) for r in values for x in r))
        return m

    def newMatrixFromVectors(self, vectors):
        # rows shorter than the longest one are zero-extended (polynomials of different degrees, e.g. boundary
        # interpolants of registers with different numbers of assertions: BoundaryConstraints.ts:84-85)
        cols = max(v.length for v in vectors)
        m = Matrix(self.backend, len(vectors), cols)
        for r, v in enumerate(vectors):
            self.backend.call('gs_copy', C.c_void_p(m.ptr + r * cols * self.elementSize), C.c_void_p(v.ptr), v.length * self.elementSize)
            if v.length < cols:
                self.backend.upload(m.ptr + (r * cols + v.length) * self.elementSize, bytes((cols - v.length) * self.elementSize))
        return m

    def matrixRowsToVectors(self, matrix):
        return [matrix.row(r) for r in range(matrix.rowCount)]

    # ---- vector operations
    def _binary(self, fn_vec, fn_scalar, a, b):
        out = Vector(self.backend, a.length)
        if isinstance(b, int):
            self.backend.call(fn_scalar, C.c_void_p(a.ptr), self.le(b % self.modulus), a.length, C.c_void_p(out.ptr))
        else:
            if a.length != b.length:
                raise GstarkError('Cannot combine vector elements: vectors have different lengths')
            self.backend.call(fn_vec, C.c_void_p(a.ptr), C.c_void_p(b.ptr), a.length, C.c_void_p(out.ptr))
        return out

    def addVectorElements(self, a, b): return self._binary('gs_vec_add', 'gs_vec_add_scalar', a, b)
    def subVectorElements(self, a, b): return self._binary('gs_vec_sub', 'gs_vec_sub_scalar', a, b)
    def mulVectorElements(self, a, b): return self._binary('gs_vec_mul', 'gs_vec_mul_scalar', a, b)

    def divVectorElements(self, a, b):
        if isinstance(b, int):
            return self.mulVectorElements(a, self.inv(b))
        if a.length != b.length:
            raise GstarkError('Cannot divide vector elements: vectors have different lengths')
        out = Vector(self.backend, a.length)
        self.backend.call('gs_vec_div', C.c_void_p(a.ptr), C.c_void_p(b.ptr), a.length, C.c_void_p(out.ptr))
        return out

    def invVectorElements(self, a):
        out = Vector(self.backend, a.length)
        self.backend.call('gs_vec_inv', C.c_void_p(a.ptr), a.length, C.c_void_p(out.ptr))
        return out

    def expVectorElements(self, a, e):
        out = Vector(self.backend, a.length)
        if e < 0:
            a, e = self.invVectorElements(a), -e
        self.backend.call('gs_vec_exp', C.c_void_p(a.ptr), self.le(e), a.length, C.c_void_p(out.ptr))
        return out

    # ---- fused forms of divisions whose denominators are known in closed form over the domain (no counterpart in galois: same
    # values as the member sequences named in include/gstark.h, one kernel each)
    fusedDomainDivisions = True          # the components ask before using the two members below (a distributed field says no)

    def zeroPolyInverses(self, rootOfUnity, domainSize, steps, xAtLastStep):
        """1/Z(x) over the evaluation domain: ZeroPolynomial.evaluateAll + CompositionPolynomial.ts:117 (den / num)."""
        out = Vector(self.backend, domainSize)
        self.backend.call('gs_zero_poly_inverses', self.le(rootOfUnity), domainSize, steps, self.le(xAtLastStep), C.c_void_p(out.ptr))
        return out

    MAX_DOMAIN_ROOTS = 4

    def divByDomainRoots(self, numerators, rootOfUnity, rootIndexes):
        """numerators[r][i] / prod_k (omega^i - omega^k), k in rootIndexes[r]: BoundaryConstraints.ts:87-92 when every divisor's
        roots are domain points (at most MAX_DOMAIN_ROOTS per row)."""
        rows, n = numerators.rowCount, numerators.colCount
        width = max(len(r) for r in rootIndexes)
        flat = (C.c_uint64 * (rows * width))(*[(r[a] if a < len(r) else 0) for r in rootIndexes for a in range(width)])
        counts = (C.c_uint32 * rows)(*[len(r) for r in rootIndexes])
        out = Matrix(self.backend, rows, n)
        self.backend.call('gs_div_by_domain_roots', C.c_void_p(numerators.ptr), rows, n, self.le(rootOfUnity), flat, counts, width, C.c_void_p(out.ptr))
        return out

    def combineVectors(self, a, b):
        if a.length != b.length:
            raise GstarkError('Cannot combine vectors: vectors have different lengths')
        buf = C.create_string_buffer(self.elementSize)
        self.backend.call('gs_combine', C.c_void_p(a.ptr), C.c_void_p(b.ptr), a.length, C.cast(buf, C.c_void_p))
        return int.from_bytes(buf.raw, 'little')

    def combineManyVectors(self, vectors, coefficients):
        ks = coefficients.toValues() if isinstance(coefficients, Vector) else list(coefficients)
        if len(ks) != len(vectors):
            raise GstarkError('Number of coefficients must be the same as the number of vectors')
        n = vectors[0].length
        out = Vector(self.backend, n)
        self.backend.call('gs_combine_many', Backend.ptr_array([v.ptr for v in vectors]),
                          b''.join(self.le(k) for k in ks), len(vectors), n, C.c_void_p(out.ptr))
        return out

    def getPowerSeries(self, base, length):
        out = Vector(self.backend, length)
        self.backend.call('gs_power_series', self.le(base), length, C.c_void_p(out.ptr))
        out.series_base = base % self.modulus
        return out

    def pluckVector(self, v, skip, times):
        out = Vector(self.backend, times)
        self.backend.call('gs_pluck', C.c_void_p(v.ptr), v.length, skip, times, C.c_void_p(out.ptr))
        return out

    def transposeVector(self, v, columns, step=1):
        if v.length % (columns * step):
            raise GstarkError('Number of columns must evenly divide vector length')
        rows = v.length // (columns * step)
        m = Matrix(self.backend, rows, columns)
        self.backend.call('gs_transpose_vector', C.c_void_p(v.ptr), v.length, columns, step, C.c_void_p(m.ptr))
        if columns == 4 and v.series_base is not None:
            m.quartic_domain = (v.series_base, v.length, step)
        return m

    # ---- matrix operations
    def transposeMatrix(self, m):
        out = Matrix(self.backend, m.colCount, m.rowCount)
        self.backend.call('gs_transpose_matrix', C.c_void_p(m.ptr), m.rowCount, m.colCount, C.c_void_p(out.ptr))
        return out

    def mulMatrixByVector(self, m, v):
        """galois FiniteField.mulMatrixByVector (examples/poseidon/utils.ts:45: `mds x state`): out[r] = sum_c m[r][c] * v[c],
        one device dot product per row."""
        if m.colCount != v.length:
            raise GstarkError('Matrix column count must be the same as vector length')
        return self.newVectorFrom([self.combineVectors(m.row(r), v) for r in range(m.rowCount)])

    def joinMatrixRows(self, m):
        return Vector(self.backend, m.rowCount * m.colCount, owner=m._owner, offset=m._offset)

    def subMatrixElementsFromVectors(self, vectors, m):
        out = Matrix(self.backend, m.rowCount, m.colCount)
        self.backend.call('gs_sub_matrix_from_vectors', Backend.ptr_array([v.ptr for v in vectors]),
                          C.c_void_p(m.ptr), m.rowCount, m.colCount, C.c_void_p(out.ptr))
        return out

    def divMatrixElements(self, a, b):
        out = Matrix(self.backend, a.rowCount, a.colCount)
        self.backend.call('gs_vec_div', C.c_void_p(a.ptr), C.c_void_p(b.ptr), a.rowCount * a.colCount, C.c_void_p(out.ptr))
        return out

    # ---- polynomials
    def _omega_of(self, roots):
        if roots.series_base is not None:
            return roots.series_base
        return roots.getValue(1) if roots.length > 1 else 1

    def evalPolyAtRoots(self, poly, roots):
        """CompositionPolynomial.ts:110 — NTT of `poly` (zero-extended) over the roots-of-unity vector."""
        if poly.length > roots.length:
            raise GstarkError('Number of roots of unity cannot be smaller than number of values')
        out = Vector(self.backend, roots.length)
        self.backend.call('gs_eval_polys_at_roots', C.c_void_p(poly.ptr), 1, poly.length, self.le(self._omega_of(roots)),
                          roots.length, C.c_void_p(out.ptr))
        return out

    def evalPolysAtRoots(self, polys, roots):
        """lib/Stark.ts:109; BoundaryConstraints.ts:87-88.  The evaluations of the SAME polynomials over a subgroup of the
        domain they were last evaluated on (the composition domain after the evaluation domain: lib/Stark.ts:109 then
        CompositionPolynomial.ts:76) are every k-th element of that result: a strided copy instead of another NTT."""
        if polys.colCount > roots.length:
            raise GstarkError('Number of roots of unity cannot be smaller than number of values')
        n, omega = roots.length, self._omega_of(roots)
        out = Matrix(self.backend, polys.rowCount, n)
        prev = self._lastEvaluation
        if prev is not None:
            ppolys, pptr, pomega, pn, pout = prev[0](), prev[1], prev[2], prev[3], prev[4]()
            if ppolys is polys and pptr == polys.ptr and pout is not None and pn > n and pn % n == 0 and \
                    pow(pomega, pn // n, self.modulus) == omega:
                for r in range(polys.rowCount):
                    self.backend.call('gs_pluck', C.c_void_p(pout.ptr + r * pn * self.elementSize), pn, pn // n, n,
                                      C.c_void_p(out.ptr + r * n * self.elementSize))
                return out
        self.backend.call('gs_eval_polys_at_roots', C.c_void_p(polys.ptr), polys.rowCount, polys.colCount, self.le(omega), n,
                          C.c_void_p(out.ptr))
        self._lastEvaluation = (weakref.ref(polys), polys.ptr, omega, n, weakref.ref(out))
        return out

    def interpolateRoots(self, roots, ys):
        """lib/Stark.ts:106; CompositionPolynomial.ts:109 — inverse NTT (Vector or Matrix of rows)."""
        n = roots.length
        if isinstance(ys, Matrix):
            if ys.colCount != n:
                raise GstarkError('Number of roots of unity must be the same as the number of y coordinates')
            out = Matrix(self.backend, ys.rowCount, n)
            rows = ys.rowCount
        else:
            if ys.length != n:
                raise GstarkError('Number of roots of unity must be the same as the number of y coordinates')
            out, rows = Vector(self.backend, n), 1
        self.backend.call('gs_interpolate_roots', C.c_void_p(ys.ptr), rows, self.le(self._omega_of(roots)), n, C.c_void_p(out.ptr))
        return out

    def evalPolyAt(self, poly, x):
        buf = C.create_string_buffer(self.elementSize)
        self.backend.call('gs_eval_poly_at', C.c_void_p(poly.ptr), poly.length, self.le(x), C.cast(buf, C.c_void_p))
        return int.from_bytes(buf.raw, 'little')

    def mulPolys(self, a, b):
        """BoundaryConstraints.ts:30 — product of two polynomials.  The reference only multiplies tiny ones (degree <= number
        of assertions): those stay on the host; larger operands go through the device NTT (evaluate both on a domain of
        >= len(a)+len(b)-1 roots, multiply pointwise, interpolate)."""
        la, lb = a.length, b.length
        if la * lb <= 4096:
            av, bv = a.toValues(), b.toValues()
            out = [0] * (la + lb - 1)
            for i, x in enumerate(av):
                for j, y in enumerate(bv):
                    out[i + j] = (out[i + j] + x * y) % self.modulus
            return self.newVectorFrom(out)
        n = 1 << (la + lb - 2).bit_length()
        roots = self.getPowerSeries(self.getRootOfUnity(n), n)
        prod = self.mulVectorElements(self.evalPolyAtRoots(a, roots), self.evalPolyAtRoots(b, roots))
        full = self.interpolateRoots(roots, prod)
        return Vector(self.backend, la + lb - 1, owner=full._owner, offset=full._offset)

    def _pad_poly(self, v, length):
        if v.length == length:
            return v
        out = Vector(self.backend, length)
        self.backend.call('gs_copy', C.c_void_p(out.ptr), C.c_void_p(v.ptr), v.length * self.elementSize)
        self.backend.upload(out.ptr + v.length * self.elementSize, bytes((length - v.length) * self.elementSize))
        return out

    def addPolys(self, a, b):
        """galois FiniteField.addPolys: coefficient-wise sum, the shorter operand zero-extended."""
        n = max(a.length, b.length)
        return self.addVectorElements(self._pad_poly(a, n), self._pad_poly(b, n))

    def subPolys(self, a, b):
        n = max(a.length, b.length)
        return self.subVectorElements(self._pad_poly(a, n), self._pad_poly(b, n))

    def mulPolyByConstant(self, a, c):
        return self.mulVectorElements(a, c % self.modulus)

    def interpolate(self, xs, ys):
        """BoundaryConstraints.ts:42; LowDegreeProver.ts:243 — Lagrange through a handful of points: O(n^2) host
        arithmetic inside the library (gs_small_interpolate), as in the reference's JS layer."""
        xv = xs if isinstance(xs, (list, tuple)) else xs.toValues()
        yv = ys if isinstance(ys, (list, tuple)) else ys.toValues()
        if len(xv) != len(yv):
            raise GstarkError('Number of x coordinates must be the same as number of y coordinates')
        return self.newVectorFrom(self.interpolateValues(xv, yv))

    def interpolateValues(self, xv, yv):
        n = len(xv)
        es = self.elementSize
        out = C.create_string_buffer(es * n)
        rc = self.backend.lib.gs_small_interpolate(b''.join(self.le(x % self.modulus) for x in xv),
                                                   b''.join(self.le(y % self.modulus) for y in yv), n, C.cast(out, C.c_void_p))
        if rc:
            raise GstarkError(f'gs_small_interpolate failed ({rc})')
        raw = out.raw
        return [int.from_bytes(raw[es * i:es * i + es], 'little') for i in range(n)]

    def evalPolyAtMany(self, poly_values, xs):
        """evalPolyAt over a list of points (LowDegreeProver.ts:246-251), host arithmetic inside the library."""
        m = len(xs)
        es = self.elementSize
        out = C.create_string_buffer(es * max(m, 1))
        rc = self.backend.lib.gs_small_eval_poly(b''.join(self.le(c) for c in poly_values), len(poly_values),
                                                 b''.join(self.le(x) for x in xs), m, C.cast(out, C.c_void_p))
        if rc:
            raise GstarkError(f'gs_small_eval_poly failed ({rc})')
        raw = out.raw
        return [int.from_bytes(raw[es * i:es * i + es], 'little') for i in range(m)]

    def interpolateQuarticBatch(self, xs, ys):
        """LowDegreeProver.ts:137,191 — one cubic per row.  When xs is the transposed power-series domain
        the prover builds (LowDegreeProver.ts:190) the x coordinates are regenerated on the fly."""
        if xs.rowCount != ys.rowCount or xs.colCount != 4 or ys.colCount != 4:
            raise GstarkError('X and Y coordinate matrixes must have the same shape with 4 columns')
        out = Matrix(self.backend, ys.rowCount, 4)
        if xs.quartic_domain is not None:
            omega, n, step = xs.quartic_domain
            self.backend.call('gs_interpolate_quartic_domain', self.le(omega), n, step, C.c_void_p(ys.ptr), ys.rowCount,
                              C.c_void_p(out.ptr))
        else:
            self.backend.call('gs_interpolate_quartic_batch', C.c_void_p(xs.ptr), C.c_void_p(ys.ptr), ys.rowCount,
                              C.c_void_p(out.ptr))
        return out

    def evalQuarticBatch(self, polys, x):
        out = Vector(self.backend, polys.rowCount)
        self.backend.call('gs_eval_quartic_batch', C.c_void_p(polys.ptr), polys.rowCount, self.le(x), C.c_void_p(out.ptr))
        return out


def createPrimeField(modulus=MODULUS, backend=None):
    return PrimeField(modulus, backend)
