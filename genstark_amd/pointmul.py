"""The reference's elliptic-curve example, examples/elliptic/pointmul.aa driven like examples/elliptic/pointMul.ts, as a GenericAir
over the 224-bit field 2^224 - 2^96 + 1 (the wide build flavour of the library, csrc/gf_wide.h): double-and-add multiplication
of a point of the curve y^2 = x^3 + a*x + b, a = p - 3 (pointmul.aa:3), by a secret 256-bit scalar.  8 registers, 256 steps per
multiplication:

    r0, r1   P    doubled every step                                   pointmul.aa:70-74
    r2, r3   Q    accumulator: += P on the steps whose scalar bit is 1   :75-80 ($computeQ :32-47)
    r4       m1   slope of the tangent at P        (3*Px^2 + a) / (2*Py) :86-87 ($computeM1 :21-30)
    r5       m2   slope of the chord through P, Q  (Py - Qy) / (Px - Qx) :88-93
    r6       1 while Q is still the point at infinity                    :81-85
    r7       the bits consumed so far as a number: sum bit_i * 2^i       :101-103

What AirAssembly writes as static-register declarations (:58-63) is restated as the columns those declarations denote (see
lib128.py): k0, k1 = the point (secret, held for the whole multiplication, rotated one step earlier), k2 = the scalar's bits
(secret, one per step), k3 = `(mask (input 0))` (1 on the last step of every multiplication), k4 = `(cycle (power 2 256))`.
The transition divides (the slopes); the constraints check the same slopes by cross-multiplication (:123-140).  Division is
x * y^(p-2) in the register machine (0^-1 = 0, as in galois).

Known answer held by the reference: pointMul.ts:22-31 — the point, the scalar and the expected product.
"""
from .air_generic import GenericAir, PackedColumn

STEPS = 256
CURVE_A = -3          # pointmul.aa:3: p - 3
DEGREES = [3, 4, 5, 6, 3, 4, 3, 3]


def add_points(p, q, m):                      # pointmul.aa:5-19 ($addPoints)
    x = m ** 2 - (p[0] + q[0])
    y = m * (p[0] - x) - p[1]
    return [x, y]


def compute_q(p, q, m, k):                    # :32-47 ($computeQ)
    s = add_points(p, q, m)
    return [q[i] * (1 - k[0]) + (p[i] * (k[0] * k[1]) + s[i] * (k[0] * (1 - k[1]))) for i in range(2)]


def point_mul_air(field, count=1, extensionFactor=16):
    """`count` independent multiplications (a power of 2) in one trace of 256 * count steps.  prove(assertions,
    air.expandInputs(raw), air.segmentSeeds(raw)) with raw = [xs, ys, bit lists] (pointMul.ts:22-26: one value each, the bits as
    toBits() orders them, least significant first).  The product of multiplication s is in registers 2, 3 at step 256*s + 255."""
    p = field.modulus
    a = CURVE_A % p
    inv = lambda e: e ** (p - 2)
    m1_of = lambda pt: (3 * pt[0] ** 2 + a) * inv(2 * pt[1])               # :21-30
    public = [[0] * (STEPS - 1) + [1], [pow(2, i, p) for i in range(STEPS)]]
    npub = len(public)

    def lib_order(k):                                                       # the order of pointmul.aa:58-63
        s = k[npub:npub + 3]
        return [s[0], s[1], s[2], k[0], k[1]]

    def init_trace(pt):                                                     # :49-56 ($initTrace)
        return [pt[0], pt[1], 0, 0, m1_of(pt), 0, 1, 0]

    def transition(r, k):                                                   # :66-103
        k = lib_order(k)
        pn = add_points(r[0:2], r[0:2], r[4])
        qn = compute_q(r[0:2], r[2:4], r[5], [k[2], r[6]])
        is_q_null = (1 - k[2]) * r[6]
        m1 = m1_of(pn)
        m2 = (pn[1] - qn[1]) * inv(pn[0] - qn[0]) * (1 - is_q_null)
        regular = pn + qn + [m1, m2, is_q_null, r[7] + k[2] * k[4]]
        return [x * k[3] + y * (1 - k[3]) for x, y in zip(init_trace(k[0:2]), regular)]

    def evaluation(r, n, k):                                                # :104-155
        k = lib_order(k)
        pn = add_points(r[0:2], r[0:2], r[4])
        qn = compute_q(r[0:2], r[2:4], r[5], [k[2], r[6]])
        is_q_null = (1 - k[2]) * r[6]
        m1_check = (3 * r[0] ** 2 + a) - (2 * r[1]) * r[4]
        m2_check = ((r[1] - r[3]) - (r[0] - r[2]) * r[5]) * (1 - r[6])
        left = [n[0], n[1], n[2], n[3], 0, 0, n[6], n[7]]
        fresh = [k[0], k[1], 0, 0, m1_check, m2_check, 1, 0]
        regular = pn + qn + [m1_check, m2_check, is_q_null, r[7] + k[2] * k[4]]
        return [l - (x * k[3] + y * (1 - k[3])) for l, x, y in zip(left, fresh, regular)]

    air = GenericAir(STEPS * count, 8, DEGREES, public, transition, evaluation, lambda seed: list(seed) + [0] * 6, extensionFactor, field,
                     secretRegisters=3, segmentLength=STEPS, maskSegments=False, initExpr=lambda x: init_trace(x[0:2]))
    es = field.elementSize

    def held(values):                       # `(input secret (shift -1))`: one value per multiplication, rotated one step earlier
        flat = b''.join(int(v % p).to_bytes(es, 'little') * STEPS for v in values)
        return PackedColumn(flat[es:] + flat[:es], es)

    def expand(raw):
        xs, ys, bits = raw
        if not (len(xs) == len(ys) == len(bits) == count) or any(len(b) != STEPS for b in bits):
            raise ValueError(f'point_mul_air: {count} points and {count} lists of {STEPS} bits expected')
        return [held(xs), held(ys), PackedColumn(b''.join(int(b % p).to_bytes(es, 'little') for bl in bits for b in bl), es)]
    air.expandInputs = expand
    air.segmentSeeds = lambda raw: [[raw[0][s], raw[1][s]] for s in range(count)]
    return air


def to_bits(value, length=STEPS):
    """pointMul.ts:64-67 (toBits): least significant bit first."""
    return [(value >> i) & 1 for i in range(length)]


def ec_multiply(p, point, scalar):
    """Plain double-and-add on the same curve (control computation on Python integers; affine, no special cases beyond infinity)."""
    a = CURVE_A % p

    def add(u, v):
        if u is None: return v
        if v is None: return u
        if u[0] == v[0]:
            if (u[1] + v[1]) % p == 0: return None
            m = (3 * u[0] * u[0] + a) * pow(2 * u[1], p - 2, p) % p
        else:
            m = (u[1] - v[1]) * pow(u[0] - v[0], p - 2, p) % p
        x = (m * m - u[0] - v[0]) % p
        return x, (m * (u[0] - x) - u[1]) % p
    acc, base = None, point
    while scalar:
        if scalar & 1:
            acc = add(acc, base)
        base = add(base, base)
        scalar >>= 1
    return acc
