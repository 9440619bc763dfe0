"""The reference's AirAssembly library for the 224-bit field 2^224 - 2^96 + 1, assembly/lib224.aa, as GenericAirs on the wide build
flavour of the library (csrc/gf_wide.h), driven like examples/assembly/lib224.ts drives them:

    ComputePoseidonHash      3 registers, Poseidon of width 3 (x^5, 8 full + 55 partial rounds), lib224.aa:329-364
    ComputeMerkleRoot        6 registers, one authentication path of single-element nodes, :367-389
    ComputeMerkleUpdate      12 registers + `bit^2 = bit`, old and new path side by side, :392-431
    VerifySchnorrSignature   14 registers, 18 constraints: s*G and R + h*P by double-and-add on y^2 = x^3 - 3x + b, :142-326

Static-register declarations are restated as the columns they denote, exactly as in lib128.py (held inputs rotated by -1, masks,
cyclic columns); secret inputs are GenericAir secret registers.  The MDS matrix is DATA of lib224.aa:9-12; round constants are
air-assembly's `prng sha256 <seed> 64` (UNVERIFIED restatement, genstark_amd/air.py: sha256_prng — the example's own control
computation uses the same generator, lib224.ts:20-24, so round trips do not depend on it).

Known answer held by the reference: lib224.ts:184-188 — a generator, a public key, a nonce point and a signature (s, h) for which
s*G = R + h*P must hold; the trace below reproduces it (registers 2, 3 = registers 9, 10 on the last row).
"""
from .air import sha256_prng
from .air_generic import GenericAir, PackedColumn, mat_vec
from .pointmul import CURVE_A, add_points, compute_q, to_bits  # noqa: F401  (the same $addPoints / $computeQ: lib224.aa:14-58)
from .poseidon import F_ROUNDS, P_ROUNDS, round_controls

MODULUS = 2**224 - 2**96 + 1
WIDTH = 3
ROUND_STEPS = F_ROUNDS + P_ROUNDS + 1      # 64
SIG_STEPS = 256
MDS = [
    [10008242661848396561239390234674179205368613625200051069729988268390, 9185485104865875859138105757221792016506107743032910484553683820527, 13022115361163578305030570201821707355852025945355679498341734719229],
    [16713671886491979986124218739510375851266164236167090939379707559363, 6297058756837203650911510387243124482364311079232836613813757262194, 9703880437758597963751121827371255747687161183939416984217938944391],
    [9528240692195379674131243771445039054419001347578518495333298828602, 5855344777004640210384918417896799913056422502669197740330508636122, 10933989424836635492481388167692471478355813503558543008871486397534],
]


def round_constant_columns(f):
    """lib224.aa:337-339 / lib224.ts:20-24: three cyclic columns of 64 values, seeds 'Hades1' .. 'Hades3'."""
    return [sha256_prng(bytes.fromhex('48616465733' + str(j + 1)), ROUND_STEPS, f) for j in range(WIDTH)]


def poseidon_hash(f, inputs):
    """examples/poseidon/utils.ts:19-49 with the round constants and MDS matrix of lib224 (lib224.ts:52-53: the control value)."""
    p = f.modulus
    cols = round_constant_columns(f)
    state = [v % p for v in inputs] + [0] * (WIDTH - len(inputs))
    for i in range(F_ROUNDS + P_ROUNDS):
        state = [(s + cols[j][i]) % p for j, s in enumerate(state)]
        if i < F_ROUNDS // 2 or i >= F_ROUNDS // 2 + P_ROUNDS:
            state = [pow(s, 5, p) for s in state]
        else:
            state[WIDTH - 1] = pow(state[WIDTH - 1], 5, p)
        state = [sum(a * b for a, b in zip(row, state)) % p for row in MDS]
    return state[:2]


def poseidon_round(state, keys, full):       # lib224.aa:78-97 ($poseidonRound)
    full_round = mat_vec(MDS, [(s + k) ** 5 for s, k in zip(state, keys)])
    part_round = mat_vec(MDS, [state[0] + keys[0], state[1] + keys[1], (state[2] + keys[2]) ** 5])
    return [a * full + b * (1 - full) for a, b in zip(full_round, part_round)]


def init_merkle_hash(p, v):                  # lib224.aa:100-106 ($initMerkleHash)
    return [p, v, 0, v, p, 0]


def _held(field, values, steps):
    """`(input ... (steps S) (shift -1))` as a PackedColumn: values[j] during steps [j*S, (j+1)*S), rotated one step earlier."""
    es = field.elementSize
    flat = b''.join(int(v % field.modulus).to_bytes(es, 'little') * steps for v in values)
    return PackedColumn(flat[es:] + flat[:es], es)


def _held_list(values, steps, total):
    return [values[((i + 1) // steps) % len(values)] for i in range(total)]


def segment_mask(steps):
    return [0] * (steps - 1) + [1]


def compute_poseidon_hash_air(field, hashes=1, extensionFactor=32):
    """lib224.aa:329-364.  prove(assertions, air.expandInputs(raw), air.segmentSeeds(raw)) with raw = the two secret input registers,
    each a list of `hashes` values (lib224.ts:56: [[42n], [43n]]).  The digest of hash number s is in registers 0, 1 at step
    64*s + 63 (lib224.ts:57-60)."""
    total = ROUND_STEPS * hashes
    public = [segment_mask(ROUND_STEPS), round_controls()] + round_constant_columns(field)      # k[0] mask, k[1] round kind, k[2..4] constants
    npub = len(public)

    def transition(r, k):                                                   # :343-352
        mask = k[0]
        rnd = poseidon_round(r, k[2:5], k[1])
        start = [k[npub], k[npub + 1], 0]
        return [s * mask + x * (1 - mask) for s, x in zip(start, rnd)]

    def evaluation(r, n, k):                                                # :353-364
        return [a - b for a, b in zip(n, transition(r, k))]

    air = GenericAir(total, WIDTH, [7] * WIDTH, public, transition, evaluation, lambda seed: list(seed) + [0], extensionFactor, field,
                     secretRegisters=2, segmentLength=ROUND_STEPS, maskSegments=False)
    air.expandInputs = lambda inputs: [_held(field, col, ROUND_STEPS) for col in inputs]
    air.segmentSeeds = lambda inputs: [[col[s] for col in inputs] for s in range(hashes)]
    return air


def _merkle_transition(r, k):                 # lib224.aa:109-139 ($merkleTransition); k: leaf, node, bit, leaf mask, node mask, round kind, 3 constants
    h1 = poseidon_round(r[0:3], k[6:9], k[5])
    h2 = poseidon_round(r[3:6], k[6:9], k[5])
    h = r[3] * k[2] + r[0] * (1 - k[2])
    a = init_merkle_hash(k[0], k[1])
    b = init_merkle_hash(h, k[1])
    return [x * k[3] + (y * ((1 - k[3]) * k[4]) + z * ((1 - k[3]) * (1 - k[4]))) for x, y, z in zip(a, b, h1 + h2)]


def compute_merkle_root_air(field, index_bits, extensionFactor=32):
    """lib224.aa:367-389 for ONE authentication path of depth len(index_bits) (a power of 2).  index_bits is the PUBLIC input
    register (already shifted as lib224.ts:87-89 does); inputs and first row from merkle_inputs().  The root is in register 0 at
    step 64*depth - 1 (lib224.ts:96-98)."""
    depth = len(index_bits)
    total = ROUND_STEPS * depth
    public = [_held_list([b % field.modulus for b in index_bits], ROUND_STEPS, total), segment_mask(total), segment_mask(ROUND_STEPS),
              round_controls()] + round_constant_columns(field)
    npub = len(public)
    lib_order = lambda k: [k[npub], k[npub + 1]] + list(k[0:npub])          # secret leaf, nodes, then the public columns
    transition = lambda r, k: _merkle_transition(r, lib_order(k))
    evaluation = lambda r, n, k: [a - b for a, b in zip(n, transition(r, k))]
    return GenericAir(total, 6, [8] * 6, public, transition, evaluation, lambda seed: list(seed), extensionFactor, field, secretRegisters=2)


def merkle_inputs(field, leaf, nodes):
    """The secret columns of ComputeMerkleRoot and its first row (lib224.aa:381-382) from an authentication path (lib224.ts:91-92)."""
    total = ROUND_STEPS * len(nodes)
    return [_held(field, [leaf], total), _held(field, list(nodes), ROUND_STEPS)], [v % field.modulus for v in init_merkle_hash(leaf, nodes[0])]


def compute_merkle_update_air(field, depth, extensionFactor=32):
    """lib224.aa:392-431: old and new leaf at the same SECRET index; the old root is in register 0 and the new root in register 6 at
    step 64*depth - 1 (lib224.ts:140-143)."""
    total = ROUND_STEPS * depth
    public = [segment_mask(total), segment_mask(ROUND_STEPS), round_controls()] + round_constant_columns(field)
    npub = len(public)

    def halves(k):
        s = k[npub:npub + 4]                  # oldLeaf, newLeaf, nodes, index bit
        shared = [s[2], s[3]] + list(k[0:npub])
        return [s[0]] + shared, [s[1]] + shared

    def transition(r, k):                     # :411-419
        old, new = halves(k)
        return _merkle_transition(r[0:6], old) + _merkle_transition(r[6:12], new)

    def evaluation(r, n, k):                  # :420-431
        bit = k[npub + 3]
        return [a - b for a, b in zip(n, transition(r, k))] + [bit ** 2 - bit]

    return GenericAir(total, 12, [8] * 12 + [2], public, transition, evaluation, lambda seed: list(seed), extensionFactor, field, secretRegisters=4)


def merkle_update_inputs(field, old_leaf, new_leaf, nodes, index_bits):
    total = ROUND_STEPS * len(nodes)
    cols = [_held(field, [old_leaf], total), _held(field, [new_leaf], total), _held(field, list(nodes), ROUND_STEPS),
            _held(field, list(index_bits), ROUND_STEPS)]
    first = init_merkle_hash(old_leaf, nodes[0]) + init_merkle_hash(new_leaf, nodes[0])
    return cols, [v % field.modulus for v in first]


class PoseidonMerkleTree:
    """examples/poseidon/utils.ts:169-192 (MerkleTree2: single-element nodes) with lib224's hash: the example's control."""

    def __init__(self, field, leaves):
        n = len(leaves)
        self.nodes = [None] * n + list(leaves)
        for i in range(n - 1, 0, -1):
            self.nodes[i] = poseidon_hash(field, [self.nodes[2 * i], self.nodes[2 * i + 1]])[0]

    @property
    def root(self):
        return self.nodes[1]

    def prove(self, index):
        index += len(self.nodes) // 2
        proof = [self.nodes[index]]
        while index > 1:
            proof.append(self.nodes[index ^ 1])
            index >>= 1
        return proof


# ---- Schnorr signature verification ---------------------------------------------------------------------------------------------
SCHNORR_DEGREES = [3, 4, 5, 6, 3, 4, 3, 3, 4, 4, 5, 3, 3, 3, 2, 2, 2, 2]


def verify_schnorr_signature_air(field, count=1, extensionFactor=16):
    """lib224.aa:142-326.  `count` signatures (a power of 2), 256 steps each.  prove(assertions, air.expandInputs(raw),
    air.segmentSeeds(raw)) with raw = [Gx, Gy, s bits, Px, Py, h bits, Rx, Ry], one entry per signature (lib224.ts:190-196).
    Registers: 0-6 the multiplication s*G (as in pointmul.py), 7-12 the accumulation R + h*P (Q starts at R), 13 the bits of h
    consumed so far as a number (lib224.ts:198-208 asserts the inputs on step 0 and h on step 255)."""
    p = field.modulus
    a = CURVE_A % p
    inv = lambda e: e ** (p - 2)
    m1_of = lambda pt: (3 * pt[0] ** 2 + a) * inv(2 * pt[1])               # :61-70 ($computeM1)
    public = [segment_mask(SIG_STEPS), [pow(2, i, p) for i in range(SIG_STEPS)]]
    npub = len(public)

    def lib_order(k):                                                       # the order of lib224.aa:145-154: 8 inputs, mask, powers of 2
        return list(k[npub:npub + 8]) + [k[0], k[1]]

    def init_trace(g, pk, r):                                               # :60-76 ($initSchnorrTrace)
        return [g[0], g[1], 0, 0, m1_of(g), 0, 1, pk[0], pk[1], r[0], r[1], m1_of(pk), (pk[1] - r[1]) * inv(pk[0] - r[0]), 0]

    def steps_of(r, k):
        p1 = add_points(r[0:2], r[0:2], r[4])
        p2 = add_points(r[7:9], r[7:9], r[11])
        q1 = compute_q(r[0:2], r[2:4], r[5], [k[2], r[6]])
        q2 = compute_q(r[7:9], r[9:11], r[12], [k[5], 0])
        return p1, p2, q1, q2, (1 - k[2]) * r[6]

    def transition(r, k):                                                   # :160-221
        k = lib_order(k)
        p1, p2, q1, q2, is_q1_null = steps_of(r, k)
        mq1 = (p1[1] - q1[1]) * inv(p1[0] - q1[0]) * (1 - is_q1_null)
        mq2 = (p2[1] - q2[1]) * inv(p2[0] - q2[0])
        regular = p1 + q1 + [m1_of(p1), mq1, is_q1_null] + p2 + q2 + [m1_of(p2), mq2, r[13] + k[5] * k[9]]
        return [x * k[8] + y * (1 - k[8]) for x, y in zip(init_trace(k[0:2], k[3:5], k[6:8]), regular)]

    def evaluation(r, n, k):                                                # :223-326
        k = lib_order(k)
        p1, p2, q1, q2, is_q1_null = steps_of(r, k)
        mp1_check = (3 * r[0] ** 2 + a) - (2 * r[1]) * r[4]
        mp2_check = (3 * r[7] ** 2 + a) - (2 * r[8]) * r[11]
        mq1_check = ((r[1] - r[3]) - (r[0] - r[2]) * r[5]) * (1 - r[6])
        mq2_check = (r[8] - r[10]) - (r[7] - r[9]) * r[12]
        left = [n[0], n[1], n[2], n[3], 0, 0, n[6], n[7], n[8], n[9], n[10], 0, 0, n[13], k[2] ** 2 - k[2], k[5] ** 2 - k[5], 0, 0]
        fresh = [k[0], k[1], 0, 0, mp1_check, mq1_check, 1, k[3], k[4], k[6], k[7], mp2_check, mq2_check, 0, 0, 0, r[2] - r[9], r[3] - r[10]]
        regular = p1 + q1 + [mp1_check, mq1_check, is_q1_null] + p2 + q2 + [mp2_check, mq2_check, r[13] + k[5] * k[9], 0, 0, 0, 0]
        return [l - (x * k[8] + y * (1 - k[8])) for l, x, y in zip(left, fresh, regular)]

    air = GenericAir(SIG_STEPS * count, 14, SCHNORR_DEGREES, public, transition, evaluation, lambda seed: list(seed) + [0] * 8, extensionFactor,
                     field, secretRegisters=8, segmentLength=SIG_STEPS, maskSegments=False,
                     initExpr=lambda x: init_trace(x[0:2], x[2:4], x[4:6]))
    es = field.elementSize

    def expand(raw):
        if len(raw) != 8 or any(len(col) != count for col in raw) or any(len(b) != SIG_STEPS for b in raw[2] + raw[5]):
            raise ValueError(f'verify_schnorr_signature_air: 8 input registers of {count} entries, bit lists of {SIG_STEPS}')
        bits = lambda lists: PackedColumn(b''.join(int(b % p).to_bytes(es, 'little') for bl in lists for b in bl), es)
        return [bits(col) if j in (2, 5) else _held(field, col, SIG_STEPS) for j, col in enumerate(raw)]
    air.expandInputs = expand
    air.segmentSeeds = lambda raw: [[raw[0][s], raw[1][s], raw[3][s], raw[4][s], raw[6][s], raw[7][s]] for s in range(count)]
    return air
