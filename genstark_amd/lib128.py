"""The reference's AirAssembly library for the 128-bit field, assembly/lib128.aa, as GenericAirs: `ComputePoseidonHash`
(6 registers, :81-123) and `ComputeMerkleRoot` (12 registers, the Poseidon Merkle-path STARK of BASELINE configs[3], :125-150),
driven like examples/assembly/lib128.ts:55-117 drives them.

What AirAssembly writes as static-register declarations is restated here as the columns those declarations denote:
  (input secret|public (steps S) (shift -1))   one value per input, held for S steps, the whole column rotated one step earlier
                                               (so that on the LAST step of a segment the registers already show the NEXT input)
  (input ... (childof 0) (steps S))            a list of values per parent input, each held for S steps
  (input ... (peerof 0))                       the same shape as input 0
  (mask (input i))                             1 on the step where input i changes (after the rotation: the last step of each of
                                               its segments), 0 elsewhere — the transition blends `init(next inputs)` in on it
  (cycle ...)                                  a cyclic column
Secret inputs are GenericAir secret registers (committed with the trace, lib/Stark.ts:113), public inputs and masks are
public static registers, which makes the transition functions below literal transcriptions of the .aa expressions.
The MDS matrix is DATA of lib128.aa:6-12; round constants are air-assembly's `prng sha256 <seed> 64` (same UNVERIFIED
restatement as the MiMC constants, genstark_amd/air.py: sha256_prng; the example's own control computation uses the same
generator, examples/assembly/lib128.ts:21-28, so the known answers below do not depend on it).
"""
from .air import sha256_prng
from .air_generic import GenericAir, PackedColumn, mat_vec
from .poseidon import F_ROUNDS, P_ROUNDS, STATE_WIDTH, round_controls

ROUND_STEPS = F_ROUNDS + P_ROUNDS + 1      # 64
MDS = [
    [214709430312099715322788202694750992687, 54066244720673262921467176400601950806, 122144641489288436529811410313120680228, 31306921464140082640306742797164216427, 175168617969612323849888177639760381562, 132141821748092528881872238908581032861],
    [83122512782280758906222839313578703456, 163244785834732434882219275190570945140, 65865044136286518938950810559808473518, 18180551964097663916757206212354824776, 249870759939216084597363298282234681285, 277848157012146393126919156748857149900],
    [12333142678723890553278650076570367543, 308304933036173868454178201249080175007, 76915505462549994902479959396659996669, 18709421677975378951783554559899201050, 194094680499515472018551780371064260782, 307996370140270198980510484883186251320],
    [208379730163689696681819863669840588820, 139228116619884689637390357571491341686, 20697300236245124157484102630323760041, 149868860475127892585325727303994541834, 267559900028452092630277575379158932918, 82085214952496693902284543143423475908],
    [15202238431155429285648564568592062486, 336660456679856225851744224588562255722, 111484781404051652919056230896785744525, 19879940832183425491077957046268887076, 12604714924249352815400732976355636593, 1385111712720963900529005819570184056],
    [56257924185444874124459580258315826298, 6609414732577910747612629775769094818, 222516026778809277319420550386007789953, 186298854479664158795006770633754553086, 83847903426790374369611045128936398695, 18289323526456896741189879358874983848],
]


def round_constant_columns(f):
    """lib128.aa:90-95 / lib128.ts:21-28: six cyclic columns of 64 values, seeds 'Hades1' .. 'Hades6'."""
    return [sha256_prng(bytes.fromhex('48616465733' + str(j + 1)), ROUND_STEPS, f) for j in range(STATE_WIDTH)]


def poseidon_hash(f, inputs):
    """examples/poseidon/utils.ts:19-49 with the round constants and MDS matrix of lib128 (lib128.ts:58: the control value)."""
    p, m = f.modulus, STATE_WIDTH
    cols = round_constant_columns(f)
    state = [v % p for v in inputs] + [0] * (m - len(inputs))
    for i in range(F_ROUNDS + P_ROUNDS):
        state = [(s + cols[j][i]) % p for j, s in enumerate(state)]
        if i < F_ROUNDS // 2 or i >= F_ROUNDS // 2 + P_ROUNDS:
            state = [pow(s, 5, p) for s in state]
        else:
            state[m - 1] = pow(state[m - 1], 5, p)
        state = [sum(a * b for a, b in zip(row, state)) % p for row in MDS]
    return state[:2]


def poseidon_round(state, keys, full):       # lib128.aa:15-36 ($poseidonRound)
    full_round = mat_vec(MDS, [(s + k) ** 5 for s, k in zip(state, keys)])
    part_round = mat_vec(MDS, [state[i] + keys[i] for i in range(5)] + [(state[5] + keys[5]) ** 5])
    return [a * full + b * (1 - full) for a, b in zip(full_round, part_round)]


def init_merkle_hash(p, v):                  # lib128.aa:39-45 ($initMerkleHash)
    return [p[0], p[1], v[0], v[1], 0, 0, v[0], v[1], p[0], p[1], 0, 0]


def held(values, steps, total, shift=-1):
    """The column of `(input ... (steps S) (shift -1))`: values[j] during steps [j*S, (j+1)*S), rotated by `shift`."""
    return [values[((i - shift) // steps) % len(values)] for i in range(total)]


def held_packed(values, steps, modulus):
    """held(values, steps, len(values) * steps) as a PackedColumn, built on bytes (one input per 64-step hash: 2^16 steps = 1 024
    values to pack instead of 65 536)."""
    words = [int(v % modulus).to_bytes(16, 'little') for v in values]
    flat = b''.join(w * steps for w in words)
    return PackedColumn(flat[16:] + flat[:16])               # the rotation by one step of `(shift -1)`


def segment_mask(steps):
    """`(mask (input i))` for an input held `steps` steps and rotated by -1: 1 on the last step of every segment."""
    return [0] * (steps - 1) + [1]


def compute_poseidon_hash_air(field, hashes=1, extensionFactor=32):
    """lib128.aa:81-123.  prove(assertions, air.expandInputs(raw), air.segmentSeeds(raw)) with raw = the four secret input
    registers, each a list of `hashes` values (lib128.ts:61: [[42n], [43n], [44n], [45n]]).  The digest of hash number s is in
    registers 0, 1 at step 64*s + 63.  The transition itself restarts every 64-step segment from the next inputs (mask
    register), so the segments are independent and the device generates them in parallel."""
    total = ROUND_STEPS * hashes
    rc = round_constant_columns(field)
    public = [segment_mask(ROUND_STEPS), round_controls()] + rc          # k[0] mask, k[1] round kind, k[2..7] round constants
    npub = len(public)

    def transition(r, k):                                                  # lib128.aa:99-108
        inputs, mask = k[npub:npub + 4], k[0]
        rnd = poseidon_round(r, k[2:8], k[1])
        start = list(inputs) + [0, 0]
        return [s * mask + x * (1 - mask) for s, x in zip(start, rnd)]

    def evaluation(r, n, k):                                               # :109-121
        return [a - b for a, b in zip(n, transition(r, k))]

    air = GenericAir(total, STATE_WIDTH, [7] * STATE_WIDTH, public, transition, evaluation, lambda seed: list(seed) + [0, 0],
                     extensionFactor, field, secretRegisters=4, segmentLength=ROUND_STEPS, maskSegments=False)
    air.expandInputs = lambda inputs: [held_packed(col, ROUND_STEPS, field.modulus) for col in inputs]
    air.segmentSeeds = lambda inputs: [[col[s] for col in inputs] for s in range(hashes)]
    return air


def compute_merkle_root_air(field, index_bits, extensionFactor=32):
    """lib128.aa:125-150 for ONE authentication path of depth len(index_bits) (a power of two, lib128.ts:77-117).  index_bits is
    the PUBLIC input register (already shifted as lib128.ts:92-95 does); prove(assertions, inputs, seed) takes inputs = the four
    secret columns built by merkle_inputs().  The root is in registers 0, 1 at step 64*depth - 1."""
    depth = len(index_bits)
    total = ROUND_STEPS * depth
    rc = round_constant_columns(field)
    # public statics: k[0] index bits, k[1] leaf mask, k[2] node mask, k[3] round kind, k[4..9] round constants; secret: leaf_1,
    # leaf_2, nodes_1, nodes_2.  lib128's $k vector order is restored below.
    public = [held([b % field.modulus for b in index_bits], ROUND_STEPS, total), segment_mask(total), segment_mask(ROUND_STEPS),
              round_controls()] + rc
    npub = len(public)

    def lib_order(k):
        s = k[npub:npub + 4]
        return [s[0], s[1], s[2], s[3], k[0], k[1], k[2], k[3]] + list(k[4:10])

    transition = lambda r, k: _merkle_transition(r, lib_order(k))
    evaluation = lambda r, n, k: [a - b for a, b in zip(n, transition(r, k))]
    return GenericAir(total, 12, [8] * 12, public, transition, evaluation, lambda seed: list(seed), extensionFactor, field, secretRegisters=4)


def _merkle_transition(r, k):                 # lib128.aa:48-77 ($merkleTransition), k in the library's order (14 values)
    h1 = poseidon_round(r[0:6], k[8:14], k[7])
    h2 = poseidon_round(r[6:12], k[8:14], k[7])
    h = [r[6] * k[4] + r[0] * (1 - k[4]), r[7] * k[4] + r[1] * (1 - k[4])]
    a = init_merkle_hash(k[0:2], k[1:3])      # as written in lib128.aa:66 (slices 0..1 and 1..2)
    b = init_merkle_hash(h, k[2:4])
    return [x * k[5] + (y * ((1 - k[5]) * k[6]) + z * ((1 - k[5]) * (1 - k[6]))) for x, y, z in zip(a, b, h1 + h2)]


def compute_merkle_update_air(field, depth, extensionFactor=32):
    """lib128.aa:152-203 (ComputeMerkleUpdate): the authentication paths of an old and a new leaf at the same (SECRET) index run
    side by side — 24 registers, 24 transition constraints + `bit^2 = bit`.  Secret columns from merkle_update_inputs(); the old
    root is in registers 0, 1 and the new root in registers 12, 13 at step 64*depth - 1 (lib128.ts:150-155)."""
    total = ROUND_STEPS * depth
    public = [segment_mask(total), segment_mask(ROUND_STEPS), round_controls()] + round_constant_columns(field)   # masks, round kind, constants
    npub = len(public)

    def halves(k):
        s = k[npub:npub + 7]                  # oldLeaf 1,2  newLeaf 1,2  nodes 1,2  index bit
        shared = [s[4], s[5], s[6]] + list(k[0:npub])
        return [s[0], s[1]] + shared, [s[2], s[3]] + shared

    def transition(r, k):                     # :179-187
        old, new = halves(k)
        return _merkle_transition(r[0:12], old) + _merkle_transition(r[12:24], new)

    def evaluation(r, n, k):                  # :188-203
        bit = k[npub + 6]
        return [a - b for a, b in zip(n, transition(r, k))] + [bit ** 2 - bit]

    return GenericAir(total, 24, [8] * 24 + [2], public, transition, evaluation, lambda seed: list(seed), extensionFactor, field, secretRegisters=7)


def merkle_update_inputs(field, old_leaf, new_leaf, nodes, index_bits):
    """Secret columns and first row (lib128.aa:175-178) of ComputeMerkleUpdate."""
    depth = len(nodes)
    total = ROUND_STEPS * depth
    one = lambda v: held([v], total, total)
    cols = [one(old_leaf[0]), one(old_leaf[1]), one(new_leaf[0]), one(new_leaf[1]), held([n[0] for n in nodes], ROUND_STEPS, total),
            held([n[1] for n in nodes], ROUND_STEPS, total), held(list(index_bits), ROUND_STEPS, total)]
    first = init_merkle_hash(old_leaf, nodes[0]) + init_merkle_hash(new_leaf, nodes[0])
    return [[v % field.modulus for v in c] for c in cols], [v % field.modulus for v in first]


def merkle_inputs(field, leaf, nodes):
    """The secret columns of ComputeMerkleRoot and its first row (lib128.aa:143: init = $initMerkleHash(leaf, first node)) from an
    authentication path: leaf = (l1, l2), nodes = [(n1, n2), ...] bottom-up (lib128.ts:97-99)."""
    depth = len(nodes)
    total = ROUND_STEPS * depth
    cols = [held([leaf[0]], total, total), held([leaf[1]], total, total), held([n[0] for n in nodes], ROUND_STEPS, total),
            held([n[1] for n in nodes], ROUND_STEPS, total)]
    first = [leaf[0], leaf[1], nodes[0][0], nodes[0][1], 0, 0, nodes[0][0], nodes[0][1], leaf[0], leaf[1], 0, 0]
    return [[v % field.modulus for v in c] for c in cols], [v % field.modulus for v in first]


class PoseidonMerkleTree:
    """examples/poseidon/utils.ts:128-168 (MerkleTree over [bigint, bigint] nodes) with lib128's hash: the example's control."""

    def __init__(self, field, leaves):
        self.field = field
        n = len(leaves)
        self.nodes = [None] * n + [tuple(v) for v in leaves]
        for i in range(n - 1, 0, -1):
            self.nodes[i] = tuple(poseidon_hash(field, list(self.nodes[2 * i]) + list(self.nodes[2 * i + 1])))

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
