"""The Poseidon 6x128 hash AIR of examples/poseidon/hash6x128.ts (6 trace registers, 8 full + 55 partial rounds, S-box x^5,
6 cyclic round-constant registers of period 64) as a GenericAir, for BASELINE configs[3] ("Poseidon ... 6 state registers").

Every parameter is derived exactly as the example derives it, from sha256 over ASCII strings (examples/poseidon/utils.ts):
round constants sha256("Hades<c>") (:52-63), the Cauchy MDS matrix 1/(x_i - y_j) from sha256("HadesMDSx<i>") and
sha256("HadesMDSy<j>") (:65-81,119-126), the full/partial schedule of hash6x128.ts:21-29.  The AirScript `for steps [1..4,
60..63]` / `[5..59]` segments (:70-79) become one more cyclic static register (1 = full round) that blends the two
transition bodies, which makes every constraint degree 6 in x (S-box degree 5 times the degree-<T control polynomial).
With steps > 64 either the permutation keeps cycling through the 64-step schedule (step 63 -> 64 is one more partial round),
or — segmented=True — every 64-step segment hashes its own inputs like the example's `for each (value1, value2)` loop over
several inputs (GenericAir.segmentLength: the device generates the segments in parallel, the transition constraints are
masked on the last step of each segment; the example's secret-input registers themselves are not modelled)."""
import hashlib

from .air_generic import GenericAir, mat_vec

STATE_WIDTH, F_ROUNDS, P_ROUNDS, ALPHA = 6, 8, 55, 5
STEPS_PER_HASH = F_ROUNDS + P_ROUNDS + 1


def _constants(f, seed, count):                      # utils.ts:119-126
    return [int.from_bytes(hashlib.sha256(f'{seed}{i}'.encode()).digest(), 'big') % f.modulus for i in range(count)]


def round_constants(f, width=STATE_WIDTH, rounds=STEPS_PER_HASH):     # utils.ts:52-63, one list of `width` values per round
    flat = _constants(f, 'Hades', width * rounds)
    return [flat[i * width:(i + 1) * width] for i in range(rounds)]


def mds_matrix(f, width=STATE_WIDTH):                # utils.ts:65-81
    xs, ys = _constants(f, 'HadesMDSx', width), _constants(f, 'HadesMDSy', width)
    if len(set(xs + ys)) != 2 * width:
        raise ValueError('MDS values are not all different')
    return [[f.inv(f.sub(x, y)) for y in ys] for x in xs]


def round_controls():                                # hash6x128.ts:21-29: 1 on full rounds, 0 on partial ones and the last step
    full = lambda i: i < F_ROUNDS // 2 or i >= F_ROUNDS // 2 + P_ROUNDS
    return [1 if full(i) else 0 for i in range(F_ROUNDS + P_ROUNDS)] + [0]


def poseidon_hash(f, inputs):
    """utils.ts:19-49 on host integers — the example's own control computation (hash6x128.ts:19)."""
    p, m = f.modulus, STATE_WIDTH
    mds, ark = mds_matrix(f), round_constants(f, m, F_ROUNDS + P_ROUNDS)
    assert 0 < len(inputs) < m
    state = [v % p for v in inputs] + [0] * (m - len(inputs))
    for i in range(F_ROUNDS + P_ROUNDS):
        state = [(s + k) % p for s, k in zip(state, ark[i])]
        if i < F_ROUNDS // 2 or i >= F_ROUNDS // 2 + P_ROUNDS:
            state = [pow(s, ALPHA, p) for s in state]
        else:
            state[m - 1] = pow(state[m - 1], ALPHA, p)
        state = [sum(a * b for a, b in zip(row, state)) % p for row in mds]
    return state[:2]


def poseidon6x128_air(steps, extensionFactor=16, field=None, segmented=False):
    """Returns the GenericAir; prove with `stark.prove(assertions, [], [v1, v2, v3, v4])` (the hashed elements).  For
    steps == 64 the digest is registers 0 and 1 of step 63 (hash6x128.ts:93-96)."""
    from .field import PrimeField
    f = field or PrimeField()
    m = STATE_WIDTH
    mds = mds_matrix(f)
    rc = round_constants(f)
    statics = [[rc[i][j] for i in range(STEPS_PER_HASH)] for j in range(m)] + [round_controls()]

    def next_state(r, k):                            # hash6x128.ts:70-79
        v = [a + b for a, b in zip(r, k[0:m])]
        full = k[m]
        s = [x + full * (x ** ALPHA - x) for x in v[:m - 1]] + [v[m - 1] ** ALPHA]
        return mat_vec(mds, s)

    def evaluation(r, n, k):                         # hash6x128.ts:83-87: transition($r) = $n
        return [a - b for a, b in zip(n, next_state(r, k))]

    def init(seed):                                  # hash6x128.ts:65-67
        if len(seed) != 4:
            raise ValueError('Poseidon6x128 hashes two pairs of elements: seed must hold 4 values')
        return list(seed) + [0, 0]

    # segmented=True: steps/64 independent hashes (`for each (value1, value2)` over several inputs, hash6x128.ts:62-81),
    # seed = [[a, b, c, d], ...], digest of input s in registers 0 and 1 of step 64*s + 63
    return GenericAir(steps, m, [ALPHA + 1] * m, statics, next_state, evaluation, init, extensionFactor, f,
                      segmentLength=STEPS_PER_HASH if segmented else None)
