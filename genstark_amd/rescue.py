"""The Rescue 4x128 hash AIR of examples/rescue/hash4x128.ts (4 trace registers, 4 degree-3 constraints, 8 cyclic static
registers of period 32) as a GenericAir, for BASELINE configs[2] ("Rescue hash-preimage STARK, 128-bit field").

The cipher parameters (alpha, inv_alpha, MDS matrix and its inverse, the 24 seed constants) are DATA of the reference's
example (hash4x128.ts:12-34,52-65); the key schedule and input injection are restated from examples/rescue/utils.ts:126-181
and hash4x128.ts:130-160.  With steps > 32 either the permutation keeps cycling through the 32-step round-constant
schedule, or — segmented=True — every 32-step segment hashes its own input pair like the example's `for each (value1,
value2)` loop over several inputs (GenericAir.segmentLength: segments generated in parallel on the device, transition
constraints masked on the last step of each segment; the example's secret-input registers themselves are not modelled)."""
from .air_generic import GenericAir, mat_vec

ALPHA = 3
INV_ALPHA = 113427455640312821154458202464371168597     # 3 * INV_ALPHA == -1 (mod p - 1): (1/x)^INV_ALPHA is the cube root
STEPS_PER_HASH = 32

MDS = [
    [340282366920938463463374607393113505064, 340282366920938463463374607393113476633, 340282366920938463463374607393112623703, 340282366920938463463374607393088807273],
    [1080, 42471, 1277640, 35708310],
    [340282366920938463463374607393113505403, 340282366920938463463374607393113491273, 340282366920938463463374607393113076364, 340282366920938463463374607393101570233],
    [40, 1210, 33880, 925771],
]
INV_MDS = [
    [236997924285633886309140921207528337986, 247254910923297358352547052529406562002, 311342028444809266296393502237594936029, 126030506267014245727175780515967965110],
    [33069997328254894416993606273702832836, 59740111947936946229464514160137230831, 88480676416265968399408181712033476738, 124630167308491865219096049621346098829],
    [336618017400133662891528246258390023400, 144341202744775798260123226512082052891, 154884404066691444097361840554534567820, 4667796528407935026932436315406220930],
    [73878794827854483309086441046605817365, 229228508225866824084614421584601165863, 125857624914110248133585690282064031000, 84953896817024417490170340940393220925],
]
SEED_CONSTANTS = [
    144517900019036866096022507193071809599, 271707809579969091656092579345468860225, 139424957805302989189422527487860690608, 126750251129487986697737866024960215983,
    271118613762407276564214152179206069413, 39384648060424157691646880565718875760, 189037434251220539428539337560615209464, 218986062987136192416421725751708413726,
    103808983578136303126641899945581033860, 198823153506012419365570940451368319246, 339599443104046223725845265111864465825, 169004341575174204803282453992954960786,
    171596418631454858790177474513731208863, 157569361262795131998922854453557743690, 211837534394685913032370295607135890739, 328609939009439440841980058678511564944,
    229628671790616575443886906286361261591, 95675137928612392156876334331168593412, 301613873771889848137714364785485714735, 278224571298089265666737094541710980794,
    140049647417493050970983064725330334359, 159594320057012289760186736637936788141, 44954493393746175043012738454844468290, 223519669575552375517628855932195463175,
]


def key_schedule(f):
    """examples/rescue/utils.ts:126-181 (unrollConstants + groupConstants) on host integers."""
    n = 4
    inv_exp = f.modulus - 1 - INV_ALPHA          # x^-INV_ALPHA
    vadd = lambda a, b: [f.add(x, y) for x, y in zip(a, b)]

    def mmul(m, v):
        return [sum(a * b for a, b in zip(row, v)) % f.modulus for row in m]

    c = list(SEED_CONSTANTS)
    i_const, c_matrix, c_const = c[:n], [c[n + i * n:n + (i + 1) * n] for i in range(n)], c[n + n * n:n + n * n + n]
    state, injection = list(i_const), i_const
    states = [list(state)]
    for _ in range(STEPS_PER_HASH + 1):
        state = [f.exp(x, inv_exp) for x in state]
        injection = vadd(mmul(c_matrix, injection), c_const)
        state = vadd(mmul(MDS, state), injection)
        states.append(list(state))
        state = [f.exp(x, ALPHA) for x in state]
        injection = vadd(mmul(c_matrix, injection), c_const)
        state = vadd(mmul(MDS, state), injection)
        states.append(list(state))
    initial = states[0] + states[1]
    rc = [[0] * STEPS_PER_HASH for _ in range(2 * n)]
    k = 2
    for i in range(STEPS_PER_HASH):
        for j in range(n):
            rc[j][i] = states[k][j]
            rc[n + j][i] = states[k + 1][j]
        k += 2
    return initial, rc


def build_inputs(f, values, initial):
    """hash4x128.ts:130-160 — inject the two input elements and apply the first half round."""
    inv_exp = f.modulus - 1 - INV_ALPHA
    r = [f.add(values[0], initial[0]), f.add(values[1], initial[1]), initial[2], initial[3]]
    a = [f.exp(x, inv_exp) for x in r]
    r = [sum(m * x for m, x in zip(row, a)) % f.modulus for row in MDS]
    return [f.add(r[j], initial[4 + j]) for j in range(4)]


def rescue4x128_air(steps, extensionFactor=16, field=None, segmented=False):
    """Returns the GenericAir; prove with `stark.prove(assertions, [], [v1, v2])` (the two hashed elements).  segmented=True:
    steps/32 independent hashes (the example's `for each (value1, value2)` over several input pairs, hash4x128.ts:60-81):
    seed = [[v1, v2], ...] one pair per 32-step segment, digest of pair s in registers 0 and 1 of step 32*s + 31."""
    from .field import PrimeField
    f = field or PrimeField()
    initial, rc = key_schedule(f)
    inv_exp = f.modulus - 1 - INV_ALPHA

    def transition(r, k):      # hash4x128.ts:94-97
        s = [a + b for a, b in zip(mat_vec(MDS, [x ** ALPHA for x in r]), k[0:4])]
        return [a + b for a, b in zip(mat_vec(MDS, [x ** inv_exp for x in s]), k[4:8])]

    def evaluation(r, n, k):   # hash4x128.ts:102-106
        s = [a + b for a, b in zip(mat_vec(MDS, [x ** ALPHA for x in r]), k[0:4])]
        nn = [x ** ALPHA for x in mat_vec(INV_MDS, [a - b for a, b in zip(n, k[4:8])])]
        return [a - b for a, b in zip(s, nn)]

    if not segmented:
        return GenericAir(steps, 4, [3, 3, 3, 3], rc, transition, evaluation, lambda seed: build_inputs(f, seed, initial), extensionFactor, f)

    def init_expr(x):          # build_inputs() as expressions: evaluated on the device for every segment
        a = [(x[0] + initial[0]) ** inv_exp, (x[1] + initial[1]) ** inv_exp,
             pow(initial[2], inv_exp, f.modulus), pow(initial[3], inv_exp, f.modulus)]     # registers 2, 3 start from constants
        return [m + k for m, k in zip(mat_vec(MDS, a), initial[4:8])]

    return GenericAir(steps, 4, [3, 3, 3, 3], rc, transition, evaluation, lambda seed: [seed[0], seed[1], 0, 0], extensionFactor, f,
                      segmentLength=STEPS_PER_HASH, initExpr=init_expr)


# ---- Rescue 2x64 (examples/rescue/hash2x64.ts): the same construction over the 64-bit field 2^64 - 21*2^30 + 1 -------------
MODULUS_2X64 = 2**64 - 21 * 2**30 + 1
INV_ALPHA_2X64 = 6148914683720324437                     # hash2x64.ts:14 (invAlpha = -6148914683720324437)
MDS_2X64 = [[18446744051160973310, 18446744051160973301], [4, 13]]                                       # :17-20
INV_MDS_2X64 = [[2049638227906774814, 6148914683720324439], [16397105823254198500, 12297829367440648875]]   # :57-60
SEED_CONSTANTS_2X64 = [1908230773479027697, 11775995824954138427, 18345613653544031596, 8765075832563166921,
                       10398013025088720944, 5494050611496560306, 17002767073604012844, 4907993559994152336]  # :23-28


def key_schedule_2x64(f):
    """examples/rescue/utils.ts:126-181 for state width 2 (same procedure as key_schedule above)."""
    n = 2
    inv_exp = f.modulus - 1 - INV_ALPHA_2X64
    vadd = lambda a, b: [f.add(x, y) for x, y in zip(a, b)]
    mmul = lambda m, v: [sum(a * b for a, b in zip(row, v)) % f.modulus for row in m]
    c = list(SEED_CONSTANTS_2X64)
    i_const, c_matrix, c_const = c[:n], [c[n + i * n:n + (i + 1) * n] for i in range(n)], c[n + n * n:n + n * n + n]
    state, injection = list(i_const), i_const
    states = [list(state)]
    for _ in range(STEPS_PER_HASH + 1):
        state = [f.exp(x, inv_exp) for x in state]
        injection = vadd(mmul(c_matrix, injection), c_const)
        state = vadd(mmul(MDS_2X64, state), injection)
        states.append(list(state))
        state = [f.exp(x, ALPHA) for x in state]
        injection = vadd(mmul(c_matrix, injection), c_const)
        state = vadd(mmul(MDS_2X64, state), injection)
        states.append(list(state))
    initial = states[0] + states[1]
    rc = [[0] * STEPS_PER_HASH for _ in range(2 * n)]
    k = 2
    for i in range(STEPS_PER_HASH):
        for j in range(n):
            rc[j][i] = states[k][j]
            rc[n + j][i] = states[k + 1][j]
        k += 2
    return initial, rc


def rescue2x64_air(steps, extensionFactor=16, field=None):
    """examples/rescue/hash2x64.ts:50-98 — 2 registers, 2 degree-3 constraints, 4 cyclic static registers of period 32, over the
    64-bit field (a PrimeField on the q64 build of the library).  prove(assertions, [], [value]); the digest is register 0 at
    step 31 (hash2x64.ts:101-107: 42 -> 14354339131598895532)."""
    from ._abi import Backend
    from .field import PrimeField
    f = field or PrimeField(backend=Backend(modulus=MODULUS_2X64))
    if f.modulus != MODULUS_2X64:
        raise ValueError('Rescue 2x64 is defined over 2^64 - 21*2^30 + 1')
    initial, rc = key_schedule_2x64(f)
    inv_exp = f.modulus - 1 - INV_ALPHA_2X64

    def transition(r, k):      # hash2x64.ts:76-79
        s = [a + b for a, b in zip(mat_vec(MDS_2X64, [x ** ALPHA for x in r]), k[0:2])]
        return [a + b for a, b in zip(mat_vec(MDS_2X64, [x ** inv_exp for x in s]), k[2:4])]

    def evaluation(r, n, k):   # :90-94
        s = [a + b for a, b in zip(mat_vec(MDS_2X64, [x ** ALPHA for x in r]), k[0:2])]
        nn = [x ** ALPHA for x in mat_vec(INV_MDS_2X64, [a - b for a, b in zip(n, k[2:4])])]
        return [a - b for a, b in zip(s, nn)]

    def init(seed):            # buildInputs, hash2x64.ts:121-135
        r = [f.add(seed[0], initial[0]), f.add(0, initial[1])]
        a = [f.exp(x, inv_exp) for x in r]
        return [f.add(sum(m * x for m, x in zip(MDS_2X64[j], a)) % f.modulus, initial[2 + j]) for j in range(2)]

    return GenericAir(steps, 2, [3, 3], rc, transition, evaluation, init, extensionFactor, f)
