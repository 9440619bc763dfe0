"""GPU parity tests proper (-m gpu): every entry point of include/gstark.h on the HIP backend
(libgstark_hip.so, hand-written gfx950 kernels) against (a) the independent expectations of abi_cases.py,
(b) the CPU oracle's bytes for the same call on the same seeded inputs, (c) the committed golden proofs,
and (d) size-independent properties at the full BASELINE sizes.  Integer work: the bar is bit-exact."""
import ctypes as C
import json
import os
import random

import pytest

import abi_cases as cases
from conftest import P, from_bytes, rand_elements, to_bytes
from genstark_amd.field import PrimeField
from genstark_amd.merkle import MerkleTree, createHash
from oracle import pyref

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'oracle_proofs.json')) as f:
    GOLDEN = json.load(f)


def test_backend_is_the_hip_library(hip_backend):
    assert hip_backend.name == 'hip-gfx950'
    assert hip_backend.stream


# ---- (a) independent expectations -----------------------------------------------------------------------
@pytest.mark.parametrize('n', [1, 2, 7, 64, 1000, 70001])
def test_pointwise(hip_backend, rng, n):
    cases.check_pointwise(hip_backend, rng, n)


@pytest.mark.parametrize('logn,logsteps', [(4, 2), (8, 4), (12, 7), (17, 13)])
def test_domain_divisions(hip_backend, rng, logn, logsteps):
    cases.check_domain_divisions(hip_backend, rng, logn, logsteps)


@pytest.mark.parametrize('logn,logsteps,nroots', [(5, 2, 1), (8, 4, 2), (12, 8, 4), (16, 12, 2)])
def test_mimc_composition(hip_backend, oracle_backend, rng, logn, logsteps, nroots):
    import random
    seed = rng.randrange(1 << 30)
    assert cases.check_mimc_composition(hip_backend, random.Random(seed), logn, logsteps, nroots) == \
        cases.check_mimc_composition(oracle_backend, random.Random(seed), logn, logsteps, nroots)


@pytest.mark.parametrize('logn,depth', [(4, 0), (8, 1), (14, 0), (14, 3), (20, 1)])
def test_fri_fold(hip_backend, oracle_backend, rng, logn, depth):
    import random
    seed = rng.randrange(1 << 30)
    got = cases.check_fri_fold(hip_backend, random.Random(seed), logn, depth)
    if logn <= 14:
        assert got == cases.check_fri_fold(oracle_backend, random.Random(seed), logn, depth)


@pytest.mark.parametrize('logm,depth,nlayers,alg', [
    (5, 0, 1, 'blake2s256'),       # the smallest layer: 2 leaves... (32 values -> 8, a tree of 2 leaves)
    (9, 0, 1, 'blake2s256'),       # 32 leaves, one workgroup, one layer
    (13, 1, 3, 'blake2s256'),      # one workgroup from the first layer on: three layers in ONE launch
    (14, 0, 4, 'sha256'),          # the same through the SHA-256 instantiation
    (17, 1, 5, 'blake2s256'),      # 8192 leaves = 32 workgroups + last-arrival top, 2048 = 8, then the single-workgroup run
    (19, 0, 6, 'blake2s256'),      # 2^15 leaves: 128 workgroups, the widest one-launch layer
    (20, 0, 2, 'blake2s256'),      # 2^16 leaves: the streaming fold + fused tree launches, then one-launch layers
    (18, 0, 5, 'sha256'),
])
def test_fri_layers(hip_backend, oracle_backend, rng, logm, depth, nlayers, alg):
    """gs_fri_layers: a run of FRI layers in as few launches as the sizes allow == the member sequence on the same device (every next
    column, leaf, node, posted root and point), and every output byte == the oracle's."""
    import random
    seed = rng.randrange(1 << 30)
    got = cases.check_fri_layers(hip_backend, random.Random(seed), logm, depth, nlayers, alg)
    assert got == cases.check_fri_layers(oracle_backend, random.Random(seed), logm, depth, nlayers, alg)


@pytest.mark.parametrize('logn,logsteps,ranks', [(8, 4, 1), (10, 6, 8), (16, 12, 4), (20, 16, 2), (20, 15, 8)])
def test_coset_divisions(hip_backend, oracle_backend, rng, logn, logsteps, ranks):
    """gs_zero_poly_inverses_coset / gs_div_by_domain_roots_coset: a rank's coset of the domain == the strided share of the whole-domain
    entries on the same device, == the oracle's values."""
    import random
    seed = rng.randrange(1 << 30)
    got = cases.check_coset_divisions(hip_backend, random.Random(seed), logn, logsteps, ranks)
    if logn <= 16:
        assert got == cases.check_coset_divisions(oracle_backend, random.Random(seed), logn, logsteps, ranks)


def test_fri_layers_repeated_launches_keep_the_arrival_counter_clean(hip_backend, rng):
    """Back-to-back multi-workgroup launches on one context: the workgroup that arrives last resets the counter for the next."""
    import random
    for k in range(6):
        cases.check_fri_layers(hip_backend, random.Random(k), 16 + k % 3, 0, 3)


@pytest.mark.parametrize('n', [64, 1 << 16])
def test_deferred_readbacks(hip_backend, rng, n):
    cases.check_deferred_readbacks(hip_backend, rng, n)


@pytest.mark.parametrize('n', [5, 257, 40000])
def test_inverse_with_zeros(hip_backend, rng, n):
    cases.check_inverse_with_zeros(hip_backend, rng, n)


@pytest.mark.parametrize('n', [1, 8, 64, 1024, 1 << 17])
def test_power_series_and_shuffles(hip_backend, rng, n):
    cases.check_power_series_and_shuffles(hip_backend, rng, n)


@pytest.mark.parametrize('n,k', [(100, 1), (1000, 5), (5000, 24)])
def test_combine_many(hip_backend, rng, n, k):
    cases.check_combine_many(hip_backend, rng, n, max(k, 2))


@pytest.mark.parametrize('logn', list(range(0, 14)))
def test_ntt_every_size_full(hip_backend, rng, logn):
    cases.check_ntt(hip_backend, rng, logn, full=True)


@pytest.mark.parametrize('logn,plen,rows', [(6, 3, 2), (8, 16, 3), (10, 64, 3), (12, 256, 2), (13, 8192, 1), (16, 4096, 2),
                                            (17, 1 << 13, 1), (20, 3, 2), (12, 9, 1), (9, 512, 5)])
def test_ntt_zero_extended_rows(hip_backend, rng, logn, plen, rows):
    cases.check_ntt(hip_backend, rng, logn, poly_len=plen, rows=rows)


def test_small_polys(hip_backend, rng):
    cases.check_small_polys(hip_backend, rng)


@pytest.mark.parametrize('logn,depth', [(8, 0), (10, 1), (12, 2), (16, 0), (16, 3)])
def test_quartic(hip_backend, rng, logn, depth):
    cases.check_quartic(hip_backend, rng, logn, depth)


@pytest.mark.parametrize('alg', ['sha256', 'blake2s256'])
@pytest.mark.parametrize('n', [64, 4096])
def test_hashing(hip_backend, rng, alg, n):
    cases.check_hashing(hip_backend, rng, alg, n)


@pytest.mark.parametrize('alg,logn', [('sha256', 1), ('blake2s256', 1), ('blake2s256', 2), ('blake2s256', 4), ('sha256', 7),
                                      ('blake2s256', 8), ('blake2s256', 9), ('blake2s256', 10), ('sha256', 11), ('blake2s256', 13), ('sha256', 15)])
def test_merkle(hip_backend, rng, alg, logn):
    cases.check_merkle(hip_backend, rng, alg, logn)


# every branch of the construction plan (csrc/hash.hip: merkle_run): one subtree workgroup (n <= 256), subtrees of 256 digests + the tree
# over their roots built by the last workgroup to arrive (n <= 2^15), streaming layers with 1..4 fused layers above the leaves, a second streaming launch (n >= 2^20)
@pytest.mark.parametrize('alg,logn,count', [('blake2s256', 1, 1), ('sha256', 1, 4), ('blake2s256', 3, 2), ('blake2s256', 10, 4), ('sha256', 10, 1),
                                            ('blake2s256', 11, 1), ('blake2s256', 13, 6), ('sha256', 12, 3)])
def test_merkle_commit_rows(hip_backend, rng, alg, logn, count):
    cases.check_merkle_commit(hip_backend, rng, alg, logn, count)


@pytest.mark.parametrize('alg,logn,count', [('blake2s256', 15, 1), ('blake2s256', 16, 4), ('sha256', 16, 1), ('blake2s256', 17, 2), ('blake2s256', 18, 1),
                                            ('blake2s256', 19, 4), ('sha256', 19, 2), ('blake2s256', 20, 1), ('blake2s256', 21, 4), ('blake2s256', 22, 1)])
def test_merkle_commit_rows_equals_oracle(hip_backend, oracle_omp_backend, alg, logn, count):
    """the fused entry on HIP against the oracle's mergeVectorRows + MerkleTree.create, every leaf digest and every node; and
    gs_merkle_build alone (digests given) on the same leaves"""
    n = 1 << logn
    r = random.Random(logn * 131 + count)
    cols = [r.randbytes(16 * n) for _ in range(count)]
    got = cases.merkle_commit_bytes(hip_backend, alg, cols, n)
    want = cases.merkle_commit_bytes(oracle_omp_backend, alg, cols, n)
    assert got[0] == want[0]
    assert got[1][32:] == want[1][32:]
    assert cases.merkle_commit_bytes(hip_backend, alg, cols, n, fused=False)[1][32:] == want[1][32:]


def test_device_record_ops(hip_backend, rng):
    for _ in range(3):
        cases.check_device_record_ops(hip_backend, rng)
        cases.check_combine_adjusted(hip_backend, rng)


def test_mimc_air(hip_backend, rng):
    cases.check_mimc_air(hip_backend, rng, 128)


@pytest.mark.parametrize('logn,logsteps,per_row,lcount,adjusted', [(8, 4, [1], 0, False), (12, 8, [2, 1], 3, True), (14, 10, [4, 1, 3], 7, True),
                                                                   (16, 12, [1, 1], 6, True), (13, 9, [1, 1], 70, False), (16, 11, [3] * 20, 2, True)])
def test_composition_tail(hip_backend, oracle_backend, logn, logsteps, per_row, lcount, adjusted):
    """gs_composition_tail on HIP: its definition, the member sequence it replaces, and the oracle's bytes."""
    seed = hash((logn, tuple(per_row), lcount)) & 0xffff
    got = cases.check_composition_tail(hip_backend, random.Random(seed), logn, logsteps, per_row, lcount, adjusted)
    if logn <= 14:
        assert got == cases.check_composition_tail(oracle_backend, random.Random(seed), logn, logsteps, per_row, lcount, adjusted)
    cases.check_composition_tail(hip_backend, random.Random(seed + 1), logn, logsteps, per_row, lcount, adjusted, with_c=False)
    got = cases.check_composition_tail(hip_backend, random.Random(seed + 2), logn, logsteps, per_row, lcount, adjusted, made=True)
    if logn <= 14:
        assert got == cases.check_composition_tail(oracle_backend, random.Random(seed + 2), logn, logsteps, per_row, lcount, adjusted, made=True)
    cases.check_composition_tail_limits(hip_backend)


@pytest.mark.parametrize('logn,logsteps,per_row,lcount,ranks', [(8, 4, [1], 2, 2), (12, 7, [2, 1], 3, 4), (16, 11, [4, 1, 3], 7, 8), (15, 11, [1, 1], 6, 2)])
def test_composition_tail_over_a_rank_s_coset(hip_backend, oracle_backend, logn, logsteps, per_row, lcount, ranks):
    seed = logn * 100 + ranks
    got = cases.check_composition_tail_coset(hip_backend, random.Random(seed), logn, logsteps, per_row, lcount, ranks)
    if logn <= 12:
        assert got == cases.check_composition_tail_coset(oracle_backend, random.Random(seed), logn, logsteps, per_row, lcount, ranks)


@pytest.mark.parametrize('jit', [0, 1])
def test_constraints_read_in_place_from_the_evaluation_domain(hip_backend, oracle_backend, jit):
    """gs_air_constraints_strided, interpreted and compiled, against the oracle's (Poseidon and Rescue segments)."""
    from genstark_amd.poseidon import poseidon6x128_air
    from genstark_amd.rescue import rescue4x128_air
    from genstark_amd._abi import Backend
    be = Backend(device=0).jit() if jit else hip_backend        # (a context of its own: the session's counts no compiled launches)
    try:
        for make, inputs in ((lambda f: poseidon6x128_air(256, 16, f, segmented=True), [[1, 2, 3, 4], [5, 6, 7, 8], [9, 9, 9, 9], [0, 1, 0, 1]]),
                             (lambda f: rescue4x128_air(128, 16, f, segmented=True), [[42, 43], [1, 2], [3, 4], [5, 6]])):
            got = cases.check_constraints_strided(be, make(PrimeField(backend=be)), inputs)
            assert got == cases.check_constraints_strided(oracle_backend, make(PrimeField(backend=oracle_backend)), inputs)
        assert not jit or be.jit_launches >= 2
    finally:
        if jit:
            be.close()


def test_kat_rescue_4x128_through_hip_kernels(hip_backend):
    cases.check_rescue_kat(hip_backend)


# ---- (b) HIP bytes == oracle bytes on the same seeded inputs ----------------------------------------------
def _both(hip_backend, oracle_backend):
    return PrimeField(backend=hip_backend), PrimeField(backend=oracle_backend)


@pytest.mark.parametrize('logn,plen,rows', [(14, None, 1), (16, None, 2), (18, None, 1), (20, None, 1), (20, 1 << 16, 1), (17, 1 << 13, 3)])
def test_ntt_bytes_equal_oracle(hip_backend, oracle_backend, logn, plen, rows):
    fh, fo = _both(hip_backend, oracle_backend)
    rng = random.Random(logn * 131 + rows)
    n = 1 << logn
    plen = plen or n
    w = fh.getRootOfUnity(n)
    raw = to_bytes([rng.randrange(P) for _ in range(plen * rows)])
    outs = []
    for f in (fh, fo):
        m = f.newMatrix(rows, plen)
        f.backend.upload(m.ptr, raw)
        roots = f.getPowerSeries(w, n)
        ev = f.evalPolysAtRoots(m, roots)
        back = f.interpolateRoots(roots, ev)
        outs.append((ev.toBuffer(), back.toBuffer()))
    assert outs[0][0] == outs[1][0]
    assert outs[0][1] == outs[1][1]


@pytest.mark.parametrize('n', [1 << 12, 1 << 18, (1 << 18) + 13])
def test_streaming_ops_bytes_equal_oracle(hip_backend, oracle_backend, n):
    fh, fo = _both(hip_backend, oracle_backend)
    rng = random.Random(n)
    a = to_bytes([rng.randrange(P) if rng.random() > 0.01 else 0 for _ in range(n)])
    b = to_bytes([rng.randrange(P) for _ in range(n)])
    res = []
    for f in (fh, fo):
        va, vb = f.newVector(n), f.newVector(n)
        f.backend.upload(va.ptr, a)
        f.backend.upload(vb.ptr, b)
        h = createHash('blake2s256', f.backend)
        out = [f.mulVectorElements(va, vb).toBuffer(), f.divVectorElements(vb, va).toBuffer(), f.invVectorElements(va).toBuffer(),
               f.combineManyVectors([va, vb, va], [3, 5, P - 1]).toBuffer(), f.getPowerSeries(12345, n).toBuffer(),
               h.mergeVectorRows([va, vb]).toBuffer(), f.combineVectors(va, vb)]
        if n % 4 == 0:
            m = f.transposeVector(va, 4)
            out.append(h.digestValues(m, 64).toBuffer())
            tree = MerkleTree.create(h.digestValues(m, 64), h)
            out.append(tree.nodes.toBuffer())
        res.append(out)
    for x, y in zip(*res):
        assert x == y


# ---- (c) end-to-end: golden proofs, byte for byte ---------------------------------------------------------
@pytest.mark.parametrize('case', GOLDEN, ids=[c['name'] for c in GOLDEN])
def test_prove_matches_golden_bytes(case, hip_backend):
    from test_host_mirror import run_golden_case
    run_golden_case(case, hip_backend)


@pytest.mark.parametrize('steps,ef', [(1 << 13, 16), (1 << 13, 8)])
def test_prove_2p13_equals_oracle_and_verifies(hip_backend, oracle_backend, steps, ef):
    """BASELINE configs[1] (MiMC-128, 2^13 steps, E=16 as in the README log and E=8 as BASELINE words it):
    proof bytes identical between the HIP path and the CPU oracle path; verify(parse(serialize)) holds."""
    import genstark_amd as ga
    opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': ef, 'exeQueryCount': 48, 'friQueryCount': 24}
    datas = []
    for be in (hip_backend, oracle_backend):
        stark = ga.instantiateMimc(steps, opts, backend=be)
        controls = ga.runMimc(stark.air.field, steps, stark.air.roundConstants, 3)
        assertions = [{'step': 0, 'register': 0, 'value': controls[0]}, {'step': steps - 1, 'register': 0, 'value': controls[-1]}]
        proof = stark.prove(assertions, [], [3])
        data = stark.serialize(proof)
        assert len(data) == stark.sizeOf(proof)
        datas.append(data)
        if be is hip_backend:
            assert stark.verify(assertions, stark.parse(data))
            assert stark.securityLevel == (96 if ef == 16 else 67)   # floor(min(48*log2(E/3), 24*log2 E, 128)), lib/Stark.ts:62-77
    assert datas[0] == datas[1]


# ---- (d) full-size properties (sizes the oracle cannot finish in seconds) ----------------------------------
@pytest.mark.parametrize('logn', [22, 24])
def test_ntt_full_size_properties(hip_backend, logn):
    """n = 2^22 / 2^24 (BASELINE configs[4] sizes): interpolate(eval(p)) == p bit-exactly, linearity
    eval(a) + eval(b) == eval(a + b), and spot values against direct evaluation of a sparse polynomial."""
    f = PrimeField(backend=hip_backend)
    n = 1 << logn
    w = f.getRootOfUnity(n)
    roots = f.getPowerSeries(w, n)
    a = f.getPowerSeries(0x1234567890abcdef1234567, n)          # dense pseudo-random-looking coefficients
    b = f.getPowerSeries(0xfedcba9876543210fedcba987654321, n)
    ea, eb = f.evalPolyAtRoots(a, roots), f.evalPolyAtRoots(b, roots)
    assert f.interpolateRoots(roots, ea).toBuffer() == a.toBuffer()
    lhs = f.addVectorElements(ea, eb)
    rhs = f.evalPolyAtRoots(f.addVectorElements(a, b), roots)
    assert lhs.toBuffer() == rhs.toBuffer()
    # zero-extended transform (LDE shape: n/16 coefficients) against direct evaluation of sampled points
    plen = n // 16
    coeffs = f.getPowerSeries(987654321987654321, plen)
    ev = f.evalPolyAtRoots(coeffs, roots)
    rng = random.Random(logn)
    # p(x) = sum c^i x^i = ((c x)^plen - 1) / (c x - 1)
    c = 987654321987654321
    for q in [0, 1, n - 1] + [rng.randrange(n) for _ in range(5)]:
        x = pow(w, q, P)
        cx = c * x % P
        want = (pow(cx, plen, P) - 1) * pow(cx - 1, P - 2, P) % P
        assert ev.getValue(q) == want


@pytest.mark.parametrize('logn', [25, 26])
def test_ntt_four_pass_sizes_properties(hip_backend, logn):
    """Beyond the BASELINE sizes: n = 2^25 / 2^26 take FOUR radix passes (1 GiB per vector at 2^26).  Round trip, linearity
    (vectors compared on the device through their Merkle roots) and closed-form spot values of a zero-extended transform."""
    f = PrimeField(backend=hip_backend)
    h = createHash('blake2s256', hip_backend)
    fingerprint = lambda v: MerkleTree.create(h.mergeVectorRows([v]), h).root
    n = 1 << logn
    w = f.getRootOfUnity(n)
    roots = f.getPowerSeries(w, n)
    a = f.getPowerSeries(0x1234567890abcdef1234567, n)
    ea = f.evalPolyAtRoots(a, roots)
    assert fingerprint(f.interpolateRoots(roots, ea)) == fingerprint(a)
    b = f.getPowerSeries(0xfedcba9876543210fedcba987654321, n)
    eb = f.evalPolyAtRoots(b, roots)
    assert fingerprint(f.addVectorElements(ea, eb)) == fingerprint(f.evalPolyAtRoots(f.addVectorElements(a, b), roots))
    del ea, eb, b
    plen, c = n // 16, 987654321987654321
    ev = f.evalPolyAtRoots(f.getPowerSeries(c, plen), roots)
    rng = random.Random(logn)
    for q in [0, 1, n - 1] + [rng.randrange(n) for _ in range(5)]:
        cx = c * pow(w, q, P) % P
        assert ev.getValue(q) == (pow(cx, plen, P) - 1) * pow(cx - 1, P - 2, P) % P


def test_prove_2p20_full_config_verifies(hip_backend):
    """BASELINE configs[4]: MiMC-128, 2^20 steps, E=16, friQueryCount 64 — the proof verifies, sizeOf matches,
    8 FRI layers with a 256-value remainder."""
    import genstark_amd as ga
    steps = 1 << 20
    opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 64}
    stark = ga.instantiateMimc(steps, opts, backend=hip_backend)
    trace = stark.generateExecutionTrace([], [3])['dTrace']
    first, last = trace.getValue(0, 0), trace.getValue(0, steps - 1)
    assertions = [{'step': 0, 'register': 0, 'value': first}, {'step': steps - 1, 'register': 0, 'value': last}]
    proof = stark.prove(assertions, [], [3])
    assert len(proof['ldProof']['components']) == 8 and len(proof['ldProof']['remainder']) == 256
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    assert stark.verify(assertions, stark.parse(data))


# ---- (e) randomized differential runs: the same call on both backends, every shape drawn at random -----------------------
@pytest.mark.parametrize('seed', range(12))
def test_randomized_differential_hip_vs_oracle(hip_backend, oracle_backend, seed):
    """40 randomly shaped calls per seed (lengths that are not powers of two, ragged NTT inputs, random register counts,
    random query sets, zeros sprinkled into inversions): byte equality of every result on the HIP library and on the oracle."""
    rng = random.Random(1000 + seed)
    fh, fo = PrimeField(backend=hip_backend), PrimeField(backend=oracle_backend)
    hh, ho = createHash('blake2s256' if seed % 2 else 'sha256', hip_backend), createHash('blake2s256' if seed % 2 else 'sha256', oracle_backend)

    def both(fn):
        a, b = fn(fh, hh), fn(fo, ho)
        for x, y in zip(a, b):
            assert x == y
    for _ in range(40):
        kind = rng.randrange(8)
        if kind == 0:       # pointwise family on an arbitrary length
            n = rng.choice([1, 2, 3, 63, 64, 65, 255, 1000, 4097, rng.randrange(1, 150000)])
            xs, ys, k, e = rand_elements(rng, n), rand_elements(rng, n, edge=0.3), rng.randrange(P), rng.choice([0, 1, 5, P - 2])
            both(lambda f, h: [f.addVectorElements(f.newVectorFrom(xs), f.newVectorFrom(ys)).toBuffer(),
                               f.subVectorElements(f.newVectorFrom(xs), k).toBuffer(),
                               f.mulVectorElements(f.newVectorFrom(xs), f.newVectorFrom(ys)).toBuffer(),
                               f.divVectorElements(f.newVectorFrom(xs), f.newVectorFrom(ys)).toBuffer(),     # zeros in ys: 0^-1 := 0
                               f.invVectorElements(f.newVectorFrom(ys)).toBuffer(),
                               f.expVectorElements(f.newVectorFrom(xs[:64]), e).toBuffer()])
        elif kind == 1:     # linear combinations / dot products
            n, k = rng.randrange(1, 30000), rng.randrange(1, 20)
            vs, cs = [rand_elements(rng, n) for _ in range(k)], rand_elements(rng, k)
            both(lambda f, h: [f.combineManyVectors([f.newVectorFrom(v) for v in vs], cs).toBuffer(),
                               f.combineVectors(f.newVectorFrom(vs[0]), f.newVectorFrom(vs[-1])).to_bytes(16, 'little')])
        elif kind == 2:     # NTT: ragged polynomial lengths, several rows, forward and inverse
            logn = rng.randrange(0, 17)
            n, rows = 1 << logn, rng.randrange(1, 4)
            plen = rng.choice([1, n, max(1, n // 16), rng.randrange(1, n + 1)])
            polys = [rand_elements(rng, plen) for _ in range(rows)]
            both(lambda f, h: [f.evalPolysAtRoots(f.newMatrixFrom(polys), f.getPowerSeries(f.getRootOfUnity(n), n)).toBuffer(),
                               f.interpolateRoots(f.getPowerSeries(f.getRootOfUnity(n), n),
                                                  f.evalPolysAtRoots(f.newMatrixFrom(polys), f.getPowerSeries(f.getRootOfUnity(n), n))).toBuffer()])
        elif kind == 3:     # power series, pluck, transposes
            n = 1 << rng.randrange(2, 15)
            base, skip = rng.randrange(2, P), rng.randrange(1, 50)
            step = rng.choice([1, 1, 4]) if n >= 64 else 1
            times = n * 2
            both(lambda f, h: [f.getPowerSeries(base, n).toBuffer(), f.pluckVector(f.getPowerSeries(base, n), skip, times).toBuffer(),
                               f.transposeVector(f.getPowerSeries(base, n), 4, step).toBuffer()])
        elif kind == 4:     # leaf / row hashing with a random number of registers
            n, regs = rng.randrange(1, 5000), rng.randrange(1, 9)
            cols = [rand_elements(rng, n) for _ in range(regs)]
            both(lambda f, h: [h.mergeVectorRows([f.newVectorFrom(c) for c in cols]).toBuffer()])
        elif kind == 5:     # Merkle tree + batch proofs over random query sets
            logn = rng.randrange(1, 13)
            n = 1 << logn
            leaves = rand_elements(rng, n)
            idx = rng.sample(range(n), rng.randrange(1, min(n, 70) + 1))

            def run(f, h):
                tree = MerkleTree.create(h.mergeVectorRows([f.newVectorFrom(leaves)]), h)
                proof = tree.proveBatch(idx)
                assert MerkleTree.verifyBatch(tree.root, idx, proof, h)
                return [tree.root, b''.join(proof['values']), b''.join(b''.join(c) for c in proof['nodes']), bytes([len(c) for c in proof['nodes']])]
            both(run)
        elif kind == 6:     # FRI rows: domain fast path, generic path, evaluation
            n = 1 << rng.randrange(4, 14)
            vals, x = rand_elements(rng, n), rng.randrange(P)

            def run(f, h):
                dom = f.getPowerSeries(f.getRootOfUnity(n), n)
                ys = f.transposeVector(f.newVectorFrom(vals), 4)
                polys = f.interpolateQuarticBatch(f.transposeVector(dom, 4), ys)
                return [polys.toBuffer(), f.evalQuarticBatch(polys, x).toBuffer(), h.digestValues(ys, 64).toBuffer()]
            both(run)
        else:               # small host-side polynomials and gathers
            m = rng.randrange(1, 40)
            xs, ys = rng.sample(range(1, 10 ** 9), m), rand_elements(rng, m)
            n = rng.randrange(m, 3000)
            vec, idx = rand_elements(rng, n), [rng.randrange(n) for _ in range(rng.randrange(1, 200))]
            both(lambda f, h: [f.interpolate(f.newVectorFrom(xs), f.newVectorFrom(ys)).toBuffer(), b''.join(f.newVectorFrom(vec).valuesAt(idx)),
                               f.mulPolys(f.newVectorFrom(vec[:90]), f.newVectorFrom(vec[-80:] if n >= 80 else vec)).toBuffer()])


@pytest.mark.parametrize('n,count', [(33, 65), (1000, 150)])
def test_more_vectors_than_one_launch_carries(hip_backend, rng, n, count):
    cases.check_many_vectors(hip_backend, rng, n, count)


# ---- (e) byte parity with the C oracle AT the headline sizes -----------------------------
# The oracle's loops are annotated for OpenMP (oracle/liboracle_omp.so: the same oracle_abi.c built with -fopenmp); on the GPU box's
# host cores its 2^24-point transforms take seconds, so the comparison below is on EVERY output byte, not on properties.
@pytest.fixture(scope='module')
def oracle_omp_backend():
    import subprocess
    from genstark_amd._abi import Backend
    lib = os.path.join(os.path.dirname(HERE), 'oracle', 'liboracle_omp.so')
    if not os.path.exists(lib):
        subprocess.check_call(['make', '-C', os.path.dirname(lib), '-s', 'liboracle_omp.so'])
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    os.environ.setdefault('OMP_NUM_THREADS', str(max(1, min(32, cpus))))      # read by libgomp when the library is loaded
    os.environ.setdefault('OMP_WAIT_POLICY', 'passive')
    return Backend(lib_path=lib, allow_test_double=True)


def _dense_input(f, n, base):
    """pseudo-random-looking dense coefficients without a host-side 2^24 loop: a power series of a generic element"""
    return f.getPowerSeries(base, n)


@pytest.mark.parametrize('logn', [21, 22, 24])
def test_ntt_bytes_equal_oracle_at_headline_sizes(hip_backend, oracle_omp_backend, logn):
    """forward (dense, un-pruned), zero-extended by 16 (the pruned LDE shape), zero-extended ragged, and inverse transforms of
    2^21 / 2^22 / 2^24 points: every output byte of the HIP passes (three-pass plans: table twiddles AND the running-product
    branch that only plans above 2^20 entries take) equals the oracle's radix-2 NTT."""
    n = 1 << logn
    fh, fo = PrimeField(backend=hip_backend), PrimeField(backend=oracle_omp_backend)
    w = fh.getRootOfUnity(n)
    assert w == fo.getRootOfUnity(n)
    wb = w.to_bytes(16, 'little')
    raw = _dense_input(fh, n, 0x1234567890abcdef1234567).toBuffer()
    vh, vo = fh.newVector(n), fo.newVector(n)
    hip_backend.upload(vh.ptr, raw); oracle_omp_backend.upload(vo.ptr, raw)
    for kind, plen in (('fwd', n), ('fwd', n // 16), ('fwd', n // 16 + 12345), ('inv', n)):
        outs = []
        for be, f, v in ((hip_backend, fh, vh), (oracle_omp_backend, fo, vo)):
            out = f.newVector(n)
            if kind == 'fwd':
                be.call('gs_eval_polys_at_roots', C.c_void_p(v.ptr), 1, plen, wb, n, C.c_void_p(out.ptr))
            else:
                be.call('gs_interpolate_roots', C.c_void_p(v.ptr), 1, wb, n, C.c_void_p(out.ptr))
            outs.append(out.toBuffer())
        assert outs[0] == outs[1], (logn, kind, plen)


def test_mimc_composition_bytes_equal_oracle_at_2p24(hip_backend, oracle_omp_backend):
    """gs_mimc_composition over the whole evaluation domain of the headline configuration (T = 2^20, E = 16), LinearCombination
    folded in: every byte against the oracle's term-by-term evaluation."""
    n, steps = 1 << 24, 1 << 20
    rng = random.Random(2024)
    fh, fo = PrimeField(backend=hip_backend), PrimeField(backend=oracle_omp_backend)
    w = fh.getRootOfUnity(n)
    raw = _dense_input(fh, n, 0xfedcba9876543210fedcba987654321).toBuffer()
    k = to_bytes(rand_elements(rng, 64))
    coeffs = to_bytes(rng.randrange(P) for _ in range(4))
    ipoly = to_bytes(rng.randrange(P) for _ in range(2))
    lc = to_bytes(rng.randrange(P) for _ in range(2))
    roots = (C.c_uint64 * 2)(0, (steps - 1) * (n // steps))
    outs = []
    for be, f in ((hip_backend, fh), (oracle_omp_backend, fo)):
        p, kv, out = f.newVector(n), f.newVector(64), f.newVector(n)
        be.upload(p.ptr, raw); be.upload(kv.ptr, k)
        be.call('gs_mimc_composition', C.c_void_p(p.ptr), n, steps, w.to_bytes(16, 'little'), C.c_void_p(kv.ptr), 64, coeffs, 2 * steps, 2 * steps,
                ipoly, roots, 2, lc, C.c_void_p(out.ptr))
        outs.append(out.toBuffer())
    assert outs[0] == outs[1]


@pytest.mark.parametrize('logn,depth', [(22, 0), (24, 0), (24, 1)])
def test_fri_fold_bytes_equal_oracle_at_headline_sizes(hip_backend, oracle_omp_backend, logn, depth):
    """gs_fri_fold on the first layers of the headline configuration: every folded value against the oracle's Lagrange
    interpolation + Horner evaluation per row (lib/components/LowDegreeProver.ts:189-198)."""
    n = 1 << logn
    step = 4 ** depth
    m = n // step
    fh, fo = PrimeField(backend=hip_backend), PrimeField(backend=oracle_omp_backend)
    w = fh.getRootOfUnity(n)
    raw = _dense_input(fh, m, 0x31415926535897932384626433).toBuffer()
    x = (0x2718281828459045235360287471352 % P).to_bytes(16, 'little')
    outs = []
    for be, f in ((hip_backend, fh), (oracle_omp_backend, fo)):
        col, out = f.newVector(m), f.newVector(m // 4)
        be.upload(col.ptr, raw)
        be.call('gs_fri_fold', w.to_bytes(16, 'little'), n, step, C.c_void_p(col.ptr), m, x, C.c_void_p(out.ptr))
        outs.append(out.toBuffer())
    assert outs[0] == outs[1]


def test_merkle_root_equals_oracle_at_2p24(hip_backend, oracle_omp_backend):
    """leaf hashing + the whole tree over 2^24 rows (lib/Stark.ts:115-118): every node of the HIP tree equals the oracle's."""
    n = 1 << 24
    fh, fo = PrimeField(backend=hip_backend), PrimeField(backend=oracle_omp_backend)
    raw = _dense_input(fh, n, 0x5eed5eed5eed5eed5eed5eed5eed).toBuffer()
    nodes = []
    for be, f in ((hip_backend, fh), (oracle_omp_backend, fo)):
        v = f.newVector(n)
        be.upload(v.ptr, raw)
        leaves, tree = be.alloc(32 * n), be.alloc(32 * n)
        ptrs = (C.c_void_p * 1)(v.ptr)
        be.call('gs_hash_merge_rows', 1, ptrs, 1, n, C.c_void_p(leaves))       # blake2s256 = 1
        be.call('gs_merkle_build', 1, C.c_void_p(leaves), n, C.c_void_p(tree))
        nodes.append(be.download(tree, 32 * n))
        be.free(leaves); be.free(tree)
    assert nodes[0][32:] == nodes[1][32:]       # node 0 is unused


def test_lazy_field_arithmetic_on_the_device():
    """tools/lazy_device_check: every routine of gf128_lazy.h (the arithmetic inside the NTT pass kernels) executed on the GPU on
    2^20 lanes, each result compared on the device with the canonical fe_mul / fe_add / fe_sub of gf128.h."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), 'tools', 'lazy_device_check')
    if not os.path.exists(exe):
        pytest.fail('tools/lazy_device_check is missing: run __graft_entry__.build()')
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and ' 0 mismatching' in r.stdout, r.stdout + r.stderr


def test_composition_tail_bytes_equal_oracle_at_2p24(hip_backend, oracle_omp_backend):
    """gs_composition_tail over the evaluation domain of the 2^20-step Poseidon statement (N = 2^24, six committed vectors, two asserted
    registers, degree adjustment; 1/Z and the powers computed in place): every byte of L against the oracle's member sequence."""
    n, steps = 1 << 24, 1 << 20
    rng = random.Random(424)
    fh, fo = PrimeField(backend=hip_backend), PrimeField(backend=oracle_omp_backend)
    w = fh.getRootOfUnity(n)
    cols = [_dense_input(fh, n, 0x1234567890abcdef1234567 + 977 * k).toBuffer() for k in range(7)]
    x_last = pow(w, (steps - 1) * 16, P)
    roots = (C.c_uint64 * 2)(0, (steps - 64) * 16)
    per_row = (C.c_uint32 * 2)(1, 1)
    ip, bk, bkp = to_bytes(rng.randrange(P) for _ in range(2)), to_bytes(rng.randrange(P) for _ in range(2)), to_bytes(rng.randrange(P) for _ in range(2))
    lk, lkp = to_bytes(rng.randrange(P) for _ in range(6)), to_bytes(rng.randrange(P) for _ in range(6))
    outs = []
    for be, f in ((hip_backend, fh), (oracle_omp_backend, fo)):
        vecs = [f.newVector(n) for _ in range(7)]
        for v, raw in zip(vecs, cols):
            be.upload(v.ptr, raw)
        bp = (C.c_void_p * 2)(vecs[1].ptr, vecs[3].ptr)
        lp = (C.c_void_p * 6)(*[v.ptr for v in vecs[1:]])
        out = f.newVector(n)
        be.call('gs_composition_tail', n, w.to_bytes(16, 'little'), C.c_void_p(vecs[0].ptr), None, steps, x_last.to_bytes(16, 'little'), bp, 2, ip, 1, roots, per_row, 1,
                bk, bkp, lp, 6, lk, lkp, None, 6 * steps, None, C.c_void_p(out.ptr))
        outs.append(out.toBuffer())
        del vecs
    assert outs[0] == outs[1]


def test_constraints_in_place_bytes_equal_oracle_at_2p22(hip_backend, oracle_omp_backend):
    """gs_air_constraints_strided, compiled, with the Poseidon 6x128 constraint program over 2^22 composition-domain points read in
    place from six columns of 2^23 elements: every byte of the six constraint columns against the oracle's interpreter."""
    from genstark_amd._abi import Backend
    from genstark_amd.poseidon import poseidon6x128_air
    from genstark_amd.field import Matrix
    nc, n, steps = 1 << 22, 1 << 23, 1 << 19
    compiled = Backend(device=0).jit()
    outs = []
    try:
        raw = None
        for be in (compiled, oracle_omp_backend):
            f = PrimeField(backend=be)
            air = poseidon6x128_air(steps, 16, f, segmented=True)
            ctx = air.initProvingContext([], [[1 + s, 2, 3 + s, 4] for s in range(steps // 64)])
            if raw is None:
                raw = [_dense_input(f, n, 0xabcdef1234567 + 131 * r).toBuffer() for r in range(6)]
            p = Matrix(be, 6, n)
            for r in range(6):
                be.upload(p.ptr + r * n * 16, raw[r])
            code, ninstr, consts, nconsts, nregs = air.evaluationProgram.abi_args(16)
            q = Matrix(be, len(air.constraintDegrees), nc)
            lens = (C.c_uint64 * len(ctx._staticLens))(*ctx._staticLens)
            be.call('gs_air_constraints_strided', code, ninstr, consts, nconsts, nregs, 6, len(air.constraintDegrees), C.c_void_p(p.ptr), n, 2, nc,
                    nc // steps, C.c_void_p(ctx._staticTables.ptr), lens, len(ctx._staticLens), C.c_void_p(q.ptr))
            outs.append(be.download(q.ptr, len(air.constraintDegrees) * nc * 16))
        assert compiled.jit_launches >= 1
    finally:
        compiled.close()
    assert outs[0] == outs[1]
