"""The native prove() driver (genstark_amd/csrc/prover.cc: the call sequence of lib/Stark.ts:81-163 issued by C++ above the C
ABI) must produce exactly the bytes the Python mirror's prove() + serialize() produce: golden MiMC proofs, the generic AIR path
(Rescue, Poseidon, many-hash segmented variants, Foo with zero FRI layers), error behaviour.  CPU: bound to the oracle's
implementation of the ABI; GPU: bound to the HIP library."""
import hashlib
import json
import os

import pytest

import genstark_amd as ga
from genstark_amd.errors import StarkError
from genstark_amd.field import PrimeField
from genstark_amd.native import NativeProver
from genstark_amd._mirror.stark import Stark

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'oracle_proofs.json')) as f:
    GOLDEN = json.load(f)


def check_golden(case, backend):
    options = {'hashAlgorithm': case['hash_algorithm'], 'extensionFactor': case['extension_factor'],
               'exeQueryCount': case['exe_query_count'], 'friQueryCount': case['fri_query_count']}
    stark = ga.instantiateMimc(case['steps'], options, None, backend=backend)
    assertions = [{'step': a['step'], 'register': a['register'], 'value': int(a['value'])} for a in case['assertions']]
    native = NativeProver(stark)
    data = native.prove_bytes(assertions, [], [case['seed']])
    assert len(data) == case['proofSize'] and hashlib.sha256(data).hexdigest() == case['proofSha256']
    assert data == stark.serialize(stark.prove(assertions, [], [case['seed']]))
    proof = native.prove(assertions, [], [case['seed']])
    assert proof['evRoot'].hex() == case['evRoot'] and stark.verify(assertions, proof)
    return native, assertions


@pytest.mark.parametrize('case', GOLDEN, ids=[c['name'] for c in GOLDEN])
def test_native_prover_matches_golden_bytes_oracle(case, oracle_backend):
    check_golden(case, oracle_backend)


def test_native_prover_errors(oracle_backend):
    native, assertions = check_golden(GOLDEN[0], oracle_backend)
    with pytest.raises(StarkError, match='conflicts with execution trace'):
        native.prove_bytes([dict(assertions[0], value=assertions[0]['value'] + 1)], [], [GOLDEN[0]['seed']])
    with pytest.raises(StarkError, match='outside of execution trace'):
        native.prove_bytes([dict(assertions[0], step=GOLDEN[0]['steps'])], [], [GOLDEN[0]['seed']])
    with pytest.raises(StarkError, match='outside of register bank'):
        native.prove_bytes([dict(assertions[0], register=1)], [], [GOLDEN[0]['seed']])
    with pytest.raises(TypeError):
        native.prove_bytes([], [], [3])


def generic_cases(backend):
    from genstark_amd import poseidon
    from genstark_amd.rescue import rescue4x128_air
    from test_generic_air import POSEIDON_OPTS, RESCUE_OPTS, degree5_air, foo_air
    f = PrimeField(backend=backend)
    out = []
    air = rescue4x128_air(64, 16, f)
    out.append(('rescue', Stark(air, RESCUE_OPTS), [42, 43], air.hostTrace([42, 43]), [(31, 0), (31, 1), (63, 3), (0, 2), (5, 1), (9, 1)]))
    air = poseidon.poseidon6x128_air(128, 16, f)
    out.append(('poseidon', Stark(air, POSEIDON_OPTS), [1, 2, 3, 4], air.hostTrace([1, 2, 3, 4]), [(63, 0), (63, 1), (127, 5)]))
    air = rescue4x128_air(128, 16, f, segmented=True)
    seeds = [[42 + s, 43 + 2 * s] for s in range(4)]
    out.append(('rescue-segmented', Stark(air, RESCUE_OPTS), seeds, air.hostTrace(seeds), [(31, 0), (127, 1), (96, 0)]))
    air = poseidon.poseidon6x128_air(128, 16, f, segmented=True)
    seeds = [[1, 2, 3, 4], [5, 6, 7, 8]]
    out.append(('poseidon-segmented', Stark(air, POSEIDON_OPTS), seeds, air.hostTrace(seeds), [(63, 0), (127, 1)]))
    air = degree5_air(f, 64)
    out.append(('degree5', Stark(air, {'hashAlgorithm': 'sha256', 'extensionFactor': 16, 'exeQueryCount': 40, 'friQueryCount': 20}), [5],
                air.hostTrace([5]), [(0, r) for r in range(6)] + [(63, 4), (32, 4)]))
    air = foo_air(f)
    out.append(('foo', Stark(air, None), [1], air.hostTrace([1]), [(0, 0), (63, 0)]))
    # assembly/lib128.aa: ComputePoseidonHash (two hashes) and ComputeMerkleRoot (depth 4) with their input registers
    from genstark_amd import lib128
    from test_lib128 import OPTS as LIB_OPTS, merkle_case
    air = lib128.compute_poseidon_hash_air(f, 2)
    raw = [[42, 52], [43, 44], [44, 44], [45, 46]]
    cols = air.expandInputs(raw)
    out.append(('lib128-hash', Stark(air, LIB_OPTS), air.segmentSeeds(raw), air.hostTrace(air.segmentSeeds(raw), inputs=cols), [(63, 0), (127, 1)], cols))
    tree, leaf, nodes, bits = merkle_case(f, 4, 5)
    air = lib128.compute_merkle_root_air(f, bits)
    cols, first = lib128.merkle_inputs(f, leaf, nodes)
    out.append(('lib128-merkle', Stark(air, LIB_OPTS), first, air.hostTrace(first, inputs=cols), [(255, 0), (255, 1)], cols))
    from test_generic_air import secret_air
    air = secret_air(f, 64)
    secrets = [[11, 22, 33, 44], [(7 * i * i + 1) % 100003 for i in range(64)]]
    out.append(('secret-registers', Stark(air, {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 30, 'friQueryCount': 20}),
                [3, 4], air.hostTrace([3, 4], inputs=secrets), [(0, 0), (63, 0), (63, 1)], secrets))
    return out


def check_generic(backend):
    digests = {}
    for name, stark, seed, trace, points, *rest in generic_cases(backend):
        inputs = rest[0] if rest else []
        assertions = [{'step': s, 'register': r, 'value': trace[s][r]} for s, r in points]
        want = stark.serialize(stark.prove(assertions, inputs, seed))
        got = NativeProver(stark).prove_bytes(assertions, inputs, seed)
        assert got == want, name
        assert stark.verify(assertions, stark.parse(got)), name
        digests[name] = hashlib.sha256(got).hexdigest()
    return digests


def test_native_prover_generic_airs_oracle(oracle_backend):
    check_generic(oracle_backend)


def check_without_the_fused_tail(backend, monkeypatch):
    """The driver's composition tail in its two forms: gs_composition_tail (default, when every asserted register has at most four
    assertions) and the member sequence it replaces — asked for with gs_prover_member_sequence, and taken on its own when a register has five
    assertions.  Same bytes as the mirror either way."""
    fused = check_generic(backend)
    some = NativeProver(generic_cases(backend)[0][1])
    some.member_sequence(True)                    # (a mode of the calling thread, like sync_phases)
    try:
        assert check_generic(backend) == fused
    finally:
        some.member_sequence(False)
    from genstark_amd import poseidon
    from test_generic_air import POSEIDON_OPTS
    f = PrimeField(backend=backend)
    air = poseidon.poseidon6x128_air(128, 16, f)
    stark = Stark(air, POSEIDON_OPTS)
    trace = air.hostTrace([1, 2, 3, 4])
    assertions = [{'step': s, 'register': r, 'value': trace[s][r]} for s, r in [(0, 0), (7, 0), (63, 0), (64, 0), (127, 0), (127, 5)]]
    want = stark.serialize(stark.prove(assertions, [], [1, 2, 3, 4]))
    assert NativeProver(stark).prove_bytes(assertions, [], [1, 2, 3, 4]) == want
    return fused


def test_native_prover_with_and_without_the_fused_tail_oracle(oracle_backend, monkeypatch):
    check_without_the_fused_tail(oracle_backend, monkeypatch)


@pytest.mark.gpu
def test_native_prover_with_and_without_the_fused_tail_hip(hip_backend, monkeypatch):
    check_without_the_fused_tail(hip_backend, monkeypatch)


@pytest.mark.gpu
def test_native_prover_hip(hip_backend, oracle_backend):
    for case in GOLDEN:
        check_golden(case, hip_backend)
    assert check_generic(hip_backend) == check_generic(oracle_backend)


@pytest.mark.gpu
def test_native_prover_2p20_config(hip_backend):
    """BASELINE configs[4] through the native driver: same bytes as the Python mirror on the same device."""
    opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 64}
    steps = 1 << 20
    stark = ga.instantiateMimc(steps, opts, None, backend=hip_backend)
    trace = stark.air.initProvingContext([], [3]).generateExecutionTrace()
    assertions = [{'step': 0, 'register': 0, 'value': trace.getValue(0, 0)}, {'step': steps - 1, 'register': 0, 'value': trace.getValue(0, steps - 1)}]
    data = NativeProver(stark).prove_bytes(assertions, [], [3])
    assert data == stark.serialize(stark.prove(assertions, [], [3]))
    assert stark.verify(assertions, stark.parse(data))


@pytest.mark.gpu
def test_headline_proof_bytes_hip_equals_oracle():
    """BASELINE configs[4] (C5: MiMC-128, 2^20 steps, E = 16, friQueryCount 64, blake2s256) END TO END: the serialized proof the
    product entry (genstark_amd.prover.Prover on the HIP library) produces against the proof of the same statement on the CPU
    oracle's implementation of the C ABI (OpenMP build, its own process) — every byte, via sha256 + length (lib/Stark.ts:81-163)."""
    import bench
    from conftest import ROOT
    from genstark_amd._abi import Backend
    from genstark_amd.prover import Prover
    import subprocess
    lib = os.path.join(ROOT, 'oracle', 'liboracle_omp.so')
    if not os.path.exists(lib):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', 'liboracle_omp.so'])
    threads = min(16, len(os.sched_getaffinity(0)))
    _, _, sha, nbytes = bench.cpu_prove(ga, lib, 20, 16, 48, 64, threads, timeout=600)        # asserts verify() in the child
    opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 64}
    be = Backend(device=0)
    stark = ga.instantiateMimc(1 << 20, opts, None, backend=be)
    data = Prover(stark.air, opts).prove_bytes(bench.assertions_for(stark, 1 << 20, 3), [], [3])
    assert len(data) == nbytes and hashlib.sha256(data).hexdigest() == sha
    assert len(data) == stark.sizeOf(stark.parse(data))


def check_native_verifier(backend):
    """csrc/verifier.h — Stark.verify (lib/Stark.ts:167-248) + LowDegreeProver.verify (LowDegreeProver.ts:70-172) natively: accepts what
    the mirror's verifier accepts (golden MiMC proofs, sha256 and blake2s256; every generic AIR case incl. secret registers), and on a
    sweep of single-bit corruptions over the whole proof gives the mirror's verdict AND the mirror's message every time."""
    import random
    for case in GOLDEN:
        options = {'hashAlgorithm': case['hash_algorithm'], 'extensionFactor': case['extension_factor'],
                   'exeQueryCount': case['exe_query_count'], 'friQueryCount': case['fri_query_count']}
        stark = ga.instantiateMimc(case['steps'], options, None, backend=backend)
        assertions = [{'step': a['step'], 'register': a['register'], 'value': int(a['value'])} for a in case['assertions']]
        nat = NativeProver(stark)
        data = nat.prove_bytes(assertions, [], [case['seed']])
        assert nat.verify_bytes(assertions, data) is True
        with pytest.raises(StarkError, match='linear combination correctness'):
            nat.verify_bytes([dict(assertions[0], value=assertions[0]['value'] + 1)] + assertions[1:], data)
    # the verdicts and messages of the two verifiers on corrupted proofs (one statement: the sweep is ~150 verifications)
    case = GOLDEN[0]
    options = {'hashAlgorithm': case['hash_algorithm'], 'extensionFactor': case['extension_factor'], 'exeQueryCount': case['exe_query_count'],
               'friQueryCount': case['fri_query_count']}
    stark = ga.instantiateMimc(case['steps'], options, None, backend=backend)
    assertions = [{'step': a['step'], 'register': a['register'], 'value': int(a['value'])} for a in case['assertions']]
    nat = NativeProver(stark)
    data = nat.prove_bytes(assertions, [], [case['seed']])
    rng = random.Random(2026)
    offsets = sorted(set([0, 31, 32, 33, len(data) - 1, len(data) - 2, len(data) - 20] + [rng.randrange(len(data)) for _ in range(140)]))
    agree = 0
    for off in offsets:
        bad = bytearray(data)
        bad[off] ^= 1 << rng.randrange(8)
        verdicts = []
        for verify in (lambda b: stark.verify(assertions, stark.parse(b)), lambda b: nat.verify_bytes(assertions, b)):
            try:
                verdicts.append(('ok', '') if verify(bytes(bad)) else ('false', ''))
            except StarkError as e:
                verdicts.append(('rejected', str(e).split(':')[0]))
        assert verdicts[0][0] == verdicts[1][0], (off, verdicts)
        agree += verdicts[0][1] == verdicts[1][1] or verdicts[0][1].startswith('Verification of low degree failed') or 'malformed' in verdicts[0][1] + verdicts[1][1]
    assert agree == len(offsets)
    # truncated and empty proofs are StarkErrors, not crashes
    for cut in (0, 1, 31, 40, len(data) // 2, len(data) - 1):
        with pytest.raises(StarkError):
            nat.verify_bytes(assertions, data[:cut])
    for name, stark, seed, trace, points, *rest in generic_cases(backend):
        inputs = rest[0] if rest else []
        assertions = [{'step': s, 'register': r, 'value': trace[s][r]} for s, r in points]
        nat = NativeProver(stark)
        data = nat.prove_bytes(assertions, inputs, seed)
        assert nat.verify_bytes(assertions, data) is True, name
        bad = bytearray(data)
        bad[len(bad) // 3] ^= 4
        with pytest.raises(StarkError):
            nat.verify_bytes(assertions, bytes(bad))


def test_native_verifier_oracle(oracle_backend):
    check_native_verifier(oracle_backend)


@pytest.mark.gpu
def test_native_verifier_hip(hip_backend):
    check_native_verifier(hip_backend)


def test_native_verifier_refuses_structures_the_prover_never_emits(oracle_backend):
    """the number of FRI layers and the remainder length are functions of the domain (LowDegreeProver.ts:179) — a proof
    with appended layers (which would floor the remainder's degree bound to zero) is malformed; a proof carrying input shapes for an AIR
    that declares no input registers is malformed (never checked against job.steps alone); a MiMC job whose round-constant count does not divide the trace is invalid."""
    import copy
    from genstark_amd._abi import GstarkError
    case = GOLDEN[0]
    options = {'hashAlgorithm': case['hash_algorithm'], 'extensionFactor': case['extension_factor'], 'exeQueryCount': case['exe_query_count'],
               'friQueryCount': case['fri_query_count']}
    stark = ga.instantiateMimc(case['steps'], options, None, backend=oracle_backend)
    assertions = [{'step': a['step'], 'register': a['register'], 'value': int(a['value'])} for a in case['assertions']]
    nat = NativeProver(stark)
    data = nat.prove_bytes(assertions, [], [case['seed']])
    assert nat.verify_bytes(assertions, data) is True
    proof = stark.parse(data)
    assert stark.serialize(proof) == data
    comps = proof['ldProof']['components']
    more = copy.deepcopy(proof)
    more['ldProof']['components'] = comps + [copy.deepcopy(comps[-1])]
    with pytest.raises(StarkError, match='FRI components'):
        nat.verify_bytes(assertions, stark.serialize(more))
    if comps:
        fewer = copy.deepcopy(proof)
        fewer['ldProof']['components'] = comps[:-1]
        with pytest.raises(StarkError, match='FRI components'):
            nat.verify_bytes(assertions, stark.serialize(fewer))
    short = copy.deepcopy(proof)
    short['ldProof']['remainder'] = proof['ldProof']['remainder'][:len(proof['ldProof']['remainder']) // 4]
    with pytest.raises(StarkError, match='remainder of'):
        nat.verify_bytes(assertions, stark.serialize(short))
    shaped = copy.deepcopy(proof)
    shaped['iShapes'] = [[4]]
    with pytest.raises(StarkError, match='1 input shapes for an AIR without input registers'):      # (sized from its shapes only if the AIR
        nat.verify_bytes(assertions, stark.serialize(shaped))                                         #  declares input registers)
    # a MiMC statement whose round constants cannot be a cyclic register of this trace
    keep = stark.air.roundConstants
    try:
        stark.air.roundConstants = keep[:3]           # (the Python front end refuses to BUILD such an AIR; the job struct can still say it)
        with pytest.raises(StarkError, match='round constants'):
            nat.verify_bytes(assertions, data)
    finally:
        stark.air.roundConstants = keep


def test_product_prover_entry(oracle_backend):
    """genstark_amd.prover.Prover: an AIR + options straight into the native driver (no mirror object), the reference's option rules
    (lib/Stark.ts:318-344), bytes of the mirror, and the CPU verifier behind verify()."""
    import sys
    from genstark_amd.air import MimcAir, runMimc
    from genstark_amd.field import PrimeField
    from genstark_amd.prover import Prover
    f = PrimeField(backend=oracle_backend)
    opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 24}
    air = MimcAir(256, 16, f)
    controls = runMimc(f, 256, air.roundConstants, 3)
    a = [{'step': 0, 'register': 0, 'value': 3}, {'step': 255, 'register': 0, 'value': controls[-1]}]
    p = Prover(air, opts)
    data = p.prove_bytes(a, [], [3])
    from genstark_amd._mirror.stark import Stark
    mirror = Stark(air, opts)
    assert data == mirror.serialize(mirror.prove(a, [], [3]))
    assert p.verify(a, data) and p.prove(a, [], [3])['evRoot'] == data[:32]
    assert Prover(air).options == {'extensionFactor': 16, 'exeQueryCount': 80, 'friQueryCount': 40, 'hashAlgorithm': 'sha256'}
    for bad, msg in (({'exeQueryCount': 129}, 'Execution sample size'), ({'friQueryCount': 65}, 'FRI sample size'), ({'hashAlgorithm': 'md5'}, 'not supported')):
        with pytest.raises(TypeError, match=msg):
            Prover(air, bad)
    # importing the product entry does not load the mirror
    import subprocess
    code = 'import sys, genstark_amd, genstark_amd.prover; assert not any(m.startswith("genstark_amd._mirror") for m in sys.modules), sorted(sys.modules)'
    assert subprocess.run([sys.executable, '-c', code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).returncode == 0


def test_packed_seed_gives_the_same_proof(oracle_backend):
    """Prover.pack_seed: the first rows of a segmented statement packed once; the proof bytes do not change."""
    import generic_cases as gc
    from genstark_amd.prover import Prover
    air, stark, seed, assertions = gc.build('rescue_seg_128', oracle_backend)
    p = Prover(air, gc.RESCUE_OPTS)
    packed = p.pack_seed(seed)
    assert packed.rows == 4 and len(packed.first) == 4 * 4 * 16
    assert p.prove_bytes(assertions, [], packed) == p.prove_bytes(assertions, [], seed) == stark.serialize(stark.prove(assertions, [], seed))


def test_remainder_check_forms_agree(oracle_backend):
    """verifyRemainder (LowDegreeProver.ts:223-252): the coefficient form the driver uses gives the reference procedure's verdict —
    valid remainders of every degree, one value off, junk at the excluded positions, degree one too high — and the mirror's."""
    import ctypes as C
    import random
    from genstark_amd.field import PrimeField
    from genstark_amd.native import _driver
    from genstark_amd._mirror.components.low_degree_prover import LowDegreeProver
    from genstark_amd.errors import StarkError
    f = PrimeField(backend=oracle_backend)
    lib, binding = _driver(oracle_backend)
    rng = random.Random(20260927)
    p = f.modulus

    def verdicts(values, ef, m, rou):
        raw = b''.join(v.to_bytes(16, 'little') for v in values)
        r = rou.to_bytes(16, 'little')
        return [lib.gs_prover_remainder_check_on(binding, raw, len(values), ef, m, r, method) for method in (0, 1)]

    def mirror(values, ef, m, rou):
        ldp = LowDegreeProver.__new__(LowDegreeProver)
        ldp.field, ldp.idxGenerator = f, type('G', (), {'extensionFactor': ef})()
        try:
            ldp.verifyRemainder(list(values), m, rou)
            return 1
        except StarkError:
            return 0

    cases = 0
    for length, ef in ((256, 16), (256, 8), (128, 16), (64, 32), (256, 32), (64, 4), (16, 16), (32, 16)):
        rou = f.getRootOfUnity(length)
        checked = length - length // ef
        for m in sorted({1, 2, 3, checked // 4, checked // 2, checked - 17, checked - 1, checked} - {0} | set()):
            if m <= 0:
                continue
            poly = [rng.randrange(p) for _ in range(m)]
            xs = [pow(rou, i, p) for i in range(length)]
            good = [sum(c * pow(x, k, p) for k, c in enumerate(poly)) % p for x in xs]
            junk = list(good)
            for i in range(0, length, ef):
                junk[i] = rng.randrange(p)                                   # the excluded positions carry anything
            assert verdicts(good, ef, m, rou) == [1, 1] and verdicts(junk, ef, m, rou) == [1, 1], (length, ef, m)
            if m < checked:
                bad = list(junk)
                at = rng.choice([i for i in range(length) if i % ef])
                bad[at] = (bad[at] + 1 + rng.randrange(p - 1)) % p
                assert verdicts(bad, ef, m, rou) == [0, 0], (length, ef, m, at)
                over = [(v + pow(x, m, p)) % p for v, x in zip(good, xs)]      # degree exactly m: one too high
                assert verdicts(over, ef, m, rou) == [0, 0], (length, ef, m)
                if cases % 3 == 0:
                    assert mirror(good, ef, m, rou) == 1 and mirror(bad, ef, m, rou) == 0 and mirror(over, ef, m, rou) == 0
            cases += 1
        rnd = [rng.randrange(p) for _ in range(length)]
        assert verdicts(rnd, ef, checked // 2, rou) == [0, 0]
        assert verdicts(rnd, ef, checked + 1, rou) == [-1, -1]                  # "Remainder degree is greater than number of remainder values"
    assert cases > 40


@pytest.mark.parametrize('degrees', [[3, 5, 3, 8], [4, 4, 3, 3], [3, 3, 3, 3], [3, 8, 8, 5], [2, 8, 8, 5]])
def test_native_prover_mixed_constraint_degrees(oracle_backend, degrees):
    """Constraint groups with degrees of their own (CompositionPolynomial.ts:83-107, :206-225): one group at the combination degree
    (not adjusted), several below it (each with its own powers) — the merged form of the driver (gs_combine_adjusted) against the
    mirror's member-by-member sequence, byte for byte.  The Rescue AIR with DECLARED degrees above the actual ones."""
    from genstark_amd.field import PrimeField
    from genstark_amd.native import NativeProver
    from genstark_amd.rescue import rescue4x128_air
    from genstark_amd._mirror.stark import Stark
    f = PrimeField(backend=oracle_backend)
    air = rescue4x128_air(64, 16, f)
    air.constraintDegrees = list(degrees)
    air.maxConstraintDegree = max(degrees)
    air.compositionFactor = 1 << (air.maxConstraintDegree - 1).bit_length()
    stark = Stark(air, {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 20, 'friQueryCount': 10})
    full = air.hostTrace([42, 43])
    assertions = [{'step': 31, 'register': 0, 'value': full[31][0]}, {'step': 63, 'register': 1, 'value': full[63][1]},
                  {'step': 5, 'register': 3, 'value': full[5][3]}]
    if degrees[0] < 3:        # a declared degree below the real one: the composition is not low-degree, BOTH drivers must say so
        from genstark_amd.errors import StarkError
        for prove in (lambda: stark.prove(assertions, [], [42, 43]), lambda: NativeProver(stark).prove_bytes(assertions, [], [42, 43])):
            with pytest.raises(StarkError, match='Remainder is not a valid degree 111 polynomial'):
                prove()
        return
    want = stark.serialize(stark.prove(assertions, [], [42, 43]))
    assert NativeProver(stark).prove_bytes(assertions, [], [42, 43]) == want
    assert stark.verify(assertions, stark.parse(want))
