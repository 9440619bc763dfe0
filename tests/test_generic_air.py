"""Generic AIR path: transition / constraint expressions compiled to the register machine of include/gstark.h,
exercised with the Rescue 4x128 hash AIR of examples/rescue/hash4x128.ts (4 registers, 4 degree-3 constraints, 8 cyclic
static registers) and with MiMC re-expressed generically (must reproduce the dedicated MiMC kernels' proof bytes)."""
import hashlib

import os

import pytest

import genstark_amd as ga
from genstark_amd.air_generic import GenericAir, Program, const, reg
from genstark_amd.errors import StarkError
from genstark_amd.field import PrimeField
from genstark_amd.rescue import rescue4x128_air
from genstark_amd._mirror.stark import Stark

RESCUE_KAT = (302524937772545017647250309501879538110, 205025454306577433144586673939030012640)   # hash4x128.ts:115-118
RESCUE_OPTS = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24}   # hash4x128.ts:41-47


def rescue_case(backend, steps):
    f = PrimeField(backend=backend)
    air = rescue4x128_air(steps, 16, f)
    stark = Stark(air, RESCUE_OPTS)
    full = air.hostTrace([42, 43])
    assertions = [{'step': 31, 'register': 0, 'value': full[31][0]}, {'step': 31, 'register': 1, 'value': full[31][1]},
                  {'step': steps - 1, 'register': 3, 'value': full[-1][3]}, {'step': 0, 'register': 2, 'value': full[0][2]}]
    return air, stark, full, assertions


def check_rescue(backend, steps):
    air, stark, full, assertions = rescue_case(backend, steps)
    assert (full[31][0], full[31][1]) == RESCUE_KAT
    trace = air.initProvingContext([], [42, 43]).generateExecutionTrace().toValues()
    assert trace == [list(r) for r in zip(*full)]
    assert (trace[0][31], trace[1][31]) == RESCUE_KAT          # the reference example's known answer, through the VM
    proof = stark.prove(assertions, [], [42, 43])
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    assert stark.verify(assertions, stark.parse(data))
    assert stark.securityLevel == 96
    bad = bytearray(data)
    bad[len(data) // 3] ^= 4
    with pytest.raises((StarkError, IndexError, ValueError, AssertionError)):
        stark.verify(assertions, stark.parse(bytes(bad)))
    wrong = [dict(a) for a in assertions]
    wrong[0]['value'] = (wrong[0]['value'] + 1) % ga.MODULUS
    with pytest.raises(StarkError):
        stark.verify(wrong, stark.parse(data))
    return data


def mimc_generic_air(field, steps, ef):
    rc = ga.sha256_prng(bytes.fromhex('4d694d43'), 64, field)
    return GenericAir(steps, 1, [3], [rc], lambda r, k: [r[0] ** 3 + k[0]], lambda r, n, k: [n[0] - (r[0] ** 3 + k[0])],
                      lambda seed: [seed[0]], ef, field)


def check_mimc_generic_equals_dedicated(backend, steps):
    f = PrimeField(backend=backend)
    opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 24}
    dedicated = ga.instantiateMimc(steps, opts, backend=backend)
    controls = ga.runMimc(f, steps, dedicated.air.roundConstants, 3)
    assertions = [{'step': 0, 'register': 0, 'value': controls[0]}, {'step': steps - 1, 'register': 0, 'value': controls[-1]}]
    want = dedicated.serialize(dedicated.prove(assertions, [], [3]))
    generic = Stark(mimc_generic_air(f, steps, 16), opts)
    got = generic.serialize(generic.prove(assertions, [], [3]))
    assert got == want
    assert generic.verify(assertions, generic.parse(got))


def test_program_compiler_reuses_registers_and_matches_python():
    x, y = reg(0), reg(1)
    e = (x + y) * (x - y) + const(7) * x ** 5
    prog = Program([e, x * x], ga.MODULUS)
    assert prog.nregs <= 6
    p = ga.MODULUS
    assert prog.run([3, 10], None, []) == [((3 + 10) * (3 - 10) + 7 * 3 ** 5) % p, 9]


@pytest.mark.parametrize('steps', [64, 256])
def test_rescue_prove_verify_oracle(oracle_backend, steps):
    check_rescue(oracle_backend, steps)


def test_mimc_generic_equals_dedicated_oracle(oracle_backend):
    check_mimc_generic_equals_dedicated(oracle_backend, 256)


@pytest.mark.gpu
@pytest.mark.parametrize('steps', [64, 1024, 1 << 13])
def test_rescue_hip_equals_oracle(hip_backend, oracle_backend, steps):
    got = check_rescue(hip_backend, steps)
    air, stark, full, assertions = rescue_case(oracle_backend, steps)
    want = stark.serialize(stark.prove(assertions, [], [42, 43]))
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()


@pytest.mark.gpu
def test_mimc_generic_equals_dedicated_hip(hip_backend):
    check_mimc_generic_equals_dedicated(hip_backend, 1 << 12)


@pytest.mark.gpu
def test_rescue_2p16_config_verifies(hip_backend):
    """BASELINE configs[2]: Rescue hash STARK, 128-bit field, 2^16 steps, blake2s256 Merkle, E = 16 (N = 2^20, 6 FRI layers)."""
    air, stark, full, assertions = None, None, None, None
    f = PrimeField(backend=hip_backend)
    steps = 1 << 16
    air = rescue4x128_air(steps, 16, f)
    stark = Stark(air, RESCUE_OPTS)
    trace = air.initProvingContext([], [42, 43]).generateExecutionTrace()
    assert (trace.getValue(0, 31), trace.getValue(1, 31)) == RESCUE_KAT
    assertions = [{'step': 31, 'register': 0, 'value': RESCUE_KAT[0]}, {'step': 31, 'register': 1, 'value': RESCUE_KAT[1]},
                  {'step': steps - 1, 'register': 2, 'value': trace.getValue(2, steps - 1)}]
    proof = stark.prove(assertions, [], [42, 43])
    assert len(proof['ldProof']['components']) == 6
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    assert stark.verify(assertions, stark.parse(data))


def degree5_air(field, steps, ef=16):
    """A synthetic Poseidon-SHAPED AIR (BASELINE configs[3] shape: 6 trace registers, degree-5 S-box => composition domain 8T,
    composition degree 7T): r_i' = r_i^5 + 3*r_{i+1} + k_{i mod 2}, with two cyclic static registers of periods 16 and 4."""
    ks = [[(7 * i + 1) % 1000003 for i in range(16)], [11, 22, 33, 44]]

    def transition(r, k):
        return [r[i] ** 5 + 3 * r[(i + 1) % 6] + k[i % 2] for i in range(6)]

    def evaluation(r, n, k):
        return [n[i] - (r[i] ** 5 + 3 * r[(i + 1) % 6] + k[i % 2]) for i in range(6)]

    return GenericAir(steps, 6, [5] * 6, ks, transition, evaluation, lambda seed: [seed[0] + i for i in range(6)], ef, field)


def check_degree5(backend, steps):
    f = PrimeField(backend=backend)
    air = degree5_air(f, steps)
    assert air.compositionFactor == 8
    stark = Stark(air, {'hashAlgorithm': 'sha256', 'extensionFactor': 16, 'exeQueryCount': 40, 'friQueryCount': 20})
    full = air.hostTrace([5])
    assertions = [{'step': 0, 'register': r, 'value': full[0][r]} for r in range(6)] + \
                 [{'step': steps - 1, 'register': 4, 'value': full[-1][4]}, {'step': steps // 2, 'register': 4, 'value': full[steps // 2][4]}]
    proof = stark.prove(assertions, [], [5])
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    assert len(proof['evProof']['values'][0]) == 6 * 16          # six registers merged per leaf (lib/Stark.ts:284-296)
    assert stark.verify(assertions, stark.parse(data))
    return data


def test_degree5_six_registers_oracle(oracle_backend):
    check_degree5(oracle_backend, 64)


@pytest.mark.gpu
def test_degree5_six_registers_hip_equals_oracle(hip_backend, oracle_backend):
    assert check_degree5(hip_backend, 1 << 10) == check_degree5(oracle_backend, 1 << 10)


# ---- Poseidon 6x128 (examples/poseidon/hash6x128.ts), BASELINE configs[3] ---------------------------------------------
POSEIDON_OPTS = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24}   # hash6x128.ts:33-39


def poseidon_case(backend, steps):
    from genstark_amd import poseidon
    f = PrimeField(backend=backend)
    air = poseidon.poseidon6x128_air(steps, 16, f)
    digest = poseidon.poseidon_hash(f, [1, 2, 3, 4])          # the example's own control computation (hash6x128.ts:19)
    assertions = [{'step': 63, 'register': 0, 'value': digest[0]}, {'step': 63, 'register': 1, 'value': digest[1]}]   # :93-96
    return air, Stark(air, POSEIDON_OPTS), digest, assertions


def check_poseidon(backend, steps, host_trace=True):
    air, stark, digest, assertions = poseidon_case(backend, steps)
    assert air.compositionFactor == 8 and air.traceRegisterCount == 6
    trace = air.initProvingContext([], [1, 2, 3, 4]).generateExecutionTrace()
    assert (trace.getValue(0, 63), trace.getValue(1, 63)) == tuple(digest)      # device-side VM trace == utils.ts createHash
    if host_trace:
        assert trace.toValues() == [list(r) for r in zip(*air.hostTrace([1, 2, 3, 4]))]
    proof = stark.prove(assertions, [], [1, 2, 3, 4])
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    assert stark.verify(assertions, stark.parse(data))
    wrong = [dict(assertions[0], value=(digest[0] + 1) % ga.MODULUS), assertions[1]]
    with pytest.raises(StarkError):
        stark.prove(wrong, [], [1, 2, 3, 4])
    with pytest.raises(StarkError):
        stark.verify(wrong, stark.parse(data))
    return data


def test_poseidon_parameters_follow_the_example():
    from genstark_amd import poseidon
    f = PrimeField.__new__(PrimeField)
    f.modulus = ga.MODULUS
    f.inv = lambda a: pow(a, ga.MODULUS - 2, ga.MODULUS)
    f.sub = lambda a, b: (a - b) % ga.MODULUS
    rc = poseidon.round_constants(f)
    assert len(rc) == 64 and all(len(r) == 6 for r in rc)
    assert rc[1][2] == int(hashlib.sha256(b'Hades8').hexdigest(), 16) % ga.MODULUS            # c = round * width + column
    mds = poseidon.mds_matrix(f)
    x1 = int(hashlib.sha256(b'HadesMDSx1').hexdigest(), 16) % ga.MODULUS
    y4 = int(hashlib.sha256(b'HadesMDSy4').hexdigest(), 16) % ga.MODULUS
    assert mds[1][4] * (x1 - y4) % ga.MODULUS == 1
    ctl = poseidon.round_controls()
    assert len(ctl) == 64 and sum(ctl) == 8 and ctl[:5] == [1, 1, 1, 1, 0] and ctl[58:] == [0, 1, 1, 1, 1, 0]


@pytest.mark.parametrize('steps', [64, 128])
def test_poseidon_prove_verify_oracle(oracle_backend, steps):
    check_poseidon(oracle_backend, steps)


@pytest.mark.gpu
@pytest.mark.parametrize('steps', [64, 1024])
def test_poseidon_hip_equals_oracle(hip_backend, oracle_backend, steps):
    assert check_poseidon(hip_backend, steps) == check_poseidon(oracle_backend, steps)


@pytest.mark.gpu
def test_poseidon_2p16_config_verifies(hip_backend):
    """BASELINE configs[3] on one GPU: 6 state registers, 2^16 steps, E = 16 (N = 2^20, composition domain 2^19)."""
    data = check_poseidon(hip_backend, 1 << 16, host_trace=False)
    assert len(data) > 100000


# ---- segmented traces: many independent hashes per proof, trace generated on the device (gs_air_trace_segments) --------
def check_segmented(backend, kind, steps):
    from genstark_amd import poseidon
    f = PrimeField(backend=backend)
    if kind == 'rescue':
        air, per = rescue4x128_air(steps, 16, f, segmented=True), 32
        seeds = [[42 + s, 43 + 2 * s] for s in range(steps // per)]
        opts = RESCUE_OPTS
    else:
        air, per = poseidon.poseidon6x128_air(steps, 16, f, segmented=True), 64
        seeds = [[1 + s, 2, 3 + s, 4] for s in range(steps // per)]
        opts = POSEIDON_OPTS
    stark = Stark(air, opts)
    trace = air.initProvingContext([], seeds).generateExecutionTrace()
    full = air.hostTrace(seeds)
    assert trace.toValues() == [list(r) for r in zip(*full)]           # device VM, one thread per segment == host integers
    if kind == 'rescue':
        assert (full[31][0], full[31][1]) == RESCUE_KAT                # segment 0 hashes (42, 43): hash4x128.ts:115-118
    else:
        for s in (0, len(seeds) - 1):
            assert full[64 * s + 63][:2] == poseidon.poseidon_hash(f, seeds[s])
    last = len(seeds) - 1
    assertions = [{'step': per - 1, 'register': 0, 'value': full[per - 1][0]}, {'step': per * last + per - 1, 'register': 1, 'value': full[per * last + per - 1][1]},
                  {'step': per * last, 'register': 0, 'value': full[per * last][0]}]
    proof = stark.prove(assertions, [], seeds)
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    assert stark.verify(assertions, stark.parse(data))
    # a trace that cheats INSIDE a segment must fail; the boundary rows are free by construction
    wrong = [dict(assertions[0], value=(assertions[0]['value'] + 1) % ga.MODULUS)] + assertions[1:]
    with pytest.raises(StarkError):
        stark.verify(wrong, stark.parse(data))
    return data


@pytest.mark.parametrize('kind,steps', [('rescue', 128), ('poseidon', 128)])
def test_segmented_hash_chains_oracle(oracle_backend, kind, steps):
    check_segmented(oracle_backend, kind, steps)


def test_segmented_air_rejects_wrong_seed_count(oracle_backend):
    air = rescue4x128_air(64, 16, PrimeField(backend=oracle_backend), segmented=True)
    with pytest.raises(Exception):
        air.initProvingContext([], [[1, 2]])


@pytest.mark.gpu
@pytest.mark.parametrize('kind,steps', [('rescue', 1024), ('poseidon', 1024)])
def test_segmented_hash_chains_hip_equals_oracle(hip_backend, oracle_backend, kind, steps):
    assert check_segmented(hip_backend, kind, steps) == check_segmented(oracle_backend, kind, steps)


@pytest.mark.gpu
@pytest.mark.parametrize('kind,steps', [('rescue', 1024), ('poseidon', 2048)])
def test_compiled_air_programs_equal_interpreted(hip_backend, kind, steps):
    """gs_air_jit: the transition function and the constraint evaluator compiled with hiprtc (csrc/air_jit.hip) give the
    interpreter's trace and the same proof bytes — and they really ran (gs_air_jit_launches)."""
    from genstark_amd._abi import Backend
    compiled = Backend(device=0).jit()
    try:
        assert compiled.jit_launches == 0
        data = check_segmented(compiled, kind, steps)            # also compares the device trace with host integers
        assert compiled.jit_launches >= 2, 'the programs were interpreted: hiprtc failed?'
        assert hip_backend.jit_launches == 0
        assert data == check_segmented(hip_backend, kind, steps)
    finally:
        compiled.close()


AUTO_MODE_SCRIPT = '''
import glob, os, sys, time
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_generic_air as t
from genstark_amd._abi import Backend
plain = Backend(device=0).jit(False)
want = t.check_segmented(plain, "poseidon", 1024)
assert plain.jit_launches == 0
auto = Backend(device=0)                         # the default of a new context: auto
first = t.check_segmented(auto, "poseidon", 1024)
assert first == want          # (interpreted unless the ~1 s background build finished between two launches of this very call)
deadline = time.time() + 150
files = []
while (auto.jit_launches < 2 or len(files) < 2) and time.time() < deadline:      # two programs: the trace kernel and the constraint evaluator
    time.sleep(0.5)
    assert t.check_segmented(auto, "poseidon", 1024) == want
    files = glob.glob(os.path.join(os.environ["GSTARK_JIT_CACHE_DIR"], "*.hsaco"))
assert auto.jit_launches >= 2, "the background build never delivered"
assert len(files) >= 2, files
print("AUTO OK", len(files))
'''


@pytest.mark.gpu
def test_air_programs_auto_mode(hip_backend, tmp_path):
    """The default mode of a context (gs_air_jit 2, "auto"), in a fresh process with an empty code-object cache: the first launches never
    wait for the compiler — they are interpreted while ONE background thread builds the programs; later launches run compiled; the code
    objects land in the disk cache; a second process finds them there and runs compiled from its first proof.  Same bytes throughout."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSTARK_JIT_CACHE_DIR=str(tmp_path))
    env.pop('GSTARK_AIR_JIT', None)
    r = subprocess.run([sys.executable, '-c', f'ROOT = {root!r}\n' + AUTO_MODE_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'AUTO OK' in r.stdout, r.stderr[-3000:]
    second = ('import os, sys\nROOT = %r\nsys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)\n'
              'import test_generic_air as t\nfrom genstark_amd._abi import Backend\nb = Backend(device=0)\n'
              't.check_segmented(b, "poseidon", 1024)\nassert b.jit_launches >= 2, b.jit_launches\nprint("WARM OK")\n') % root
    r = subprocess.run([sys.executable, '-c', second], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'WARM OK' in r.stdout, r.stderr[-3000:]


@pytest.mark.gpu
def test_exit_during_first_background_build(tmp_path):
    """a short-lived process in the default (auto) mode reaches exit() while the background builder is still inside hiprtc.
    The builder is joined before the HIP / comgr libraries are torn down: the process must end cleanly (exit code 0, no signal)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSTARK_JIT_CACHE_DIR=str(tmp_path / 'cache'))
    env.pop('GSTARK_AIR_JIT', None)
    script = ('import os, sys\nROOT = %r\nsys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)\n'
              'import test_generic_air as t\nfrom genstark_amd._abi import Backend\nb = Backend(device=0)\n'
              't.check_segmented(b, "poseidon", 1024)\nprint("PROVED", b.jit_launches)\n') % root      # ... and straight to exit()
    for _ in range(2):           # second round: one program may already be cached, the other still building
        r = subprocess.run([sys.executable, '-c', script], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and 'PROVED' in r.stdout, (r.returncode, r.stderr[-3000:])


@pytest.mark.gpu
def test_generic_suite_in_default_auto_mode(tmp_path):
    """The rest of this suite pins GSTARK_AIR_JIT=0 so that its results do not depend on what earlier processes left in the disk cache;
    the PRODUCT default is auto.  This runs the generic-AIR GPU tests once more in a child pytest with the variable unset (programs
    interpreted at first, compiled as the background builds land) and a cache directory of its own."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSTARK_JIT_CACHE_DIR=str(tmp_path / 'cache'), GSTARK_TEST_AUTO_CHILD='1')
    env.pop('GSTARK_AIR_JIT', None)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_generic_air.py'), os.path.join(root, 'tests', 'test_lib128.py'),
                        '-m', 'gpu', '-q', '-x', '-k', 'not auto_mode and not background_build and not compiled_air_programs_equal_interpreted', '-p', 'no:cacheprovider'],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['rescue', 'poseidon'])
def test_segmented_2p16_configs_verify(hip_backend, kind):
    """BASELINE configs[2] / configs[3] in the reference's sense: 2^16 steps = 2048 Rescue hashes / 1024 Poseidon hashes."""
    check_segmented(hip_backend, kind, 1 << 16)


# ---- BASELINE configs[0]: "Foo" (README.md:17-60) — the smallest STARK the reference shows -----------------------------
def foo_air(field, steps=64):
    """x_{n+1} = x_n + 2: one register, one degree-1 constraint => extension factor 4, evaluation domain 256, which is not
    larger than the FRI remainder bound: the low-degree proof has NO layers (LowDegreeProver.ts:179, getComponentCount = 0).
    The README runs it over 2^32 - 3*2^25 + 1; the arithmetic here is the 128-bit field's (the only one this build accelerates),
    the protocol path is the same."""
    return GenericAir(steps, 1, [1], [], lambda r, k: [r[0] + 2], lambda r, n, k: [n[0] - (r[0] + 2)], lambda seed: [seed[0]], None, field)


def check_foo(backend):
    f = PrimeField(backend=backend)
    air = foo_air(f)
    assert air.extensionFactor == 4 and air.compositionFactor == 1
    stark = Stark(air, None)                                   # default options: sha256, exe 80, fri 40 (lib/Stark.ts:17-23)
    assert stark.hash.algorithm == 'sha256'
    assertions = [{'step': 0, 'register': 0, 'value': 1}, {'step': 63, 'register': 0, 'value': 127}]     # README.md:42-45
    proof = stark.prove(assertions, [], [1])
    assert proof['ldProof']['components'] == [] and len(proof['ldProof']['remainder']) == 256
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    assert stark.verify(assertions, stark.parse(data))
    with pytest.raises(StarkError):
        stark.verify([assertions[0], {'step': 63, 'register': 0, 'value': 126}], stark.parse(data))
    with pytest.raises(StarkError):
        stark.prove([{'step': 63, 'register': 0, 'value': 128}], [], [1])
    return data


def test_foo_readme_example_oracle(oracle_backend):
    check_foo(oracle_backend)


@pytest.mark.gpu
def test_foo_readme_example_hip_equals_oracle(hip_backend, oracle_backend):
    assert check_foo(hip_backend) == check_foo(oracle_backend)


def check_poseidon_hash_through_facade(backend):
    """examples/poseidon/utils.ts:19-49 executed member by member on the galois-shaped field object (addVectorElements,
    expVectorElements, mulMatrixByVector, exp, newVectorFrom/newMatrixFrom, toValues) must give the digest the host-integer
    restatement gives."""
    from genstark_amd import poseidon
    f = PrimeField(backend=backend)
    m, rf, rp = poseidon.STATE_WIDTH, poseidon.F_ROUNDS, poseidon.P_ROUNDS
    mds = f.newMatrixFrom(poseidon.mds_matrix(f))
    ark = [f.newVectorFrom(v) for v in poseidon.round_constants(f, m, rf + rp)]
    for inputs in ([1, 2, 3, 4], [ga.MODULUS - 1, 0, 7]):
        state = f.newVectorFrom(list(inputs) + [0] * (m - len(inputs)))
        for i in range(rf + rp):
            state = f.addVectorElements(state, ark[i])
            if i < rf // 2 or i >= rf // 2 + rp:
                state = f.expVectorElements(state, 5)
            else:
                values = state.toValues()
                values[m - 1] = f.exp(values[m - 1], 5)
                state = f.newVectorFrom(values)
            state = f.mulMatrixByVector(mds, state)
        assert state.toValues()[:2] == poseidon.poseidon_hash(f, inputs)


def test_poseidon_hash_through_facade_oracle(oracle_backend):
    check_poseidon_hash_through_facade(oracle_backend)


@pytest.mark.gpu
def test_poseidon_hash_through_facade_hip(hip_backend):
    check_poseidon_hash_through_facade(hip_backend)


# ---- secret registers (AirScript `secret input`, examples/mimc/mimc128.ts:38; lib/Stark.ts:113-114, 284-313) -------------
def secret_air(field, steps=64, ef=16):
    """MiMC-like chain keyed by a SECRET cyclic register: x' = x^3 + k_public + s_secret, y' = y + s_secret * x.  Two trace
    registers, one public static register (period 8), two secret registers (periods 4 and steps): the leaves of the evaluation
    tree hold P_0 | P_1 | S_0 | S_1 and the verifier takes S(x) from the proof."""
    pub = [[(5 * i + 3) % 1009 for i in range(8)]]

    def transition(r, k):
        return [r[0] ** 3 + k[0] + k[1], r[1] + k[1] * r[0] + k[2]]

    def evaluation(r, n, k):
        t = transition(r, k)
        return [n[0] - t[0], n[1] - t[1]]

    return GenericAir(steps, 2, [3, 2], pub, transition, evaluation, lambda seed: [seed[0], seed[1]], ef, field, secretRegisters=2)


def check_secret_registers(backend, steps=64):
    f = PrimeField(backend=backend)
    air = secret_air(f, steps)
    assert air.secretInputCount == 2
    stark = Stark(air, {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 30, 'friQueryCount': 20})
    secrets = [[11, 22, 33, 44], [(7 * i * i + 1) % 100003 for i in range(steps)]]
    full = air.hostTrace([3, 4], inputs=secrets)
    trace = air.initProvingContext(secrets, [3, 4]).generateExecutionTrace().toValues()
    assert trace == [list(r) for r in zip(*full)]
    assertions = [{'step': 0, 'register': 0, 'value': 3}, {'step': steps - 1, 'register': 0, 'value': full[-1][0]},
                  {'step': steps - 1, 'register': 1, 'value': full[-1][1]}]
    proof = stark.prove(assertions, secrets, [3, 4])
    assert len(proof['evProof']['values'][0]) == 4 * 16            # P_0 | P_1 | S_0 | S_1 per leaf (lib/Stark.ts:284-296)
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    parsed = stark.parse(data)
    assert stark.verify(assertions, parsed)
    # the secret columns are bound: another secret gives another trace, and a proof with a doctored secret value is rejected
    other = stark.prove([assertions[0]], [[11, 22, 33, 45], secrets[1]], [3, 4])
    assert other['evRoot'] != proof['evRoot']
    leaf = bytearray(parsed['evProof']['values'][0])
    leaf[2 * 16] ^= 1                                               # first byte of S_0 at the first queried position
    parsed['evProof']['values'][0] = bytes(leaf)
    with pytest.raises(StarkError):
        stark.verify(assertions, parsed)
    with pytest.raises(Exception):
        stark.prove(assertions, [secrets[0]], [3, 4])               # one secret register missing
    return data


def test_secret_registers_oracle(oracle_backend):
    check_secret_registers(oracle_backend)


@pytest.mark.gpu
def test_secret_registers_hip_equals_oracle(hip_backend, oracle_backend):
    assert check_secret_registers(hip_backend, 256) == check_secret_registers(oracle_backend, 256)
