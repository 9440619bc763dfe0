"""AirAssembly source -> STARK (`instantiate(source, component, options)`, index.ts:18-33; genstark_amd/airassembly.py).
* Own fixtures (tests/golden/aa/*.aa, written for these tests): a cube chain whose proofs must equal the dedicated MiMC path byte
  for byte, and a module using every input-register feature (secret / public, peerof / childof, steps, shift, mask, cycles,
  matrix product, division) checked against a plain Python model of the same recurrence.
* In the build container the loader is also run on the REFERENCE's own sources (assembly/lib128.aa, assembly/lib224.aa,
  examples/elliptic/pointmul.aa, the inline module of examples/mimc/mimc128Assembly.ts) and must give, byte for byte, the proofs
  of this repository's hand transcriptions of those files (lib128.py, lib224.py, pointmul.py) — which in turn reproduce the
  reference examples' known answers."""
import os
import re

import pytest

import genstark_amd as ga
from conftest import ROOT
from genstark_amd import airassembly, lib128, lib224
from genstark_amd._abi import GstarkError
from genstark_amd.errors import StarkError
from genstark_amd.field import PrimeField
from genstark_amd.hostfield import HostField
from genstark_amd.pointmul import point_mul_air, to_bits
from genstark_amd._mirror.stark import Stark

AA = os.path.join(ROOT, 'tests', 'golden', 'aa')
REF = '/root/reference'
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'assembly')), reason='needs the genSTARK checkout (build container only)')
MIMC_OPTS = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 24}


def without_shapes(stark, proof):
    """The hand transcriptions are fixed-shape GenericAirs (no input shapes in the proof); everything else must match."""
    return stark.serialize(dict(proof, iShapes=[]))


def check_native_shaped(stark, options, assertions, inputs, seed, public, data, sweep=60):
    """The product entry on an air-assembly component WITH input registers (genstark_amd.prover.Prover -> csrc/prover.cc, csrc/verifier.h):
    * prove_bytes == the mirror's serialized proof, input shapes included (lib/Stark.ts:157-162);
    * verify_native sizes the trace from the shapes IN THE PROOF (initVerificationContext(proof.iShapes, publicInputs), lib/Stark.ts:176),
      lays the public registers out itself and gives the mirror's verdict — and the mirror's message — on the valid proof, on wrong
      public inputs, and on a sweep of corruptions over the whole proof, its shape bytes included."""
    import random
    from genstark_amd.prover import Prover
    p = Prover(stark.air, options)
    assert p.prove_bytes(assertions, inputs, seed) == data
    assert p.verify_native(assertions, data, public) is True

    def verdict(verify, blob):
        try:
            return ('ok', '') if verify(blob) else ('false', '')
        except (StarkError, GstarkError) as e:
            return ('rejected', str(e).split(':')[0])
        except (IndexError, ValueError, OverflowError, MemoryError) as e:      # the mirror's parser on a truncated proof
            return ('rejected', 'malformed')
    mirror = lambda blob: stark.verify(assertions, stark.parse(blob), public)
    native = lambda blob: p.verify_native(assertions, blob, public)
    rng = random.Random(len(data))
    offsets = sorted(set([0, 31, 32, 33, len(data) - 1] + [rng.randrange(len(data)) for _ in range(sweep)]))
    agree = 0
    for off in offsets:
        bad = bytearray(data)
        bad[off] ^= 1 << rng.randrange(8)
        got = [verdict(v, bytes(bad)) for v in (mirror, native)]
        assert got[0][0] == got[1][0] == 'rejected', (off, got)
        agree += got[0][1] == got[1][1] or got[0][1].startswith('Verification of low degree failed') or 'malformed' in got[0][1] + got[1][1]
    assert agree == len(offsets)
    for cut in (0, 1, 40, len(data) // 2, len(data) - 1, len(data) - 5):
        assert verdict(native, data[:cut])[0] == 'rejected'
    return p


def shape_bytes(stark, data):
    """(offset, length) of the serialized input shapes at the end of a proof (lib/Serializer.ts:70-78)"""
    n = 1 + sum(1 + 4 * len(sh) for sh in stark.parse(data)['iShapes'])
    return len(data) - n, n


# ---- own fixtures ---------------------------------------------------------------------------------------------------------------------
def check_cube_chain(backend):
    f = PrimeField(backend=backend)
    stark = airassembly.instantiate(os.path.join(AA, 'cube_chain.aa'), 'chain', MIMC_OPTS, field=f)
    assert stark.air.constraintDegrees == [3] and stark.air.secretInputCount == 0
    dedicated = ga.instantiateMimc(64, MIMC_OPTS, backend=backend)
    control = ga.runMimc(f, 64, dedicated.air.roundConstants, 3)
    assertions = [{'step': 0, 'register': 0, 'value': 3}, {'step': 63, 'register': 0, 'value': control[-1]}]
    data = stark.serialize(stark.prove(assertions, [], [3]))
    assert data == dedicated.serialize(dedicated.prove(assertions, [], [3]))          # the same statement, the same bytes
    assert stark.verify(assertions, stark.parse(data))
    return data


def ledger_model(p, balances, factors, deposits):
    rows = []
    for b, fac, deps in zip(balances, factors, deposits):
        r = [b % p, fac % p, 0]
        for t in range(2 * len(deps)):
            rows.append(r)
            step = len(rows) - 1
            dep = deps[min((t + 1) // 2, len(deps) - 1)]          # (shift -1): a step sees the value of the step after it
            n0 = (r[0] + dep * (1, 2, 3, 4)[step % 4]) % p
            n1 = (r[1] * fac + pow(2, step % 8, p)) % p
            s = (n0 + 2 * n1) % p
            r = [n0, n1, s * s * pow(3, p - 2, p) % p]
    return rows


def check_ledger(backend, runs):
    f = PrimeField(backend=backend)
    p = f.modulus
    stark = airassembly.instantiate(os.path.join(AA, 'ledger.aa'), 'default', {'hashAlgorithm': 'sha256', 'exeQueryCount': 24, 'friQueryCount': 12}, field=f)
    air = stark.air
    assert air.constraintDegrees == [3, 3, 5] and air.extensionFactor == 16 and air.secretInputCount == 2
    balances = [100 + 7 * i for i in range(runs)]
    factors = [3 + i for i in range(runs)]
    deposits = [[5 + i + 2 * j for j in range(4)] for i in range(runs)]
    inputs = [balances, factors, deposits]
    model = ledger_model(p, balances, factors, deposits)
    trace = air.initProvingContext(inputs).generateExecutionTrace().toValues()
    assert [list(r) for r in zip(*trace)] == model
    last = 8 * runs - 1
    assertions = [{'step': 0, 'register': 0, 'value': balances[0]}, {'step': last, 'register': 2, 'value': model[last][2]},
                  {'step': last, 'register': 0, 'value': model[last][0]}]
    proof = stark.prove(assertions, inputs)
    assert proof['iShapes'] == [[runs], [runs], [runs, 4]]
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    assert stark.verify(assertions, stark.parse(data), [deposits])                       # the verifier supplies the PUBLIC register
    hv = airassembly.instantiate(os.path.join(AA, 'ledger.aa'), 'default', {'hashAlgorithm': 'sha256', 'exeQueryCount': 24, 'friQueryCount': 12},
                                 field=HostField(p))
    assert hv.verify(assertions, hv.parse(data), [deposits])                             # ... also without a GPU
    other = [list(d) for d in deposits]
    other[0][1] += 1
    with pytest.raises(StarkError):
        stark.verify(assertions, stark.parse(data), [other])
    with pytest.raises(StarkError):
        stark.verify([dict(assertions[1], value=model[last][2] ^ 1)], stark.parse(data), [deposits])
    # ---- the product entry: native prove() with the shapes in the proof, native verify() sized from them
    opts = {'hashAlgorithm': 'sha256', 'exeQueryCount': 24, 'friQueryCount': 12}
    p = check_native_shaped(stark, opts, assertions, inputs, None, [deposits], data)
    for wrong, msg in (([other], 'linear combination correctness'), ([], 'public input registers are needed'), ([deposits[:-1]], 'values expected')):
        with pytest.raises(StarkError, match=msg):
            p.verify_native(assertions, data, wrong)
    with pytest.raises(StarkError, match='linear combination correctness'):
        p.verify_native([dict(assertions[1], value=model[last][2] ^ 1)], data, [deposits])
    # every byte of the serialized shapes matters: a changed rank, dimension or count is a different layout (or none at all)
    at, n = shape_bytes(stark, data)
    assert n == 1 + (1 + 4) + (1 + 4) + (1 + 8)
    for k in range(n):
        bad = bytearray(data)
        bad[at + k] ^= 1
        with pytest.raises(StarkError):
            p.verify_native(assertions, bytes(bad), [deposits])
    return data


def test_cube_chain_equals_dedicated_mimc_oracle(oracle_backend):
    check_cube_chain(oracle_backend)


@pytest.mark.parametrize('runs', [2, 4])
def test_ledger_oracle(oracle_backend, runs):
    check_ledger(oracle_backend, runs)


def test_source_errors(oracle_backend):
    f = PrimeField(backend=oracle_backend)
    src = open(os.path.join(AA, 'ledger.aa')).read()
    with pytest.raises(GstarkError, match='not exported'):
        airassembly.AssemblyAir(src, 'nope', None, f)
    with pytest.raises(GstarkError, match='unknown operation'):
        airassembly.AssemblyAir(src.replace('(exp (load.local $sum) (scalar 2))', '(frobnicate (load.local $sum))'), 'default', None, f)
    with pytest.raises(GstarkError, match='constraints declared'):
        airassembly.AssemblyAir(src.replace('(constraints 3)', '(constraints 4)'), 'default', None, f)
    air = airassembly.AssemblyAir(src, 'default', None, f)
    with pytest.raises(GstarkError, match='one entry'):
        air.initProvingContext([[1], [2]])
    with pytest.raises(GstarkError, match='ragged'):
        air.initProvingContext([[1, 2], [3, 4], [[1, 2, 3, 4], [1, 2]]])
    with pytest.raises(GstarkError, match='peer'):
        air.initProvingContext([[1, 2], [3], [[1, 2, 3, 4], [1, 2, 3, 4]]])
    with pytest.raises(GstarkError, match='power of 2'):
        air.initProvingContext([[1, 2, 3], [3, 4, 5], [[1, 2, 3, 4]] * 3])
    with pytest.raises(GstarkError, match='field'):
        airassembly.AssemblyAir(src.replace('340282366920938463463374607393113505793', '96769'), 'default', None, f)
    assert airassembly.parse('(a (b 1) # comment\n c)') == [['a', ['b', '1'], 'c']]


@pytest.mark.gpu
def test_airassembly_hip_equals_oracle(hip_backend, oracle_backend):
    assert check_cube_chain(hip_backend) == check_cube_chain(oracle_backend)
    assert check_ledger(hip_backend, 4) == check_ledger(oracle_backend, 4)
    assert check_ledger(hip_backend, 16) == check_ledger(oracle_backend, 16)             # 16 runs: device-side trace segments


# ---- the reference's own sources (build container) ---------------------------------------------------------------------------------------
LIB_OPTS = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 32, 'exeQueryCount': 44, 'friQueryCount': 20}


@needs_reference
def test_reference_lib128_source_equals_hand_transcription(oracle_backend):
    from test_lib128 import merkle_case
    f = PrimeField(backend=oracle_backend)
    src = open(os.path.join(REF, 'assembly', 'lib128.aa')).read()
    hashes = 2
    raw = [[42 + 10 * s for s in range(hashes)], [43 + s for s in range(hashes)], [44] * hashes, [45 + s * s for s in range(hashes)]]
    air = lib128.compute_poseidon_hash_air(f, hashes)
    hand = Stark(air, LIB_OPTS)
    assertions = []
    for s in range(hashes):
        d = lib128.poseidon_hash(f, [c[s] for c in raw])
        assertions += [{'step': 64 * s + 63, 'register': 0, 'value': d[0]}, {'step': 64 * s + 63, 'register': 1, 'value': d[1]}]
    want = without_shapes(hand, hand.prove(assertions, air.expandInputs(raw), air.segmentSeeds(raw)))
    stark = airassembly.instantiate(src, 'ComputePoseidonHash', LIB_OPTS, field=f)
    assert stark.air.constraintDegrees == [7] * 6
    proof = stark.prove(assertions, raw)                                                  # lib128.ts:61-64: prove(assertions, inputs)
    assert without_shapes(stark, proof) == want and proof['iShapes'] == [[2]] * 4
    assert stark.verify(assertions, stark.parse(stark.serialize(proof)))
    check_native_shaped(stark, LIB_OPTS, assertions, raw, None, None, stark.serialize(proof), sweep=25)      # four secret input registers
    # ComputeMerkleRoot with its public index-bit register (lib128.ts:97-112)
    tree, leaf, nodes, bits = merkle_case(f, 4, 5)
    air = lib128.compute_merkle_root_air(f, bits)
    inputs, first = lib128.merkle_inputs(f, leaf, nodes)
    hand = Stark(air, LIB_OPTS)
    assertions = [{'step': 255, 'register': 0, 'value': tree.root[0]}, {'step': 255, 'register': 1, 'value': tree.root[1]}]
    want = without_shapes(hand, hand.prove(assertions, inputs, first))
    stark = airassembly.instantiate(src, 'ComputeMerkleRoot', LIB_OPTS, field=f)
    raw = [[leaf[0]], [leaf[1]], [[n[0] for n in nodes]], [[n[1] for n in nodes]], [bits]]
    proof = stark.prove(assertions, raw)
    assert without_shapes(stark, proof) == want
    data = stark.serialize(proof)
    assert stark.verify(assertions, stark.parse(data), [[bits]])                         # lib128.ts:112: verify(assertions, proof, [[indexBits]])
    with pytest.raises(StarkError):
        stark.verify(assertions, stark.parse(data), [[[bits[0], 1 - bits[1]] + bits[2:]]])
    nat = check_native_shaped(stark, LIB_OPTS, assertions, raw, None, [[bits]], data, sweep=25)               # ... and a public one, nested
    with pytest.raises(StarkError, match='linear combination correctness'):
        nat.verify_native(assertions, data, [[[bits[0], 1 - bits[1]] + bits[2:]]])
    assert airassembly.AssemblyAir(src, 'ComputeMerkleUpdate', 32, f).constraintDegrees == [8] * 24 + [2]


@needs_reference
def test_reference_mimc_assembly_source_equals_dedicated_path(oracle_backend):
    f = PrimeField(backend=oracle_backend)
    ts = open(os.path.join(REF, 'examples', 'mimc', 'mimc128Assembly.ts')).read()
    src = re.search(r'Buffer\.from\(`(.*?)`', ts, re.S).group(1).replace('${steps}', '64').replace('${constantCount}', '64')
    stark = airassembly.instantiate(src, 'mimc', MIMC_OPTS, field=f)
    dedicated = ga.instantiateMimc(64, MIMC_OPTS, backend=oracle_backend)
    control = ga.runMimc(f, 64, dedicated.air.roundConstants, 3)
    assertions = [{'step': 0, 'register': 0, 'value': 3}, {'step': 63, 'register': 0, 'value': control[-1]}]
    assert stark.serialize(stark.prove(assertions, [], [3])) == dedicated.serialize(dedicated.prove(assertions, [], [3]))


@needs_reference
def test_reference_224_bit_sources_equal_hand_transcriptions():
    from test_lib224 import SIG_G, SIG_H, SIG_P, SIG_R, SIG_S
    from test_wide_fields import EC_OPTIONS, EC_POINT, EC_PRODUCT, EC_SCALAR, oracle_for
    f = PrimeField(backend=oracle_for('p224'))
    air = point_mul_air(f)
    raw = [[EC_POINT[0]], [EC_POINT[1]], [to_bits(EC_SCALAR)]]
    assertions = [{'step': 255, 'register': 2, 'value': EC_PRODUCT[0]}, {'step': 255, 'register': 3, 'value': EC_PRODUCT[1]}]     # pointMul.ts:33-36
    hand = Stark(air, EC_OPTIONS)
    want = without_shapes(hand, hand.prove(assertions, air.expandInputs(raw), air.segmentSeeds(raw)))
    stark = airassembly.instantiate(os.path.join(REF, 'examples', 'elliptic', 'pointmul.aa'), 'default', EC_OPTIONS, field=f)
    proof = stark.prove(assertions, raw)                                                  # pointMul.ts:39
    assert without_shapes(stark, proof) == want and proof['iShapes'] == [[1], [1], [1, 256]]
    check_native_shaped(stark, EC_OPTIONS, assertions, raw, None, None, stark.serialize(proof), sweep=10)      # the 224-bit flavour of driver and verifier
    src = open(os.path.join(REF, 'assembly', 'lib224.aa')).read()
    degrees = {c: airassembly.AssemblyAir(src, c, 32, f).constraintDegrees for c in ('ComputePoseidonHash', 'ComputeMerkleRoot', 'ComputeMerkleUpdate', 'VerifySchnorrSignature')}
    assert degrees == {'ComputePoseidonHash': [7] * 3, 'ComputeMerkleRoot': [8] * 6, 'ComputeMerkleUpdate': [8] * 12 + [2], 'VerifySchnorrSignature': lib224.SCHNORR_DEGREES}
    # VerifySchnorrSignature: the same trace and the same constraint values as the hand transcription (a full proof over the
    # _BitInt oracle takes half a minute; the GPU suite proves it: test_lib224.py)
    import random
    rng = random.Random(9)
    hand = lib224.verify_schnorr_signature_air(f)
    raw = [[SIG_G[0]], [SIG_G[1]], [to_bits(SIG_S)], [SIG_P[0]], [SIG_P[1]], [to_bits(SIG_H)], [SIG_R[0]], [SIG_R[1]]]
    loaded = airassembly.AssemblyAir(src, 'VerifySchnorrSignature', 16, f)
    context = loaded.initProvingContext(raw)
    trace = context.generateExecutionTrace().toValues()
    assert trace == hand.initProvingContext(hand.expandInputs(raw), hand.segmentSeeds(raw)).generateExecutionTrace().toValues()
    assert (trace[2][255], trace[3][255]) == (trace[9][255], trace[10][255]) and trace[13][255] == SIG_H      # lib224.ts:184-208
    inner = context.air
    p = f.modulus
    for _ in range(5):
        r, n, k = ([rng.randrange(p) for _ in range(count)] for count in (14, 14, 10))
        assert inner.evaluationProgram.run(r, n, k) == hand.evaluationProgram.run(r, n, k)


# ---- the node side: compile(source) / AirSchema of the air-assembly drop-in (js/air_assembly.js -> js/aa_loader.js) -------------------
import hashlib
import json
import shutil
import subprocess

from conftest import ORACLE_LIB

HERE = os.path.dirname(os.path.abspath(__file__))
needs_node_and_reference = pytest.mark.skipif(
    not (os.path.isdir(os.path.join(REF, 'bin', 'lib')) and shutil.which('node') and os.path.exists('/usr/include/node/node_api.h')),
    reason='needs the genSTARK checkout and node (build container only)')
with open(os.path.join(HERE, 'golden', 'reference_index_proofs.json')) as _f:
    INDEX_PROOFS = {r['name']: r for r in json.load(_f)}


def index_cases(backend):
    """Statements proved through the reference's bin/index.js: instantiate(SOURCE, component, options) — this repository's own modules
    (the reference's inline MiMC module is added by the live test, which reads it from the checkout): the cube chain (no inputs) and the
    ledger (secret and public input registers, nested shapes: the proof carries iShapes, the verifier supplies the public register)."""
    f = PrimeField(backend=backend)
    chain = open(os.path.join(AA, 'cube_chain.aa')).read()
    rc = ga.instantiateMimc(64, MIMC_OPTS, backend=backend).air.roundConstants
    out = []
    for steps in (64, 1024):
        control = ga.runMimc(f, steps, rc, 3)
        out.append({'name': f'cube_chain_{steps}', 'source': chain.replace('(steps 64)', f'(steps {steps})'), 'component': 'chain', 'options': MIMC_OPTS, 'seed': ['3'],
                    'assertions': [{'step': 0, 'register': 0, 'value': '3'}, {'step': steps - 1, 'register': 0, 'value': str(control[-1])}]})
    runs = 4
    balances, factors = [100 + 7 * i for i in range(runs)], [3 + i for i in range(runs)]
    deposits = [[5 + i + 2 * j for j in range(4)] for i in range(runs)]
    model = ledger_model(f.modulus, balances, factors, deposits)
    last = 8 * runs - 1
    strs = lambda x: [strs(v) for v in x] if isinstance(x, list) else str(x)
    out.append({'name': 'ledger_4_runs', 'source': open(os.path.join(AA, 'ledger.aa')).read(), 'component': 'default',
                'options': {'hashAlgorithm': 'sha256', 'exeQueryCount': 24, 'friQueryCount': 12}, 'inputs': strs([balances, factors, deposits]), 'publicInputs': strs([deposits]),
                'assertions': [{'step': 0, 'register': 0, 'value': str(balances[0])}, {'step': last, 'register': 2, 'value': str(model[last][2])},
                               {'step': last, 'register': 0, 'value': str(model[last][0])}]})
    return out


def python_bytes(case, backend):
    ints = lambda x: [ints(v) for v in x] if isinstance(x, list) else int(x)
    stark = airassembly.instantiate(case['source'], case['component'], case['options'], field=PrimeField(backend=backend))
    a = [dict(x, value=int(x['value'])) for x in case['assertions']]
    return stark.serialize(stark.prove(a, ints(case.get('inputs', [])), ints(case['seed']) if 'seed' in case else None))


def check_index_fixture(backend):
    for case in index_cases(backend):
        data = python_bytes(case, backend)
        rec = INDEX_PROOFS[case['name']]
        assert (len(data), hashlib.sha256(data).hexdigest()) == (rec['proofSize'], rec['proofSha256']), case['name']


def check_js_onecall(backend, lib_env, tmp_path):
    """js/prover.js: proveAssemblySerialized / verifyAssemblySerialized — the cube chain (no inputs) and the ledger (2 secret + 1 public
    input register, nested shapes) through ONE native call each from node: the bytes the Python host proves from the same source
    (shapes included), verified natively from node with the public inputs, tampering and wrong public inputs refused."""
    cases = index_cases(backend)
    cin, cout = tmp_path / 'cases.json', tmp_path / 'out.json'
    cin.write_text(json.dumps(cases))
    subprocess.check_call(['bash', os.path.join(ROOT, 'napi', 'build.sh')])
    r = subprocess.run(['node', os.path.join(HERE, 'js_assembly_onecall.js'), str(cin), str(cout)], env=dict(os.environ, **lib_env), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'js one-call assembly OK: 3 statements' in r.stdout, r.stderr[-3000:]
    got = {rec['name']: rec for rec in json.loads(cout.read_text())}
    for case in cases:
        assert bytes.fromhex(got[case['name']]['proofHex']) == python_bytes(case, backend), case['name']
        assert got[case['name']]['verified'] and got[case['name']]['tamperRejected']


@pytest.mark.skipif(not (shutil.which('node') and os.path.exists('/usr/include/node/node_api.h')), reason='node or its headers are not in this image')
def test_js_onecall_assembly_statements_on_oracle_double(oracle_backend, tmp_path):
    check_js_onecall(oracle_backend, {'GSTARK_LIB': ORACLE_LIB, 'GSTARK_ALLOW_TEST_DOUBLE': '1'}, tmp_path)


@pytest.mark.gpu
def test_js_onecall_assembly_statements_on_hip(hip_backend, tmp_path):
    check_js_onecall(hip_backend, {}, tmp_path)


def test_python_host_reproduces_what_the_reference_index_js_proved(oracle_backend):
    """tests/golden/reference_index_proofs.json: proofs the reference's own bin/index.js produced from AirAssembly SOURCE over the drop-in
    modules (generated in the build container by the live test below) == what the Python host proves from the same source."""
    check_index_fixture(oracle_backend)


@pytest.mark.gpu
def test_hip_backend_reproduces_what_the_reference_index_js_proved(hip_backend):
    check_index_fixture(hip_backend)


@needs_node_and_reference
def test_reference_index_js_compiles_airassembly_source_live(oracle_backend, tmp_path):
    """index.ts:18-33 unmodified: `instantiate(source)` -> `compileAirAssembly(source)` (the shim's compile(): an AirSchema from this
    repository's loader) -> `new Stark(schema, component, options)`; prove / serialize / parse / verify are the reference's.  Bytes ==
    the Python host's from the same source == the committed fixture; the reference's inline MiMC module (mimc128Assembly.ts:28-51) ==
    the dedicated MiMC path."""
    subprocess.check_call(['bash', os.path.join(ROOT, 'napi', 'build.sh')])
    ts = open(os.path.join(REF, 'examples', 'mimc', 'mimc128Assembly.ts')).read()
    inline = re.search(r'Buffer\.from\(`(.*?)`', ts, re.S).group(1).replace('${steps}', '256').replace('${constantCount}', '64')
    control = ga.runMimc(PrimeField(backend=oracle_backend), 256, ga.instantiateMimc(64, MIMC_OPTS, backend=oracle_backend).air.roundConstants, 3)
    cases = index_cases(oracle_backend) + [{'name': 'reference_inline_mimc_256', 'source': inline, 'component': 'mimc', 'options': MIMC_OPTS, 'seed': ['3'],
                                            'assertions': [{'step': 0, 'register': 0, 'value': '3'}, {'step': 255, 'register': 0, 'value': str(control[-1])}]}]
    cin, cout = tmp_path / 'cases.json', tmp_path / 'out.json'
    cin.write_text(json.dumps(cases))
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, 'js', 'shims'), GSTARK_LIB=ORACLE_LIB, GSTARK_ALLOW_TEST_DOUBLE='1')
    r = subprocess.run(['node', os.path.join(HERE, 'golden', 'run_reference_index.js'), os.path.join(REF, 'bin'), str(cin), str(cout)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = {rec['name']: rec for rec in json.loads(cout.read_text())}
    for case in cases:
        rec = out[case['name']]
        assert rec['verified'] is True and rec['tamperRejected'] is True
        assert rec['iShapes'] == ([[4], [4], [4, 4]] if case['name'].startswith('ledger') else [])
        data = bytes.fromhex(rec['proofHex'])
        assert data == python_bytes(case, oracle_backend), case['name']
        if case['name'] in INDEX_PROOFS:
            assert (len(data), hashlib.sha256(data).hexdigest()) == (INDEX_PROOFS[case['name']]['proofSize'], INDEX_PROOFS[case['name']]['proofSha256'])
    dedicated = ga.instantiateMimc(256, MIMC_OPTS, backend=oracle_backend)
    a = [dict(x, value=int(x['value'])) for x in cases[-1]['assertions']]
    assert bytes.fromhex(out['reference_inline_mimc_256']['proofHex']) == dedicated.serialize(dedicated.prove(a, [], [3]))
    if os.environ.get('GSTARK_WRITE_FIXTURES') == '1':
        with open(os.path.join(HERE, 'golden', 'reference_index_proofs.json'), 'w') as fh:
            json.dump([{'name': n, 'proofSize': len(bytes.fromhex(out[n]['proofHex'])), 'proofSha256': hashlib.sha256(bytes.fromhex(out[n]['proofHex'])).hexdigest(),
                        'securityLevel': out[n]['securityLevel'], 'friLayers': out[n]['friLayers'],
                        'generated_by': 'tests/golden/run_reference_index.js: the reference\'s bin/index.js over js/shims (GSTARK_WRITE_FIXTURES=1 pytest tests/test_airassembly.py -k live)'}
                       for n in out if not n.startswith('reference_')], fh, indent=1)


@needs_node_and_reference
@pytest.mark.parametrize('script,lib,expect', [
    ('examples/mimc/mimc128Assembly.js', 'liboracle.so', ['STARK verified in', 'STARK security level: 96', 'Computed 48 evaluation spot checks']),
    ('examples/assembly/lib128.js', 'liboracle.so', ['STARK verified in', 'Security level: 88', 'Computed 44 evaluation spot checks']),
    ('examples/assembly/lib224.js', 'liboracle_p224.so', ['STARK verified in', 'Security level: 88']),
    ('examples/elliptic/pointMul.js', 'liboracle_p224.so', ['STARK verified in', 'STARK security level: 67']),
])
def test_reference_example_script_runs_unmodified(oracle_backend, script, lib, expect):
    """`node bin/examples/...js` of the genSTARK checkout, not a byte changed — every example of the reference that is written in AirAssembly:
    requires ../../index (-> lib/Stark, and the four @guildofweavers packages = js/shims), compiles its module (inline, or assembly/*.aa /
    pointmul.aa read from the checkout), builds its control values with the example's OWN code (the package's prng.sha256; Poseidon hash and
    Merkle tree of examples/poseidon/utils.js; k*G of pointMul.js), proves — secret and public input registers, 12 trace registers in the
    Merkle update, the 224-bit field in two of them — serializes where the script does, and VERIFIES with the reference's verifier against
    those control values.  (The other twelve examples are AirScript sources: that compiler is out of scope, js/shims/.../air-script says so.)"""
    from conftest import _build_oracle
    _build_oracle()
    subprocess.check_call(['bash', os.path.join(ROOT, 'napi', 'build.sh')])
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, 'js', 'shims'), GSTARK_LIB=os.path.join(ROOT, 'oracle', lib), GSTARK_ALLOW_TEST_DOUBLE='1')
    r = subprocess.run(['node', os.path.join(REF, 'bin', script)], env=env, cwd=os.path.join(REF, 'bin'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    for line in expect:
        assert line in r.stdout, (line, r.stdout[-1500:])


# ---- node: compile(source) -> the native driver through the addon (no reference code involved: runs on the GPU box as well) -------------
NODE_COMPILE_JS = """
const path = require('path'), fs = require('fs'), crypto = require('crypto');
const repo = process.argv[2];
const { compile, instantiate, AirSchema } = require(path.join(repo, 'js', 'shims', '@guildofweavers', 'air-assembly'));
const { proveGenericSerialized, verifyGenericSerialized } = require(path.join(repo, 'js', 'prover.js'));
const c = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
const schema = compile(Buffer.from(c.source));
if (!(schema instanceof AirSchema)) throw new Error('compile() must return an AirSchema');
const air = instantiate(schema, c.component, c.options);
const assertions = c.assertions.map(a => ({ step: a.step, register: a.register, value: BigInt(a.value) }));
const bytes = proveGenericSerialized(air.generic, c.options, assertions, c.seed.map(BigInt));
if (verifyGenericSerialized(air.generic, c.options, assertions, bytes) !== true) throw new Error('native verifier refused the proof');
let refused = false;
try { compile('(module (field prime 97) (export x (registers 1)'); } catch (e) { refused = /AirAssembly/.test(e.message); }
console.log(JSON.stringify({ size: bytes.length, sha256: crypto.createHash('sha256').update(bytes).digest('hex'), malformedRefused: refused }));
"""


def check_node_compile_to_native_driver(backend, lib, allow_double, tmp_path):
    from shutil import which
    if not which('node') or not os.path.exists(os.path.join(ROOT, 'napi', 'gstark_napi.node')):
        pytest.skip('node / the addon is not available')
    script, cj = tmp_path / 'compile.js', tmp_path / 'case.json'
    script.write_text(NODE_COMPILE_JS)
    for case in index_cases(backend)[:2]:
        cj.write_text(json.dumps(case))
        env = dict(os.environ, NODE_PATH=os.path.join(ROOT, 'js', 'shims'))
        if lib:
            env['GSTARK_LIB'] = lib
        if allow_double:
            env['GSTARK_ALLOW_TEST_DOUBLE'] = '1'
        r = subprocess.run(['node', str(script), ROOT, str(cj)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads(r.stdout.strip().splitlines()[-1])
        rec = INDEX_PROOFS[case['name']]
        assert (out['size'], out['sha256']) == (rec['proofSize'], rec['proofSha256']) and out['malformedRefused'] is True, case['name']


def test_node_compile_to_native_driver_on_oracle_double(oracle_backend, tmp_path):
    """compile(source) of the air-assembly drop-in -> AirSchema -> instantiate -> ONE call of the native driver through the N-API addon:
    the bytes the reference's bin/index.js produced for the same source (tests/golden/reference_index_proofs.json); a malformed module is
    refused with the loader's message."""
    subprocess.check_call(['bash', os.path.join(ROOT, 'napi', 'build.sh')])
    check_node_compile_to_native_driver(oracle_backend, ORACLE_LIB, True, tmp_path)


@pytest.mark.gpu
def test_node_compile_to_native_driver_on_hip(hip_backend, tmp_path):
    check_node_compile_to_native_driver(hip_backend, None, False, tmp_path)


# ---- the two loaders: js/aa_loader.js (what the node side runs) against genstark_amd/airassembly.py through aa_json.handle ----------------
def loader_requests():
    """Every request kind of js/air_assembly.js on this repository's modules and, where the checkout is present, on the reference's own
    assembly sources: check / info / describe / plan (random inputs, several shapes) / verify, and requests both loaders must refuse."""
    import random
    rnd = random.Random(11)
    p = 340282366920938463463374607393113505793
    S = lambda x: [S(v) for v in x] if isinstance(x, list) else str(x)
    led, cc = (open(os.path.join(AA, name)).read() for name in ('ledger.aa', 'cube_chain.aa'))
    reqs = []
    for src, comp in ((led, 'default'), (cc, 'chain')):
        reqs.append({'op': 'check', 'source': src})
        reqs += [{'op': 'info', 'source': src, 'component': comp, 'extensionFactor': ef} for ef in (None, 16, 32)]
    reqs.append({'op': 'describe', 'source': cc, 'component': 'chain', 'extensionFactor': None})
    reqs.append({'op': 'plan', 'source': cc, 'component': 'chain', 'extensionFactor': None, 'inputs': [], 'seed': ['3']})
    for runs in (1, 2, 4, 8):
        inputs = [[rnd.randrange(p) for _ in range(runs)], [rnd.randrange(p) for _ in range(runs)], [[rnd.randrange(p) for _ in range(4)] for _ in range(runs)]]
        if runs == 4:
            inputs[2] = [[7, 9, 7, 9]] * runs              # a public column with a short period: it shrinks to a cyclic register of 4 values
        reqs.append({'op': 'plan', 'source': led, 'component': 'default', 'extensionFactor': None, 'inputs': S(inputs), 'seed': None})
        reqs.append({'op': 'verify', 'source': led, 'component': 'default', 'extensionFactor': None, 'inputShapes': [[runs], [runs], [runs, 4]], 'publicInputs': S([inputs[2]])})
    for shift in ('', '(shift 1)', '(shift -2)'):         # other rotations of the input columns than the module's own (shift -1)
        src = led.replace('(input secret (shift -1))', f'(input secret {shift})').replace('(steps 2) (shift -1))', f'(steps 2) {shift})')
        inputs = [[rnd.randrange(p) for _ in range(2)], [rnd.randrange(p) for _ in range(2)], [[rnd.randrange(p) for _ in range(4)] for _ in range(2)]]
        reqs.append({'op': 'plan', 'source': src, 'component': 'default', 'extensionFactor': None, 'inputs': S(inputs), 'seed': None})
    # refused alike
    reqs.append({'op': 'plan', 'source': led, 'component': 'default', 'extensionFactor': None, 'inputs': S([[1, 2], [3], [[1, 2, 3, 4], [1, 2, 3, 4]]]), 'seed': None})
    reqs.append({'op': 'plan', 'source': led, 'component': 'default', 'extensionFactor': None, 'inputs': S([[1, 2, 3], [3, 4, 5], [[1, 2, 3, 4]] * 3]), 'seed': None})
    reqs.append({'op': 'plan', 'source': led, 'component': 'default', 'extensionFactor': None, 'inputs': S([[1, 2], [3, 4], [[1, 2, 3, 4], [1, 2, 3]]]), 'seed': None})
    reqs.append({'op': 'info', 'source': led, 'component': 'nope', 'extensionFactor': None})
    reqs.append({'op': 'info', 'source': led, 'component': 'default', 'extensionFactor': 2})
    reqs.append({'op': 'info', 'source': led.replace('(constraints 3)', '(constraints 4)'), 'component': 'default', 'extensionFactor': None})
    reqs.append({'op': 'info', 'source': led.replace('(exp (load.local $sum) (scalar 2))', '(frobnicate (load.local $sum))'), 'component': 'default', 'extensionFactor': None})
    reqs.append({'op': 'check', 'source': '(module (const $a scalar 1))'})
    if os.path.isdir(os.path.join(REF, 'assembly')):
        lib = open(os.path.join(REF, 'assembly', 'lib128.aa')).read()
        reqs.append({'op': 'check', 'source': lib})
        reqs += [{'op': 'info', 'source': lib, 'component': comp, 'extensionFactor': 32}
                 for comp in ('ComputePoseidonHash', 'ComputeMerkleRoot', 'ComputeMerkleUpdate', 'VerifySchnorrSignature')]
        for n in (1, 2):
            reqs.append({'op': 'plan', 'source': lib, 'component': 'ComputePoseidonHash', 'extensionFactor': 32,
                         'inputs': S([[rnd.randrange(p) for _ in range(n)] for _ in range(4)]), 'seed': None})
            inputs = [[rnd.randrange(p) for _ in range(n)], [rnd.randrange(p) for _ in range(n)]] + \
                     [[[rnd.randrange(p) for _ in range(4)] for _ in range(n)] for _ in range(2)] + [[[rnd.randrange(2) for _ in range(4)] for _ in range(n)]]
            reqs.append({'op': 'plan', 'source': lib, 'component': 'ComputeMerkleRoot', 'extensionFactor': 32, 'inputs': S(inputs), 'seed': None})
        for path in (('assembly', 'lib224.aa'), ('examples', 'elliptic', 'pointmul.aa')):
            src = open(os.path.join(REF, *path)).read()
            reqs.append({'op': 'check', 'source': src})
            reqs += [{'op': 'info', 'source': src, 'component': name, 'extensionFactor': None} for name in aa_exports(src)]
    return reqs


def aa_exports(src):
    from genstark_amd import airassembly
    return list(airassembly.Module(src).exports)


@pytest.mark.skipif(not shutil.which('node'), reason='node is not in this image')
def test_javascript_loader_answers_like_the_python_loader(tmp_path):
    """js/aa_loader.js — what compile() / instantiate() of the air-assembly drop-in run on the node side — gives the answers of
    genstark_amd/aa_json.handle object for object: exports, degrees, extension factors, the register-machine programs (code words,
    constant pools, scratch registers), static registers, first rows, secret columns, input shapes; and refuses what it refuses, with
    its words."""
    from genstark_amd import aa_json
    reqs = loader_requests()
    (tmp_path / 'reqs.json').write_text(json.dumps(reqs))
    js = ("const { handle } = require(process.argv[1]); const reqs = JSON.parse(require('fs').readFileSync(process.argv[2], 'utf8'));"
          "process.stdout.write(JSON.stringify(reqs.map(r => { try { return handle(r); } catch (e) { return { error: e.message }; } })));")
    r = subprocess.run(['node', '-e', js, os.path.join(ROOT, 'js', 'aa_loader.js'), str(tmp_path / 'reqs.json')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout)
    assert len(got) == len(reqs)
    refused = 0
    for req, g in zip(reqs, got):
        try:
            want = json.loads(json.dumps(aa_json.handle(req)))
        except Exception as e:   # noqa: BLE001
            want = {'error': f'{type(e).__name__}: {e}'}
        refused += 'error' in want
        if 'error' in want and want['error'].startswith('GstarkError'):
            assert g == want, (req['op'], req.get('component'))
        elif 'error' in want:
            assert 'error' in g, (req['op'], want['error'], g)        # (a Python exception other than the loader's own: refused, other words)
        else:
            assert g == want, (req['op'], req.get('component'))
    assert refused >= 8


@pytest.mark.skipif(not shutil.which('node'), reason='node is not in this image')
def test_node_side_can_still_ask_the_python_loader(tmp_path):
    """GSTARK_AA_LOADER=python: the child-process route the node side used before it had a loader of its own gives the same schema and
    the same AIR (kept for comparisons of the two)."""
    js = ("const a = require(process.argv[1]); const src = require('fs').readFileSync(process.argv[2], 'utf8');"
          "const s = a.compile(src); const air = new a.AssemblyAir(s, 'chain', {});"
          "console.log(JSON.stringify({ modulus: String(s.modulus), exports: s.exports, degrees: air.constraintDegrees, ef: air.extensionFactor }));")
    out = []
    for loader in ('', 'python'):
        r = subprocess.run(['node', '-e', js, os.path.join(ROOT, 'js', 'air_assembly.js'), os.path.join(AA, 'cube_chain.aa')],
                           env=dict(os.environ, GSTARK_AA_LOADER=loader, GSTARK_ALLOW_TEST_DOUBLE='1', GSTARK_LIB=ORACLE_LIB), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert out[0] == out[1] and out[0]['degrees'] == [3]


@pytest.mark.skipif(not shutil.which('node'), reason='node is not in this image')
def test_javascript_loader_on_random_modules(tmp_path):
    """80 random modules (tests/aa_fuzz.py: constants of every type, cycles of every generator, a function, locals, expression trees with
    scalar-vector broadcasting, slices, matrix products, divisions by constants): `info` and `plan` of both loaders, object for object —
    the programs' code words, constant pools and scratch-register counts included.  (1 150 modules in the session that wrote the
    loader: 0 differences.)"""
    import random
    from aa_fuzz import P, gen_module
    from genstark_amd import aa_json
    rnd = random.Random(20261001)
    reqs = []
    for _ in range(80):
        src, registers = gen_module(rnd)
        reqs.append({'op': 'info', 'source': src, 'component': 'main', 'extensionFactor': 32})
        reqs.append({'op': 'plan', 'source': src, 'component': 'main', 'extensionFactor': 32, 'inputs': [], 'seed': [str(rnd.randrange(P)) for _ in range(registers)]})
    (tmp_path / 'reqs.json').write_text(json.dumps(reqs))
    js = ("const { handle } = require(process.argv[1]); const reqs = JSON.parse(require('fs').readFileSync(process.argv[2], 'utf8'));"
          "process.stdout.write(JSON.stringify(reqs.map(r => { try { return handle(r); } catch (e) { return { error: e.message }; } })));")
    r = subprocess.run(['node', '-e', js, os.path.join(ROOT, 'js', 'aa_loader.js'), str(tmp_path / 'reqs.json')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    answered = 0
    for req, g in zip(reqs, json.loads(r.stdout)):
        try:
            want = json.loads(json.dumps(aa_json.handle(req)))
            answered += 1
        except Exception as e:   # noqa: BLE001
            want = {'error': f'{type(e).__name__}: {e}'}
        assert g == want, req['source']
    assert answered >= 140
