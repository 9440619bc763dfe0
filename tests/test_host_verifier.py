"""Verification without any backend (genstark_amd/hostfield.py): proofs produced on the device-side field — golden MiMC proofs,
generic AIRs with static / secret registers — must verify with a Stark built on HostField (host integers + hashlib only), and
tampered proofs / wrong assertions must be rejected exactly as with the device-side field."""
import json
import os

import pytest

import genstark_amd as ga
from genstark_amd.air import MimcAir
from genstark_amd.errors import StarkError
from genstark_amd.hostfield import HostField
from genstark_amd._mirror.stark import Stark

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'oracle_proofs.json')) as f:
    GOLDEN = json.load(f)


def test_host_verifier_constructs_without_any_library(monkeypatch):
    import genstark_amd._abi as abi
    monkeypatch.setattr(abi, 'load_library', lambda *a, **k: (_ for _ in ()).throw(AssertionError('the verifier must not load a library')))
    stark = Stark(MimcAir(64, 16, HostField()), {'hashAlgorithm': 'sha256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 24})
    assert stark.securityLevel > 0
    with pytest.raises(AttributeError, match='prover member'):
        stark.air.field.evalPolysAtRoots


@pytest.mark.parametrize('case', GOLDEN, ids=[c['name'] for c in GOLDEN])
def test_host_verifier_accepts_golden_proofs(case, oracle_backend):
    options = {'hashAlgorithm': case['hash_algorithm'], 'extensionFactor': case['extension_factor'],
               'exeQueryCount': case['exe_query_count'], 'friQueryCount': case['fri_query_count']}
    assertions = [{'step': a['step'], 'register': a['register'], 'value': int(a['value'])} for a in case['assertions']]
    prover = ga.instantiateMimc(case['steps'], options, None, backend=oracle_backend)
    data = prover.serialize(prover.prove(assertions, [], [case['seed']]))
    verifier = Stark(MimcAir(case['steps'], case['extension_factor'], HostField()), options)
    assert verifier.verify(assertions, verifier.parse(data))
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 1
    with pytest.raises((StarkError, ValueError, IndexError)):
        verifier.verify(assertions, verifier.parse(bytes(bad)))
    with pytest.raises(StarkError):
        verifier.verify([dict(assertions[0], value=(assertions[0]['value'] + 1) % ga.MODULUS)] + assertions[1:], verifier.parse(data))


def test_host_verifier_generic_airs(oracle_backend):
    """Rescue, Poseidon (single chain and segmented), the degree-5 AIR, Foo (no FRI layers), secret registers and the lib128
    Merkle-path AIR: the same AIR definition instantiated over HostField verifies what the device-side instance proved."""
    from test_native_prover import generic_cases
    from genstark_amd.field import PrimeField
    from genstark_amd import lib128, poseidon
    from genstark_amd.rescue import rescue4x128_air
    from test_generic_air import POSEIDON_OPTS, RESCUE_OPTS, degree5_air, foo_air, secret_air
    from test_lib128 import OPTS as LIB_OPTS, merkle_case
    h = HostField()
    tree, leaf, nodes, bits = merkle_case(PrimeField(backend=oracle_backend), 4, 5)
    host_airs = {
        'rescue': (rescue4x128_air(64, 16, h), RESCUE_OPTS), 'poseidon': (poseidon.poseidon6x128_air(128, 16, h), POSEIDON_OPTS),
        'rescue-segmented': (rescue4x128_air(128, 16, h, segmented=True), RESCUE_OPTS),
        'poseidon-segmented': (poseidon.poseidon6x128_air(128, 16, h, segmented=True), POSEIDON_OPTS),
        'degree5': (degree5_air(h, 64), {'hashAlgorithm': 'sha256', 'extensionFactor': 16, 'exeQueryCount': 40, 'friQueryCount': 20}),
        'foo': (foo_air(h), None), 'lib128-hash': (lib128.compute_poseidon_hash_air(h, 2), LIB_OPTS),
        'lib128-merkle': (lib128.compute_merkle_root_air(h, bits), LIB_OPTS),
        'secret-registers': (secret_air(h, 64), {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 30, 'friQueryCount': 20}),
    }
    seen = set()
    for name, stark, seed, trace, points, *rest in generic_cases(oracle_backend):
        inputs = rest[0] if rest else []
        assertions = [{'step': s, 'register': r, 'value': trace[s][r]} for s, r in points]
        data = stark.serialize(stark.prove(assertions, inputs, seed))
        air, opts = host_airs[name]
        verifier = Stark(air, opts)
        assert verifier.verify(assertions, verifier.parse(data)), name
        with pytest.raises(StarkError):
            verifier.verify([dict(assertions[0], value=(assertions[0]['value'] + 1) % ga.MODULUS)] + assertions[1:], verifier.parse(data))
        seen.add(name)
    assert seen == set(host_airs)
