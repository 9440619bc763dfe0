"""tests/golden/config_digests.json anchors the proof digests of the bench line (`configs[].proof_sha256`): the statements of
tools/config_runs.py — every BASELINE.json configuration as bench.py proves it — proved once on the CPU oracle
(tests/golden/make_config_digests.py).  CPU tier: the 2^6 .. 2^16-step statements reproduce on the oracle (C4 — ten seconds of interpreted Poseidon — in the GPU tier only), and the committed bench lines
of this round carry exactly the committed digests.  GPU tier: the product entry on the HIP library reproduces every one of them,
the two 2^20-step statements included."""
import glob
import hashlib
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
WANT = {r['name']: r for r in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'config_digests.json')))}


def test_every_configuration_is_anchored():
    import config_runs
    assert list(WANT) == list(config_runs.CONFIGS)
    assert all(len(r['proof_sha256']) == 64 and r['proof_bytes'] > 0 for r in WANT.values())


@pytest.mark.parametrize('name', ['C1_foo', 'C2_E8', 'C2_E16', 'C3', 'X_shaped'])
def test_digest_on_the_oracle(oracle_backend, name):
    import make_config_digests
    got = make_config_digests.digest(name)
    assert (got['proof_bytes'], got['proof_sha256']) == (WANT[name]['proof_bytes'], WANT[name]['proof_sha256'])


def test_committed_bench_lines_carry_the_anchored_digests():
    """every bench line of round 6 onwards under profiles/ (the builder's copies of what the driver reads)"""
    seen = 0
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0[6-9]_*bench_line*.json'))):
        line = json.loads([l for l in open(path).read().splitlines() if l.startswith('{')][-1])
        for c in line.get('configs') or []:
            assert c['proof_sha256'] == WANT[c['name']]['proof_sha256'] and c['proof_bytes'] == WANT[c['name']]['proof_bytes'], (path, c['name'])
            seen += 1
    assert seen >= 7


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(WANT))
def test_digest_on_hip(hip_backend, name):
    import config_runs
    from genstark_amd._abi import Backend

    def hip(modulus, jit):
        be = Backend(device=0) if modulus is None else Backend(device=0, modulus=modulus)
        return be.jit() if jit else be
    be, p, a, inputs, seed, public = config_runs.statement(name, hip)
    data = p.prove_bytes(a, inputs, seed)
    assert (len(data), hashlib.sha256(data).hexdigest()) == (WANT[name]['proof_bytes'], WANT[name]['proof_sha256'])
    if name != 'C4_long':
        assert p.verify_native(a, data, public) is True
