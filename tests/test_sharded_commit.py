"""The multi-GPU path (genstark_amd/sharded.py: register-sharded NTTs -> one exchange -> index-sharded hashing and
Merkle subtrees -> all-gather of sub-roots) on CPU: world_size 2 and 4 over gloo against the oracle double, checked
against the single-device root and leaf digests.  Shapes follow BASELINE configs[3] (6 registers) and the 12-register
Poseidon Merkle-proof AIR (README.md:217)."""
import os
import subprocess
import sys

import pytest

from conftest import ORACLE_LIB, ROOT


def launch(world, registers, log_t, ef, alg, port, env_extra=None, mode='registers'):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', **(env_extra or {}))
    worker = os.path.join(ROOT, 'tests', 'sharded_worker.py')
    if world == 1:
        cmd = [sys.executable, worker]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
               '--master-port', str(port), worker]
    return subprocess.run(cmd + [str(registers), str(log_t), str(ef), alg, mode], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize('world,registers,log_t,ef,alg', [(1, 6, 6, 16, 'blake2s256'), (2, 6, 6, 16, 'blake2s256'),
                                                          (4, 6, 5, 16, 'sha256'), (2, 12, 5, 8, 'blake2s256'), (4, 2, 5, 16, 'blake2s256')])
def test_sharded_commit_gloo(oracle_backend, world, registers, log_t, ef, alg):
    r = launch(world, registers, log_t, ef, alg, 29540 + world + registers, {'GSTARK_TEST_LIB': ORACLE_LIB})
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'sharded commit OK' in r.stdout


@pytest.mark.parametrize('world,registers,log_t,ef', [(1, 1, 6, 16), (2, 1, 6, 16), (4, 1, 6, 16), (8, 1, 6, 16), (4, 6, 5, 16), (2, 3, 5, 8)])
def test_domain_sharded_commit_gloo(oracle_backend, world, registers, log_t, ef):
    """Single-register (MiMC) and multi-register traces, every rank busy: coset NTTs without communication, one exchange."""
    r = launch(world, registers, log_t, ef, 'blake2s256', 29600 + world * 7 + registers, {'GSTARK_TEST_LIB': ORACLE_LIB}, mode='domain')
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'sharded commit OK: mode=domain' in r.stdout


@pytest.mark.gpu
def test_domain_sharded_commit_single_gpu():
    r = launch(1, 1, 14, 16, 'blake2s256', 0, mode='domain')
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'sharded commit OK: mode=domain' in r.stdout


@pytest.mark.gpu
def test_sharded_commit_single_gpu():
    """world_size 1 on the HIP backend (the GPU box has one device; N > 1 is exercised over gloo above)."""
    r = launch(1, 6, 10, 16, 'blake2s256', 0)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'sharded commit OK' in r.stdout
