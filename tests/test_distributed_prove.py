"""ONE proof across several ranks (genstark_amd/distributed.py): every rank runs the unchanged prover
over distributed vectors and must produce the byte-identical proof of the single-device prover — MiMC (1 register), Rescue
(4 registers) and Poseidon (6 registers, BASELINE configs[3]); world sizes 1..8 over gloo with the oracle's implementation of
the C ABI on CPU, and two ranks sharing the one GPU of the box with the HIP kernels."""
import os
import subprocess
import sys

import pytest

from conftest import ORACLE_LIB, ROOT


def launch(world, kind, log_t, ef, alg, port, env_extra=None):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', **(env_extra or {}))
    worker = os.path.join(ROOT, 'tests', 'dist_prove_worker.py')
    if world == 1:
        cmd = [sys.executable, worker]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
               '--master-port', str(port), worker]
    return subprocess.run(cmd + [kind, str(log_t), str(ef), alg], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)


@pytest.mark.parametrize('world,kind,log_t,ef,alg', [
    (2, 'mimc', 8, 16, 'blake2s256'),
    (4, 'mimc', 9, 8, 'sha256'),            # extension factor 8: the composition polynomial is longer than a rank's coset
    (8, 'mimc', 7, 16, 'blake2s256'),       # odd log2 of the domain: the last FRI trees are too small to shard
    (4, 'poseidon', 7, 16, 'blake2s256'),   # 6 registers, degree-6 constraints
    (2, 'rescue', 7, 16, 'blake2s256'),
    (8, 'poseidon', 11, 16, 'blake2s256'),  # BASELINE configs[3]'s AIR on 8 ranks (its full size, 2^16 steps, is the product driver's: tests/test_native_dist.py::test_native_dist_c4_full_size)
])
def test_distributed_prove_gloo(oracle_backend, world, kind, log_t, ef, alg):
    r = launch(world, kind, log_t, ef, alg, 29800 + 10 * world + log_t, {'GSTARK_TEST_LIB': ORACLE_LIB})
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count(' OK kind=') == world


def test_batch_proof_plan_matches_library(oracle_backend):
    """The host plan the sharded tree uses must select exactly the digests gs_merkle_prove_batch returns."""
    import random
    from genstark_amd._mirror.distributed import batch_proof_plan
    from genstark_amd.field import PrimeField
    from genstark_amd.merkle import MerkleTree, createHash
    f = PrimeField(backend=oracle_backend)
    h = createHash('blake2s256', oracle_backend)
    rng = random.Random(9)
    for n in (4, 64, 512):
        leaves = h.mergeVectorRows([f.getPowerSeries(77, n)])
        tree = MerkleTree.create(leaves, h)
        lraw, nraw = leaves.toBuffer(), tree.nodes.toBuffer()
        for count in (1, 2, min(n, 17)):
            idx = rng.sample(range(n), count)
            proof = tree.proveBatch(idx)
            cols = batch_proof_plan(n, idx)
            want = [[(lraw if kind == 'leaf' else nraw)[32 * i:32 * i + 32] for kind, i in col] for col in cols]
            assert proof['nodes'] == want


@pytest.mark.gpu
@pytest.mark.parametrize('world,kind,log_t', [(2, 'mimc', 12), (4, 'poseidon', 10)])
def test_distributed_prove_hip_ranks_sharing_one_gpu(world, kind, log_t):
    r = launch(world, kind, log_t, 16, 'blake2s256', 29900 + world, {'GSTARK_SHARE_GPU': '1'})
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count(' OK kind=') == world
