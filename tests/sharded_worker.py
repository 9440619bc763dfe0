"""Worker for tests/test_sharded_commit.py: one rank of the register-sharded trace commitment, on the backend given by
GSTARK_TEST_LIB (oracle double on CPU + gloo) or on the HIP library (GPU + nccl)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField
from genstark_amd.merkle import MerkleTree, createHash
from genstark_amd._mirror.sharded import domain_sharded_commit, owned_registers, sharded_commit

P = 2**128 - 9 * 2**32 + 1


def main():
    registers, log_t, ef, alg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else 'registers'
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    lib = os.environ.get('GSTARK_TEST_LIB')
    if lib:
        backend = Backend(lib_path=lib, allow_test_double=True)
        if world > 1:
            dist.init_process_group('gloo')
    else:
        local = int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(local)
        backend = Backend(device=local)
        if world > 1:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    field = PrimeField(backend=backend)
    h = createHash(alg, backend)
    t = 1 << log_t
    rng = random.Random(1234)
    traces = [[rng.randrange(P) for _ in range(t)] for _ in range(registers)]   # same on every rank
    n = t * ef
    w = field.getRootOfUnity(n)
    polys = field.interpolateRoots(field.getPowerSeries(field.exp(w, ef), t), field.newMatrixFrom(traces))
    if mode == 'domain':
        root, leaves, tree, cols = domain_sharded_commit(field, h, polys, ef)
    else:
        mine = owned_registers(registers, rank, world)
        root, leaves, tree, cols = sharded_commit(field, h, {r: field.newVectorFrom(traces[r]) for r in mine}, registers, t, ef)
    # single-device reference on this rank: all registers, one tree
    ev = field.evalPolysAtRoots(polys, field.getPowerSeries(w, n))
    ref_leaves = h.mergeVectorRows(field.matrixRowsToVectors(ev))
    ref_root = MerkleTree.create(ref_leaves, h).root
    assert root == ref_root, f'rank {rank}: sharded root differs from the single-device root'
    shard = n // world
    assert leaves.toBuffer() == ref_leaves.toBuffer(rank * shard, shard), f'rank {rank}: leaf digests of my rows differ'
    for r in range(registers):
        assert cols[r].toBuffer() == ev.row(r).toBuffer(rank * shard, shard), f'rank {rank}: evaluations of register {r} differ'
    if world > 1:
        roots = [None] * world
        dist.all_gather_object(roots, root.hex())
        assert len(set(roots)) == 1
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(f'sharded commit OK: mode={mode} R={registers} T=2^{log_t} E={ef} {alg} world={world} root={root.hex()[:16]}')


if __name__ == '__main__':
    main()
