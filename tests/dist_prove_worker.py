"""Worker for tests/test_distributed_prove.py: one rank of ONE proof computed across `WORLD_SIZE` ranks through the
distributed-array facade (genstark_amd/distributed.py), on the backend given by GSTARK_TEST_LIB (oracle double on CPU + gloo)
or on the HIP library (GPU + nccl).  Every rank must end up with the proof bytes the single-device prover produces."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.air import MimcAir
from genstark_amd._mirror.distributed import DistField
from genstark_amd.field import PrimeField
from genstark_amd._mirror.stark import Stark


def build(kind, steps, ef, field):
    if kind == 'mimc':
        return MimcAir(steps, ef, field), [3]
    if kind == 'poseidon':
        from genstark_amd.poseidon import poseidon6x128_air
        return poseidon6x128_air(steps, ef, field), [1, 2, 3, 4]
    from genstark_amd.rescue import rescue4x128_air
    return rescue4x128_air(steps, ef, field), [42, 43]


def main():
    kind, log_t, ef, alg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    lib = os.environ.get('GSTARK_TEST_LIB')
    if lib:
        backend = Backend(lib_path=lib, allow_test_double=True)
        if world > 1:
            dist.init_process_group('gloo')
    else:
        share = os.environ.get('GSTARK_SHARE_GPU') == '1'      # all ranks on cuda:0, exchanges staged through gloo
        local = 0 if share else int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(local)
        backend = Backend(device=local)
        if world > 1:
            if share:
                dist.init_process_group('gloo')
            else:
                dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    steps = 1 << log_t
    opts = {'hashAlgorithm': alg, 'extensionFactor': ef, 'exeQueryCount': 40, 'friQueryCount': 24}
    # single-device prover on this rank (the expectation)
    ref_air, seed = build(kind, steps, ef, PrimeField(backend=backend))
    ref = Stark(ref_air, opts)
    trace = ref_air.initProvingContext([], seed).generateExecutionTrace()
    regs = ref_air.traceRegisterCount
    assertions = [{'step': 0, 'register': 0, 'value': trace.getValue(0, 0)},
                  {'step': steps - 1, 'register': regs - 1, 'value': trace.getValue(regs - 1, steps - 1)},
                  {'step': steps // 2, 'register': 0, 'value': trace.getValue(0, steps // 2)}]
    want = ref.serialize(ref.prove(assertions, [], seed))
    # the same proof across the ranks
    dfield = DistField(backend, steps * ef)
    dair, _ = build(kind, steps, ef, dfield)
    dstark = Stark(dair, opts)
    proof = dstark.prove(assertions, [], seed)
    got = dstark.serialize(proof)
    assert got == want, f'rank {rank}: distributed proof differs from the single-device proof'
    assert ref.verify(assertions, ref.parse(got))
    if world > 1:
        digests = [None] * world
        dist.all_gather_object(digests, hashlib.sha256(got).hexdigest())
        assert len(set(digests)) == 1
        dist.barrier()
        dist.destroy_process_group()
    print(f'rank {rank}/{world} OK kind={kind} steps=2^{log_t} sha256={hashlib.sha256(got).hexdigest()[:16]}')


if __name__ == '__main__':
    main()
