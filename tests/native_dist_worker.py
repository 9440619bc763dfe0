"""Worker of tests/test_native_dist.py::test_native_dist_gloo: one rank (= one process) of ONE proof through the native driver's
distributed mode, collectives over torch.distributed (gloo) via genstark_amd.comm.TorchComm.  GSTARK_TEST_LIB = the oracle double
(CPU tier); without it the HIP library with every rank on cuda:0 (ranks sharing the test box's single GPU)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch.distributed as dist

import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.comm import TorchComm
from genstark_amd.native import NativeProver


def main():
    name = sys.argv[1]
    lib = os.environ.get('GSTARK_TEST_LIB')
    backend = Backend(lib_path=lib, allow_test_double=True) if lib else Backend(device=0)
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    if name.startswith('mimc'):
        steps = 1 << int(name[4:])
        opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 24}
        stark = ga.instantiateMimc(steps, opts, backend=backend)
        controls = ga.runMimc(stark.air.field, steps, stark.air.roundConstants, 3)
        assertions, seed = [{'step': 0, 'register': 0, 'value': controls[0]}, {'step': steps - 1, 'register': 0, 'value': controls[-1]}], [3]
    else:
        import generic_cases as gc
        _, stark, seed, assertions = gc.build(name, backend)
    nat = NativeProver(stark)
    want = nat.prove_bytes(assertions, [], seed)
    comm = TorchComm(backend)
    comm.comm.solo_below = int(os.environ.get('GSTARK_TEST_SOLO_BELOW', '1'))      # 1: shard whatever the size; 0: the driver's default
    got = nat.prove_bytes(assertions, [], seed, comm=comm.comm)
    assert comm.error is None, comm.error
    assert got == want, f'rank {rank}: distributed proof differs from the single-device proof'
    colls = nat.last_collectives()
    assert colls and all(c['bytes'] > 0 for c in colls)
    digests = [None] * world
    dist.all_gather_object(digests, hashlib.sha256(got).hexdigest())
    assert len(set(digests)) == 1
    dist.barrier()
    dist.destroy_process_group()
    print(f'rank {rank}/{world} OK case={name} collectives={len(colls)} sha256={digests[0][:16]}')


if __name__ == '__main__':
    main()
