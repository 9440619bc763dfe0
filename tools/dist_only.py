"""tools/dist_only.py — N proofs of ONE statement through the native distributed driver with G ranks sharing the box's one GPU (ranks =
threads, the thread communicator of the tests), or through the single-device driver (G = 0), and nothing else: a rocprofv3 target.
Every proof issues the same launches, so a kernel's time per proof is its total over the trace divided by the proofs run
(warm-up included; `proofs_run` is printed).  usage: python tools/dist_only.py [c4|c4long|c5] [G] [proofs=6] [full]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.prover import Prover

which = sys.argv[1] if len(sys.argv) > 1 else 'c4'
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
FULL = len(sys.argv) > 4 and sys.argv[4] == 'full'      # shard whatever the size (gs_comm::solo_below = 1)
WARM = 2


def statement(be):
    if which == 'c5':
        opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 64}
        p = ga.mimcProver(1 << 20, opts, backend=be)
        return p, [{'step': 0, 'register': 0, 'value': 3}], [], [3]
    from genstark_amd.poseidon import poseidon6x128_air
    from genstark_amd.field import PrimeField
    t4 = 1 << (20 if which == 'c4long' else 16)
    air = poseidon6x128_air(t4, 16, PrimeField(backend=be), segmented=True)
    seed = [[1 + s, 2, 3 + s, 4] for s in range(t4 // 64)]
    p = Prover(air, {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 24})
    return p, [{'step': 0, 'register': 0, 'value': 1}], [], p.pack_seed(seed)


if G == 0:
    be = Backend(device=0).jit()
    p, a, i, s = statement(be)
    for _ in range(WARM):
        data = p.prove_bytes(a, i, s)
    be.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        data = p.prove_bytes(a, i, s)
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f'{which} single-device driver: {dt:.3f} ms per proof, {len(data)} bytes; proofs_run {WARM + n}')
    for k, v in p.last_stats()['phases_ms'].items():
        print(f'  {v:8.3f}  {k}')
    sys.exit(0)

from dist_helpers import thread_comms
bes = [Backend(device=0).jit() for _ in range(G)]
sts = [statement(be) for be in bes]
comms, keep = thread_comms(bes[0], G)
for r in range(G):
    comms[r].solo_below = 1 if FULL else 0
outs = [None] * G


import ctypes as C
TURNS = os.environ.get('GSTARK_COMM_TAKE_TURNS') == '1'      # measuring mode of the thread communicator: one rank on the device at a time
if TURNS:
    keep.gs_threads_comm_begin.argtypes = [C.c_void_p]
    keep.gs_threads_comm_end.argtypes = [C.c_void_p, C.c_void_p]


def run(r, reps):
    p, a, i, s = sts[r]
    for _ in range(reps):
        if TURNS:
            keep.gs_threads_comm_begin(C.byref(comms[r]))
        outs[r] = p.prove_bytes(a, i, s, comm=comms[r])
        if TURNS:
            keep.gs_threads_comm_end(C.byref(comms[r]), bes[r].ctx)


for reps in (WARM, n):
    ths = [threading.Thread(target=run, args=(r, reps)) for r in range(G)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = (time.perf_counter() - t0) / reps * 1e3
assert all(o == outs[0] for o in outs)
print(f'{which} G={G}{" (sharded whatever the size)" if FULL else ""}{" (ranks take turns on the device)" if TURNS else ""}: {dt:.3f} ms per proof with the ranks sharing one GPU, '
      f'{len(outs[0])} bytes; proofs_run {WARM + n}')
for k, v in sts[0][0].last_stats()['phases_ms'].items():
    print(f'  {v:8.3f}  {k}')
cs = sts[0][0].last_collectives()
print('  collectives of rank 0:', [(c['label'], c['kind'], c['bytes']) for c in cs])
