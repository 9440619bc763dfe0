"""tools/dist_phases.py — the native distributed driver with G ranks SHARING the box's one GPU (ranks = threads, thread communicator of
the tests): wall time per proof, rank 0's phase clock and the collectives of one proof, for C4 (Poseidon 2^16 steps as 1 024 chains)
and C5 (MiMC 2^20).  On one GPU the ranks' kernels serialise, so the wall time is the SUM of the ranks' device work + exchanges: it
shows how much total work the distributed form adds, not a speed-up.  usage: python tools/dist_phases.py [c4|c4long|c5] [full] [G ...]   (full: shard whatever the size)   (c4long: the same AIR at 2^20 steps; programs compiled)"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.native import NativeProver
from genstark_amd.prover import Prover
from dist_helpers import thread_comms

which = sys.argv[1] if len(sys.argv) > 1 else 'c4'
FULL = 'full' in sys.argv[2:]          # shard whatever the size (gs_comm::solo_below = 1); default: the driver decides (small statements: rank 0 alone)
Gs = [int(x) for x in sys.argv[2:] if x != 'full'] or [1, 2, 4, 8]


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


for G in Gs:
    bes = [Backend(device=0).jit() for _ in range(G)]
    sts = [statement(be) for be in bes]
    comms, keep = thread_comms(bes[0], G)
    for r in range(G):
        comms[r].solo_below = 1 if FULL else 0
    single = sts[0][0].prove_bytes(*sts[0][1:])
    outs = [None] * G

    def run(r, reps):
        p, a, i, s = sts[r]
        for _ in range(reps):
            outs[r] = p.prove_bytes(a, i, s, comm=comms[r])
    for reps in (1, 5):
        ths = [threading.Thread(target=run, args=(r, reps)) for r in range(G)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = (time.perf_counter() - t0) / reps * 1e3
    assert all(o == single for o in outs)
    t0 = time.perf_counter()
    for _ in range(5):
        sts[0][0].prove_bytes(*sts[0][1:])
    one = (time.perf_counter() - t0) / 5 * 1e3
    st = sts[0][0]
    run(0, 0)
    print(f'== {which} G={G}{" (sharded whatever the size)" if FULL else ""}: {dt:.3f} ms per proof with the ranks sharing one GPU (single-device driver on the same GPU: {one:.3f} ms); bytes equal')
    ths = [threading.Thread(target=run, args=(r, 1)) for r in range(1, G)]
    for t in ths: t.start()
    run(0, 1)
    for t in ths: t.join()
    print('   phases (rank 0):', {k: v for k, v in st.last_stats()['phases_ms'].items()})
    colls = st.last_collectives()
    agg = {}
    for c in colls:
        key = c['label'].split(':')[-1].strip() if 'tree' in c['label'] else c['label']
        a = agg.setdefault((key, c['kind']), [0, 0]); a[0] += 1; a[1] += c['bytes']
    print('   collectives:', len(colls), {f'{k[0]} ({k[1]})': f'{v[0]} x, {v[1]} B' for k, v in agg.items()})
    for be in bes: be.close()
