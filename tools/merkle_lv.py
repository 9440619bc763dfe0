"""tools/merkle_lv.py — how many node layers one k_merkle_fused<ALG, 0> launch should take (experiments build: tools/build_experiments.sh):
gs_merkle_build over n = 2^16 .. 2^23 digests with GSTARK_MERKLE_NODE_MIN_GROUPS = 0 (round 4's rule: as many layers as stay
above 2^15 digests, at most four) and = 15 .. 20 (at least 2^K threads per launch; the product keeps 2^17).  usage: python tools/merkle_lv.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genstark_amd._abi import Backend
be = Backend(device=0, lib_path=os.path.join(ROOT, 'tools', 'ab', 'libgstark_hip_exp.so'))
print('us per tree (blake2s256), columns: round-4 rule | min groups 2^15 .. 2^20 (the product: 2^17)')
for logn in range(16, 24):
    n = 1 << logn
    leaves, nodes = be.alloc(32 * n), be.alloc(32 * n)
    be.upload(leaves, os.urandom(32 * (1 << 12)) * (n >> 12))
    row = []
    for knob in (0, 15, 16, 17, 18, 19, 20):
        os.environ['GSTARK_MERKLE_NODE_MIN_GROUPS'] = str(knob)
        go = lambda: be.call('gs_merkle_build', 1, C.c_void_p(leaves), n, C.c_void_p(nodes))
        for _ in range(3): go()
        be.sync(); t0 = time.perf_counter()
        reps = 30
        for _ in range(reps): go()
        be.sync(); row.append((time.perf_counter() - t0) / reps * 1e6)
    print(f'2^{logn}: ' + ' | '.join(f'{v:7.1f}' for v in row), flush=True)
    be.free(leaves); be.free(nodes)
