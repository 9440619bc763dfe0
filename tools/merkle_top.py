"""tools/merkle_top.py — what the TOP of a Merkle tree costs: gs_merkle_build over n digests for n = 2^6 .. 2^18 (blake2s256), back to back
on one stream (a tree's launches are dependent anyway): microseconds per tree and per level.  The small end is a chain of dependent
compressions, not throughput: this is the part of every commitment that more bandwidth cannot shorten.
usage: python tools/merkle_top.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genstark_amd._abi import Backend
be = Backend()
for alg, name in ((1, 'blake2s256'), (0, 'sha256')):
    for logn in range(6, 19):
        n = 1 << logn
        leaves, nodes = be.alloc(32 * n), be.alloc(32 * n)
        be.upload(leaves, os.urandom(32 * min(n, 1 << 12)) * max(1, n >> 12))
        go = lambda: be.call('gs_merkle_build', alg, C.c_void_p(leaves), n, C.c_void_p(nodes))
        for _ in range(5): go()
        be.sync(); t0 = time.perf_counter()
        reps = 50
        for _ in range(reps): go()
        be.sync(); dt = (time.perf_counter() - t0) / reps
        print(f'{name:10s} 2^{logn:<2d} digests: {dt * 1e6:7.1f} us per tree, {dt * 1e6 / logn:5.2f} us per level', flush=True)
        be.free(leaves); be.free(nodes)
