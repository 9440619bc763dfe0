"""Merkle commitment timing: the in-tree library's fused gs_merkle_commit_rows against gs_hash_merge_rows + gs_merkle_build of the
same library and (optionally) of another build.  usage: python tools/merkle_ab.py [other libgstark_hip.so]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genstark_amd._abi import Backend

libs = [('new', Backend())]
if len(sys.argv) > 1:
    libs.append(('old', Backend(lib_path=sys.argv[1])))
ALG = 1  # blake2s256
for logn, count in ((24, 1), (22, 4), (20, 4), (20, 6), (18, 4), (16, 4), (14, 4), (12, 4), (10, 4), (8, 4), (6, 4)):
    n = 1 << logn
    for name, be in libs:
        cols = [be.alloc(16 * n) for _ in range(count)]
        for p in cols:
            be.upload(p, os.urandom(16 * min(n, 1 << 16)) * max(1, n >> 16))
        leaves, nodes = be.alloc(32 * n), be.alloc(32 * n)
        arr = (C.c_void_p * count)(*cols)
        modes = ['split'] + (['fused'] if hasattr(be.lib, 'gs_merkle_commit_rows') else [])
        for mode in modes:
            def go():
                if mode == 'fused':
                    be.call('gs_merkle_commit_rows', ALG, arr, count, n, C.c_void_p(leaves), C.c_void_p(nodes))
                else:
                    be.call('gs_hash_merge_rows', ALG, arr, count, n, C.c_void_p(leaves))
                    be.call('gs_merkle_build', ALG, C.c_void_p(leaves), n, C.c_void_p(nodes))
            for _ in range(3): go()
            be.sync(); t0 = time.perf_counter()
            reps = 20
            for _ in range(reps): go()
            be.sync(); dt = (time.perf_counter() - t0) / reps
            comp = n * ((count + 3) // 4) + n - 1
            print(f'{name:3s} {mode:5s} 2^{logn} x {count} cols  {dt * 1e6:9.1f} us   {comp / dt / 1e9:6.2f} G compressions/s', flush=True)
        for p in cols + [leaves, nodes]:
            be.free(p)
