"""tools/ntt_bench.py — quick NTT / hashing / inversion timings on the GPU box through the C ABI (development aid)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genstark_amd._abi import Backend  # noqa: E402
from genstark_amd.field import PrimeField  # noqa: E402
from genstark_amd.merkle import MerkleTree, createHash  # noqa: E402

be = Backend()
f = PrimeField(backend=be)
h = createHash('blake2s256', be)


def timeit(fn, reps=10):
    fn(); be.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    be.sync()
    return (time.perf_counter() - t) / reps * 1e3


for logn in (12, 16, 20, 22, 24):
    n = 1 << logn
    w = f.getRootOfUnity(n)
    roots = f.getPowerSeries(w, n)
    a = f.getPowerSeries(0x123456789abcdef123, n)
    out = f.newVector(n)
    args = (C.c_void_p(a.ptr), 1, n, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))
    ms = timeit(lambda: be.call('gs_eval_polys_at_roots', *args))
    print(f'ntt 2^{logn}: {ms:8.3f} ms  {n / ms / 1e6:8.2f} Gelem/s  {32 * n / ms / 1e6:8.1f} GB/s algorithmic')
    if logn >= 16:
        args2 = (C.c_void_p(a.ptr), 1, n // 16, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))
        ms = timeit(lambda: be.call('gs_eval_polys_at_roots', *args2))
        print(f'  lde x16 -> 2^{logn}: {ms:8.3f} ms')
    ms = timeit(lambda: be.call('gs_interpolate_roots', C.c_void_p(a.ptr), 1, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr)))
    print(f'  intt: {ms:8.3f} ms')
    ms = timeit(lambda: be.call('gs_vec_mul', C.c_void_p(a.ptr), C.c_void_p(roots.ptr), n, C.c_void_p(out.ptr)))
    print(f'  vec_mul: {ms:8.3f} ms  {48 * n / ms / 1e6:8.1f} GB/s')
    ms = timeit(lambda: be.call('gs_vec_inv', C.c_void_p(a.ptr), n, C.c_void_p(out.ptr)), reps=3)
    print(f'  vec_inv: {ms:8.3f} ms')
    ms = timeit(lambda: be.call('gs_power_series', (12345).to_bytes(16, 'little'), n, C.c_void_p(out.ptr)))
    print(f'  power_series: {ms:8.3f} ms')
    dg = h.mergeVectorRows([a])
    ms = timeit(lambda: h.mergeVectorRows([a]), reps=5)
    print(f'  leaf hash: {ms:8.3f} ms  {n / ms / 1e6:8.2f} Ghash/s')
    ms = timeit(lambda: MerkleTree.create(dg, h), reps=5)
    print(f'  merkle build: {ms:8.3f} ms')
    del roots, a, out, dg
