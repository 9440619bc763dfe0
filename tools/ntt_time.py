import ctypes as C, os, sys, time
sys.path.insert(0, '/root/repo')
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField
be = Backend(); f = PrimeField(backend=be)
for logn in (24, 22, 20, 16):
    n = 1 << logn
    w = f.getRootOfUnity(n)
    a = f.getPowerSeries(0x123456789abcdef123, n); out = f.newVector(n)
    args = (C.c_void_p(a.ptr), 1, n, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))
    for _ in range(3): be.call('gs_eval_polys_at_roots', *args)
    be.sync(); t0 = time.perf_counter()
    reps = 20
    for _ in range(reps): be.call('gs_eval_polys_at_roots', *args)
    be.sync(); print(logn, round((time.perf_counter() - t0) / reps * 1e3, 4), 'ms')
