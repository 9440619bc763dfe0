import ctypes as C, os, sys, time, random
sys.path.insert(0, '/root/repo')
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField
be = Backend(); f = PrimeField(backend=be)
P = f.modulus
for logn in (14, 17, 20, 24):
    n = 1 << logn
    a = f.getPowerSeries(0x123456789abcdef123, n); out = f.newVector(n)
    args = (C.c_void_p(a.ptr), n, C.c_void_p(out.ptr))
    for _ in range(3): be.call('gs_vec_inv', *args)
    be.sync(); t0 = time.perf_counter()
    for _ in range(20): be.call('gs_vec_inv', *args)
    be.sync(); dt = (time.perf_counter() - t0) / 20
    # check a few
    av, ov = a.toBuffer(), out.toBuffer()
    for i in [0, 1, n - 1, n // 3]:
        x = int.from_bytes(av[16*i:16*i+16], 'little'); y = int.from_bytes(ov[16*i:16*i+16], 'little')
        assert x * y % P == 1, (logn, i)
    print(f'vec_inv 2^{logn}: {dt*1e3:.4f} ms  {32*n/dt/1e9:.0f} GB/s algorithmic')
