"""tools/ntt_only.py [log n = 24] [library = the product one] — a few 2^24-point NTTs and nothing else (target for rocprofv3 passes)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genstark_amd._abi import Backend  # noqa: E402
from genstark_amd.field import PrimeField  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
be = Backend(lib_path=sys.argv[2]) if len(sys.argv) > 2 else Backend()
f = PrimeField(backend=be)
n = 1 << logn
w = f.getRootOfUnity(n)
a = f.getPowerSeries(0x123456789abcdef123, n)
out = f.newVector(n)
for _ in range(4):
    be.call('gs_eval_polys_at_roots', C.c_void_p(a.ptr), 1, n, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))
be.sync()
print('done')
