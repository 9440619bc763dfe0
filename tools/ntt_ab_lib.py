"""A/B of two builds of the 128-bit library on the GPU: bytes of every NTT shape 2^8..2^max must agree, then timings of both.
usage: python tools/ntt_ab_lib.py <other libgstark_hip.so> [max_log]   (the first library is the in-tree build)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField

other = sys.argv[1]
max_log = int(sys.argv[2]) if len(sys.argv) > 2 else 24
libs = [('new', Backend()), ('old', Backend(lib_path=other))]
fields = [PrimeField(backend=be) for _, be in libs]


def fwd(be, a, rows, poly_len, w, n, out):
    be.call('gs_eval_polys_at_roots', C.c_void_p(a.ptr), rows, poly_len, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))


def inv(be, a, rows, w, n, out):
    be.call('gs_interpolate_roots', C.c_void_p(a.ptr), rows, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))


bad = 0
for logn in range(8, max_log + 1):
    n = 1 << logn
    rows = 3 if logn <= 16 else 1
    cases = [('fwd', n), ('fwd', n // 2 + 3), ('fwd', max(9, n // 16)), ('fwd', max(9, n // 16 - 5)), ('inv', n)]
    for kind, plen in cases:
        outs = []
        for (name, be), f in zip(libs, fields):
            w = f.getRootOfUnity(n)
            a = f.getPowerSeries(0x123456789abcdef123 + logn, n * rows)
            out = f.newVector(n * rows)
            if kind == 'fwd':
                fwd(be, a, rows, plen, w, n, out)
            else:
                inv(be, a, rows, w, n, out)
            be.sync()
            outs.append(out.toBuffer())
            del a, out
        if outs[0] != outs[1]:
            bad += 1
            k = next(i for i in range(0, len(outs[0]), 16) if outs[0][i:i + 16] != outs[1][i:i + 16]) // 16
            print(f'MISMATCH logn={logn} {kind} len={plen} rows={rows} first bad element {k}')
    print(f'logn {logn}: {"ok" if not bad else "BAD so far: %d" % bad}', flush=True)
print('mismatches:', bad)
for (name, be), f in zip(libs, fields):
    for logn in (24, 22, 21, 20, 16):
        if logn > max_log:
            continue
        n = 1 << logn
        w = f.getRootOfUnity(n)
        a = f.getPowerSeries(0x123456789abcdef123, n); out = f.newVector(n)
        for kind in ('fwd', 'lde16', 'inv'):
            def go():
                if kind == 'fwd': fwd(be, a, 1, n, w, n, out)
                elif kind == 'lde16': fwd(be, a, 1, n // 16, w, n, out)
                else: inv(be, a, 1, w, n, out)
            for _ in range(3): go()
            be.sync(); t0 = time.perf_counter()
            reps = 30
            for _ in range(reps): go()
            be.sync(); dt = (time.perf_counter() - t0) / reps
            print(f'{name} 2^{logn} {kind:6s} {dt * 1e3:8.4f} ms  {n / dt / 1e9:7.2f} G el/s', flush=True)
        del a, out
sys.exit(1 if bad else 0)
