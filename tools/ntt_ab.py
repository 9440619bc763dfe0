"""A/B of the two NTT pass kernels of the 128-bit library on the GPU: bytes of the lazy-limb kernel (default) against the
canonical-limb kernel (GSTARK_NTT_LAZY=0) for every size 2^8..2^24, forward / inverse / zero-extended / multi-row, then timings.
usage: python tools/ntt_ab.py [max_log]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField

EXP = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ab', 'libgstark_hip_exp.so')   # tools/build_experiments.sh: the build with the A/B switches
be = Backend(lib_path=EXP); f = PrimeField(backend=be)
max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 24


def run(fn, *args):
    be.call(fn, *args)


def fwd(a, rows, poly_len, w, n, out):
    run('gs_eval_polys_at_roots', C.c_void_p(a.ptr), rows, poly_len, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))


def inv(a, rows, w, n, out):
    run('gs_interpolate_roots', C.c_void_p(a.ptr), rows, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))


def download(v, count):
    return be.download(v.ptr, count) if hasattr(be, 'download') else v.toBuffer()


bad = 0
for logn in range(8, max_log + 1):
    n = 1 << logn
    w = f.getRootOfUnity(n)
    rows = 3 if logn <= 16 else 1
    a = f.getPowerSeries(0x123456789abcdef123 + logn, n * rows)
    cases = [('fwd', n), ('fwd', n // 2 + 3), ('fwd', max(9, n // 16)), ('fwd', max(9, n // 16 - 5)), ('inv', n)]
    for kind, plen in cases:
        outs = []
        for lazy in ('1', '0'):
            os.environ['GSTARK_NTT_LAZY'] = lazy
            out = f.newVector(n * rows)
            if kind == 'fwd':
                fwd(a, rows, plen, w, n, out)
            else:
                inv(a, rows, w, n, out)
            be.sync()
            outs.append(out.toBuffer())
        ok = outs[0] == outs[1]
        if not ok:
            bad += 1
            k = next(i for i in range(0, len(outs[0]), 16) if outs[0][i:i + 16] != outs[1][i:i + 16]) // 16
            print(f'MISMATCH logn={logn} {kind} len={plen} rows={rows} first bad element {k}')
    print(f'logn {logn}: {"ok" if not bad else "BAD so far: %d" % bad}', flush=True)

print('mismatches:', bad)
for lazy in ('1', '0'):
    os.environ['GSTARK_NTT_LAZY'] = lazy
    for logn in (24, 22, 20, 16):
        if logn > max_log:
            continue
        n = 1 << logn
        w = f.getRootOfUnity(n)
        a = f.getPowerSeries(0x123456789abcdef123, n); out = f.newVector(n)
        for kind in ('fwd', 'lde16', 'inv'):
            def go():
                if kind == 'fwd': fwd(a, 1, n, w, n, out)
                elif kind == 'lde16': fwd(a, 1, n // 16, w, n, out)
                else: inv(a, 1, w, n, out)
            for _ in range(3): go()
            be.sync(); t0 = time.perf_counter()
            reps = 20
            for _ in range(reps): go()
            be.sync(); dt = (time.perf_counter() - t0) / reps
            print(f'lazy={lazy} 2^{logn} {kind:6s} {dt * 1e3:8.4f} ms  {n / dt / 1e9:7.2f} G el/s', flush=True)
sys.exit(1 if bad else 0)
