"""A/B of the matrix-core radix-256 passes (GSTARK_NTT_MFMA=1, tools/ntt_mfma.h) against the default kernels on the GPU: bytes for
2^16 and 2^24 (the sizes whose plans are made of radix-256 passes only) — forward, inverse, zero-extended, low-degree extension,
several rows — then timings.   usage: python tools/ntt_mfma_ab.py [24]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField

EXP = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ab', 'libgstark_hip_exp.so')   # tools/build_experiments.sh: the build with the A/B switches
be = Backend(lib_path=EXP); f = PrimeField(backend=be)
sizes = [16, 24] if (len(sys.argv) < 2 or int(sys.argv[1]) >= 24) else [16]


def fwd(a, rows, poly_len, w, n, out):
    be.call('gs_eval_polys_at_roots', C.c_void_p(a.ptr), rows, poly_len, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))


def inv(a, rows, w, n, out):
    be.call('gs_interpolate_roots', C.c_void_p(a.ptr), rows, w.to_bytes(16, 'little'), n, C.c_void_p(out.ptr))


bad = 0
for logn in sizes:
    n = 1 << logn
    w = f.getRootOfUnity(n)
    rows = 3 if logn <= 16 else 1
    a = f.getPowerSeries(0x123456789abcdef123 + logn, n * rows)
    for kind, plen in [('fwd', n), ('fwd', n // 2 + 3), ('fwd', n // 16), ('fwd', n // 16 - 5), ('fwd', n // 16 + 7), ('inv', n)]:
        outs = []
        for mf in ('1', '0'):
            os.environ['GSTARK_NTT_MFMA'] = mf
            out = f.newVector(n * rows)
            fwd(a, rows, plen, w, n, out) if kind == 'fwd' else inv(a, rows, w, n, out)
            be.sync()
            outs.append(out.toBuffer())
        if outs[0] != outs[1]:
            bad += 1
            diff = [i // 16 for i in range(0, len(outs[0]), 16) if outs[0][i:i + 16] != outs[1][i:i + 16]]
            print(f'MISMATCH logn={logn} {kind} len={plen} rows={rows}: {len(diff)} of {n * rows} elements differ, first {diff[:8]}')
        else:
            print(f'logn={logn} {kind} len={plen} rows={rows}: identical', flush=True)
print('mismatches:', bad)
for mf in ('1', '0'):
    os.environ['GSTARK_NTT_MFMA'] = mf
    for logn in sizes[::-1]:
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
            print(f'mfma={mf} 2^{logn} {kind:6s} {dt * 1e3:8.4f} ms  {n / dt / 1e9:7.2f} G el/s', flush=True)
sys.exit(1 if bad else 0)
