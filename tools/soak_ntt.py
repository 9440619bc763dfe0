"""tools/soak_ntt.py [count] [seed] — randomized soak of the NTT entry points on the GPU: random size 2^8..2^22, rows, input length
(zero-extension at arbitrary lengths) and direction; the default kernels against the canonical-limb kernels (GSTARK_NTT_LAZY=0), against
the opt-in matrix-core passes where they apply (GSTARK_NTT_MFMA=1), and against the CPU oracle up to 2^16 points."""
import ctypes as C, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField

count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
hip = Backend(device=0, lib_path=os.path.join(root, 'tools', 'ab', 'libgstark_hip_exp.so'))   # tools/build_experiments.sh: the build with the A/B switches
orc = Backend(lib_path=os.path.join(root, 'oracle', 'liboracle.so'), allow_test_double=True)
fh, fo = PrimeField(backend=hip), PrimeField(backend=orc)
bad, t0 = 0, time.time()
for case in range(count):
    logn = rng.choice(list(range(8, 21)) + [16, 16, 21, 22])
    n = 1 << logn
    rows = rng.choice([1, 1, 2, 3]) if logn <= 18 else 1
    inverse = rng.random() < 0.3
    plen = n if inverse else rng.choice([n, n, rng.randrange(1, n + 1), max(1, n // 16), max(1, n // 16 + rng.randrange(-3, 4)), rng.randrange(1, 64)])
    w = fh.getRootOfUnity(n)
    wb = w.to_bytes(16, 'little')
    src = fh.getPowerSeries(rng.randrange(2, 1 << 127), plen * rows)
    raw = src.toBuffer()

    def run(be, f, v):
        out = f.newVector(n * rows)
        if inverse:
            be.call('gs_interpolate_roots', C.c_void_p(v.ptr), rows, wb, n, C.c_void_p(out.ptr))
        else:
            be.call('gs_eval_polys_at_roots', C.c_void_p(v.ptr), rows, plen, wb, n, C.c_void_p(out.ptr))
        return out.toBuffer()
    os.environ['GSTARK_NTT_LAZY'], os.environ['GSTARK_NTT_MFMA'] = '1', '0'
    ref = run(hip, fh, src)
    checks = []
    os.environ['GSTARK_NTT_LAZY'] = '0'
    checks.append(('canonical kernels', run(hip, fh, src)))
    os.environ['GSTARK_NTT_LAZY'], os.environ['GSTARK_NTT_MFMA'] = '1', '1'
    checks.append(('matrix-core passes', run(hip, fh, src)))
    os.environ['GSTARK_NTT_MFMA'] = '0'
    if logn <= 16:
        vo = fo.newVector(plen * rows)
        orc.upload(vo.ptr, raw)
        checks.append(('oracle', run(orc, fo, vo)))
    wrong = [name for name, got in checks if got != ref]
    bad += 1 if wrong else 0
    print(f'{case:3d} n=2^{logn} rows={rows} {"inv" if inverse else "fwd"} len={plen} vs {len(checks)} paths: {"ok" if not wrong else "MISMATCH " + ", ".join(wrong)}', flush=True)
print(f'{count} cases, {bad} failures, {time.time() - t0:.1f} s')
sys.exit(1 if bad else 0)
