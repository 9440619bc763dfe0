"""tools/soak_generic.py [count] [seed] — randomized differential soak of the register-machine AIR path on the GPU: Rescue 4x128 /
Poseidon 6x128, plain or segmented, random sizes, seeds and options; the native driver on HIP (interpreted AND compiled programs), the
Python mirror on HIP and the Python mirror on the CPU oracle must produce the same proof bytes, and the proof must verify."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from genstark_amd import poseidon
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField
from genstark_amd.native import NativeProver
from genstark_amd.rescue import rescue4x128_air
from genstark_amd._mirror.stark import Stark

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
hip, hip_jit = Backend(device=0), Backend(device=0)
hip_jit.jit()
oracle = Backend(lib_path=os.path.join(root, 'oracle', 'liboracle.so'), allow_test_double=True)
bad, t0 = 0, time.time()
for case in range(count):
    kind = rng.choice(['rescue', 'poseidon'])
    per = 32 if kind == 'rescue' else 64
    segmented = rng.random() < 0.6
    steps = per << (rng.randrange(0, 7) if segmented else rng.randrange(0, 3))
    opts = {'hashAlgorithm': rng.choice(['blake2s256', 'sha256']), 'extensionFactor': rng.choice([16, 32]),
            'exeQueryCount': rng.randrange(1, 80), 'friQueryCount': rng.randrange(1, 40)}
    if kind == 'rescue':
        seed = [[rng.randrange(1 << 128), rng.randrange(1 << 128)] for _ in range(steps // per)] if segmented else [rng.randrange(1 << 128), rng.randrange(1 << 128)]
    else:
        seed = [[rng.randrange(1 << 128) for _ in range(4)] for _ in range(steps // per)] if segmented else [rng.randrange(1 << 128) for _ in range(4)]
    datas, assertions = [], None
    try:
        for be, native in ((hip, True), (hip_jit, True), (hip, False), (oracle, False)):
            f = PrimeField(backend=be)
            air = rescue4x128_air(steps, opts['extensionFactor'], f, segmented=segmented) if kind == 'rescue' else \
                poseidon.poseidon6x128_air(steps, opts['extensionFactor'], f, segmented=segmented)
            stark = Stark(air, opts)
            if assertions is None:
                full = air.hostTrace(seed)
                picks = sorted(set([per - 1, steps - 1] + [rng.randrange(steps) for _ in range(rng.randrange(0, 3))]))
                assertions = [{'step': s, 'register': rng.randrange(air.traceRegisterCount), 'value': None} for s in picks]
                for a in assertions:
                    a['value'] = full[a['step']][a['register']]
            datas.append(NativeProver(stark).prove_bytes(assertions, [], seed) if native else stark.serialize(stark.prove(assertions, [], seed)))
        ok = datas[0] == datas[1] == datas[2] == datas[3]
        if ok:
            ok = stark.verify(assertions, stark.parse(datas[0])) is True
    except Exception as e:
        ok, datas = False, [repr(e)]
    bad += 0 if ok else 1
    print(f'{case:3d} {kind:8s} steps={steps:5d} {"segmented" if segmented else "plain    "} E={opts["extensionFactor"]} {opts["hashAlgorithm"]:10s} exe={opts["exeQueryCount"]:2d} '
          f'fri={opts["friQueryCount"]:2d} bytes={len(datas[0]) if ok else datas[0][:100]} {"ok" if ok else "MISMATCH"}', flush=True)
print(f'{count} cases, {bad} failures, {time.time() - t0:.1f} s')
sys.exit(1 if bad else 0)
