"""tools/soak_prove.py [count] [seed] — randomized differential soak of the whole MiMC prove() path on the GPU: for random
(steps, extension factor, query counts, hash, seed, assertion set) the native driver on HIP, the Python mirror on HIP and the Python
mirror on the CPU oracle must produce the same proof bytes, and the proof must verify.  Prints one line per case and a summary."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.native import NativeProver

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20260927)
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
hip = Backend(device=0)
oracle = Backend(lib_path=os.path.join(root, 'oracle', 'liboracle.so'), allow_test_double=True)
bad = 0
t0 = time.time()
for case in range(count):
    logt = rng.choice([6, 7, 8, 9, 10, 11, 12, 13, 14])
    ef = rng.choice([8, 16, 32])
    steps = 1 << logt
    opts = {'hashAlgorithm': rng.choice(['blake2s256', 'sha256']), 'extensionFactor': ef,
            'exeQueryCount': rng.randrange(1, 90), 'friQueryCount': rng.randrange(1, 64)}
    seed = rng.randrange(1 << 128)
    datas = []
    try:
        for be, native in ((hip, True), (hip, False), (oracle, False)):
            stark = ga.instantiateMimc(steps, opts, backend=be)
            if not datas:
                control = ga.runMimc(stark.air.field, steps, stark.air.roundConstants, seed % stark.air.field.modulus)
                picks = sorted(set([0, steps - 1] + [rng.randrange(steps) for _ in range(rng.randrange(0, 4))]))
                assertions = [{'step': s, 'register': 0, 'value': control[s]} for s in picks]
            if native:
                datas.append(NativeProver(stark).prove_bytes(assertions, [], [seed]))
            else:
                datas.append(stark.serialize(stark.prove(assertions, [], [seed])))
        ok = datas[0] == datas[1] == datas[2]
        if ok:
            stark = ga.instantiateMimc(steps, opts, backend=hip)
            ok = stark.verify(assertions, stark.parse(datas[0])) is True
    except Exception as e:                                       # an option set the reference rejects must be rejected by all three alike
        ok, datas = False, [repr(e)]
    bad += 0 if ok else 1
    print(f'{case:3d} steps=2^{logt} E={ef} {opts["hashAlgorithm"]:10s} exe={opts["exeQueryCount"]:2d} fri={opts["friQueryCount"]:2d} asserts={len(assertions)} '
          f'bytes={len(datas[0]) if ok else datas[0][:80]} {"ok" if ok else "MISMATCH"}', flush=True)
print(f'{count} cases, {bad} failures, {time.time() - t0:.1f} s')
sys.exit(1 if bad else 0)
