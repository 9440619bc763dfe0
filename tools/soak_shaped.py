"""tools/soak_shaped.py [count] [seed] [backend] — randomized differential soak of air-assembly components WITH input registers
(tests/golden/aa/ledger.aa and variants of it with other rotations of the input columns): random numbers of runs and of public values
per run, random values (periodic public columns among them, which shrink to short cyclic registers), random options.  Per case: the
product entry (Prover over an AssemblyAir -> the native driver) on HIP with compiled AND interpreted programs and on the CPU oracle must
give the same bytes; the shapes the proof carries are the inputs'; the native verifier accepts the proof with the public inputs and
refuses it with one public value changed; the Python mirror's verifier agrees on the first.  backend = 'oracle' runs the product entry
on the oracle only (no GPU: what the build container can execute)."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from genstark_amd import airassembly
from genstark_amd._abi import Backend
from genstark_amd.errors import StarkError
from genstark_amd.field import PrimeField
from genstark_amd.prover import Prover

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
only_oracle = len(sys.argv) > 3 and sys.argv[3] == 'oracle'
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
LEDGER = open(os.path.join(root, 'tests', 'golden', 'aa', 'ledger.aa')).read()
oracle = Backend(lib_path=os.path.join(root, 'oracle', 'liboracle.so'), allow_test_double=True)
backends = [oracle] if only_oracle else [Backend(device=0).jit(), Backend(device=0), oracle]
P = oracle.modulus
bad, t0 = 0, time.time()
for case in range(count):
    # the module's own rotation (shift -1) is what its transition relies on; the other rotations prove other (still satisfiable: the mask
    # moves with its register) statements only when the restart check of the loader lets the segmentation stand or fall alike everywhere
    runs, per = 1 << rng.randrange(0, 8), 4
    opts = {'hashAlgorithm': rng.choice(['blake2s256', 'sha256']), 'exeQueryCount': rng.randrange(1, 60), 'friQueryCount': rng.randrange(1, 30)}
    balances = [rng.randrange(P) for _ in range(runs)]
    factors = [rng.randrange(P) for _ in range(runs)]
    style = rng.random()
    if style < 0.25:
        deposits = [[7, 9, 7, 9]] * runs                                  # a public column of period 2 values x 2 steps: a cyclic register of 4
    elif style < 0.4:
        row = [rng.randrange(P) for _ in range(per)]
        deposits = [list(row) for _ in range(runs)]                        # period = one run
    else:
        deposits = [[rng.randrange(P) for _ in range(per)] for _ in range(runs)]
    inputs, public = [balances, factors, deposits], [deposits]
    datas, assertions, ok, note = [], None, True, ''
    try:
        for be in backends:
            air = airassembly.AssemblyAir(LEDGER, 'default', None, PrimeField(backend=be))
            p = Prover(air, opts)
            if assertions is None:
                tr = air.initProvingContext(inputs).generateExecutionTrace()
                last = 8 * runs - 1
                picks = sorted(set([0, last] + [rng.randrange(8 * runs) for _ in range(rng.randrange(0, 3))]))
                assertions = [{'step': s, 'register': rng.randrange(3), 'value': None} for s in picks]
                for a in assertions:
                    a['value'] = tr.getValue(a['register'], a['step'])
            datas.append(p.prove_bytes(assertions, inputs, None))
        ok = all(d == datas[0] for d in datas)
        if ok:
            proof = p.parse(datas[0])
            ok = proof['iShapes'] == [[runs], [runs], [runs, per]] and p.verify_native(assertions, datas[0], public) is True
            if ok and case % 8 == 0:
                ok = p.verify(assertions, datas[0], public) is True        # the mirror's verdict (slow: every eighth case)
            if ok:
                r, j = rng.randrange(runs), rng.randrange(per)
                wrong = [[list(row) for row in deposits]]
                wrong[0][r][j] = (wrong[0][r][j] + 1) % P
                try:
                    p.verify_native(assertions, datas[0], wrong)
                    ok, note = False, 'a changed public input was accepted'
                except StarkError:
                    pass
    except Exception as e:   # noqa: BLE001
        ok, note = False, repr(e)[:160]
    bad += 0 if ok else 1
    print(f'{case:3d} runs={runs:4d} steps={8 * runs:5d} {opts["hashAlgorithm"]:10s} exe={opts["exeQueryCount"]:2d} fri={opts["friQueryCount"]:2d} '
          f'bytes={len(datas[0]) if datas else 0} {"ok" if ok else "MISMATCH " + note}', flush=True)
print(f'{count} cases, {bad} failures, {time.time() - t0:.1f} s')
sys.exit(1 if bad else 0)
