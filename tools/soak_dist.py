"""tools/soak_dist.py — randomized differential soak of the native distributed driver: random statements (MiMC, Rescue, Poseidon, plain
and segmented), trace lengths, extension factors, hash algorithms, query counts and rank counts; ONE proof across G ranks (threads, the
tests' thread communicator) must equal the single-device driver's bytes on every rank and verify.  Backend: the oracle's implementation
of the C ABI by default (runs anywhere), `hip` as first argument for the HIP library with the ranks sharing the GPU.
usage: python tools/soak_dist.py [hip] [cases=100] [seed=1]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('GSTARK_AIR_JIT', '0')
from genstark_amd._abi import Backend
from genstark_amd.air import MimcAir, runMimc
from genstark_amd.field import PrimeField
from genstark_amd.native import NativeProver
from genstark_amd.poseidon import poseidon6x128_air
from genstark_amd.prover import Prover
from genstark_amd.rescue import rescue4x128_air
from dist_helpers import thread_comms
import threading

args = [a for a in sys.argv[1:]]
hip = bool(args and args[0] == 'hip')
if hip: args = args[1:]
cases = int(args[0]) if args else 100
rng = random.Random(int(args[1]) if len(args) > 1 else 1)
mk = (lambda: Backend(device=0)) if hip else (lambda: Backend(lib_path=os.path.join(ROOT, 'oracle', 'liboracle.so'), allow_test_double=True))
t0 = time.time()
done = 0
while done < cases:
    kind = rng.choice(['mimc', 'mimc', 'rescue', 'rescue_seg', 'poseidon', 'poseidon_seg'])
    G = rng.choice([1, 2, 2, 4, 4, 8])
    ef = rng.choice([8, 16, 16, 32]) if kind == 'mimc' else rng.choice([16, 32]) if 'poseidon' in kind else rng.choice([8, 16, 32])
    per = 64 if 'poseidon' in kind else 32
    log_t = rng.randrange(6 if kind == 'mimc' else per.bit_length() - 1 + ('seg' in kind), 11 if not hip else 14)
    steps = 1 << log_t
    if ef % G or (steps * ef) % (4 * G * G) or steps * ef < 128:
        continue
    opts = {'hashAlgorithm': rng.choice(['sha256', 'blake2s256']), 'extensionFactor': ef, 'exeQueryCount': rng.randrange(1, 81), 'friQueryCount': rng.randrange(1, 41)}

    def make(be):
        f = PrimeField(backend=be)
        if kind == 'mimc':
            return Prover(MimcAir(steps, ef, f), opts)
        if kind.startswith('rescue'):
            return Prover(rescue4x128_air(steps, ef, f, segmented='seg' in kind), opts)
        return Prover(poseidon6x128_air(steps, ef, f, segmented='seg' in kind), opts)
    if kind == 'mimc':
        seed = [rng.randrange(1, 2 ** 128 - 9 * 2 ** 32 + 1)]
    elif 'seg' in kind:
        seed = [[rng.randrange(1 << 64) for _ in range(4 if 'poseidon' in kind else 2)] for _ in range(steps // per)]
    else:
        seed = [rng.randrange(1 << 64) for _ in range(4 if 'poseidon' in kind else 2)]
    bes = [mk() for _ in range(G)]
    provers = [make(be) for be in bes]
    air = provers[0].air
    if kind == 'mimc':
        trace0 = runMimc(air.field, steps, air.roundConstants, seed[0])
        pick = lambda st, reg: trace0[st]
        regs = 1
    else:
        full = air.hostTrace(seed)
        pick = lambda st, reg: full[st][reg]
        regs = air.traceRegisterCount
    assertions = []
    for _ in range(rng.randrange(1, 4)):
        st, reg = rng.randrange(steps), rng.randrange(regs)
        if not any(a['step'] == st and a['register'] == reg for a in assertions):
            assertions.append({'step': st, 'register': reg, 'value': pick(st, reg)})
    try:
        want = provers[0].prove_bytes(assertions, [], seed)
    except Exception as e:      # noqa: BLE001  (e.g. more queries than the domain allows: same error on both paths, not a case)
        for be in bes: be.close()
        continue
    comms, keep = thread_comms(bes[0], G)
    gather_below = rng.choice([0, 1, 1 << 10, 1 << 14])          # FRI tail: default threshold, never gathered, gathered at various depths
    solo_below = rng.choice([1, 1, 1, 0, 1 << 12])               # mostly sharded whatever the size; sometimes the default / a low bar (rank 0 alone)
    for r in range(G):
        comms[r].fri_gather_below = gather_below
        comms[r].solo_below = solo_below
    outs, errs = [None] * G, [None] * G

    def run(r):
        try:
            outs[r] = provers[r].prove_bytes(assertions, [], seed, comm=comms[r])
        except BaseException as e:      # noqa: BLE001
            errs[r] = e
    ths = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in ths: t.start()
    for t in ths: t.join(600)
    assert not any(errs), (kind, steps, ef, G, opts, errs)
    assert all(o == want for o in outs), (kind, steps, ef, G, opts)
    assert provers[0].verify(assertions, want)
    for be in bes: be.close()
    done += 1
    if done % 10 == 0:
        print(f'{done} cases ok ({time.time() - t0:.0f} s); last: {kind} 2^{log_t} E={ef} G={G} {opts["hashAlgorithm"]} exe {opts["exeQueryCount"]} fri {opts["friQueryCount"]}', flush=True)
print(f'soak_dist: {done} random statements, every rank equal to the single-device driver, every proof verified ({"hip" if hip else "oracle"} backend, {time.time() - t0:.0f} s)')
