"""tools/prove_generic_only.py — N proofs of a segmented hash AIR (BASELINE configs[2] / configs[3] as many hash chains) through the
product entry with compiled programs and a packed seed, and nothing else (rocprofv3 target).
usage: python tools/prove_generic_only.py [poseidon|rescue] [proofs=10] [log_steps=16]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField
from genstark_amd.poseidon import poseidon6x128_air
from genstark_amd.prover import Prover
from genstark_amd.rescue import rescue4x128_air
which = sys.argv[1] if len(sys.argv) > 1 else 'poseidon'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 16)
be = Backend(device=0).jit()
f = PrimeField(backend=be)
opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24}
if which == 'poseidon':
    air, seeds = poseidon6x128_air(t, 16, f, segmented=True), [[1 + s, 2, 3 + s, 4] for s in range(t // 64)]
    a = [{'step': 0, 'register': 0, 'value': 1}, {'step': t - 64, 'register': 2, 'value': 3 + t // 64 - 1}]
else:
    air, seeds = rescue4x128_air(t, 16, f, segmented=True), [[42 + s, 43 + 2 * s] for s in range(t // 32)]
    tr = air.initProvingContext([], seeds).generateExecutionTrace()      # (its init block transforms the inputs: read the cells back)
    a = [{'step': 31, 'register': 0, 'value': tr.getValue(0, 31)}, {'step': t - 1, 'register': 1, 'value': tr.getValue(1, t - 1)}]
p = Prover(air, opts)
seed = p.pack_seed(seeds)
for i in range(3):
    data = p.prove_bytes(a, [], seed)      # compiles the programs (or loads the cached code objects), fills the block cache
be.sync()
print('MARK timed proofs start', flush=True)
t0 = time.perf_counter()
for i in range(n):
    data = p.prove_bytes(a, [], seed)
print(f'{which} {t} steps: {n} proofs, {(time.perf_counter() - t0) / n * 1e3:.3f} ms each, {len(data)} bytes, driver {p.last_stats()["total_ms"]:.3f} ms, '
      f'compiled launches {be.jit_launches}')
for k, v in p.last_stats()['phases_ms'].items():
    print(f'  {v:8.3f}  {k}')
