"""tools/prove_only.py — N proofs of the headline statement through the native driver and nothing else (rocprofv3 target: the kernel
table divided by N is one proof).  usage: python tools/prove_only.py [proofs=10] [log_steps=20] [friQueryCount=64] [extensionFactor=16]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.native import NativeProver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
logt = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fri = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ef = int(sys.argv[4]) if len(sys.argv) > 4 else 16
be = Backend(device=0)
opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': ef, 'exeQueryCount': 48, 'friQueryCount': fri}
stark = ga.instantiateMimc(1 << logt, opts, backend=be)
nat = NativeProver(stark)
a = [{'step': 0, 'register': 0, 'value': 3}]
for i in range(3):
    nat.prove_bytes(a, [], [3])          # plans, block cache
t0 = time.perf_counter()
for i in range(n):
    data = nat.prove_bytes(a, [], [3])
print(f'MiMC-128 2^{logt} steps, E={ef}, fri {fri}: {n} proofs, {(time.perf_counter() - t0) / n * 1e3:.3f} ms each, {len(data)} bytes, driver {nat.last_stats()["total_ms"]:.3f} ms')
for k, v in nat.last_stats()['phases_ms'].items():
    print(f'  {v:8.3f}  {k}')
