"""tools/prove_only.py — N proofs of the headline statement through the native driver and nothing else (rocprofv3 target: the kernel
table divided by N is one proof).  usage: python tools/prove_only.py [proofs=10] [log_steps=20]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.native import NativeProver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
logt = int(sys.argv[2]) if len(sys.argv) > 2 else 20
be = Backend(device=0)
opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 64}
stark = ga.instantiateMimc(1 << logt, opts, backend=be)
nat = NativeProver(stark)
a = [{'step': 0, 'register': 0, 'value': 3}]
t0 = time.perf_counter()
for i in range(n):
    data = nat.prove_bytes(a, [], [3])
print(f'{n} proofs, {(time.perf_counter() - t0) / n * 1e3:.3f} ms each, {len(data)} bytes')
