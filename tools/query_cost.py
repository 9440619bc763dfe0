"""tools/query_cost.py — how much of a native prove() is the query tail (Merkle batch proofs + row gathers: device round trips)?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.native import NativeProver
be = Backend(device=0)
steps = 1 << 20
for exe, fri in ((48, 64), (48, 2), (2, 64), (2, 2)):
    opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': exe, 'friQueryCount': fri}
    stark = ga.instantiateMimc(steps, opts, backend=be)
    c = ga.runMimc(stark.air.field, 4, stark.air.roundConstants, 3)
    a = [{'step': 0, 'register': 0, 'value': 3}]
    nat = NativeProver(stark)
    for _ in range(2): nat.prove_bytes(a, [], [3])
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); nat.prove_bytes(a, [], [3]); t.append((time.perf_counter() - t0) * 1e3)
    print(exe, fri, round(min(t), 2), round(sum(t) / len(t), 2))
