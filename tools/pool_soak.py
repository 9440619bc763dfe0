"""tools/pool_soak.py — throughput-mode soak: rounds of identical MiMC-128 2^20 proofs through native-driver lanes on one GPU; every
proof compared with the first one, GPU memory after each round (markdown)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import genstark_amd as ga
from genstark_amd.pipeline import ProverPool
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds, per = 5, 160
opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 64}
a = [{'step': 0, 'register': 0, 'value': 3}]
rows = []
with ProverPool(lambda backend: ga.instantiateMimc(1 << 20, opts, backend=backend), lanes=lanes, native=True) as pool:
    first = pool.prove_many_bytes([(a, [], [3])] * lanes)[0]
    for r in range(rounds):
        t0 = time.perf_counter()
        out = pool.prove_many_bytes([(a, [], [3])] * per)
        dt = time.perf_counter() - t0
        assert all(o == first for o in out)
        free, total = torch.cuda.mem_get_info()
        rows.append(f'| {r} | {per} | {dt / per * 1e3:.2f} | {(total - free) / 1e9:.1f} GB |')
print(f'# Throughput-mode soak: {rounds * per} proofs of MiMC-128 2^20 / E=16 / fri 64 through {lanes} native-driver lanes on one MI355X\n')
print(f'command: `python tools/pool_soak.py {lanes}` (after one warm-up proof per lane; every returned proof compared byte for byte with the first one)\n')
print('| round | proofs | ms per proof | GPU memory in use |\n|---:|---:|---:|---:|')
print('\n'.join(rows))
