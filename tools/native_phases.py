"""tools/native_phases.py — host wall-clock of the native driver's phases (GSTARK_PROVER_TIMING=1), MiMC-128 2^20 steps."""
import os, sys
os.environ['GSTARK_PROVER_TIMING'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.native import NativeProver
be = Backend(device=0)
opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 64}
stark = ga.instantiateMimc(1 << 20, opts, backend=be)
nat = NativeProver(stark)
a = [{'step': 0, 'register': 0, 'value': 3}]
for i in range(4):
    sys.stderr.write(f'--- proof {i}\n')
    nat.prove_bytes(a, [], [3])
