"""tools/native_phases_generic.py — phase clock of the native driver (GSTARK_PROVER_TIMING=1) for the segmented Poseidon / Rescue configs."""
import os, sys
os.environ['GSTARK_PROVER_TIMING'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField
from genstark_amd.native import NativeProver
from genstark_amd.poseidon import poseidon6x128_air
from genstark_amd.rescue import rescue4x128_air
from genstark_amd._mirror.stark import Stark
be = Backend(device=0)
if len(sys.argv) > 1 and sys.argv[1] == 'jit':
    be.jit()
f = PrimeField(backend=be)
opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24}
for name, air, seeds, last in (('poseidon 1024 x 64', poseidon6x128_air(1 << 16, 16, f, segmented=True), [[1 + s, 2, 3 + s, 4] for s in range(1024)], 63),
                               ('rescue 2048 x 32', rescue4x128_air(1 << 16, 16, f, segmented=True), [[42 + s, 43 + 2 * s] for s in range(2048)], 31)):
    tr = air.initProvingContext([], seeds).generateExecutionTrace()
    a = [{'step': last, 'register': 0, 'value': tr.getValue(0, last)}, {'step': 65535, 'register': 1, 'value': tr.getValue(1, 65535)}]
    nat = NativeProver(Stark(air, opts))
    for i in range(3):
        sys.stderr.write(f'--- {name}, proof {i}\n')
        nat.prove_bytes(a, [], seeds)
