"""tools/profile_generic.py — a few native-driver proofs of the many-hash Rescue / Poseidon AIRs (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField
from genstark_amd.native import NativeProver
from genstark_amd.poseidon import poseidon6x128_air
from genstark_amd.rescue import rescue4x128_air
from genstark_amd._mirror.stark import Stark

f = PrimeField(backend=Backend(device=0))
opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24}
for kind in sys.argv[1:] or ['rescue', 'poseidon']:
    if kind == 'rescue':
        air, seeds = rescue4x128_air(1 << 16, 16, f, segmented=True), [[42 + s, 43 + 2 * s] for s in range(2048)]
    else:
        air, seeds = poseidon6x128_air(1 << 16, 16, f, segmented=True), [[1 + s, 2, 3 + s, 4] for s in range(1024)]
    tr = air.initProvingContext([], seeds).generateExecutionTrace()
    a = [{'step': 65535, 'register': 1, 'value': tr.getValue(1, 65535)}]
    nat = NativeProver(Stark(air, opts))
    for _ in range(5):
        nat.prove_bytes(a, [], seeds)
