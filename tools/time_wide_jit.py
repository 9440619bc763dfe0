"""Point multiplication over the 224-bit field, 16 multiplications (4096 steps): device trace + constraints, interpreted vs compiled."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from test_wide_fields import hip_for, EC_POINT, EC_SCALAR, EC_OPTIONS
from genstark_amd.field import PrimeField
from genstark_amd.pointmul import point_mul_air, to_bits
from genstark_amd._mirror.stark import Stark

for jit in (False, True):
    b = hip_for('p224')
    if jit:
        b.jit()
    f = PrimeField(backend=b)
    air = point_mul_air(f, 16)
    raw = [[EC_POINT[0]] * 16, [EC_POINT[1]] * 16, [to_bits(EC_SCALAR + 7 * i) for i in range(16)]]
    inputs, seeds = air.expandInputs(raw), air.segmentSeeds(raw)
    stark = Stark(air, EC_OPTIONS)
    for rep in range(3):
        t0 = time.perf_counter()
        ctx = air.initProvingContext(inputs, seeds)
        tr = ctx.generateExecutionTrace(); b.call('gs_sync')
        t1 = time.perf_counter()
        proof = stark.prove([{'step': 255, 'register': 7, 'value': EC_SCALAR}], inputs, seeds)
        t2 = time.perf_counter()
        print(f'jit={jit} rep={rep} trace {1e3 * (t1 - t0):.1f} ms  prove {1e3 * (t2 - t1):.1f} ms  launches {b.jit_launches}', flush=True)
