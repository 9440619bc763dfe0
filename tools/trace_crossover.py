"""Trace generation of segmented AIRs, host interpreter vs one device thread per segment (interpreted and compiled), by number of
segments: where is the crossover?  GSTARK_HOST_TRACE_SEGMENTS moves the dispatch threshold of gs_air_trace_segments.
Usage: python tools/trace_crossover.py  (runs itself once per setting)"""
import os, subprocess, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')


def run(which):
    from genstark_amd._abi import Backend, MODULUS_224
    from genstark_amd.field import PrimeField
    if which == 'ec':
        from genstark_amd.pointmul import point_mul_air, to_bits
        from test_wide_fields import EC_POINT, EC_SCALAR
        b = Backend(device=0, modulus=MODULUS_224)
        f = PrimeField(backend=b)
        sizes = [4, 16, 64, 256]
        def make(n):
            air = point_mul_air(f, n)
            raw = [[EC_POINT[0]] * n, [EC_POINT[1]] * n, [to_bits(EC_SCALAR + 7 * i) for i in range(n)]]
            return air, air.expandInputs(raw), air.segmentSeeds(raw)
    else:
        from genstark_amd import poseidon
        from test_generic_air import rescue4x128_air
        b = Backend(device=0)
        f = PrimeField(backend=b)
        sizes = [4, 16, 64, 256, 1024]
        def make(n):
            if which == 'rescue':
                return rescue4x128_air(32 * n, 16, f, segmented=True), [], [[42 + s, 43 + 2 * s] for s in range(n)]
            return poseidon.poseidon6x128_air(64 * n, 16, f, segmented=True), [], [[1 + s, 2, 3 + s, 4] for s in range(n)]
    for n in sizes:
        air, inputs, seeds = make(n)
        ts = []
        for rep in range(4):
            ctx = air.initProvingContext(inputs, seeds)
            b.call('gs_sync')
            t0 = time.perf_counter()
            ctx.generateExecutionTrace(); b.call('gs_sync')
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f'{which:9s} segments {n:5d}  {os.environ.get("MODE"):12s} {min(ts[1:]):9.2f} ms', flush=True)


if len(sys.argv) > 1:
    run(sys.argv[1])
else:
    for which in ('ec', 'rescue', 'poseidon'):
        for mode, env in (('host', {'GSTARK_HOST_TRACE_SEGMENTS': '1000000'}), ('device', {'GSTARK_HOST_TRACE_SEGMENTS': '0'}),
                          ('device-jit', {'GSTARK_HOST_TRACE_SEGMENTS': '0', 'GSTARK_AIR_JIT': '1'})):
            subprocess.run([sys.executable, __file__, which], env=dict(os.environ, MODE=mode, **env))
