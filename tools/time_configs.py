"""tools/time_configs.py — prove() wall-clock of the BASELINE parity configurations on the HIP backend (markdown table).
usage: python tools/time_configs.py > profiles/xxx.md"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField
from genstark_amd.native import NativeProver
from genstark_amd.poseidon import poseidon6x128_air, poseidon_hash
from genstark_amd.rescue import rescue4x128_air
from genstark_amd._mirror.stark import Stark

be = Backend(device=0).jit(False)          # the plain columns are INTERPRETED programs (a new context's default is auto)
f = PrimeField(backend=be)
rows = []


def run(name, stark, assertions, seed, reps=5, make_air=None):
    log = ga.Logger(echo=False, sync=be.sync)
    for _ in range(2):
        proof = stark.prove(assertions, [], seed)
    gc.collect()
    gc.freeze()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        proof = stark.prove(assertions, [], seed)
        be.sync()
        t.append((time.perf_counter() - t0) * 1e3)
    data = stark.serialize(proof)
    nat = NativeProver(stark)
    tn = []
    for i in range(reps + 2):
        t0 = time.perf_counter()
        nd = nat.prove_bytes(assertions, [], seed)
        if i >= 2:
            tn.append((time.perf_counter() - t0) * 1e3)
    assert nd == data
    t0 = time.perf_counter()
    assert stark.verify(assertions, stark.parse(data))
    tv = (time.perf_counter() - t0) * 1e3
    th = None
    if name.startswith('MiMC'):          # the same proof through the GPU-free verifier (genstark_amd/hostfield.py)
        from genstark_amd.air import MimcAir
        from genstark_amd.hostfield import HostField
        hv = Stark(MimcAir(stark.air.steps, stark.air.extensionFactor, HostField()), stark_opts[name])
        hv.verify(assertions, hv.parse(data))
        t0 = time.perf_counter()
        assert hv.verify(assertions, hv.parse(data))
        th = (time.perf_counter() - t0) * 1e3
    tj = None
    if make_air is not None:             # the same statement with the AIR's programs compiled (gs_air_jit) instead of interpreted
        bj = Backend(device=0).jit()
        sj = Stark(make_air(PrimeField(backend=bj)), stark_opts[name])
        nj = NativeProver(sj)
        tj = []
        tjd = []
        for i in range(reps + 2):
            t0 = time.perf_counter()
            dj = nj.prove_bytes(assertions, [], seed)
            if i >= 2:
                tj.append((time.perf_counter() - t0) * 1e3)
                tjd.append(nj.last_stats()['total_ms'])
        assert dj == data and bj.jit_launches > 0
    s2 = Stark(stark.air, stark_opts[name], log)
    s2.prove(assertions, [], seed)
    phases = {k.strip(): v for k, v in log.phases}
    trace_ms = phases.get('Generated execution trace', 0.0)
    ths = f'{th:.1f}' if th else '-'
    tjs = f'{min(tj):.2f} ({min(tjd):.2f} inside the driver)' if tj else '-'
    rows.append(f'| {name} | {min(tn):.2f} | {tjs} | {sum(tn) / len(tn):.2f} | {min(t):.2f} | {trace_ms:.2f} | {tv:.1f} | {ths} | {len(data)} | {stark.securityLevel} |')


stark_opts = {}
# configs[1]: MiMC 2^13, E = 8 and E = 16 (README log)
for ef, fri in ((8, 24), (16, 24)):
    name = f'MiMC-128 2^13 steps, E={ef}, exe 48 / fri {fri}, blake2s256'
    stark_opts[name] = {'hashAlgorithm': 'blake2s256', 'extensionFactor': ef, 'exeQueryCount': 48, 'friQueryCount': fri}
    st = ga.instantiateMimc(1 << 13, stark_opts[name], backend=be)
    tr = st.air.initProvingContext([], [3]).generateExecutionTrace()
    a = [{'step': 0, 'register': 0, 'value': tr.getValue(0, 0)}, {'step': (1 << 13) - 1, 'register': 0, 'value': tr.getValue(0, (1 << 13) - 1)}]
    run(name, st, a, [3])
# configs[2]: Rescue 2^16
name = 'Rescue 4x128 2^16 steps (4 registers, degree 3), E=16, exe 68 / fri 24, blake2s256'
stark_opts[name] = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24}
air = rescue4x128_air(1 << 16, 16, f)
tr = air.initProvingContext([], [42, 43]).generateExecutionTrace()
a = [{'step': 31, 'register': 0, 'value': tr.getValue(0, 31)}, {'step': 31, 'register': 1, 'value': tr.getValue(1, 31)}]
run(name, Stark(air, stark_opts[name]), a, [42, 43])
# configs[3]: Poseidon 2^16, 6 registers (one GPU)
name = 'Poseidon 6x128 2^16 steps (6 registers, degree 6), E=16, exe 68 / fri 24, blake2s256'
stark_opts[name] = dict(stark_opts['Rescue 4x128 2^16 steps (4 registers, degree 3), E=16, exe 68 / fri 24, blake2s256'])
air = poseidon6x128_air(1 << 16, 16, f)
d = poseidon_hash(f, [1, 2, 3, 4])
a = [{'step': 63, 'register': 0, 'value': d[0]}, {'step': 63, 'register': 1, 'value': d[1]}]
run(name, Stark(air, stark_opts[name]), a, [1, 2, 3, 4])
# configs[2] / configs[3] as MANY hashes (one per 32- / 64-step segment; trace generated on the device, one thread per segment)
name = 'Rescue 4x128, 2048 hashes x 32 steps = 2^16 (segmented), E=16, exe 68 / fri 24, blake2s256'
stark_opts[name] = stark_opts['Rescue 4x128 2^16 steps (4 registers, degree 3), E=16, exe 68 / fri 24, blake2s256']
air = rescue4x128_air(1 << 16, 16, f, segmented=True)
seeds = [[42 + s, 43 + 2 * s] for s in range(2048)]
tr = air.initProvingContext([], seeds).generateExecutionTrace()
a = [{'step': 31, 'register': 0, 'value': tr.getValue(0, 31)}, {'step': 65535, 'register': 1, 'value': tr.getValue(1, 65535)}]
run(name, Stark(air, stark_opts[name]), a, seeds, make_air=lambda fj: rescue4x128_air(1 << 16, 16, fj, segmented=True))
name = 'Poseidon 6x128, 1024 hashes x 64 steps = 2^16 (segmented), E=16, exe 68 / fri 24, blake2s256'
stark_opts[name] = stark_opts['Rescue 4x128 2^16 steps (4 registers, degree 3), E=16, exe 68 / fri 24, blake2s256']
air = poseidon6x128_air(1 << 16, 16, f, segmented=True)
seeds = [[1 + s, 2, 3 + s, 4] for s in range(1024)]
tr = air.initProvingContext([], seeds).generateExecutionTrace()
a = [{'step': 63, 'register': 0, 'value': tr.getValue(0, 63)}, {'step': 65535, 'register': 1, 'value': tr.getValue(1, 65535)}]
run(name, Stark(air, stark_opts[name]), a, seeds, make_air=lambda fj: poseidon6x128_air(1 << 16, 16, fj, segmented=True))
# configs[4]: MiMC 2^20 (the bench workload) for reference
name = 'MiMC-128 2^20 steps, E=16, exe 48 / fri 64, blake2s256'
stark_opts[name] = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 64}
st = ga.instantiateMimc(1 << 20, stark_opts[name], backend=be)
tr = st.air.initProvingContext([], [3]).generateExecutionTrace()
a = [{'step': 0, 'register': 0, 'value': tr.getValue(0, 0)}, {'step': (1 << 20) - 1, 'register': 0, 'value': tr.getValue(0, (1 << 20) - 1)}]
run(name, st, a, [3])

print('# prove() wall-clock of the BASELINE configurations, 1 x MI355X (HIP backend), host-side trace generation included\n')
print('command: `python tools/time_configs.py` (2 warm-up proofs, 5 timed; native driver = csrc/prover.cc, Python mirror = stark.prove(), same bytes asserted; verify() timed once on the host)\n')
print('| configuration | prove() native driver best ms | same with compiled AIR programs (gs_air_jit); in brackets the clock of the driver itself, i.e. without the job packing of the Python binding | mean ms | Python mirror best ms | of which trace generation ms | verify() ms (device-side field) | verify() ms (HostField, no GPU) | proof bytes | security level |')
print('|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
print('\n'.join(rows))
