"""tools/config_runs.py — prove() of every BASELINE.json configuration on the HIP backend through the product entry (native driver,
compiled AIR programs, packed seeds): wall-clock per proof, device-busy time per proof (sum of kernel durations, rocprofv3
--kernel-trace of a second run of the same child), launches per proof, proof size.  bench.py embeds the JSON (`configs`).

    python tools/config_runs.py                 -> one JSON list on stdout (parent: runs the children below)
    python tools/config_runs.py --child NAME N  -> N timed proofs of one configuration, one JSON object on stdout
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {   # name -> (what BASELINE.json calls it, timed proofs)
    'C1_foo': ('configs[0]: Foo (x <- x + 2), 64 steps over 2^32 - 3*2^25 + 1, 1 register, defaults (sha256, 80 / 40 queries): the q32 build of library and driver', 20),
    'C2_E8': ('configs[1]: MiMC-128, 2^13 steps, extensionFactor 8, exe 48, fri 24, blake2s256', 30),
    'C2_E16': ('configs[1] as the README log runs it: MiMC-128, 2^13 steps, extensionFactor 16, exe 48, fri 24, blake2s256', 30),
    'C3': ('configs[2]: Rescue 4x128, 2^16 steps = 2048 hash chains, E = 16, exe 68, fri 24, blake2s256', 20),
    'C4': ('configs[3]: Poseidon 6x128, 2^16 steps = 1024 hash chains, E = 16, exe 48, fri 24, blake2s256', 20),
    'C4_long': ('configs[3] at 2^20 steps = 16384 hash chains (the strong-scaling statement of --gpus N)', 5),
    'C5': ('configs[4]: MiMC-128, 2^20 steps, extensionFactor 16, exe 48, fri 64, blake2s256 (the headline)', 5),
    # not a BASELINE configuration: an air-assembly component WITH input registers (the shape of the reference's assembly/lib128.aa
    # statements), so that prove_ms / verify_native_ms also cover a proof that carries input shapes (lib/Stark.ts:161,176)
    'X_shaped': ('extra: tests/golden/aa/ledger.aa (this repository\'s module: 2 secret + 1 public input register, nested shapes, masks), 4096 runs = 2^15 steps, '
                 'E = 16, sha256, exe 24, fri 12; the proof carries iShapes and verify_native sizes the trace from them', 10),
}


NOTES = {
    'C3': 'measures a dependency chain, not the GPU: half of the device time is gs_jit_trace — 2 048 independent Rescue chains (x 4 registers on a 65 536-lane chip), '
          'each step an inverse S-box of 128 dependent squarings + 13 products at the floor of one product per 0.16 us (tools/microbench5: lz_sqr chain); more lanes do not shorten it',
    'C1_foo': 'plumbing check (64 steps, no FRI layer): every kernel is a launch latency',
    'C2_E8': 'latency-bound: 16 dependent launches for 2^16 points; k_fri_layers (a chain of dependent BLAKE2s compressions per layer) is the largest share',
    'C5': 'the headline: 70 % of prove_ms is the serial x^3 + k recurrence on one host core (gs_mimc_trace); the device side is 3.3 ms in 35 launches',
}


def statement(name, backend_for):
    """The statement of configuration `name` on a backend: (backend, Prover, assertions, inputs, seed).  backend_for(modulus or None, jit)
    -> Backend: the HIP library here, the CPU oracle's implementation of the C ABI in tests/golden/make_config_digests.py (the same
    statements proved once on the checker: tests/golden/config_digests.json anchors every configs[].proof_sha256 of the bench line)."""
    from genstark_amd._abi import MODULUS_32
    from genstark_amd.field import PrimeField
    from genstark_amd.prover import Prover
    opts = lambda ef, exe, fri: {'hashAlgorithm': 'blake2s256', 'extensionFactor': ef, 'exeQueryCount': exe, 'friQueryCount': fri}
    inputs, public = [], None
    if name == 'X_shaped':
        from genstark_amd import airassembly
        be = backend_for(None, True)
        f = PrimeField(backend=be)
        o = {'hashAlgorithm': 'sha256', 'exeQueryCount': 24, 'friQueryCount': 12}
        air = airassembly.AssemblyAir(open(os.path.join(ROOT, 'tests', 'golden', 'aa', 'ledger.aa')).read(), 'default', None, f)
        runs = 4096
        balances, factors = [100 + 7 * i for i in range(runs)], [3 + i for i in range(runs)]
        deposits = [[5 + i + 2 * j for j in range(4)] for i in range(runs)]
        inputs, public, seed = [balances, factors, deposits], [deposits], None
        tr = air.initProvingContext(inputs).generateExecutionTrace()
        last = 8 * runs - 1
        a = [{'step': 0, 'register': 0, 'value': balances[0]}, {'step': last, 'register': 2, 'value': tr.getValue(2, last)}]
        p = Prover(air, o)
    elif name == 'C1_foo':
        from genstark_amd.air_generic import GenericAir
        be = backend_for(MODULUS_32, True)
        f = PrimeField(backend=be)
        air = GenericAir(64, 1, [1], [], lambda r, k: [r[0] + 2], lambda r, n, k: [n[0] - (r[0] + 2)], lambda seed: [seed[0]], None, f)
        p = Prover(air, {'hashAlgorithm': 'sha256', 'extensionFactor': air.extensionFactor})      # README.md:17-60: defaults (sha256, 80 / 40 queries)
        a, seed = [{'step': 0, 'register': 0, 'value': 1}, {'step': 63, 'register': 0, 'value': 127}], [1]
    elif name in ('C2_E8', 'C2_E16', 'C5'):
        import genstark_amd as ga
        be = backend_for(None, False)
        steps, ef, fri = {'C2_E8': (1 << 13, 8, 24), 'C2_E16': (1 << 13, 16, 24), 'C5': (1 << 20, 16, 64)}[name]
        st = ga.instantiateMimc(steps, opts(ef, 48, fri), backend=be)
        p = Prover(st.air, opts(ef, 48, fri))
        # the statement bench.py's headline proves: the first and the last step asserted (examples/mimc/mimc128.ts:64-67)
        tr = st.generateExecutionTrace([], [3])['dTrace']
        a, seed = [{'step': 0, 'register': 0, 'value': tr.getValue(0, 0)}, {'step': steps - 1, 'register': 0, 'value': tr.getValue(0, steps - 1)}], [3]
    else:
        be = backend_for(None, True)
        f = PrimeField(backend=be)
        t = 1 << (20 if name == 'C4_long' else 16)
        if name == 'C3':
            from genstark_amd.rescue import rescue4x128_air
            air, seeds = rescue4x128_air(t, 16, f, segmented=True), [[42 + s, 43 + 2 * s] for s in range(t // 32)]
            tr = air.initProvingContext([], seeds).generateExecutionTrace()
            a = [{'step': 31, 'register': 0, 'value': tr.getValue(0, 31)}, {'step': t - 1, 'register': 1, 'value': tr.getValue(1, t - 1)}]
            p = Prover(air, opts(16, 68, 24))
        else:
            from genstark_amd.poseidon import poseidon6x128_air
            air, seeds = poseidon6x128_air(t, 16, f, segmented=True), [[1 + s, 2, 3 + s, 4] for s in range(t // 64)]
            a = [{'step': 0, 'register': 0, 'value': 1}, {'step': t - 64, 'register': 2, 'value': 3 + t // 64 - 1}]
            p = Prover(air, opts(16, 48, 24))
        seed = p.pack_seed(seeds)
    return be, p, a, inputs, seed, public


def child(name, reps):
    from genstark_amd._abi import Backend

    def hip(modulus, jit):
        be = Backend(device=0) if modulus is None else Backend(device=0, modulus=modulus)
        return be.jit() if jit else be
    be, p, a, inputs, seed, public = statement(name, hip)
    for _ in range(3):
        data = p.prove_bytes(a, inputs, seed)       # plans, block cache, compiled programs (hiprtc or the disk cache)
    be.sync()
    each = []
    for _ in range(reps):
        t0 = time.perf_counter()
        data = p.prove_bytes(a, inputs, seed)
        each.append((time.perf_counter() - t0) * 1e3)
    each.sort()
    ms = each[len(each) // 2]                          # the median proof (one host hiccup in five proofs moves a mean by 20 %)
    st = p.last_stats()
    p.sync_phases(True)
    p.prove_bytes(a, inputs, seed)
    readme = p.last_stats().get('phases_readme')
    p.sync_phases(False)
    # what every kernel of one proof HAD to move (gs_traffic_enable: algorithmic bytes and work units per kernel name; the parent joins
    # them with the kernel trace of the second child)
    be.traffic(True)
    p.prove_bytes(a, inputs, seed)
    be.sync()
    be.traffic(False)
    traffic = be.traffic()
    p.sync_phases(True)
    p.prove_bytes(a, inputs, seed)                    # (the measuring proof stays the LAST one of the child: busy_from_db steps back over it)
    tv = time.perf_counter()
    for _ in range(5):
        assert p.verify_native(a, data, public) is True        # Stark.verify natively (csrc/verifier.h): CPU only
    verify_ms = (time.perf_counter() - tv) / 5 * 1e3
    import hashlib
    print(json.dumps({'name': name, 'proof_sha256': hashlib.sha256(data).hexdigest(), 'assertions': len(a), 'traffic': traffic, 'prove_ms': round(ms, 4), 'prove_ms_min_max': [round(each[0], 4), round(each[-1], 4)], 'verify_native_ms': round(verify_ms, 4), 'driver_ms': st['total_ms'], 'proof_bytes': len(data), 'proofs_timed': reps,
                      'compiled_program_launches': int(getattr(be, 'jit_launches', 0)), 'phases_readme_ms': readme}), flush=True)


def busy_from_db(db_path):
    """(kernel time of the last proof in ms, its launches, the kernel with the largest share and that share) from a rocprofv3 rocpd db:
    proofs of one statement issue the same launch sequence; the last period of the sequence of kernel names is one proof."""
    import sqlite3
    rows = list(sqlite3.connect(db_path).execute('select name, start, end from kernels order by start'))
    names = [r[0] for r in rows]
    per = next((q for q in range(3, len(names) // 2 + 1) if names[-q:] == names[-2 * q:-q]), None)
    if per is None:
        return None
    # the synchronised measuring proof is the LAST one in the child: step back over it, then average over up to three plain proofs
    k = 0
    while k < 3 and len(names) >= (k + 3) * per and names[-(k + 2) * per:-(k + 1) * per] == names[-(k + 3) * per:-(k + 2) * per]:
        k += 1
    k = max(k, 1)
    sel = rows[-(k + 1) * per:-per]
    busy = sum(e - s for _, s, e in sel) / k / 1e6
    by, calls = {}, {}
    for n, s, e in sel:
        by[n] = by.get(n, 0) + (e - s)
        calls[n] = calls.get(n, 0) + 1
    top = max(by, key=by.get)
    return {'device_busy_ms': round(busy, 4), 'launches_per_proof': per, 'dominant_kernel': top[:80], 'dominant_share': round(by[top] / sum(by.values()), 3),
            'dominant_kernel_ms_per_proof': round(by[top] / k / 1e6, 4),
            '_per_kernel': {n: (calls[n] / k, by[n] / k / 1e6) for n in by}}


HBM_PEAK_GBS = 8000.0
B2S_COMPRESSIONS_PER_S = 39.5e9          # chip-wide BLAKE2s compression issue roof (tools/microbench_hash.hip, profiles/r03_b_*)
VALU_CLASS_NS = {'cheap': 1.1, 'vop3': 1.8, 'carry': 1.95, 'mad64': 2.0}      # profiles/r02_a_instruction_costs.txt, 4 waves per SIMD


def kernel_table(per_kernel, traffic, counters=None):
    """roofline.kernels[]: every kernel of one proof — calls, ms, the bytes it HAD to move (the library's own tally, SURVEY 8d's
    algorithmic bytes per launch), GB/s and the fraction of the 8 TB/s HBM roof; for the kernels whose own roof is known, that roof and
    the fraction of it: hash kernels against the chip's BLAKE2s compression issue rate, NTT passes against the issue time of their own
    static VALU instruction mix (csrc/ntt_isa_mix.json x the measured cost of each instruction class, 1024 SIMDs)."""
    import re
    try:
        mix = json.load(open(os.path.join(ROOT, 'genstark_amd', 'csrc', 'ntt_isa_mix.json')))
    except Exception:   # noqa: BLE001
        mix = {}
    out = []
    for full, (calls, ms) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        base = re.sub(r'^void ', '', full)
        base = re.sub(r'\(.*$', '', base).strip()
        t = traffic.get(base)
        row = {'kernel': base, 'calls': round(calls, 2), 'ms': round(ms, 4)}
        if t and ms > 0:
            per = max(t['launches'], 1)                 # the tally is of ONE proof; calls from the trace should agree
            row['algorithmic_MB'] = round(t['bytes'] / 1e6, 3)
            row['GBs'] = round(t['bytes'] / (ms * 1e-3) / 1e9, 1)
            row['frac_hbm'] = round(t['bytes'] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            row['calls_tallied'] = per
            if ('merkle' in base or 'hash' in base or 'fri_layers' in base) and '<1' in base:
                row['own_roof'] = 'BLAKE2s compression issue, 39.5 G/s chip-wide'
                row['compressions'] = t['units']
                row['frac_own_roof'] = round(t['units'] / (ms * 1e-3) / B2S_COMPRESSIONS_PER_S, 4)
            m = re.match(r'k_ntt_(wave|pass_lz)<(\d), (\d)>', base)
            if m:
                key = next((k for k in mix if f'k_ntt_{m.group(1)}ILi{m.group(2)}ELi{m.group(3)}E' in k), None)
                if key:
                    waves = t['units'] / 16 / 64                      # every thread owns 16 elements, whatever the workgroup shape
                    ns = sum(mix[key][c] * VALU_CLASS_NS[c] for c in VALU_CLASS_NS) * waves / 1024
                    row['own_roof'] = 'VALU issue of the kernel\'s own instruction mix'
                    row['valu_per_wave'] = mix[key]['valu']
                    row['frac_own_roof'] = round(ns * 1e-6 / ms, 4)
                    if row['frac_own_roof'] > 1.0:      # a zero-extending first pass (>= 16x) skips its first network: the static count overstates it
                        row['frac_own_roof'] = None
                        row['own_roof'] += ' (not applicable: the pruned first pass of a >= 16x extension skips the first radix-16 network)'
        cn = (counters or {}).get(base)
        if cn and 'valu_issue_utilisation' in cn:
            row['valu_issue_utilisation'] = cn['valu_issue_utilisation']
            row['valu_insts_per_launch'] = cn['valu_insts_per_launch']
            if row.get('frac_own_roof') is None and 'own_roof' not in row:
                # neither a hash kernel nor an NTT pass: the kernel's own roof is whichever of its two resources it uses more of — the
                # HBM stream it has to move, or the issue slots of its own instruction stream (counters of THIS run)
                hbm = row.get('frac_hbm') or 0.0
                if cn['valu_issue_utilisation'] >= hbm:
                    row['own_roof'] = 'VALU issue of its own instruction stream (SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, this run)'
                    row['frac_own_roof'] = min(1.0, cn['valu_issue_utilisation'])      # (the counter pass runs at its own clocks: a few % over 1 is measurement)
                else:
                    row['own_roof'] = 'HBM (8 TB/s)'
                    row['frac_own_roof'] = hbm
        out.append(row)
    if counters and 'error' in counters:
        out.append({'kernel': '(issue counters)', 'calls': 0, 'ms': 0.0, 'error': counters['error']})
    return out


def issue_counters(name, reps, env):
    """Third child of a configuration: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE (one pass; counters alone with
    --kernel-trace, as the pool's rule for PMC runs requires) -> per kernel name {valu_insts_per_launch, valu_issue_utilisation}:
    SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs, so
    utilisation = active x 4 / 1024 / (gui / 8): the share of its own cycles a SIMD spent issuing vector instructions in that launch."""
    import csv
    import glob
    import re
    d = f'/tmp/gs_pmc_{os.getpid()}_{name}'
    out = {}
    try:
        subprocess.run(['rocprofv3', '--pmc', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'GRBM_GUI_ACTIVE', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--',
                        sys.executable, os.path.abspath(__file__), '--child', name, str(min(reps, 2))], capture_output=True, text=True, timeout=300, env=env, cwd='/tmp')
        files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
        acc = {}
        for row in csv.DictReader(open(files[0])) if files else []:
            base = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name'])).strip()
            acc.setdefault(base, {}).setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
        for base, c in acc.items():
            v, a, g = c.get('SQ_INSTS_VALU'), c.get('SQ_ACTIVE_INST_VALU'), c.get('GRBM_GUI_ACTIVE')
            if v and a and g and sum(g) > 0:
                out[base] = {'valu_insts_per_launch': round(sum(v) / len(v)), 'valu_issue_utilisation': round((sum(a) / len(a)) * 4 / 1024 / ((sum(g) / len(g)) / 8), 4),
                             'launches_sampled': len(v)}
    except Exception as e:   # noqa: BLE001
        out = {'error': repr(e)[:200]}
    subprocess.run(['rm', '-rf', d])
    return out


def parent(names, rocprof=True):
    out = []
    env = dict(os.environ, TMPDIR='/tmp')
    for name in names:
        what, reps = CONFIGS[name]
        rec = {'name': name, 'config': what}
        if name in NOTES:
            rec['note'] = NOTES[name]
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', name, str(reps)], capture_output=True, text=True, timeout=180, env=env)
            rec.update(json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1]))
        except Exception as e:   # noqa: BLE001
            rec['error'] = repr(e)[:200]
            out.append(rec)
            continue
        if rocprof:
            d = f'/tmp/gs_cfg_{os.getpid()}_{name}'
            try:
                subprocess.run(['rocprofv3', '--kernel-trace', '-d', d, '-o', 'c', '--', sys.executable, os.path.abspath(__file__), '--child', name, str(min(reps, 4))],
                               capture_output=True, text=True, timeout=240, env=env, cwd='/tmp')
                db = next((os.path.join(dp, fn) for dp, _, fns in os.walk(d) for fn in fns if fn.endswith('_results.db')), None)
                got = busy_from_db(db) or {'device_busy_ms': None}
                per_kernel = got.pop('_per_kernel', None)
                rec.update(got)
                if per_kernel and rec.get('traffic') is not None:
                    # the two long statements also get the issue-side counters of every kernel (their own roof for the kernels that are
                    # neither hash- nor HBM-bound: how much of its own time a SIMD spent issuing vector instructions)
                    counters = issue_counters(name, reps, env) if name in ('C5', 'C4_long') else None
                    rec['kernels'] = kernel_table(per_kernel, rec['traffic'], counters)
            except Exception as e:   # noqa: BLE001
                rec['rocprof_error'] = repr(e)[:200]
            subprocess.run(['rm', '-rf', d])
        rec.pop('traffic', None)
        out.append(rec)
    return out


if __name__ == '__main__':
    if len(sys.argv) >= 4 and sys.argv[1] == '--child':
        child(sys.argv[2], int(sys.argv[3]))
    else:
        names = [a for a in sys.argv[1:] if a in CONFIGS] or list(CONFIGS)
        print(json.dumps(parent(names, rocprof='--no-rocprof' not in sys.argv)))
