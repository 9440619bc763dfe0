"""tools/pmc_traffic.py — HBM bytes per launch of the NTT pass kernels from the PMC counters, as MI355X_MICROARCH.md's HBM section
prescribes: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (they do not fit one), kernel-trace only beside them (a third pass
reads SQ_INSTS_VALU: the wave-level VALU instruction count bench.py's second roof is checked against);
on gfx950 FETCH_SIZE counts 128-byte requests of a wide coalesced streaming read at 64 bytes, so it is doubled; WRITE_SIZE as
reported.  Both are in KiB.  Prints one JSON object; bench.py runs this on rank 0 at N = 1.
usage: python tools/pmc_traffic.py [log2 n = 24] [outdir]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
keep = sys.argv[2] if len(sys.argv) > 2 else None
rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
out = {'n': 1 << logn, 'commands': []}
per_kernel = {}
work = tempfile.mkdtemp(prefix='pmc_', dir='/tmp')
env = dict(os.environ, TMPDIR='/tmp')
GROUPS = {'FETCH_SIZE': ['FETCH_SIZE'], 'WRITE_SIZE': ['WRITE_SIZE'],
          # the issue side, one pass: wave-level VALU instructions, the (quad-)cycles a SIMD spent issuing them, the kernel's own cycle count
          'SQ_INSTS_VALU': ['SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'GRBM_GUI_ACTIVE']}
for counter in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU'):
    d = os.path.join(work, counter)
    cmd = [rocprof, '--pmc', *GROUPS[counter], '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--',
           sys.executable, os.path.join(ROOT, 'tools', 'ntt_only.py'), str(logn)]
    out['commands'].append(' '.join(cmd[:cmd.index('--')] + ['--', 'python', 'tools/ntt_only.py', str(logn)]))
    subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no counter file under {d}')
    for row in csv.DictReader(open(files[0])):
        if 'k_ntt_' not in row['Kernel_Name'] or row['Counter_Name'] not in GROUPS[counter]:
            continue
        name = row['Kernel_Name'].split('(')[0].replace('void ', '')
        per_kernel.setdefault(name, {}).setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
    if keep:
        os.makedirs(keep, exist_ok=True)
        shutil.copy(files[0], os.path.join(keep, f'pmc_{counter.lower()}_ntt_2p{logn}.csv'))
kernels = {}
for name, c in per_kernel.items():
    f = sum(c.get('FETCH_SIZE', [0])) / max(1, len(c.get('FETCH_SIZE', [])))
    w = sum(c.get('WRITE_SIZE', [0])) / max(1, len(c.get('WRITE_SIZE', [])))
    v = c.get('SQ_INSTS_VALU', [])
    kernels[name] = {'launches_sampled': len(c.get('FETCH_SIZE', [])), 'FETCH_SIZE_kib_avg': f, 'WRITE_SIZE_kib_avg': w,
                     'hbm_bytes_per_launch': (2 * f + w) * 1024, 'SQ_INSTS_VALU_avg': (sum(v) / len(v)) if v else None}
    act, gui = c.get('SQ_ACTIVE_INST_VALU', []), c.get('GRBM_GUI_ACTIVE', [])
    if act and gui:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
        kernels[name]['valu_issue_utilisation'] = round((sum(act) / len(act)) * 4 / 1024 / ((sum(gui) / len(gui)) / 8), 4)
        kernels[name]['kernel_cycles'] = round((sum(gui) / len(gui)) / 8)
out['kernels'] = kernels
out['correction'] = 'bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes)'
if kernels:
    out['hbm_bytes_per_launch'] = sum(k['hbm_bytes_per_launch'] for k in kernels.values()) / len(kernels)
shutil.rmtree(work, ignore_errors=True)
print(json.dumps(out))
