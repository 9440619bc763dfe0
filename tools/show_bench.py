"""tools/show_bench.py <bench_detail.json | bench line> — the fields of a bench record worth a glance."""
import json, sys
text = open(sys.argv[1]).read().strip()
j = json.loads(text) if text.startswith('{\n') else json.loads(text.splitlines()[-1])
r = j['roofline']
print('prove_ms', round(j['ms_per_step'], 3), 'value', round(j['value'] / 1e9, 3), 'G el/s; NTT', r['achieved'], 'GB/s frac', r['frac'], 'transform_ms', r['transform_ms'],
      'second_roof frac', r['second_roof'].get('frac'), 'traffic', r.get('traffic'))
print('proof dominant kernel:', r.get('proof_dominant_kernel'))
for l, m in (j.get('phases_readme') or {}).get('ms') or []:
    print('   %8.3f  %s' % (m, l))
for c in j.get('configs') or []:
    print('  ', {k: c.get(k) for k in ('name', 'prove_ms', 'verify_native_ms', 'device_busy_ms', 'launches_per_proof', 'proof_bytes', 'dominant_kernel', 'dominant_share', 'error')})
cb = j.get('cpu_baseline') or {}
print('cpu_baseline', cb.get('value'), cb.get('cores'), 'same_bytes_as_gpu', cb.get('same_bytes_as_gpu'), '| pipelined', (j.get('pipelined') or {}).get('ms_per_proof'))
