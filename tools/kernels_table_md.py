"""tools/kernels_table_md.py — roofline.kernels[] of a bench.py line (or the output of tools/config_runs.py) as markdown tables.
usage: python tools/kernels_table_md.py <bench line .json | config_runs .json>"""
import json, sys
text = open(sys.argv[1]).read().strip()
try:
    d = json.loads(text)                       # bench_detail.json (indented) or a one-line record
except ValueError:
    d = json.loads(text.splitlines()[-1])
tables = d['roofline']['kernels'] if isinstance(d, dict) else {c['name']: c['kernels'] for c in d if c.get('kernels')}
for name, rows in tables.items():
    busy = sum(r['ms'] for r in rows)
    print(f'\n**{name}** — {len(rows)} kernels, {busy:.3f} ms of kernels per proof\n')
    print('| kernel | calls | ms | share | algorithmic MB | GB/s | of 8 TB/s | VALU issue util. | own roof | of own roof |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|---|---:|')
    for r in rows:
        own = r.get('own_roof', '—').split(' (not')[0].replace("VALU issue of the kernel's own instruction mix", 'VALU issue (own mix)').replace('BLAKE2s compression issue, 39.5 G/s chip-wide', 'BLAKE2s issue 39.5 G/s').split(' (SQ_ACTIVE')[0]
        fo = r.get('frac_own_roof')
        print(f"| `{r['kernel']}` | {r['calls']:g} | {r['ms']:.4f} | {100 * r['ms'] / busy:.1f} % | {r.get('algorithmic_MB', '—')} | {r.get('GBs', '—')} | {r.get('frac_hbm', '—')} | {r.get('valu_issue_utilisation', '—')} | {own} | {'—' if fo is None else fo} |")
