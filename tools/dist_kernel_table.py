"""tools/dist_kernel_table.py — where the distributed driver's added work is: kernels of the single-device proof beside the SUM over
the ranks of the same proof through the distributed driver (ranks sharing the one GPU), per proof, from two rocprofv3 kernel traces.
usage: python tools/dist_kernel_table.py <single_results.db> <proofs> <dist_results.db> <proofs> [title]"""
import re
import sqlite3
import sys


def load(path, proofs):
    db = sqlite3.connect(path)
    out = {}
    for name, calls, total in db.execute('select name, count(*), sum(end-start) from kernels group by name'):
        name = re.sub(r'\(.*$', '', name)
        c, t = out.get(name, (0, 0))
        out[name] = (c + calls / proofs, t + total / 1e6 / proofs)
    copies = 0.0
    try:
        for (total,) in db.execute('select sum(end-start) from memory_copies'):
            copies = (total or 0) / 1e6 / proofs
    except sqlite3.Error:
        pass
    span = list(db.execute('select min(start), max(end) from kernels'))[0]
    return out, copies, (span[1] - span[0]) / 1e6 / proofs


a, acopy, aspan = load(sys.argv[1], float(sys.argv[2]))
b, bcopy, bspan = load(sys.argv[3], float(sys.argv[4]))
title = sys.argv[5] if len(sys.argv) > 5 else ''
names = sorted(set(a) | set(b), key=lambda k: -(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]))
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print(f'# kernels per proof: single-device driver vs the distributed driver, summed over its ranks {title}\n')
print(f'kernel time per proof: single {ta:.3f} ms, distributed (all ranks) {tb:.3f} ms, added {tb - ta:+.3f} ms; device copies {acopy:.3f} / {bcopy:.3f} ms; '
      f'trace span per proof {aspan:.3f} / {bspan:.3f} ms\n')
print('| kernel | single calls | single ms | dist calls | dist ms | added ms |')
print('|---|---:|---:|---:|---:|---:|')
for k in names:
    ca, ma = a.get(k, (0, 0))
    cb, mb = b.get(k, (0, 0))
    if abs(mb - ma) < 0.0005 and ma < 0.002 and mb < 0.002:
        continue
    print(f'| `{k[:80]}` | {ca:.1f} | {ma:.3f} | {cb:.1f} | {mb:.3f} | {mb - ma:+.3f} |')
