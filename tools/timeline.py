"""tools/timeline.py — the kernels of the LAST proof in a rocprofv3 kernel trace, in issue order: start offset, duration and the idle
gap before each (how much of a proof's device time is dependent-launch latency rather than work).
usage: python tools/timeline.py <..._results.db> [launches per proof; default: the period of the launch sequence] > profiles/xxx.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute('select name, start, end from kernels order by start'))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if not per:                # proofs of one statement issue the same launches: the shortest period of the sequence of kernel names at its end
    names = [r[0] for r in rows]
    per = next((p for p in range(4, len(names) // 2 + 1) if names[-p:] == names[-2 * p:-p]), len(names))
rows = rows[-per:]
t0 = rows[0][1]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - t0
print(f'# device timeline of one proof ({sys.argv[1]}): {len(rows)} launches, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle between launches {(span - busy) / 1e3:.1f} us\n')
print('| # | kernel | start us | duration us | idle before us |')
print('|---:|---|---:|---:|---:|')
prev = t0
for i, (name, s, e) in enumerate(rows):
    print(f'| {i} | `{name[:70]}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {max(0, s - prev) / 1e3:.1f} |')
    prev = e
