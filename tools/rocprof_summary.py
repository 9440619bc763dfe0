"""tools/rocprof_summary.py — per-kernel statistics (calls, total/avg/min/max duration, share) from a
rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes NAME_results.db).
usage: python tools/rocprof_summary.py gpurun_out/prof/r01_results.db [command words...] > profiles/xxx.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                        'from kernels group by name order by 3 desc'))
tot = sum(r[2] for r in rows) or 1
print(f'# rocprofv3 --kernel-trace --stats summary ({sys.argv[1]})')
if len(sys.argv) > 2:
    print('\ncommand: `' + ' '.join(sys.argv[2:]) + '`')
print(f'\ntotal kernel time: {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n')
print('| kernel | calls | total ms | avg us | min us | max us | % |')
print('|---|---:|---:|---:|---:|---:|---:|')
for name, calls, total, avg, mn, mx in rows:
    print(f'| `{name[:90]}` | {calls} | {total / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * total / tot:.1f} |')
extra = list(cur.execute('select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, count(*), avg(end-start) '
                         'from kernels group by name, grid_x, workgroup_x, lds_size order by name'))
print('\n## launch shapes (NTT / hash / inversion kernels)\n')
print('| kernel | grid_x | wg_x | lds bytes | vgpr | agpr | sgpr | scratch | calls | avg us |')
print('|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
for r in extra:
    if 'ntt' in r[0] or 'hash' in r[0] or 'batch_inv' in r[0] or 'merkle' in r[0]:
        print(f'| `{r[0][:60]}` | ' + ' | '.join(str(x) for x in r[1:9]) + f' | {r[9] / 1e3:.2f} |')
