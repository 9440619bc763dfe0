#!/bin/bash
# instruction-fetch side of the NTT pass kernels (straight-line kernels of 40-57 KB against a 64 KB instruction cache shared by two CUs):
# SQC instruction-cache requests / hits / misses and the SQ's wait counters, per kernel, for a few 2^LOG-point transforms
LOG=${1:-24}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_if; mkdir -p $R/gpurun_out/pmc_if
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_if/available.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_if/a -o p -- python $R/tools/ntt_only.py $LOG > $R/gpurun_out/pmc_if/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_if/b -o p -- python $R/tools/ntt_only.py $LOG > $R/gpurun_out/pmc_if/b.log 2>&1
python - <<PY
import csv, glob, collections
print(open('$R/gpurun_out/pmc_if/available.txt').read())
for d in ('a', 'b'):
    for f in glob.glob('$R/gpurun_out/pmc_if/%s/**/*counter_collection.csv' % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'k_ntt' not in k: continue
            acc[k[:40]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in sorted(acc.items()):
            print(k)
            for c, v in sorted(cs.items()):
                print('   %-28s %14.0f  (x%d)' % (c, sum(v) / len(v), len(v)))
PY
tail -3 $R/gpurun_out/pmc_if/a.log | cut -c1-300
