#!/bin/bash
# SQ counters of the NTT pass kernels of a few 2^LOG-point transforms (default 24): two counter-only passes, per-kernel averages on stdout
LOG=${1:-24}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_ntt; mkdir -p $R/gpurun_out/pmc_ntt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ntt/a -o p -- python $R/tools/ntt_only.py $LOG > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ntt/b -o p -- python $R/tools/ntt_only.py $LOG > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ('a', 'b'):
    for f in glob.glob('$R/gpurun_out/pmc_ntt/%s/**/*counter_collection.csv' % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'k_ntt' not in k: continue
            acc[k[:40]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in acc.items():
            print(k)
            for c, v in sorted(cs.items()):
                print('   %-28s %14.0f  (x%d)' % (c, sum(v) / len(v), len(v)))
PY
