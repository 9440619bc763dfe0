#!/bin/bash
# memory side of the NTT pass kernels: address translation (UTCL1 hits / misses / stalls), vector-cache stalls, L2 hit rate and
# read latency, per kernel, for a few 2^LOG-point transforms.  Counter-only passes (--pmc with --kernel-trace), one group per run.
LOG=${1:-24}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_mem; mkdir -p $R/gpurun_out/pmc_mem
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mem/$N -o p -- python $R/tools/ntt_only.py $LOG > $R/gpurun_out/pmc_mem/$N.log 2>&1; }
N=a; run TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS
N=b; run TCP_UTCL1_THRASHING_STALL TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS
N=c; run TCP_PENDING_STALL_CYCLES TCP_TA_ADDR_STALL_CYCLES TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES
N=d; run TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY
N=e; run TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL
N=f; run TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_BUSY
N=g; run SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES TA_BUSY GRBM_GUI_ACTIVE
python - <<PY
import csv, glob, collections
for d in 'abcdefg':
    for f in glob.glob('$R/gpurun_out/pmc_mem/%s/**/*counter_collection.csv' % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'k_ntt' not in k: continue
            acc[k[:28]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in sorted(acc.items()):
            print(k, ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
grep -l "rror" $R/gpurun_out/pmc_mem/*.log | head
