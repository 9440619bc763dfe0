#!/bin/bash
# per-kernel timings of a few 2^LOG-point NTTs (default 24): rocprofv3 --kernel-trace --stats, summary on stdout
LOG=${1:-24}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_ntt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ntt -o p -- python $R/tools/ntt_only.py $LOG > /dev/null 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/prof_ntt/p_results.db | grep -E "k_ntt|total kernel"
