#!/bin/bash
# tools/collect_dist_kernels.sh <tag> [statement=c4long] [G=8] [full] — kernel tables of one statement: single-device driver vs the distributed
# driver with G ranks sharing the GPU (tools/dist_only.py under rocprofv3 --kernel-trace), joined by tools/dist_kernel_table.py
set -u
tag=${1:-r05/d}; which=${2:-c4long}; G=${3:-8}; n=6
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export GSTARK_COMM_TAKE_TURNS=1     # one rank on the device at a time: every kernel at its uncontended duration (oracle/comm_threads.c)
for g in 0 $G; do
  rm -rf /tmp/dk_$g
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/dk_$g -o t -- python3 $root/tools/dist_only.py $which $g $n ${4:-} > $out/run_${which}_G$g.log 2>&1
  grep -E "ms per proof|^ +[0-9.]+  |collectives" $out/run_${which}_G$g.log
done
a=$(find /tmp/dk_0 -name 't_results.db' | head -1); b=$(find /tmp/dk_$G -name 't_results.db' | head -1)
python3 $root/tools/dist_kernel_table.py $a $((n + 2)) $b $((n + 2)) "($which, G = $G)" > $out/dist_kernel_table_${which}_G$G.md
head -40 $out/dist_kernel_table_${which}_G$G.md
