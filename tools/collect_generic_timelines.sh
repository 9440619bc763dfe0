#!/bin/bash
# tools/collect_generic_timelines.sh <tag> — device timelines + kernel tables of the generic-AIR statements: Poseidon 2^16 (C4), Rescue 2^16 (C3),
# Poseidon 2^20 (C4-long).  Output: gpurun_out/<tag>/timeline_*.md, stats_*.md
set -u
tag=${1:-r04/t}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {   # <name> <cmd...>
  local name=$1; shift
  rm -rf /tmp/tl_$name
  rocprofv3 --kernel-trace --stats -d /tmp/tl_$name -o t -- "$@" > $out/run_$name.log 2>&1
  local db=$(find /tmp/tl_$name -name 't_results.db' | head -1)
  python3 $root/tools/timeline.py $db > $out/timeline_$name.md 2>> $out/run_$name.log
  python3 $root/tools/rocprof_summary.py $db > $out/stats_$name.md 2>> $out/run_$name.log || true
  grep -E "proofs,|^ +[0-9.]+  " $out/run_$name.log | head -20
}
run poseidon_2p16 python3 $root/tools/prove_generic_only.py poseidon 10
run rescue_2p16 python3 $root/tools/prove_generic_only.py rescue 10
run poseidon_2p20 python3 $root/tools/prove_generic_only.py poseidon 4 20
