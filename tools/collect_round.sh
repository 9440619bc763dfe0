#!/bin/bash
# tools/collect_round.sh <tag> — the round-end evidence in one call on the GPU box: GPU test tier + default bench line (checkpoint.sh),
# per-proof device timelines of the BASELINE statements (collect_timelines.sh, collect_generic_timelines.sh), the kernel table of the 2^24
# transform, and `rocprofv3 --kernel-trace --stats` of the bench command itself (its NTT kernels' average durations are what
# roofline.achieved is computed from).  Output: gpurun_out/<tag>/
tag=${1:-r04/z}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
bash tools/checkpoint.sh $tag
# the sanitizer tier (CPU: ASAN + UBSAN builds of driver, verifier and addon on corrupted input; tests/test_sanitizers.py)
(time python3 -m pytest tests/test_sanitizers.py -q) > $out/sanitizer_tier.txt 2>&1; grep -E "passed|failed|skipped" $out/sanitizer_tier.txt | tail -1
bash tools/collect_timelines.sh $tag 2>&1 | grep -E "proofs,"
bash tools/collect_generic_timelines.sh $tag 2>&1 | grep -E "proofs,"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_ntt /tmp/tr_bench
rocprofv3 --kernel-trace --stats -d /tmp/tr_ntt -o t -- python3 $root/tools/ntt_only.py 24 > /dev/null 2>&1
python3 $root/tools/rocprof_summary.py $(find /tmp/tr_ntt -name t_results.db | head -1) rocprofv3 --kernel-trace --stats -- python tools/ntt_only.py 24 > $out/kernel_stats_ntt_2p24.md 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/tr_bench -o b -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --lanes 0 --no-configs > $out/bench_under_rocprof.json 2>/dev/null
python3 $root/tools/rocprof_summary.py $(find /tmp/tr_bench -name b_results.db | head -1) rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --lanes 0 --no-configs > $out/kernel_stats_bench.md 2>&1
head -12 $out/kernel_stats_ntt_2p24.md
