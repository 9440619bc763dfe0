mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_native_prover.py tests/test_native_dist.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
GSTARK_AIR_JIT=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03/trace_c4 -o b -- python $R/tools/dist_phases.py c4 1 > $R/gpurun_out/r03/c4_g1.txt 2>/dev/null
python $R/tools/rocprof_summary.py $R/gpurun_out/r03/trace_c4/b_results.db > $R/gpurun_out/r03/kernel_stats_c4_dist_g1.md 2>&1
rm -rf $R/gpurun_out/r03/trace_c4
head -60 $R/gpurun_out/r03/kernel_stats_c4_dist_g1.md | cut -c1-200
