mkdir -p gpurun_out/r03
python tools/merkle_ab.py > gpurun_out/r03/merkle_ab.txt 2>&1; cat gpurun_out/r03/merkle_ab.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03/trace_prove -o b -- python $R/tools/prove_only.py 10 > $R/gpurun_out/r03/prove_only.txt 2>/dev/null
python $R/tools/rocprof_summary.py $R/gpurun_out/r03/trace_prove/b_results.db rocprofv3 --kernel-trace --stats -- python tools/prove_only.py 10 > $R/gpurun_out/r03/kernel_stats_prove_only.md 2>&1
rm -rf $R/gpurun_out/r03/trace_prove
cat $R/gpurun_out/r03/prove_only.txt; head -45 $R/gpurun_out/r03/kernel_stats_prove_only.md; grep merkle $R/gpurun_out/r03/kernel_stats_prove_only.md | tail -40
