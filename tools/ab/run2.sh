set -x
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "merkle or hashing" 2>&1 | tail -5
python tools/merkle_ab.py tools/ab/libgstark_hip_r02.so > gpurun_out/r03/merkle_ab.txt 2>&1; cat gpurun_out/r03/merkle_ab.txt
timeout 600 python -m pytest tests/test_native_prover.py tests/test_gpu_parity.py -m gpu -x -q -k "native or golden or prove" 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > gpurun_out/r03/bench_quick.json 2> gpurun_out/r03/bench_quick.err; cat gpurun_out/r03/bench_quick.json; tail -3 gpurun_out/r03/bench_quick.err
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03/trace_bench -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --lanes 0 > $R/gpurun_out/r03/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocprof_summary.py $R/gpurun_out/r03/trace_bench/b_results.db rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --lanes 0 > $R/gpurun_out/r03/kernel_stats_bench.md 2>&1
rm -rf $R/gpurun_out/r03/trace_bench
head -60 $R/gpurun_out/r03/kernel_stats_bench.md
