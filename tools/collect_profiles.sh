#!/bin/bash
# Collects the round's measured evidence on the GPU box into gpurun_out/rNN/ (copied to profiles/ afterwards).
# usage (through gpurun): tools/collect_profiles.sh r02
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. instruction costs and arithmetic cores (the second roof)
$R/tools/microbench4 > $O/${TAG}_a_instruction_costs.txt 2>&1
$R/tools/microbench5 > $O/${TAG}_a_arithmetic_cores.txt 2>&1
# 2. NTT kernels: A/B against the canonical-limb kernel, per-kernel trace, SQ counters, HBM traffic
python $R/tools/ntt_ab.py 24 > $O/${TAG}_b_ntt_ab.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_ntt -o t -- python $R/tools/ntt_only.py 24 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $O/trace_ntt/t_results.db rocprofv3 --kernel-trace --stats -- python tools/ntt_only.py 24 > $O/${TAG}_c_kernel_stats_ntt_2p24.md 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python $R/tools/ntt_only.py 24 > /dev/null 2>&1
cp $O/pmc_sq/s_counter_collection.csv $O/${TAG}_c_pmc_sq_ntt_2p24.csv 2>/dev/null
python $R/tools/pmc_traffic.py 24 $O > $O/${TAG}_c_ntt_pass_traffic.json 2>&1
# 3. the benchmark itself: the JSON line, and the kernel trace of the same command
python $R/bench.py > $O/${TAG}_d_bench_default.json 2> $O/${TAG}_d_bench_default.err
rocprofv3 --kernel-trace --stats -d $O/trace_bench -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --lanes 0 > $O/${TAG}_e_bench_under_rocprof.json 2>/dev/null
python $R/tools/rocprof_summary.py $O/trace_bench/b_results.db rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --lanes 0 > $O/${TAG}_e_kernel_stats_bench.md 2>&1
rm -rf $O/trace_ntt $O/trace_bench $O/pmc_sq
ls -la $O
