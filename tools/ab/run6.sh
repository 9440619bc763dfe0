mkdir -p gpurun_out/r03
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; tail -c 3000 gpurun_out/r03/bench_default.json; tail -3 gpurun_out/r03/bench_default.err
