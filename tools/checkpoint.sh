#!/bin/bash
# tools/checkpoint.sh <tag> — the round-end sequence on the GPU box: the GPU test tier, then the default bench line
tag=${1:-r04/g}
mkdir -p gpurun_out/$tag
(time python -m pytest tests -m gpu -x -q) > gpurun_out/$tag/gputests.log 2>&1
grep -E "passed|failed|error" gpurun_out/$tag/gputests.log | tail -3
python bench.py --steps 5 --warmup 2 --detail gpurun_out/$tag/bench_detail.json > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -c 300 gpurun_out/$tag/bench.err
echo "bench line: $(tail -n 1 gpurun_out/$tag/bench.json | wc -c) bytes (the driver's bound: 6000)"
python tools/show_bench.py gpurun_out/$tag/bench_detail.json
