mkdir -p gpurun_out/r03
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; python - <<'PY'
import json
j=json.load(open('gpurun_out/r03/bench_default.json'))
print({k:j[k] for k in ('value','ms_per_step','per_step_ms')}); print(j['phases_ms']); r=j['roofline']; print({k:r[k] for k in ('achieved','frac','traffic','launch_ms','transform_ms')}); sr=r['second_roof']; print(sr.get('peak'), sr.get('frac'), sr.get('from_counters',{}).get('peak'), sr.get('from_counters',{}).get('frac'), [p.get('SQ_INSTS_VALU_per_launch') for p in sr.get('from_counters',{}).get('passes',[])]); print(j['pipelined']); print(j['cpu_baseline']['single_thread'])
PY
tail -2 gpurun_out/r03/bench_default.err
