mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_generic_air.py tests/test_pipeline.py tests/test_wide_fields.py tests/test_native_dist.py tests/test_lib128.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
python tools/time_configs.py > gpurun_out/r03/prove_ms_baseline_configs.md 2>&1; cat gpurun_out/r03/prove_ms_baseline_configs.md | tail -25
