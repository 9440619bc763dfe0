tools/lazy_device_check
timeout 1200 python -m pytest tests/test_generic_air.py tests/test_gpu_parity.py tests/test_lib128.py tests/test_native_prover.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/native_phases_generic.py jit 2>&1 | grep -A12 "proof 2" | grep -E "proof 2|execution trace|serialized"
