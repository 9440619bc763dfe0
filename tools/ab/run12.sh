python tools/binv_time.py
python - <<'PY'
import ctypes as C, sys, time
sys.path.insert(0, '.')
from genstark_amd._abi import Backend
from genstark_amd.field import PrimeField
be = Backend(lib_path='tools/ab/libgstark_hip_r02.so'.replace('_r02','_r02')) if False else None
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inverse or pointwise or domain_div" 2>&1 | grep -E "passed|failed|rror" | tail -2
