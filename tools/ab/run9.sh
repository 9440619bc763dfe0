timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "merkle or hashing or record_ops or golden" 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/merkle_ab.py 2>&1 | grep -E "2\^(16|14|12|10|8|6) "
python tools/prove_only.py 10
