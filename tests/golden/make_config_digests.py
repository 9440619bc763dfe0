"""tests/golden/make_config_digests.py — the statements of tools/config_runs.py (every BASELINE.json configuration as bench.py proves it)
proved ONCE on the CPU oracle's implementation of the C ABI (oracle/liboracle*.so: the checker, OpenMP build for the two 2^20-step
statements) through the same native driver; writes tests/golden/config_digests.json = [{name, proof_bytes, proof_sha256}].

    python tests/golden/make_config_digests.py            (C4_long takes a few minutes on 8 cores)

What the file anchors: every configs[].proof_sha256 of a bench line (BENCH_rNN.json) must equal the committed value — a reader can tell
a right digest from a wrong one.  tests/test_config_digests.py reproduces the small ones on the oracle in the CPU
tier and all of them on the HIP library in the GPU tier."""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def oracle_for(omp=False):
    from genstark_amd._abi import MODULUS_32, Backend

    def make(modulus, jit):
        if modulus is None:
            lib = 'liboracle_omp.so' if omp else 'liboracle.so'
        elif modulus == MODULUS_32:
            lib = 'liboracle_q32.so'
        else:
            raise ValueError(modulus)
        path = os.path.join(ROOT, 'oracle', lib)
        if not os.path.exists(path):
            subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', lib])
        return Backend(lib_path=path, allow_test_double=True, **({} if modulus is None else {'modulus': modulus}))
    return make


def digest(name, omp=False):
    import config_runs
    be, p, a, inputs, seed, public = config_runs.statement(name, oracle_for(omp))
    data = p.prove_bytes(a, inputs, seed)
    assert p.verify_native(a, data, public) is True
    return {'name': name, 'proof_bytes': len(data), 'proof_sha256': hashlib.sha256(data).hexdigest()}


if __name__ == '__main__':
    import config_runs
    out = []
    for name in config_runs.CONFIGS:
        rec = digest(name, omp=name in ('C4_long', 'C5'))
        rec['config'] = config_runs.CONFIGS[name][0]
        print(rec, flush=True)
        out.append(rec)
    with open(os.path.join(HERE, 'config_digests.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
        fh.write('\n')
