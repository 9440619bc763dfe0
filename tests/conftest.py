import os
import random
import subprocess
import sys

import pytest

# AIR programs: the library's default is "auto" (compiled when a code object already exists, built in the background otherwise), which
# depends on what earlier processes left in the on-disk cache.  The tests pin the mode instead: interpreted unless a test asks for
# compiled programs (Backend.jit(True) / .jit('auto')), code objects in a directory of their own.
import os as _os
import tempfile as _tempfile
if not _os.environ.get('GSTARK_TEST_AUTO_CHILD'):      # tests/test_generic_air.py::test_generic_suite_in_default_auto_mode runs a child suite in the product default
    _os.environ.setdefault('GSTARK_AIR_JIT', '0')
_os.environ.setdefault('GSTARK_JIT_CACHE_DIR', _os.path.join(_tempfile.gettempdir(), f'gstark_jit_tests_{_os.getuid()}'))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

P = 2**128 - 9 * 2**32 + 1
ORACLE_LIB = os.path.join(ROOT, 'oracle', 'liboracle.so')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _build_oracle():
    src = [os.path.join(ROOT, 'oracle', f) for f in ('oracle_abi.c', 'hashes.c', 'gf128.h', 'gf_small.h', 'gf_wide.h', 'hashes.h')]
    libs = [ORACLE_LIB] + [os.path.join(ROOT, 'oracle', f'liboracle_{n}.so') for n in ('q64', 'q32', 'q17', 'p256', 'p224')]
    if any(not os.path.exists(l) or any(os.path.getmtime(s) > os.path.getmtime(l) for s in src) for l in libs):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s'])
    return ORACLE_LIB


@pytest.fixture(scope='session')
def oracle_backend():
    """The CPU oracle's implementation of include/gstark.h (test double; never used by the product)."""
    from genstark_amd._abi import Backend
    return Backend(lib_path=_build_oracle(), allow_test_double=True)


@pytest.fixture(scope='session')
def hip_backend():
    """The product backend: libgstark_hip.so on cuda:0 (raises without an MI355X)."""
    from genstark_amd._abi import Backend
    return Backend(device=0)


def rand_elements(rng, n, edge=0.1):
    edges = [0, 1, 2, P - 1, P - 2, 2**32 - 1, 2**32, 2**64 - 1, 2**64, 2**96, 2**127, 9 * 2**32 - 1, P - 9 * 2**32, P >> 1]
    out = []
    for _ in range(n):
        out.append(rng.choice(edges) if rng.random() < edge else rng.randrange(P))
    return out


def to_bytes(vals):
    return b''.join(int(v).to_bytes(16, 'little') for v in vals)


def from_bytes(raw):
    return [int.from_bytes(raw[i:i + 16], 'little') for i in range(0, len(raw), 16)]


@pytest.fixture
def rng():
    return random.Random(0x4d694d43)


def native_same_bytes(stark, assertions, inputs, seed, data):
    """The product's native driver (csrc/prover.cc, the build for this field flavour, bound to this backend's ABI library) on the
    statement a mirror Stark has just proved: the same serialized bytes."""
    from genstark_amd.native import NativeProver
    nat = NativeProver(stark)
    got = nat.prove_bytes(assertions, inputs, seed)
    assert got == data, 'native driver and mirror disagree'
    # ... and the native verifier (csrc/verifier.h) accepts it, and rejects it with a bit flipped in the middle
    assert nat.verify_bytes(assertions, got) is True
    from genstark_amd.errors import StarkError
    bad = bytearray(got)
    bad[len(bad) // 2] ^= 0x10
    with pytest.raises(StarkError):
        nat.verify_bytes(assertions, bytes(bad))
    return got
