"""createPrimeField(modulus) for ANY prime (index.ts:14; examples/assembly/lib128.ts:12): the runtime-modulus build flavours —
libgstark_hip_rt.so (gf_wide.h with GS_WIDE_BITS=0: constants in device constant memory, a product = two word-serial Montgomery
reductions, canonical 32-byte elements), its checker liboracle_rt.so (C23 _BitInt, plain `%`), and the native driver's
libgstark_prover_rt.so, which adopts the modulus of the library it is bound to.  One modulus per process (gs_set_modulus), so every
modulus runs in a worker process (tests/runtime_modulus_worker.py).

CPU tier: six NTT-friendly primes of 31 / 61 / 127 / 255 bits on the oracle flavour — vector members and NTT against Python integers,
two STARKs proved by mirror and native driver (same bytes), verified three ways.  GPU tier: 24 primes, the HIP library's proof
bytes == the oracle's for every one."""
import json
import os
import random
import subprocess
import sys

import pytest

from conftest import ROOT, _build_oracle

WORKER = os.path.join(ROOT, 'tests', 'runtime_modulus_worker.py')


def ntt_friendly_primes(per_size=6):
    """deterministic: primes k * 2^s + 1 of exactly `bits` bits (s = two-adicity at least 16 / 24 / 32 / 32: room for a 2^10-point domain)"""
    from sympy import isprime
    rng = random.Random(0x6753)
    out = []
    for bits, s in ((31, 16), (61, 24), (127, 32), (255, 32)):
        found = []
        while len(found) < per_size:
            k = rng.getrandbits(bits - s) | (1 << (bits - s - 1)) | 1
            p = (k << s) + 1
            if p.bit_length() == bits and isprime(p) and p not in found:
                found.append(p)
        out += found
    return out


PRIMES = ntt_friendly_primes()


def run_worker(q, which, timeout=600):
    r = subprocess.run([sys.executable, WORKER, str(q), which], cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (q, r.stderr[-2500:])
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec['modulus'] == str(q) and len(rec['proofs']) == 2
    return rec


def test_prime_list_is_what_it_says():
    assert len(PRIMES) == 24 and sorted({p.bit_length() for p in PRIMES}) == [31, 61, 127, 255]
    assert all((p - 1) % (1 << 16) == 0 for p in PRIMES) and len(set(PRIMES)) == 24


@pytest.mark.parametrize('q', [PRIMES[0], PRIMES[6], PRIMES[7], PRIMES[12], PRIMES[18], PRIMES[23]])
def test_runtime_modulus_on_the_oracle(q):
    _build_oracle()
    assert run_worker(q, 'oracle')['backends'] == ['oracle']


def test_fixed_builds_accept_only_their_own_modulus(oracle_backend):
    from genstark_amd._abi import GS_OK
    p = oracle_backend.modulus
    assert oracle_backend.lib.gs_set_modulus(p.to_bytes(16, 'little'), 16) == GS_OK
    assert oracle_backend.lib.gs_set_modulus((p - 2).to_bytes(16, 'little'), 16) == -3          # GS_ERR_UNSUPPORTED
    assert oracle_backend.lib.gs_set_modulus(b'', 0) != GS_OK


def test_runtime_library_refuses_even_and_tiny_moduli():
    _build_oracle()
    code = ('import sys; sys.path.insert(0, %r)\n'
            'import ctypes as C, os\n'
            'lib = C.CDLL(os.path.join(%r, "oracle", "liboracle_rt.so"))\n'
            'ctx = C.c_void_p()\n'
            'assert lib.gs_ctx_create(0, None, C.byref(ctx)) != 0          # no modulus yet\n'
            'for bad in (0, 1, 2, 4, 1 << 64):\n'
            '    assert lib.gs_set_modulus(bad.to_bytes(32, "little"), 32) != 0, bad\n'
            'assert lib.gs_set_modulus((97).to_bytes(32, "little"), 33) != 0      # longer than an element\n'
            'assert lib.gs_set_modulus((97).to_bytes(4, "little"), 4) == 0        # short encodings are zero-extended\n'
            'assert lib.gs_set_modulus((97).to_bytes(32, "little"), 32) == 0 and lib.gs_set_modulus((101).to_bytes(32, "little"), 32) == -3\n'
            'assert lib.gs_ctx_create(0, None, C.byref(ctx)) == 0\n') % (ROOT, ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize('q', PRIMES)
def test_runtime_modulus_hip_equals_oracle(q):
    _build_oracle()
    assert run_worker(q, 'both')['backends'] == ['hip', 'oracle']
