"""The C-ABI shared library loads and exports every symbol include/gstark.h declares (no compute calls:
runs without a GPU).  Also unit-tests the device arithmetic header on the host."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import P, ROOT, rand_elements, to_bytes
from genstark_amd import _abi


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'gstark.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gs_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_abi.EXPORTED_SYMBOLS)


def test_hip_library_exports_every_symbol():
    assert os.path.exists(_abi.HIP_LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_abi.HIP_LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    lib.gs_backend_name.restype = ctypes.c_char_p
    assert lib.gs_backend_name() == b'hip-gfx950'
    assert lib.gs_abi_version() == 1
    buf = ctypes.create_string_buffer(16)
    lib.gs_field_modulus(buf)
    assert int.from_bytes(buf.raw, 'little') == P


def test_field_flavours_export_every_symbol():
    """The q64 / q32 builds (csrc/gf_small.h) and the p256 / p224 builds (csrc/gf_wide.h) of the same sources: same ABI, their
    own modulus and element size."""
    for modulus, path in _abi.HIP_LIB_PATHS.items():
        assert os.path.exists(path), f'{path}: run __graft_entry__.build() first'
        lib = ctypes.CDLL(path)
        for name in header_symbols():
            assert hasattr(lib, name), (path, name)
        lib.gs_backend_name.restype = ctypes.c_char_p
        assert lib.gs_backend_name() == b'hip-gfx950'
        assert lib.gs_element_size() == (16 if modulus < 2**128 else 32)
        buf = ctypes.create_string_buffer(lib.gs_element_size())
        lib.gs_field_modulus(buf)
        assert int.from_bytes(buf.raw, 'little') == modulus


def test_small_field_device_header_on_host(tmp_path, rng):
    """gf_small.h compiled for the host (the same GF_HD functions the kernels inline) against Python integers, both moduli."""
    src = tmp_path / 'h.cpp'
    src.write_text('#include <stdint.h>\n#include "%s"\n'
                   'extern "C" void ops(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *add, uint64_t *sub, uint64_t *mul, uint64_t *inv) {\n'
                   '  for (uint64_t i = 0; i < n; i++) { fe x = fe_from(a[i]), y = fe_from(b[i]);\n'
                   '    add[i] = fe_u64(fe_add(x, y)); sub[i] = fe_u64(fe_sub(x, y)); mul[i] = fe_u64(fe_mul(x, y)); inv[i] = fe_u64(fe_inv(x)); } }\n'
                   % os.path.join(ROOT, 'genstark_amd', 'csrc', 'gf_small.h'))
    for q in (_abi.MODULUS_64, _abi.MODULUS_32, _abi.MODULUS_17):
        so = str(tmp_path / f'h_{q}.so')
        subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', f'-DGS_SMALL_Q={q}ull', '-o', so, str(src)])
        lib = ctypes.CDLL(so)
        n = 5000
        a = [0, 1, q - 1, q - 2] + [rng.randrange(q) for _ in range(n - 4)]
        b = [q - 1, q - 1, q - 1, 2] + [rng.randrange(q) for _ in range(n - 4)]
        arr = lambda v: (ctypes.c_uint64 * n)(*v)
        A, B, o = arr(a), arr(b), [(ctypes.c_uint64 * n)() for _ in range(4)]
        lib.ops(A, B, ctypes.c_uint64(n), *o)
        assert list(o[0]) == [(x + y) % q for x, y in zip(a, b)]
        assert list(o[1]) == [(x - y) % q for x, y in zip(a, b)]
        assert list(o[2]) == [x * y % q for x, y in zip(a, b)]
        assert list(o[3]) == [pow(x, -1, q) if x else 0 for x in a]


def test_wide_field_device_header_on_host(tmp_path, rng):
    """gf_wide.h compiled for the host (the same GF_HD functions the kernels inline) against Python integers, both moduli."""
    src = tmp_path / 'w.cpp'
    src.write_text('#include <stdint.h>\n#include <string.h>\n#include "%s"\n'
                   'extern "C" void ops(const uint8_t *a, const uint8_t *b, uint64_t n, uint8_t *add, uint8_t *sub, uint8_t *mul, uint8_t *inv) {\n'
                   '  for (uint64_t i = 0; i < n; i++) { fe x, y, r; memcpy(&x, a + 32 * i, 32); memcpy(&y, b + 32 * i, 32);\n'
                   '    r = fe_add(x, y); memcpy(add + 32 * i, &r, 32); r = fe_sub(x, y); memcpy(sub + 32 * i, &r, 32);\n'
                   '    r = fe_mul(x, y); memcpy(mul + 32 * i, &r, 32); if (i < 200) { r = fe_inv(x); memcpy(inv + 32 * i, &r, 32); } } }\n'
                   % os.path.join(ROOT, 'genstark_amd', 'csrc', 'gf_wide.h'))
    for bits, p in ((256, _abi.MODULUS_256), (224, _abi.MODULUS_224)):
        so = str(tmp_path / f'w_{bits}.so')
        subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-Wno-unknown-pragmas', f'-DGS_WIDE_BITS={bits}', '-o', so, str(src)])
        lib = ctypes.CDLL(so)
        n = 20000
        edge = [0, 1, 2, p - 1, p - 2, 2**128 - 1, 2**128, 1 << (bits - 1), p >> 1, 2**32 - 1, p - 2**32, p - 2**96]
        a = [rng.choice(edge) if rng.random() < 0.3 else rng.randrange(p) for _ in range(n)]
        b = [rng.choice(edge) if rng.random() < 0.3 else rng.randrange(p) for _ in range(n)]
        pack = lambda v: b''.join(x.to_bytes(32, 'little') for x in v)
        outs = [ctypes.create_string_buffer(32 * n) for _ in range(4)]
        lib.ops(pack(a), pack(b), ctypes.c_uint64(n), *outs)
        dec = lambda o, m=n: [int.from_bytes(o.raw[32 * i:32 * i + 32], 'little') for i in range(m)]
        assert dec(outs[0]) == [(x + y) % p for x, y in zip(a, b)]
        assert dec(outs[1]) == [(x - y) % p for x, y in zip(a, b)]
        assert dec(outs[2]) == [x * y % p for x, y in zip(a, b)]
        assert dec(outs[3], 200) == [pow(x, p - 2, p) for x in a[:200]]


def test_product_refuses_the_oracle_backend(oracle_backend):
    from conftest import ORACLE_LIB
    with pytest.raises(_abi.GstarkError, match='refusing backend'):
        _abi.Backend(lib_path=ORACLE_LIB)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_abi.GstarkError, match='no CPU fallback|not found'):
        _abi.Backend(lib_path=str(tmp_path / 'nope.so'))


@pytest.mark.parametrize('compiler', ['g++', '/opt/rocm/lib/llvm/bin/clang++'])
def test_device_field_header_on_host(tmp_path, compiler, rng):
    """genstark_amd/csrc/gf128.h (the arithmetic every kernel uses) compiled for the host: both the
    portable carry path (g++) and the __builtin_addc path the device build takes (clang)."""
    if not os.path.exists(compiler) and compiler != 'g++':
        pytest.skip('ROCm clang not present')
    so = str(tmp_path / 'gf128_host.so')
    subprocess.check_call([compiler, '-O2', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'tests', 'host_harness', 'gf128_host.cpp')])
    lib = ctypes.CDLL(so)
    n = 20000
    a, b = rand_elements(rng, n, edge=0.3), rand_elements(rng, n, edge=0.3)
    A, B = to_bytes(a), to_bytes(b)

    def run(fn, count, *bufs):
        o = ctypes.create_string_buffer(16 * count)
        getattr(lib, fn)(*bufs, o, count)
        return [int.from_bytes(o.raw[16 * i:16 * i + 16], 'little') for i in range(count)]

    assert run('h_mul', n, A, B) == [x * y % P for x, y in zip(a, b)]
    assert run('h_add', n, A, B) == [(x + y) % P for x, y in zip(a, b)]
    assert run('h_sub', n, A, B) == [(x - y) % P for x, y in zip(a, b)]
    m = 300
    assert run('h_inv', m, A[:16 * m]) == [pow(x, P - 2, P) for x in a[:m]]
    assert run('h_pow', m, A[:16 * m], B[:16 * m]) == [pow(x, y, P) for x, y in zip(a[:m], b[:m])]


def test_prover_library_exports_its_header():
    """include/gstark_prover.h <-> libgstark_prover.so: the two entry points exist, the header is valid C, and the driver refuses
    to run unbound."""
    import ctypes as C
    import re
    import subprocess
    from genstark_amd.native import PROVER_LIB_PATH
    header = os.path.join(ROOT, 'include', 'gstark_prover.h')
    subprocess.check_call(['gcc', '-fsyntax-only', '-x', 'c', '-std=c11', header])
    names = set(re.findall(r'^int\s+(gs_prover_\w+)\s*\(', open(header).read(), flags=re.M))
    assert names == {'gs_prover_bind', 'gs_prover_open', 'gs_prover_element_size', 'gs_prover_abi_version', 'gs_prover_prove', 'gs_prover_prove_on', 'gs_prover_last_stats',
                     'gs_prover_remainder_check', 'gs_prover_remainder_check_on', 'gs_prover_verify', 'gs_prover_verify_on', 'gs_prover_input_layout'}
    if not os.path.exists(PROVER_LIB_PATH):
        pytest.skip('libgstark_prover.so not built')
    import shutil
    import tempfile
    private = os.path.join(tempfile.mkdtemp(), 'libgstark_prover_unbound.so')       # a fresh image: never bound
    shutil.copy(PROVER_LIB_PATH, private)
    lib = C.CDLL(private)
    for name in names:
        assert hasattr(lib, name)
    lib.gs_prover_prove.restype = C.c_int
    n = C.c_uint64()
    assert lib.gs_prover_prove(None, None, None, C.c_uint64(0), C.byref(n), None, C.c_uint64(0)) != 0
    assert lib.gs_prover_bind(None) != 0
    assert lib.gs_prover_element_size() == 16 and lib.gs_prover_abi_version() == 2
    # one build of the driver per field flavour; each refuses an ABI library of another field (here: the 128-bit oracle)
    from genstark_amd.native import PROVER_LIB_PATHS
    from conftest import ORACLE_LIB
    oracle = C.CDLL(ORACLE_LIB)
    for modulus, path in PROVER_LIB_PATHS.items():
        assert os.path.exists(path), f'{path}: run __graft_entry__.build() first'
        drv = C.CDLL(path)
        assert drv.gs_prover_element_size() == (16 if modulus < 2**128 else 32)
        b = C.c_void_p()
        rc = drv.gs_prover_open(C.c_void_p(oracle._handle), C.byref(b))
        assert (rc == 0) == (modulus == P), (modulus, rc)


def test_traffic_tally_is_a_no_op_on_the_oracle(oracle_backend):
    """gs_traffic_enable / gs_traffic_read (include/gstark.h): a measurement aid of the HIP library; the checker launches no kernels."""
    assert oracle_backend.traffic(True).traffic() == {}
    oracle_backend.traffic(False)


@pytest.mark.gpu
def test_traffic_tally_counts_what_the_launches_had_to_move(hip_backend):
    """While enabled, every launch of the hot path adds its algorithmic bytes and work units under its kernel's name (what bench.py joins
    with a kernel trace into roofline.kernels): an NTT pass reads every element once and writes it once; a committed column costs its
    bytes in, one digest per leaf and per node out, one compression each."""
    import ctypes as C
    from genstark_amd.field import PrimeField
    be = hip_backend
    f = PrimeField(backend=be)
    n, rows = 1 << 16, 3
    src, dst = f.newVector(rows * n), f.newVector(rows * n)
    w = f.getRootOfUnity(n).to_bytes(16, 'little')
    be.traffic(True)
    be.call('gs_eval_polys_at_roots', C.c_void_p(src.ptr), rows, n, w, n, C.c_void_p(dst.ptr))
    tally = be.traffic()
    ntt = {k: v for k, v in tally.items() if k.startswith('k_ntt_')}
    passes = sum(v['launches'] for v in ntt.values())
    assert passes == 2 and sum(v['bytes'] for v in ntt.values()) == passes * rows * 2 * n * 16 and sum(v['units'] for v in ntt.values()) == passes * rows * n
    be.traffic(True)                                           # a fresh tally
    leaves, nodes = be.alloc(32 * n), be.alloc(32 * n)
    cols = (C.c_void_p * 1)(src.ptr)
    be.call('gs_merkle_commit_rows', 1, cols, 1, n, C.c_void_p(leaves), C.c_void_p(nodes))
    be.sync()
    tally = be.traffic()
    be.traffic(False)
    assert all('merkle' in k for k in tally) and tally
    assert sum(v['units'] for v in tally.values()) == n + n - 1            # one compression per 16-byte leaf, one per node
    assert sum(v['bytes'] for v in tally.values()) >= n * (16 + 32) + (n - 1) * 32
    be.free(leaves); be.free(nodes)
    assert be.traffic() == tally                                # reading does not clear; enable(1) does
    assert be.traffic(True).traffic() == {}
    be.traffic(False)
