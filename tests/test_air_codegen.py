"""The code generator of csrc/air_jit.hip, checked where there is no GPU: gs_air_jit_check turns an AIR program into HIP source and
runs the compiler (hiprtc cross-compiles gfx950 without a device) — the "does it build" tier for what a serving prover compiles at
run time.  The values the compiled programs compute are compared with the interpreter's in the gpu tier
(test_generic_air.py::test_compiled_air_programs_equal_interpreted, test_pipeline.py)."""
import ctypes as C
import os

import pytest

from genstark_amd import lib224, poseidon
from genstark_amd._abi import GS_OK, HIP_LIB_PATHS, MODULUS_128, MODULUS_224, load_library
from genstark_amd.pointmul import point_mul_air
from genstark_amd.field import PrimeField
from test_generic_air import rescue4x128_air
from test_wide_fields import oracle_for


@pytest.fixture(scope='module')
def hip_libs():
    """The product libraries, loaded without a context (no device needed for the compiler)."""
    return {m: load_library(HIP_LIB_PATHS[m]) for m in (MODULUS_128, MODULUS_224)}


def builds(air, lib):
    results = air.compileCheck(lib)
    assert [name for name, _, _ in results] == (['trace', 'constraints'] if air.segmentLength is not None else ['constraints'])
    for name, ok, log in results:
        assert ok, f'{name} program does not build:\n{log}'


def test_example_air_programs_build_for_gfx950(oracle_backend, hip_libs):
    f = PrimeField(backend=oracle_backend)              # the AIR objects only need a field to reduce their constants with
    builds(rescue4x128_air(1024, 16, f, segmented=True), hip_libs[MODULUS_128])      # examples/rescue/hash4x128.ts
    builds(poseidon.poseidon6x128_air(2048, 16, f, segmented=True), hip_libs[MODULUS_128])   # examples/poseidon/hash6x128.ts
    builds(rescue4x128_air(64, 16, f), hip_libs[MODULUS_128])                        # unsegmented: constraints only


def test_wide_field_air_programs_build_for_gfx950(hip_libs):
    f = PrimeField(backend=oracle_for('p224'))
    builds(point_mul_air(f), hip_libs[MODULUS_224])                                  # examples/elliptic/pointmul.aa (divisions: POWC)
    builds(lib224.verify_schnorr_signature_air(f), hip_libs[MODULUS_224])            # assembly/lib224.aa, 14 registers


def test_unknown_instruction_is_refused_with_a_reason(hip_libs):
    lib = hip_libs[MODULUS_128]
    log = C.create_string_buffer(256)
    bad = (C.c_uint32 * 4)(99, 0, 0, 0)
    assert lib.gs_air_jit_check(1, bad, 1, None, 0, None, 0, 4, 1, None, 0, log, len(log)) != GS_OK
    assert b'instruction' in log.value
    assert lib.gs_air_jit_check(1, None, 0, None, 0, None, 0, 4, 1, None, 0, log, len(log)) != GS_OK       # no program


def generated_trace_source(air, lib, tmp_path, monkeypatch):
    tmp_path.mkdir(parents=True, exist_ok=True)
    """The HIP source the generator writes for the AIR's trace program (GSTARK_AIR_JIT_DUMP keeps it)."""
    monkeypatch.setenv('GSTARK_AIR_JIT_DUMP', str(tmp_path))
    builds(air, lib)
    files = sorted(p for p in tmp_path.iterdir() if p.name.startswith('gs_jit_trace_'))
    assert files, 'no trace kernel was generated'
    return files[-1].read_text()


def test_trace_kernels_spread_products_over_lanes(oracle_backend, hip_libs, tmp_path, monkeypatch):
    """Shape of the generated trace kernels (csrc/air_jit.hip): products of one depth share a round over the lanes of
    a segment's group; S-box layers are one member per lane; operands are picked limb-wise (never `if (sub == i) x = v`, which the
    compiler turns into scratch traffic); the lanes exchange results through LDS."""
    f = PrimeField(backend=oracle_backend)
    src = generated_trace_source(poseidon.poseidon6x128_air(2048, 16, f, segmented=True), hip_libs[MODULUS_128], tmp_path / 'p', monkeypatch)
    # the 6 x 6 MDS layer is SIX fused rows (ssa_fuse_dots: a row = one lane's 6 x 25 v_mad + one fold), not 36 products over 16 lanes
    assert '#define GS_LANES 8u' in src
    body = src[src.index('for (unsigned long long k = 0'):]
    assert body.count('gs_mul(') + body.count('gs_sqr(') == 3   # per lane and step: the x^5 chain; the MDS is a dot, and ...
    # ... the full / partial round flag (a static register of zeros and ones) selects instead of multiplying, on every lane, between
    # the two rounds: a step is TWO exchanges (S-box layer, MDS rows)
    assert body.count('gs_blend(v') == 5 and body.count('gs_swap[threadIdx.x] =') == 2
    assert body.count('gs_dot_acc(') == 6 and body.count('gs_dot_end(') == 1 and 'gs_pick' not in body[body.index('gs_dot_begin'):body.index('gs_dot_end')]
    assert 'gs_pick(sub ==' in body and 'gs_swap[threadIdx.x]' in body
    assert [line for line in body.split('\n') if 'if (sub ==' in line and 'out[' not in line] == []
    head = src[:src.index('for (unsigned long long k = 0')]
    assert head.count('gs_dotk_make(vd') == 6 and 'sub == 5u ? consts[' in head          # per-lane MDS constants in W-form, built before the step loop
    assert body.count('fe_add(') <= 14                          # the row sums are inside the dots: no tree of 30 additions on every lane

    src = generated_trace_source(rescue4x128_air(1024, 16, f, segmented=True), hip_libs[MODULUS_128], tmp_path / 'r', monkeypatch)
    assert '#define GS_LANES 4u' in src or '#define GS_LANES 8u' in src      # 4 x 4 MDS layers as four fused rows each; four S-box members
    body = src[src.index('for (unsigned long long k = 0'):]
    assert body.count('#pragma nounroll') >= 1                  # the inverse S-box: squaring runs of the fixed addition chain
    assert body.count('gs_swap[threadIdx.x] = x;') >= 2         # S-box layers (x^3 and x^(1/3)): one member per lane
    assert body.count('gs_dot_end(') == 2 and body.count('gs_dot_acc(') >= 8      # two MDS layers per step (the rows' additive constants ride along as terms by one)


def test_point_multiplication_inversions_share_a_round(hip_libs, tmp_path, monkeypatch):
    f = PrimeField(backend=oracle_for('p224'))
    tmp_path.mkdir(exist_ok=True)
    src = generated_trace_source(point_mul_air(f), hip_libs[MODULUS_224], tmp_path, monkeypatch)
    assert '#define GS_LANES 4u' in src                         # three independent inversions per step: the two slopes and the one of
    body = src[src.index('for (unsigned long long k = 0'):]     # the next multiplication's first row
    assert body.count('const fe p3 = gs_mul(p1, p2);') == 1     # ONE Fermat chain in the step: the inversions share a round


def test_small_field_constraint_program_builds_for_gfx950():
    """The generator in a small-field flavour (64-bit elements in 16-byte slots): the constraint program of Rescue 2x64."""
    import os
    from conftest import ROOT, _build_oracle
    from genstark_amd._abi import Backend, MODULUS_64
    from genstark_amd.rescue import rescue2x64_air
    _build_oracle()
    f = PrimeField(backend=Backend(lib_path=os.path.join(ROOT, 'oracle', 'liboracle_q64.so'), allow_test_double=True))
    builds(rescue2x64_air(32, 16, f), load_library(HIP_LIB_PATHS[MODULUS_64]))


def _probe(env_extra, drop=()):
    """genstark_amd/csrc/air_jit.hip: gs_jit_cache_path_probe in a child process with a controlled environment."""
    import subprocess
    import sys
    from genstark_amd import _abi
    code = ('import ctypes, sys\nlib = ctypes.CDLL(%r)\nbuf = ctypes.create_string_buffer(4096)\n'
            'assert lib.gs_jit_cache_path_probe(b"source", buf, ctypes.c_uint64(4096)) == 0\nprint("PATH=" + buf.value.decode())\n') % _abi.HIP_LIB_PATH
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.startswith('PATH=')][0][5:]


def test_code_object_cache_directory_must_be_private(tmp_path):
    """a code object found in the cache is loaded into the proving path, so the cache directory must belong to the user:
    created 0700; a directory writable by group / others, a symlink, or no HOME and no GSTARK_JIT_CACHE_DIR -> no cache at all."""
    import stat
    fresh = tmp_path / 'fresh' / 'cache'
    path = _probe({'GSTARK_JIT_CACHE_DIR': str(fresh)})
    assert path.startswith(str(fresh) + '/') and path.endswith('.hsaco')
    assert stat.S_IMODE(os.stat(fresh).st_mode) == 0o700
    shared = tmp_path / 'shared'
    shared.mkdir()
    os.chmod(shared, 0o777)
    assert _probe({'GSTARK_JIT_CACHE_DIR': str(shared)}) == ''
    os.chmod(shared, 0o775)
    assert _probe({'GSTARK_JIT_CACHE_DIR': str(shared)}) == ''
    os.chmod(shared, 0o755)
    assert _probe({'GSTARK_JIT_CACHE_DIR': str(shared)}) != ''
    link = tmp_path / 'link'
    os.symlink(fresh, link)
    assert _probe({'GSTARK_JIT_CACHE_DIR': str(link)}) == ''
    assert _probe({}, drop=('HOME', 'GSTARK_JIT_CACHE_DIR')) == ''
    home = tmp_path / 'home'
    home.mkdir()
    assert _probe({'HOME': str(home)}, drop=('GSTARK_JIT_CACHE_DIR',)).startswith(str(home) + '/.cache/gstark_jit/')
    assert _probe({'GSTARK_JIT_CACHE_DIR': 'off'}) == ''


HELPER_SCRIPT = r'''
import ctypes, os, sys
lib = ctypes.CDLL(sys.argv[1])
lib.gs_jit_helper_probe.restype = ctypes.c_int
good = b'#define GS_SMALL_Q 18446744051160973313ull\n#include "gs_field.h"\nextern "C" __global__ void gs_probe(fe *o) { o[0] = fe_add(o[1], o[2]); }\n' if sys.argv[3] == 'plain' else \
       b'#include "gf128_lazy.h"\nextern "C" __global__ void gs_probe(fe *o) { o[0] = fe_add(o[1], o[2]); }\n'
out = ctypes.create_string_buffer(1 << 20)
size = ctypes.c_uint64(0)
r = lib.gs_jit_helper_probe(good, b'gs_probe', sys.argv[2].encode(), out, ctypes.c_uint64(len(out)), ctypes.byref(size))
print('GOOD', r, size.value, out.raw[:4] == b'\x7fELF', os.path.exists(sys.argv[2]) and os.path.getsize(sys.argv[2]) == size.value)
r = lib.gs_jit_helper_probe(b'this is not HIP\n', b'gs_probe', b'', out, ctypes.c_uint64(len(out)), ctypes.byref(size))
print('BAD', r, b'error' in out.raw[:size.value])
'''


def test_background_builds_run_in_the_helper_process(tmp_path):
    """genstark_amd/csrc/jitc.cc: the default (auto) mode never compiles inside the proving process — a host that exits during a
    build would have the compiler's statics destroyed under the builder thread (seen on the GPU box: 'LLVM ERROR' + abort at exit).
    One program through gstark_jitc: a gfx950 code object comes back AND is in the cache; a source the compiler refuses comes back
    as a log; a missing helper is reported as such (-1: the caller then compiles in-process)."""
    import subprocess
    import sys
    from genstark_amd import _abi
    helper = os.path.join(os.path.dirname(_abi.HIP_LIB_PATH), 'gstark_jitc')
    assert os.access(helper, os.X_OK), 'genstark_amd/csrc/build.sh builds gstark_jitc next to the libraries'
    for lib_path, kind in ((_abi.HIP_LIB_PATH, 'lazy'), (HIP_LIB_PATHS[_abi.MODULUS_64], 'plain')):
        cache = tmp_path / ('probe_%s.hsaco' % kind)
        r = subprocess.run([sys.executable, '-c', HELPER_SCRIPT, lib_path, str(cache), kind], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = dict((l.split()[0], l.split()[1:]) for l in r.stdout.splitlines() if l.split() and l.split()[0] in ('GOOD', 'BAD'))
        assert lines['GOOD'][0] == '1' and int(lines['GOOD'][1]) > 1000 and lines['GOOD'][2:] == ['True', 'True'], r.stdout
        assert lines['BAD'] == ['0', 'True'], r.stdout
    r = subprocess.run([sys.executable, '-c', HELPER_SCRIPT, _abi.HIP_LIB_PATH, str(tmp_path / 'none.hsaco'), 'lazy'],
                       env=dict(os.environ, GSTARK_JITC=str(tmp_path / 'no_such_helper')), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'GOOD -1' in r.stdout, r.stdout + r.stderr[-1000:]


def chain_exponent(block):
    """The exponent a generated lazy-form chain computes, by following its statements with integers (x = p1 has exponent 1)."""
    import re
    val = {'p1': 1}
    lines = block.split('\n')
    i = 0
    while i < len(lines):
        ln = lines[i].strip()
        i += 1
        m = re.match(r'for \(int q = 0; q < (\d+); q\+\+\) (\w+) = lz_sqr\((\w+), K\);', ln)
        if m:
            assert m.group(2) == m.group(3)
            val[m.group(2)] <<= int(m.group(1))
            continue
        m = re.match(r'(?:const lz |lz )?(\w+) = lz_sqr\((\w+), K\);', ln)
        if m:
            val[m.group(1)] = 2 * val[m.group(2)]
            continue
        m = re.match(r'(?:const lz |lz )?(\w+) = lz_mul_v\((\w+), (\w+), K\);', ln)
        if m:
            val[m.group(1)] = val[m.group(2)] + val[m.group(3)]
            continue
        m = re.match(r'(?:const lz |lz )?(\w+) = (\w+);$', ln)
        if m and m.group(2) in val:
            val[m.group(1)] = val[m.group(2)]
    return val['acc'], sum(1 for l in lines if 'lz_mul_v(' in l), val


def test_inverse_sbox_chain_uses_the_periodic_pattern(oracle_backend, hip_libs, tmp_path, monkeypatch):
    """Rescue's inverse S-box exponent (2p - 1)/3 = 0xaaaa...a4aaaaaaab: its top 92 bits repeat "10", so r_k = x^("10" k times) by
    doubling (r_{a+b} = r_a^(4^b) r_b) replaces five-bit windows there, and the r_j on the way serve the low bits: the generated chain
    must compute exactly that exponent, with 127 + 1 squarings and a dozen products instead of 32."""
    f = PrimeField(backend=oracle_backend)
    src = generated_trace_source(rescue4x128_air(1024, 16, f, segmented=True), hip_libs[MODULUS_128], tmp_path / 'r', monkeypatch)
    body = src[src.index('for (unsigned long long k = 0'):]
    start = body.index('const lz p1 = lz_unpack(')
    block = body[start:body.index('lz_pack(acc)', start)]
    e = (2 * MODULUS_128 - 1) // 3
    got, products, val = chain_exponent(block)
    assert got == e, hex(got)
    assert 'g46' in val and val['g46'] == int('10' * 46, 2)
    assert products <= 14 and block.count('lz_sqr(') <= 20          # (squaring runs are loops: few call sites, 128 squarings executed)
