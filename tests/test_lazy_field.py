"""genstark_amd/csrc/gf128_lazy.h — the five-limb radix-2^26 "lazy" field arithmetic of the NTT butterfly networks — compiled
for the host and checked against Python integers: values (mod p), the documented limb bounds of every result, and the
worst-case inputs the kernels can produce (sums and differences of up to 16 near-normalised values).  CPU only."""
import ctypes
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2**128 - 9 * 2**32 + 1
B = 1 << 26


def value(l):
    return sum(int(x) * B**i for i, x in enumerate(l))


def is_nn(l):
    return (-(1 << 8) < l[0] < B + (1 << 8) and -(1 << 17) < l[1] < B + (1 << 17) and 0 <= l[2] < B and 0 <= l[3] < B
            and 0 <= l[4] < (1 << 24))


def rand_nn(rng, extreme):
    """a near-normalised limb vector; `extreme` pushes every limb to one end of its allowed interval"""
    if extreme:
        return [rng.choice([-(1 << 8) + 1, B + (1 << 8) - 1]), rng.choice([-(1 << 17) + 1, B + (1 << 17) - 1]),
                rng.choice([0, B - 1]), rng.choice([0, B - 1]), rng.choice([0, (1 << 24) - 1])]
    return [rng.randrange(-(1 << 8) + 1, B + (1 << 8)), rng.randrange(-(1 << 17) + 1, B + (1 << 17)), rng.randrange(B), rng.randrange(B),
            rng.randrange(1 << 24)]


def lazy_combo(rng, terms, extreme):
    """sum of `terms` NN values with random signs (what a butterfly network hands to a product or to lz_pack)"""
    acc = [0] * 5
    if extreme:          # same sign pattern per limb: the worst case for every column at once
        sign = [rng.choice([-1, 1]) for _ in range(5)]
        for _ in range(terms):
            v = rand_nn(rng, True)
            for i in range(5):
                hi = max(abs(v[i]), 1)
                acc[i] += sign[i] * (hi if rng.random() < 0.9 else v[i])
        return acc
    for _ in range(terms):
        v = rand_nn(rng, False)
        s = rng.choice([-1, 1])
        for i in range(5):
            acc[i] += s * v[i]
    return acc


@pytest.fixture(scope='module', params=['g++', '/opt/rocm/lib/llvm/bin/clang++'])
def lib(request, tmp_path_factory):
    compiler = request.param
    if compiler != 'g++' and not os.path.exists(compiler):
        pytest.skip('ROCm clang not present')
    so = str(tmp_path_factory.mktemp('lazy') / 'gf128_lazy_host.so')
    subprocess.check_call([compiler, '-O2', '-shared', '-fPIC', '-fwrapv', '-o', so, os.path.join(ROOT, 'tests', 'host_harness', 'gf128_lazy_host.cpp')])
    return ctypes.CDLL(so)


def limbs_buf(rows):
    flat = [x for r in rows for x in r]
    return (ctypes.c_int32 * len(flat))(*flat)


def out_limbs(n):
    return (ctypes.c_int32 * (5 * n))()


def rows_of(buf, n):
    return [list(buf[5 * i:5 * i + 5]) for i in range(n)]


def elements(rng, n):
    edge = [0, 1, 2, P - 1, P - 2, (1 << 64) - 1, 1 << 64, (1 << 104) - 1, 1 << 104, (1 << 127), P - (1 << 32), 9 * 2**32 - 1, 9 * 2**32 - 2]
    return [rng.choice(edge) if rng.random() < 0.2 else rng.randrange(P) for _ in range(n)]


def test_unpack_pack_roundtrip(lib):
    rng = random.Random(1)
    n = 5000
    a = elements(rng, n)
    raw = b''.join(x.to_bytes(16, 'little') for x in a)
    o = out_limbs(n)
    lib.z_unpack(raw, o, n)
    rows = rows_of(o, n)
    assert all(value(r) == x and is_nn(r) and r[0] >= 0 and r[1] < B for r, x in zip(rows, a))
    back = ctypes.create_string_buffer(16 * n)
    lib.z_pack(limbs_buf(rows), back, n)
    assert back.raw == raw


@pytest.mark.parametrize('terms', [1, 2, 4, 8, 16])
def test_pack_is_canonical_for_every_lazy_value(lib, terms):
    rng = random.Random(2 + terms)
    n = 4000
    rows = [lazy_combo(rng, terms, extreme=(i % 3 == 0)) for i in range(n)]
    # values whose reduction sits right at the wrap-around points: r + t*C just below / above 2^128, r >= p
    for k in range(200):
        t = rng.randrange(0, 40)
        target = rng.choice([P - 1, P, P + 1, 2**128 - 1, 2**128, 2**128 + 1, 0, 1, 9 * 2**32 - 1]) + rng.randrange(-3, 4)
        v = target + t * P if target + t * P >= 0 else target + (t + 1) * P
        l = [v % B, (v >> 26) % B, (v >> 52) % B, (v >> 78) % B, v >> 104]
        if l[4] < (1 << 28):
            rows[k] = l
    back = ctypes.create_string_buffer(16 * n)
    lib.z_pack(limbs_buf(rows), back, n)
    got = [int.from_bytes(back.raw[16 * i:16 * i + 16], 'little') for i in range(n)]
    assert got == [value(r) % P for r in rows]
    # the weak form (intermediate passes): any representative below 2^128; unpacking it gives exactly normalised limbs again
    lib.z_pack_weak(limbs_buf(rows), back, n)
    weak = [int.from_bytes(back.raw[16 * i:16 * i + 16], 'little') for i in range(n)]
    assert [w % P for w in weak] == got
    o = out_limbs(n)
    lib.z_unpack(back.raw, o, n)
    assert all(value(r) == w and is_nn(r) for r, w in zip(rows_of(o, n), weak))


@pytest.mark.parametrize('terms', [1, 4, 16])
def test_norm(lib, terms):
    rng = random.Random(5 + terms)
    n = 4000
    rows = [lazy_combo(rng, terms, extreme=(i % 2 == 0)) for i in range(n)]
    o = out_limbs(n)
    lib.z_norm(limbs_buf(rows), o, n)
    for r, y in zip(rows, rows_of(o, n)):
        assert value(y) % P == value(r) % P and is_nn(y), (r, y)


@pytest.mark.parametrize('terms', [1, 2, 4, 6])
def test_products(lib, terms):
    """both product forms on inputs as large as the networks make them (a difference of two sums of `terms`/2 values),
    multipliers at the ends of their intervals included"""
    rng = random.Random(11 + terms)
    n = 6000
    xs = [lazy_combo(rng, terms, extreme=(i % 2 == 0)) for i in range(n)]
    ws = elements(rng, n)
    wraw = b''.join(w.to_bytes(16, 'little') for w in ws)
    o = out_limbs(n)
    lib.z_mul_u(limbs_buf(xs), wraw, o, n)
    for x, w, y in zip(xs, ws, rows_of(o, n)):
        assert value(y) % P == value(x) * w % P and is_nn(y), ('mul_u', x, w, y)
    wl = [rand_nn(rng, extreme=(i % 2 == 1)) for i in range(n)]
    lib.z_mul_v(limbs_buf(xs), limbs_buf(wl), o, n)
    for x, w, y in zip(xs, wl, rows_of(o, n)):
        assert value(y) % P == value(x) * value(w) % P and is_nn(y), ('mul_v', x, w, y)
    lib.z_mul_u_rows(limbs_buf(xs), limbs_buf(wl), o, n)
    for x, w, y in zip(xs, wl, rows_of(o, n)):
        assert value(y) % P == value(x) * value(w) % P and is_nn(y), ('mul_u_rows', x, w, y)


def test_square_and_exponentiation_chain(lib):
    """lz_sqr on NN inputs (ends of the limb intervals included) and k squarings + one product with no packing in between — what the
    long exponentiations of compiled AIR programs are made of (csrc/air_jit.hip: emit_pow)"""
    rng = random.Random(123)
    n = 6000
    xs = [rand_nn(rng, extreme=(i % 2 == 0)) for i in range(n)]
    o = out_limbs(n)
    lib.z_sqr(limbs_buf(xs), o, n)
    for x, y in zip(xs, rows_of(o, n)):
        assert value(y) % P == value(x) ** 2 % P and is_nn(y), ('sqr', x, y)
    for k in (1, 5, 31):
        lib.z_sqr_chain(limbs_buf(xs[:1500]), k, o, 1500)
        for x, y in zip(xs[:1500], rows_of(o, 1500)):
            assert value(y) % P == pow(value(x), 2 ** k + 1, P) and is_nn(y), ('chain', k, x, y)


def test_product_of_unnormalised_sums_by_signed_digit_multiplier(lib):
    """the exchange stage of the pass kernel: network outputs (sums of up to 9 NN values) times table entries recoded to signed
    digits |limb| <= 2^25 (k_build_lz_table), no normalisation in between"""
    rng = random.Random(77)
    n = 6000
    xs = [lazy_combo(rng, 9, extreme=(i % 2 == 0)) for i in range(n)]
    ws = []
    for i in range(n):
        w = rng.randrange(P) if i % 5 else rng.choice([P - 1, (1 << 128) - (1 << 103), sum(((1 << 25) - 1) << (26 * k) for k in range(5)) % P,
                                                         sum((1 << 25) << (26 * k) for k in range(4))])
        l = [w % B, (w >> 26) % B, (w >> 52) % B, (w >> 78) % B, w >> 104]
        for k in range(4):
            if l[k] >= 1 << 25:
                l[k] -= 1 << 26
                l[k + 1] += 1
        assert value(l) == w and all(abs(t) <= 1 << 25 for t in l)
        ws.append(l)
    o = out_limbs(n)
    lib.z_mul_v(limbs_buf(xs), limbs_buf(ws), o, n)
    for x, w, y in zip(xs, ws, rows_of(o, n)):
        assert value(y) % P == value(x) * value(w) % P and is_nn(y), (x, w, y)
    # interval check of the columns for this class of inputs
    nn_hi = [B + (1 << 8), B + (1 << 17), B, B, 1 << 24]
    x9 = [9 * h for h in nn_hi]
    cols = [sum(x9[i] * (1 << 25) for i in range(5) if 0 <= k - i < 5) for k in range(9)]
    assert max(cols) + 2 * (2304 * (1 << 31) + 147456 * (1 << 25) + (1 << 40)) < (1 << 57) - (1 << 44)


def test_montgomery_product(lib):
    """lz_mul_vm: x * w * B^-5 by REDC from the bottom (p == 1 mod B: the quotient digit is -(c_i mod B)); with the multiplier stored
    premultiplied by R = 2^130 the result is x * w.  Every input class the pass kernels feed it: exact NN data times NN multipliers,
    sums of up to 9 NN values times signed-digit table entries (the exchange stage), extreme limbs; result NN, value exact."""
    rng = random.Random(4242)
    R = pow(2, 130, P)
    Rinv = pow(R, -1, P)
    buf = ctypes.create_string_buffer(32)
    lib.z_mont_consts(buf)
    assert int.from_bytes(buf.raw[:16], 'little') == R and int.from_bytes(buf.raw[16:], 'little') == R * R % P
    n = 6000
    o = out_limbs(n)
    for terms in (1, 2, 4, 9):
        xs = [lazy_combo(rng, terms, extreme=(i % 2 == 0)) for i in range(n)]
        wl = [rand_nn(rng, extreme=(i % 2 == 1)) for i in range(n)]
        if terms == 9:           # signed-digit multipliers, |limb| <= 2^25, as k_build_lz_table stores the exchange twiddles
            wl = []
            for i in range(n):
                w = rng.randrange(P) if i % 5 else rng.choice([P - 1, (1 << 128) - (1 << 103), sum(((1 << 25) - 1) << (26 * k) for k in range(5)) % P])
                l = [w % B, (w >> 26) % B, (w >> 52) % B, (w >> 78) % B, w >> 104]
                for k in range(4):
                    if l[k] >= 1 << 25:
                        l[k] -= 1 << 26
                        l[k + 1] += 1
                wl.append(l)
        lib.z_mul_vm(limbs_buf(xs), limbs_buf(wl), o, n)
        for x, w, y in zip(xs, wl, rows_of(o, n)):
            assert value(y) % P == value(x) * value(w) * Rinv % P and is_nn(y), ('mul_vm', terms, x, w, y)
    # premultiplied multiplier: the ordinary product comes out
    ws = elements(rng, n)
    xs = [lazy_combo(rng, 2, extreme=False) for _ in range(n)]
    wl = []
    for w in ws:
        v = w * R % P
        wl.append([v % B, (v >> 26) % B, (v >> 52) % B, (v >> 78) % B, v >> 104])
    lib.z_mul_vm(limbs_buf(xs), limbs_buf(wl), o, n)
    for x, w, y in zip(xs, ws, rows_of(o, n)):
        assert value(y) % P == value(x) * w % P
    # column interval: the widest input class (9 NN values x signed digits) leaves the 2^51 of slack the REDC steps need
    nn_hi = [B + (1 << 8), B + (1 << 17), B, B, 1 << 24]
    x9 = [9 * h for h in nn_hi]
    cols = [sum(x9[i] * (1 << 25) for i in range(5) if 0 <= k - i < 5) for k in range(9)]
    assert max(cols) < (1 << 57) - (1 << 51)
    x4 = [4 * h for h in nn_hi]
    cols = [sum(x4[i] * nn_hi[k - i] for i in range(5) if 0 <= k - i < 5) for k in range(9)]
    assert max(cols) < (1 << 57) - (1 << 51)


def test_shift_limb(lib):
    rng = random.Random(3)
    n = 4000
    rows = [rand_nn(rng, extreme=(i % 2 == 0)) for i in range(n)]
    o = out_limbs(n)
    lib.z_shift(limbs_buf(rows), o, n)
    for r, y in zip(rows, rows_of(o, n)):
        assert value(y) % P == value(r) * B % P and is_nn(y)


def test_column_bounds_of_the_products():
    """interval arithmetic over the documented input classes: every 64-bit column stays below 2^57 in magnitude (so that a
    carry fits 32 bits) for any combination of 6 NN values with a tabulated multiplier, and for a difference of two sums of
    two NN values (what a radix-16 DIF network multiplies at its third level) with a per-lane multiplier."""
    nn_hi = [B + (1 << 8), B + (1 << 17), B, B, 1 << 24]
    x = [6 * h for h in nn_hi]                                  # |limb| of a sum/difference of 6 NN values
    w_can = [B, B, B, B, 1 << 24]                               # canonical multiplier rows (W-form)
    col_u = [sum(x[i] * w_can[j] for i in range(5)) for j in range(5)]
    fold = 2304 * (1 << 31) + 147456 * (1 << 25) + (1 << 40)
    assert max(col_u) + fold < (1 << 57) - (1 << 44)
    x4 = [4 * h for h in nn_hi]                                 # what the DIF network really multiplies: 4 values apart
    col_v = [sum(x4[i] * nn_hi[k - i] for i in range(5) if 0 <= k - i < 5) for k in range(9)]
    assert max(col_v) + 2 * fold < (1 << 57) - (1 << 44)


def test_fp64_fma_product_experiment(tmp_path):
    """tools/gf128_f64.h (an experiment beside the library): the radix-2^43 product on fp64 FMAs — values mod p and
    the normalised-limb bounds, for normalised multipliers and multiplicands as lazy as a radix-16 network leaves them (|limb| < 2^47)."""
    so = str(tmp_path / 'gf128_f64_host.so')
    subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'tests', 'host_harness', 'gf128_f64_host.cpp')])
    lib = ctypes.CDLL(so)
    R = 1 << 43
    rng = random.Random(43)
    D3 = ctypes.c_double * 3

    def val(l):
        assert all(float(x).is_integer() for x in l)
        return sum(int(x) * R**i for i, x in enumerate(l))

    def mul(x, w):
        out = D3()
        lib.fz_mul_host(D3(*[float(v) for v in x]), D3(*[float(v) for v in w]), out)
        return list(out)

    for case in range(20000):
        extreme = case % 4 == 0
        if extreme:
            w = [rng.choice([-(R // 2), R // 2, R - 1, 0]) for _ in range(3)]
            x = [rng.choice([-1, 1]) * ((1 << 47) - 1 - rng.randrange(4)) for _ in range(3)]
        else:
            w = [rng.randrange(R) for _ in range(2)] + [rng.randrange(1 << 42)]
            x = [rng.randrange(-(1 << 47) + 1, 1 << 47) for _ in range(3)]
        y = mul(x, w)
        assert val(y) % P == val(x) * val(w) % P, (x, w, y)
        assert abs(y[0]) < 2 * R and abs(y[1]) <= R // 2 and abs(y[2]) <= R // 2, y
        z = mul(y, w)                                  # a result is a valid multiplicand AND (signed digits) a valid multiplier
        assert val(z) % P == val(y) * val(w) % P
        z2 = mul(x, y)
        assert val(z2) % P == val(x) * val(y) % P
    words = (ctypes.c_uint32 * 4)()
    for _ in range(2000):
        v = rng.randrange(P)
        for i in range(4):
            words[i] = (v >> (32 * i)) & 0xffffffff
        out = D3()
        lib.fz_unpack_host(words, out)
        assert val(list(out)) == v and all(0 <= x < R for x in out)
