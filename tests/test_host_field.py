"""The HOST-side field arithmetic of the library (csrc/host_field.h, host_field_wide.h, host_field_small.h, host_pow.h) against
Python integers, without a GPU: the serial traces (MiMC, single-chain AIRs, AIRs with a few segments) are computed with it on a host
core, and the GPU tier only sees it through whole proofs.  tools/host_field_check.cpp wraps it as a filter."""
import os
import random
import subprocess

import pytest

from conftest import ROOT
from genstark_amd._abi import MODULUS_128, MODULUS_224, MODULUS_256, MODULUS_64

FLAVOURS = {'p128': (MODULUS_128, []), 'p224': (MODULUS_224, ['-DGS_WIDE_BITS=224']), 'p256': (MODULUS_256, ['-DGS_WIDE_BITS=256']),
            'q64': (MODULUS_64, [f'-DGS_SMALL_Q={MODULUS_64}ull'])}


@pytest.mark.parametrize('name', sorted(FLAVOURS))
def test_host_products_inverses_powers(name, tmp_path):
    p, flags = FLAVOURS[name]
    exe = str(tmp_path / f'host_field_check_{name}')
    subprocess.check_call(['g++', '-O2', '-std=c++17', *flags, os.path.join(ROOT, 'tools', 'host_field_check.cpp'), '-o', exe])
    rng = random.Random(name)
    edges = sorted({v % p for v in (0, 1, 2, 3, p - 1, p - 2, p >> 1, (p >> 1) + 1, 2**32 - 1, 2**32, 2**64 - 1, 2**64, 2**96 - 1, 2**96,
                                    2**112, 2**127, 2**128 - 1, 2**160 - 1, 2**192 + 5, 2**223, 2**223 + 2**96, 2**255, p - 2**32, p - 2**96)})
    exps = [0, 1, 2, 3, 5, 0xffff, 0x10000, 0x10001, p - 2, p - 1, (2 * p - 1) // 3, (p - 1) // 2, 2**100 + 1, 0xaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa % p,
            2**127 % p, (2**128 - 1) % p]
    cases = [(a, b, exps[(i + j) % len(exps)]) for i, a in enumerate(edges) for j, b in enumerate(edges)]
    cases += [(rng.randrange(p), rng.randrange(p), rng.choice(exps + [rng.randrange(p)])) for _ in range(1500)]
    text = ''.join(f'{a:064x} {b:064x} {e:064x}\n' for a, b, e in cases)
    out = subprocess.run([exe], input=text, capture_output=True, text=True, check=True).stdout.split('\n')
    assert len(out) >= len(cases)
    for (a, b, e), line in zip(cases, out):
        got = [int(v, 16) for v in line.split()]
        inv = pow(a, p - 2, p)                                          # 0 -> 0, as galois
        m1 = (pow(a, 3, p) + b) % p
        m1 = (pow(m1, 3, p) + b) % p
        if name == 'p128':                      # the 128-bit chain takes ANY 128-bit representative (host_field.h: hf_cube_add_weak)
            m2 = (pow((~a) & (2**128 - 1), 3, p) + b) % p
            m2 = (pow(m2, 3, p) + b) % p
        else:
            m2 = m1
        assert got == [a * b % p, inv, pow(a, e, p), pow(b, e, p), m1, m2], (name, hex(a), hex(b), hex(e))
