"""The layout rules of air-assembly input registers, twice: genstark_amd/airassembly.py `_Layout` (the loader) and csrc/prover.cc
`input_layout` (what the native prove() validates the shapes it serializes with, and what the native verify() sizes the trace with from
the shapes a proof carries — lib/Stark.ts:176).  Thousands of random declarations (childof / peerof chains, steps, shifts, secret flags)
and shapes, valid and invalid: the same trace length, or the same refusal with the same message."""
import ctypes as C
import random

import pytest

from genstark_amd._abi import GstarkError
from genstark_amd.airassembly import _Layout
from genstark_amd.native import PROVER_LIB_PATH, _InputRegister


@pytest.fixture(scope='module')
def driver():
    lib = C.CDLL(PROVER_LIB_PATH)
    lib.gs_prover_input_layout.argtypes = [C.POINTER(_InputRegister), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64]
    lib.gs_prover_input_layout.restype = C.c_int
    return lib


def native_layout(lib, decls, shapes):
    arr = (_InputRegister * max(len(decls), 1))()
    for j, d in enumerate(decls):
        arr[j].parent = -1 if d['parent'] is None else d['parent']
        arr[j].peer = -1 if d['peer'] is None else d['peer']
        arr[j].steps, arr[j].shift, arr[j].secret = d['steps'] or 0, d['shift'], 1 if d['secret'] else 0
    flat = [w for sh in shapes for w in [len(sh)] + list(sh)]
    length, err = C.c_uint64(), C.create_string_buffer(512)
    rc = lib.gs_prover_input_layout(arr, len(decls), (C.c_uint32 * max(len(flat), 1))(*flat), C.byref(length), err, 512)
    return (length.value, None) if rc == 0 else (None, err.value.decode())


def python_layout(decls, shapes):
    statics = [dict(d, kind='input') for d in decls]
    try:
        return _Layout(statics, shapes).length, None
    except GstarkError as e:
        return None, str(e)


def random_case(rng):
    n = rng.randrange(1, 6)
    decls, depth = [], []
    for j in range(n):
        d = {'parent': None, 'peer': None, 'steps': None, 'shift': rng.choice([0, 0, 0, -1, 1, 3]), 'secret': rng.random() < 0.5}
        kind = rng.random()
        if j and kind < 0.35:
            d['parent'] = rng.randrange(j)
        elif j and kind < 0.6:
            d['peer'] = rng.randrange(j)
        if rng.random() < 0.6:
            d['steps'] = rng.choice([1, 2, 4, 8, 16, 3])
        ref = d['parent'] if d['parent'] is not None else d['peer']
        depth.append(0 if ref is None else depth[ref] + (1 if d['parent'] is not None else 0))
        decls.append(d)
    # shapes: mostly consistent with the declarations, sometimes not
    top = rng.choice([1, 2, 4, 8, 3])
    shapes = []
    for j, d in enumerate(decls):
        if d['peer'] is not None and rng.random() < 0.9:
            sh = list(shapes[d['peer']])
        elif d['parent'] is not None and rng.random() < 0.9:
            sh = list(shapes[d['parent']]) + [rng.choice([1, 2, 4, 2, 4, 5])]
        else:
            sh = [top] + [rng.choice([1, 2, 4]) for _ in range(depth[j])]
        if rng.random() < 0.06:
            sh = sh[:-1] if rng.random() < 0.5 else sh + [2]
        if sh and rng.random() < 0.03:
            sh[rng.randrange(len(sh))] = 0
        shapes.append(sh)
    return decls, shapes          # (a missing shape is a count mismatch the flat job encoding cannot express: the verifier's own count
                                  #  check is exercised through proofs, tests/test_airassembly.py and the fuzz tier)


def test_layout_rules_agree_on_random_declarations(driver):
    rng = random.Random(20261001)
    valid = refused = 0
    messages = set()
    for _ in range(6000):
        decls, shapes = random_case(rng)
        want, got = python_layout(decls, shapes), native_layout(driver, decls, shapes)
        assert want == got, (decls, shapes, want, got)
        if want[1] is None:
            valid += 1
        else:
            refused += 1
            messages.add(want[1].split(':')[0][:40])
    assert valid > 500 and refused > 500 and len(messages) >= 6, (valid, refused, messages)


def test_layout_of_the_fixtures(driver):
    # tests/golden/aa/ledger.aa: two top-level secret registers, a public register nested under the first (runs x 4 deposits, 2 steps each)
    decls = [{'parent': None, 'peer': None, 'steps': None, 'shift': 0, 'secret': True}, {'parent': None, 'peer': 0, 'steps': None, 'shift': 0, 'secret': True},
             {'parent': 0, 'peer': None, 'steps': 2, 'shift': -1, 'secret': False}]
    for runs in (1, 2, 4, 4096):
        shapes = [[runs], [runs], [runs, 4]]
        assert native_layout(driver, decls, shapes) == python_layout(decls, shapes) == (8 * runs, None)
    assert native_layout(driver, decls, [[3], [3], [3, 4]])[1] == 'the inputs make a trace of 24 steps: a power of 2 is required'
    assert native_layout(driver, [], []) == (0, None)
    # a register whose values would be held for 0 steps (its child list is empty) while another register lays out a real trace: refused by
    # both — the native verifier would otherwise build a column with a span of 0 steps (found by the randomized comparison above)
    hole = [{'parent': None, 'peer': None, 'steps': None, 'shift': 0, 'secret': False}, {'parent': 0, 'peer': None, 'steps': 2, 'shift': 0, 'secret': True},
            {'parent': None, 'peer': None, 'steps': 4, 'shift': 0, 'secret': False}]
    assert native_layout(driver, hole, [[2], [2, 0], [2]]) == python_layout(hole, [[2], [2, 0], [2]]) == (None, 'input registers imply different trace lengths')
    # products beyond anything provable are refused, not wrapped around
    big = [[1 << 31], [1 << 31], [1 << 31, 1 << 31]]
    assert native_layout(driver, decls, big)[1].startswith('the inputs make a trace of')
