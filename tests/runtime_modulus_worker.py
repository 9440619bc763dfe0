"""tests/runtime_modulus_worker.py <modulus> <oracle|hip|both> — ONE modulus per process (the runtime-modulus libraries fix their field once:
gs_set_modulus, include/gstark.h).  On each backend asked for: every vector member and the NTT against Python integers
(test_small_fields.check_arithmetic), then two STARKs over the field — MiMC (x <- x^3 + k, 64 cyclic constants) and a two-register
degree-5 program AIR — proved by the mirror AND by the native driver's runtime-modulus build (same bytes), verified by the mirror, by
the native verifier and by the device-free verifier (HostField: Python integers).  `both`: the HIP library's bytes == the oracle's.
Prints one JSON line {"modulus", "backends", "proofs": [sha256...]}."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from genstark_amd._abi import Backend                         # noqa: E402
from genstark_amd.air import MimcAir, runMimc                  # noqa: E402
from genstark_amd.air_generic import GenericAir                # noqa: E402
from genstark_amd.errors import StarkError                     # noqa: E402
from genstark_amd.field import PrimeField                      # noqa: E402
from genstark_amd.hostfield import HostField                   # noqa: E402
from genstark_amd.native import NativeProver                   # noqa: E402
from genstark_amd._mirror.stark import Stark                   # noqa: E402

OPTS = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 24, 'friQueryCount': 12}


def quintic_air(field, steps):
    ks = [[(7 * i + 3) % field.modulus for i in range(8)]]
    return GenericAir(steps, 2, [5, 5], ks, lambda r, k: [(r[0] + k[0]) ** 5 + r[1], r[0] + 2 * r[1]],
                      lambda r, n, k: [n[0] - ((r[0] + k[0]) ** 5 + r[1]), n[1] - (r[0] + 2 * r[1])], lambda seed: [seed[0], seed[1]], None, field)


def run(backend, q):
    from test_small_fields import check_arithmetic
    assert backend.element_size == 32 and backend.modulus == q
    check_arithmetic(backend, q, q % 1000)
    f = PrimeField(backend=backend)
    out = []
    steps = 64
    air = MimcAir(steps, 16, f)
    control = runMimc(f, steps, air.roundConstants, 3)
    a = [{'step': 0, 'register': 0, 'value': 3}, {'step': steps - 1, 'register': 0, 'value': control[-1]}]
    cases = [(air, a, [3], lambda hf: MimcAir(steps, 16, hf))]
    qa = quintic_air(f, 64)
    full = qa.hostTrace([5, 9])
    cases.append((qa, [{'step': 0, 'register': 0, 'value': 5}, {'step': 63, 'register': 1, 'value': full[63][1]}], [5, 9], lambda hf: quintic_air(hf, 64)))
    for air, a, seed, host_air in cases:
        stark = Stark(air, OPTS)
        data = stark.serialize(stark.prove(a, [], seed))
        assert stark.verify(a, stark.parse(data))
        nat = NativeProver(stark)                               # libgstark_prover_rt.so: adopts the modulus of the library it is bound to
        assert nat.prove_bytes(a, [], seed) == data
        assert nat.verify_bytes(a, data) is True
        hv = Stark(host_air(HostField(q, elementSize=32)), OPTS)   # the device-free verifier: plain Python integers (32-byte elements, like the runtime flavour)
        assert hv.verify(a, hv.parse(data))
        bad = bytearray(data)
        bad[len(bad) // 2] ^= 2
        for verify in (lambda b: stark.verify(a, stark.parse(b)), lambda b: nat.verify_bytes(a, b)):
            try:
                verify(bytes(bad))
                raise AssertionError('a corrupted proof was accepted')
            except StarkError:
                pass
        out.append(data)
    return out


if __name__ == '__main__':
    q, which = int(sys.argv[1]), sys.argv[2]
    results = {}
    if which in ('oracle', 'both'):
        results['oracle'] = run(Backend(lib_path=os.path.join(ROOT, 'oracle', 'liboracle_rt.so'), allow_test_double=True, modulus=q), q)
    if which in ('hip', 'both'):
        hip = Backend(device=0, modulus=q)
        assert hip.name == 'hip-gfx950'
        results['hip'] = run(hip, q)
    if which == 'both':
        assert results['hip'] == results['oracle'], 'HIP and oracle proofs differ'
    # one modulus per process: a second one is refused, the same one is accepted again
    lib = os.path.join(ROOT, 'oracle', 'liboracle_rt.so') if which != 'hip' else None
    from genstark_amd._abi import GstarkError
    try:
        Backend(lib_path=lib, allow_test_double=True, modulus=q + 2) if lib else Backend(device=0, modulus=q + 2)
        raise AssertionError('a second modulus was accepted')
    except GstarkError as e:
        assert 'ONE modulus per process' in str(e) or 'gs_set_modulus' in str(e), e
    first = next(iter(results.values()))
    print(json.dumps({'modulus': str(q), 'backends': sorted(results), 'proofs': [hashlib.sha256(d).hexdigest() for d in first]}))
