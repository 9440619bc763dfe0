"""The small prime fields of the reference's examples on the build flavours of the library (genstark_amd/csrc/gf_small.h; SURVEY
8f-3): 2^64 - 21*2^30 + 1 (examples/rescue/hash2x64.ts) and 2^32 - 3*2^25 + 1 (examples/demo/fibonacci.ts, the README's Foo).
Same kernels, same 16-byte element layout, plain arithmetic.  Known answers: the Rescue 2x64 trace the example prints
(hash2x64.ts:137-215), the Fibonacci results of fibonacci.ts:9-11, Foo 1 -> 127.  CPU: the oracle flavours; GPU: HIP flavours,
byte-identical proofs."""
import os
import random

import pytest

from conftest import ROOT, _build_oracle, native_same_bytes
from genstark_amd._abi import HIP_LIB_PATHS, MODULUS_17, MODULUS_32, MODULUS_64, Backend
from genstark_amd.air_generic import GenericAir
from genstark_amd.errors import StarkError
from genstark_amd.field import PrimeField
from genstark_amd.hostfield import HostField
from genstark_amd.rescue import rescue2x64_air
from genstark_amd._mirror.stark import Stark

FLAVOURS = {'q64': MODULUS_64, 'q32': MODULUS_32, 'q17': MODULUS_17}
# the STARK trace column of hash2x64.ts:149-213 (rounds 1, 2, 3, 5, 6, 7) and the digest of 42 (:101)
RESCUE_2X64_ROWS = {0: (6192394074115262567, 6362103795149910654), 1: (4443483495863871585, 18213808804803479104),
                    2: (12298482428329212698, 17330962085246333408), 4: (8313646796226318584, 11641010825224956624),
                    5: (978482924564259844, 1504772570823547853), 6: (5186520612742714234, 12963908037192828019)}
RESCUE_2X64_DIGEST = 14354339131598895532
FIBONACCI = {2**6: 1783540607, 2**13: 203257732, 2**17: 2391373091}          # examples/demo/fibonacci.ts:9-11


# examples/demo/staticVariables.ts:56-120: the execution trace the example tabulates (V0 per step; K0 = 1..4, K1 = 1..8 cyclic)
DEMO_TRACE = [1, 5, 12, 22, 35, 47, 62, 80, 101, 105, 112, 122, 135, 147, 162, 180, 201, 205, 212, 222, 235, 247, 262, 280, 301, 305, 312, 322,
              335, 347, 362, 380, 401, 405, 412, 422, 435, 447, 462, 480, 501, 505, 512, 522, 535, 547, 562, 580, 601, 605, 612, 622, 635, 647,
              662, 680, 701, 705, 712, 722, 735, 747, 762, 780]


def demo_air(field):
    """examples/demo/staticVariables.ts:10-33: one register, two cyclic static registers, r' = r + 1 + k0 + 2*k1."""
    return GenericAir(64, 1, [1], [[1, 2, 3, 4], [1, 2, 3, 4, 5, 6, 7, 8]], lambda r, k: [r[0] + 1 + k[0] + 2 * k[1]],
                      lambda r, n, k: [n[0] - (r[0] + 1 + k[0] + 2 * k[1])], lambda seed: [seed[0]], None, field)


def oracle_for(name):
    _build_oracle()
    return Backend(lib_path=os.path.join(ROOT, 'oracle', f'liboracle_{name}.so'), allow_test_double=True)


def hip_for(name):
    return Backend(device=0, modulus=FLAVOURS[name])


def check_arithmetic(backend, q, seed):
    """Every vector member against Python integers, NTT against direct evaluation."""
    f = PrimeField(backend=backend)
    assert f.modulus == q
    rng = random.Random(seed)
    edge = [0, 1, q - 1, q - 2, 2, (q + 1) // 2]
    n = 3000
    xs = edge + [rng.randrange(q) for _ in range(n - len(edge))]
    ys = [rng.randrange(q) for _ in range(n - 3)] + [0, 1, q - 1]
    vx, vy = f.newVectorFrom(xs), f.newVectorFrom(ys)
    assert f.addVectorElements(vx, vy).toValues() == [(a + b) % q for a, b in zip(xs, ys)]
    assert f.subVectorElements(vx, vy).toValues() == [(a - b) % q for a, b in zip(xs, ys)]
    assert f.mulVectorElements(vx, vy).toValues() == [a * b % q for a, b in zip(xs, ys)]
    assert f.mulVectorElements(vx, q - 1).toValues() == [(-a) % q for a in xs]
    inv = lambda a: pow(a, -1, q) if a else 0
    assert f.invVectorElements(vx).toValues() == [inv(a) for a in xs]
    assert f.divVectorElements(vy, vx).toValues() == [b * inv(a) % q for a, b in zip(xs, ys)]
    assert f.expVectorElements(f.newVectorFrom(xs[:100]), q - 2).toValues() == [inv(a) for a in xs[:100]]
    base = rng.randrange(2, q)
    assert f.getPowerSeries(base, 500).toValues() == [pow(base, i, q) for i in range(500)]
    cs = [rng.randrange(q) for _ in range(3)]
    assert f.combineManyVectors([vx, vy, vx], cs).toValues() == [(cs[0] * a + cs[1] * b + cs[2] * a) % q for a, b in zip(xs, ys)]
    for logn, plen in ((0, 1), (3, 8), (6, 5), (9, 512), (12, 256)):
        m = 1 << logn
        if (q - 1) % m:
            continue                         # 96769 - 1 = 2^9 * 189: no roots of unity of order above 512
        w = f.getRootOfUnity(m)
        assert pow(w, m, q) == 1 and (m == 1 or pow(w, m // 2, q) == q - 1)
        coeffs = [rng.randrange(q) for _ in range(plen)]
        ev = f.evalPolyAtRoots(f.newVectorFrom(coeffs), f.getPowerSeries(w, m))
        got = ev.toValues()
        for i in [0, 1, m - 1, rng.randrange(m)]:
            x = pow(w, i % m, q)
            assert got[i % m] == sum(c * pow(x, k, q) for k, c in enumerate(coeffs)) % q
        assert f.interpolateRoots(f.getPowerSeries(w, m), ev).toValues() == coeffs + [0] * (m - plen)


def fibonacci_air(field, steps):
    """examples/demo/fibonacci.ts:22-40: two registers, a' = a + b, b' = a + 2b (two Fibonacci steps per row)."""
    return GenericAir(steps, 2, [1, 1], [], lambda r, k: [r[0] + r[1], r[0] + 2 * r[1]],
                      lambda r, n, k: [n[0] - (r[0] + r[1]), n[1] - (r[0] + 2 * r[1])], lambda seed: [seed[0], seed[1]], None, field)


def foo_air(field):
    return GenericAir(64, 1, [1], [], lambda r, k: [r[0] + 2], lambda r, n, k: [n[0] - (r[0] + 2)], lambda seed: [seed[0]], None, field)


def check_starks(backend, name):
    """Full prove / serialize / parse / verify over the small field; the GPU-free verifier accepts the same bytes."""
    q = FLAVOURS[name]
    f = PrimeField(backend=backend)
    out = []
    if name == 'q17':
        air = demo_air(f)                                                    # staticVariables.ts, its own 17-bit field
        assert air.extensionFactor == 4
        assert air.initProvingContext([], [1]).generateExecutionTrace().toValues() == [DEMO_TRACE]      # the table of :56-120
        stark = Stark(air, None)
        assertions = [{'step': 0, 'register': 0, 'value': 1}, {'step': 63, 'register': 0, 'value': 780}]    # :38-41
        proof = stark.prove(assertions, [], [1])
        data = stark.serialize(proof)
        assert len(data) == stark.sizeOf(proof) and stark.verify(assertions, stark.parse(data))
        hv = Stark(demo_air(HostField(q)), None)
        assert hv.verify(assertions, hv.parse(data))
        with pytest.raises(StarkError):
            stark.verify([assertions[0], dict(assertions[1], value=781)], stark.parse(data))
        native_same_bytes(stark, assertions, [], [1], data)
        out.append(data)
    elif name == 'q32':
        air = foo_air(f)                                                     # README.md:17-60, its own field
        stark = Stark(air, None)
        assertions = [{'step': 0, 'register': 0, 'value': 1}, {'step': 63, 'register': 0, 'value': 127}]
        proof = stark.prove(assertions, [], [1])
        assert proof['ldProof']['components'] == []
        data = stark.serialize(proof)
        assert len(data) == stark.sizeOf(proof) and stark.verify(assertions, stark.parse(data))
        hv = Stark(foo_air(HostField(q)), None)
        assert hv.verify(assertions, hv.parse(data))
        native_same_bytes(stark, assertions, [], [1], data)                  # BASELINE configs[0] (Foo) through the native driver's q32 build
        out.append(data)
        steps = 2**6
        air = fibonacci_air(f, steps)
        assert air.extensionFactor == 4
        trace = air.initProvingContext([], [1, 1]).generateExecutionTrace()
        assert trace.getValue(1, steps - 1) == FIBONACCI[steps]             # fibonacci.ts:9-11
        big = fibonacci_air(f, 2**13).initProvingContext([], [1, 1]).generateExecutionTrace()
        assert big.getValue(1, 2**13 - 1) == FIBONACCI[2**13]
        assert fibonacci_air(f, 2**17).initProvingContext([], [1, 1]).generateExecutionTrace().getValue(1, 2**17 - 1) == FIBONACCI[2**17]
        stark = Stark(air, {'hashAlgorithm': 'sha256', 'extensionFactor': 16, 'exeQueryCount': 32, 'friQueryCount': 16})
        assertions = [{'step': 0, 'register': 0, 'value': 1}, {'step': 0, 'register': 1, 'value': 1},
                      {'step': steps - 1, 'register': 1, 'value': FIBONACCI[steps]}]
        data = stark.serialize(stark.prove(assertions, [], [1, 1]))
        assert stark.verify(assertions, stark.parse(data))
        with pytest.raises(StarkError):
            stark.verify(assertions[:2] + [dict(assertions[2], value=FIBONACCI[steps] - 1)], stark.parse(data))
        native_same_bytes(stark, assertions, [], [1, 1], data)
        out.append(data)
    else:
        air = rescue2x64_air(32, 16, f)
        trace = air.initProvingContext([], [42]).generateExecutionTrace().toValues()
        for step, (r0, r1) in RESCUE_2X64_ROWS.items():                     # the trace the example prints, through the device VM
            assert (trace[0][step], trace[1][step]) == (r0, r1)
        assert trace[0][31] == RESCUE_2X64_DIGEST
        assert [list(r) for r in zip(*air.hostTrace([42]))] == trace
        stark = Stark(air, {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24})   # hash2x64.ts:38-44
        assertions = [{'step': 31, 'register': 0, 'value': RESCUE_2X64_DIGEST}]                                        # :105-107
        proof = stark.prove(assertions, [], [42])
        data = stark.serialize(proof)
        assert len(data) == stark.sizeOf(proof) and stark.verify(assertions, stark.parse(data))
        hv = Stark(rescue2x64_air(32, 16, HostField(q)), {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24})
        assert hv.verify(assertions, hv.parse(data))
        with pytest.raises(StarkError):
            stark.prove([{'step': 31, 'register': 0, 'value': RESCUE_2X64_DIGEST - 1}], [], [42])
        native_same_bytes(stark, assertions, [], [42], data)                 # hash2x64.ts through the native driver's q64 build
        out.append(data)
        air = rescue2x64_air(256, 16, f)                                     # longer chain: more FRI layers
        full = air.hostTrace([42])
        stark = Stark(air, {'hashAlgorithm': 'sha256', 'extensionFactor': 16, 'exeQueryCount': 40, 'friQueryCount': 20})
        assertions = [{'step': 31, 'register': 0, 'value': RESCUE_2X64_DIGEST}, {'step': 255, 'register': 1, 'value': full[255][1]}]
        data = stark.serialize(stark.prove(assertions, [], [42]))
        assert stark.verify(assertions, stark.parse(data))
        native_same_bytes(stark, assertions, [], [42], data)
        out.append(data)
    return out


@pytest.mark.parametrize('name', ['q64', 'q32', 'q17'])
def test_small_field_arithmetic_oracle(name):
    check_arithmetic(oracle_for(name), FLAVOURS[name], 7)


@pytest.mark.parametrize('name', ['q64', 'q32', 'q17'])
def test_small_field_starks_oracle(name):
    check_starks(oracle_for(name), name)


def test_field_and_library_must_agree(oracle_backend):
    from genstark_amd._abi import GstarkError
    with pytest.raises(GstarkError):
        PrimeField(MODULUS_64, oracle_backend)                 # the default oracle library is the 128-bit one
    with pytest.raises(GstarkError):
        Backend(modulus=2**61)                                 # no fixed build, and the runtime-modulus build takes odd moduli only
    assert set(HIP_LIB_PATHS) == {2**128 - 9 * 2**32 + 1, MODULUS_64, MODULUS_32, MODULUS_17, 2**256 - 351 * 2**32 + 1, 2**224 - 2**96 + 1}


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['q64', 'q32', 'q17'])
def test_small_field_hip(name):
    hip = hip_for(name)
    assert hip.name == 'hip-gfx950' and hip.modulus == FLAVOURS[name]
    check_arithmetic(hip, FLAVOURS[name], 11)
    assert check_starks(hip, name) == check_starks(oracle_for(name), name)
