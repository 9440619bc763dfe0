"""The two multi-limb prime fields of the reference's examples on the build flavours of the library (genstark_amd/csrc/gf_wide.h;
SURVEY 8f-3): 2^256 - 351*2^32 + 1 (examples/mimc/mimc256.ts:13) and 2^224 - 2^96 + 1 (assembly/lib224.aa:3).  Same kernels,
32-byte elements.  The reference pins nothing for these fields beyond prove -> verify round trips (mimc256.ts:70-85), so the
checks are: every vector member against Python integers, NTT against direct evaluation, MiMC-256 with the example's options
proved, serialized, parsed, verified (also by the GPU-free verifier), tampering rejected.  CPU: the oracle flavours (C23 _BitInt
arithmetic, nothing shared with the device code); GPU: the HIP flavours, byte-identical proofs."""
import os

import pytest

from conftest import ROOT, _build_oracle, native_same_bytes
from genstark_amd._abi import MODULUS_224, MODULUS_256, Backend
from genstark_amd.air import MimcAir, runMimc
from genstark_amd.air_generic import GenericAir
from genstark_amd.errors import StarkError
from genstark_amd.field import PrimeField
from genstark_amd.hostfield import HostField
from genstark_amd.pointmul import ec_multiply, point_mul_air, to_bits
from genstark_amd._mirror.stark import Stark
from test_small_fields import check_arithmetic

FLAVOURS = {'p256': MODULUS_256, 'p224': MODULUS_224}
MIMC256_OPTIONS = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 40, 'friQueryCount': 24}    # mimc256.ts:22-28


def oracle_for(name):
    _build_oracle()
    return Backend(lib_path=os.path.join(ROOT, 'oracle', f'liboracle_{name}.so'), allow_test_double=True)


def hip_for(name):
    return Backend(device=0, modulus=FLAVOURS[name])


# examples/elliptic/pointMul.ts:22-31: the point, the scalar, the expected product (the reference's own known answer)
EC_POINT = (19277929113566293071110308034699488026831934219452440156649784352033, 19926808758034470970197974370888749184205991990603949537637343198772)
EC_SCALAR = 21628546220445634706341881427918508772248629391536891476641575405363
EC_PRODUCT = (5326626235735428056996404471396244610891648579045949976641038973984, 6753729428472267765045584530315486521937702623726344079323769311058)
EC_OPTIONS = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 24}             # pointMul.ts:11-17


def check_point_mul(backend):
    """examples/elliptic/pointmul.aa over the 224-bit field: the trace the register machine generates holds the product the
    reference expects; the proof round-trips like pointMul.ts:39-56 and the GPU-free verifier accepts it."""
    f = PrimeField(backend=backend)
    air = point_mul_air(f)
    raw = [[EC_POINT[0]], [EC_POINT[1]], [to_bits(EC_SCALAR)]]
    inputs, seeds = air.expandInputs(raw), air.segmentSeeds(raw)
    trace = air.initProvingContext(inputs, seeds).generateExecutionTrace().toValues()
    assert (trace[2][255], trace[3][255]) == EC_PRODUCT                      # pointMul.ts:28-37
    assert trace[7][255] == EC_SCALAR and trace[6][0] == 1 and trace[6][255] == 0
    assert [list(r) for r in zip(*air.hostTrace(seeds, inputs=inputs))] == trace
    stark = Stark(air, EC_OPTIONS)
    assertions = [{'step': 255, 'register': 2, 'value': EC_PRODUCT[0]}, {'step': 255, 'register': 3, 'value': EC_PRODUCT[1]}]
    proof = stark.prove(assertions, inputs, seeds)
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof) and stark.verify(assertions, stark.parse(data))
    hv = Stark(point_mul_air(HostField(MODULUS_224)), EC_OPTIONS)
    assert hv.verify(assertions, hv.parse(data))
    native_same_bytes(stark, assertions, inputs, seeds, data)
    with pytest.raises(StarkError):
        stark.verify([assertions[0], dict(assertions[1], value=EC_PRODUCT[1] ^ 1)], stark.parse(data))
    # two multiplications in one trace (two device threads), the second by another scalar
    k2 = 0x1234567890abcdef1234567890abcdef1234567890abcdef12345678
    air2 = point_mul_air(f, 2)
    raw2 = [[EC_POINT[0], EC_PRODUCT[0]], [EC_POINT[1], EC_PRODUCT[1]], [to_bits(EC_SCALAR), to_bits(k2)]]
    inputs2, seeds2 = air2.expandInputs(raw2), air2.segmentSeeds(raw2)
    want2 = ec_multiply(MODULUS_224, EC_PRODUCT, k2)
    stark2 = Stark(air2, EC_OPTIONS)
    assertions2 = assertions + [{'step': 511, 'register': 2, 'value': want2[0]}, {'step': 511, 'register': 3, 'value': want2[1]}]
    data2 = stark2.serialize(stark2.prove(assertions2, inputs2, seeds2))
    assert stark2.verify(assertions2, stark2.parse(data2))
    if backend.name == 'hip-gfx950':
        # 16 multiplications: past the host-side threshold, one device thread per multiplication; same rows as the host interpreter
        air16 = point_mul_air(f, 16)
        ks = [EC_SCALAR + 7 * i for i in range(16)]
        raw16 = [[EC_POINT[0]] * 16, [EC_POINT[1]] * 16, [to_bits(k) for k in ks]]
        inputs16, seeds16 = air16.expandInputs(raw16), air16.segmentSeeds(raw16)
        t16 = air16.initProvingContext(inputs16, seeds16).generateExecutionTrace().toValues()
        for i in (0, 5, 15):
            assert (t16[2][256 * i + 255], t16[3][256 * i + 255]) == ec_multiply(MODULUS_224, EC_POINT, ks[i])
    return [data, data2]


def quintic_air(field, steps):
    """A two-register degree-5 chain with a cyclic constant (the S-box of lib224.aa's Poseidon, x^5): exercises the generic
    trace / constraint VM, exponentiation by a constant and static registers over the wide field."""
    ks = [[(7 * i + 3) % field.modulus for i in range(8)]]
    return GenericAir(steps, 2, [5, 5], ks, lambda r, k: [(r[0] + k[0]) ** 5 + r[1], r[0] + 2 * r[1]],
                      lambda r, n, k: [n[0] - ((r[0] + k[0]) ** 5 + r[1]), n[1] - (r[0] + 2 * r[1])], lambda seed: [seed[0], seed[1]], None, field)


def check_starks(backend, name, steps=2**7):
    q = FLAVOURS[name]
    f = PrimeField(backend=backend)
    assert f.elementSize == 32 and f.modulus == q
    out = []
    # MiMC over the wide field (mimc256.ts:30-56: x <- x^3 + k, 64 cyclic round constants)
    air = MimcAir(steps, 16, f)
    control = runMimc(f, steps, air.roundConstants, 3)
    trace = air.initProvingContext([], [3]).generateExecutionTrace()
    assert trace.toValues()[0] == control
    stark = Stark(air, MIMC256_OPTIONS)
    assertions = [{'step': 0, 'register': 0, 'value': 3}, {'step': steps - 1, 'register': 0, 'value': control[-1]}]    # mimc256.ts:60-63
    proof = stark.prove(assertions, [], [3])
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)                                   # mimc256.ts:76
    assert stark.verify(assertions, stark.parse(data))
    hv = Stark(MimcAir(steps, 16, HostField(q)), MIMC256_OPTIONS)
    assert hv.verify(assertions, hv.parse(data))
    with pytest.raises(StarkError):
        stark.verify([assertions[0], dict(assertions[1], value=(control[-1] + 1) % q)], stark.parse(data))
    with pytest.raises(StarkError):
        stark.prove([assertions[0], dict(assertions[1], value=(control[-1] + 1) % q)], [], [3])
    native_same_bytes(stark, assertions, [], [3], data)                      # the native driver's build for this field (README.md:213-214 rows)
    out.append(data)
    # generic AIR (register machine) over the wide field, sha256 leaves
    air = quintic_air(f, 64)
    full = air.hostTrace([5, 9])
    assert [list(r) for r in zip(*full)] == air.initProvingContext([], [5, 9]).generateExecutionTrace().toValues()
    stark = Stark(air, {'hashAlgorithm': 'sha256', 'extensionFactor': 16, 'exeQueryCount': 30, 'friQueryCount': 16})
    assertions = [{'step': 0, 'register': 0, 'value': 5}, {'step': 63, 'register': 0, 'value': full[63][0]},
                  {'step': 63, 'register': 1, 'value': full[63][1]}]
    data = stark.serialize(stark.prove(assertions, [], [5, 9]))
    assert stark.verify(assertions, stark.parse(data))
    hv = Stark(quintic_air(HostField(q), 64), {'hashAlgorithm': 'sha256', 'extensionFactor': 16, 'exeQueryCount': 30, 'friQueryCount': 16})
    assert hv.verify(assertions, hv.parse(data))
    native_same_bytes(stark, assertions, [], [5, 9], data)
    out.append(data)
    return out


@pytest.mark.parametrize('name', ['p256', 'p224'])
def test_wide_field_arithmetic_oracle(name):
    b = oracle_for(name)
    assert b.element_size == 32
    check_arithmetic(b, FLAVOURS[name], 7)


@pytest.mark.parametrize('name', ['p256', 'p224'])
def test_wide_field_starks_oracle(name):
    check_starks(oracle_for(name), name)


def test_point_multiplication_oracle():
    assert ec_multiply(MODULUS_224, EC_POINT, EC_SCALAR) == EC_PRODUCT       # the plain control computation agrees with the reference
    check_point_mul(oracle_for('p224'))


def test_wide_field_moduli_are_the_examples():
    from sympy import isprime
    assert isprime(MODULUS_256) and isprime(MODULUS_224)
    assert (MODULUS_256 - 1) % 2**32 == 0 and (MODULUS_224 - 1) % 2**96 == 0


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['p256', 'p224'])
def test_wide_field_hip(name):
    hip = hip_for(name)
    assert hip.name == 'hip-gfx950' and hip.modulus == FLAVOURS[name] and hip.element_size == 32
    check_arithmetic(hip, FLAVOURS[name], 11)
    assert check_starks(hip, name) == check_starks(oracle_for(name), name)
    # a size where the NTT runs its multi-pass radix-256 path and the Merkle tree its streaming levels
    big = check_starks(hip, name, steps=2**12)
    assert len(big) == 2


@pytest.mark.gpu
@pytest.mark.parametrize('log_steps', [13, 17])
def test_native_driver_mimc256_at_the_readme_sizes(log_steps):
    """MiMC over the 256-bit field at the two sizes the reference's README publishes for it (README.md:213-214; mimc256.ts options)
    through the PRODUCT entry — genstark_amd.prover.Prover = the native driver's p256 build on the HIP p256 library: bytes of the mirror
    on the same device, of the oracle flavour at 2^13 steps, and the GPU-free verifier accepts them."""
    from genstark_amd.prover import Prover
    steps = 1 << log_steps
    hip = hip_for('p256')
    f = PrimeField(backend=hip)
    air = MimcAir(steps, 16, f)
    last = air.initProvingContext([], [3]).generateExecutionTrace().getValue(0, steps - 1)
    assertions = [{'step': 0, 'register': 0, 'value': 3}, {'step': steps - 1, 'register': 0, 'value': last}]
    data = Prover(air, MIMC256_OPTIONS).prove_bytes(assertions, [], [3])
    mirror = Stark(air, MIMC256_OPTIONS)
    assert data == mirror.serialize(mirror.prove(assertions, [], [3]))
    if log_steps == 13:
        fo = PrimeField(backend=oracle_for('p256'))
        assert data == Prover(MimcAir(steps, 16, fo), MIMC256_OPTIONS).prove_bytes(assertions, [], [3])
        hv = Stark(MimcAir(steps, 16, HostField(MODULUS_256)), MIMC256_OPTIONS)
        assert hv.verify(assertions, hv.parse(data))
    else:
        assert mirror.verify(assertions, mirror.parse(data))


@pytest.mark.gpu
def test_point_multiplication_hip():
    assert check_point_mul(hip_for('p224')) == check_point_mul(oracle_for('p224'))


def point_mul16(backend):
    """16 multiplications in one trace: the size where the device runs the transition program (one thread per multiplication)."""
    f = PrimeField(backend=backend)
    air = point_mul_air(f, 16)
    ks = [EC_SCALAR + 7 * i for i in range(16)]
    raw = [[EC_POINT[0]] * 16, [EC_POINT[1]] * 16, [to_bits(k) for k in ks]]
    inputs, seeds = air.expandInputs(raw), air.segmentSeeds(raw)
    trace = air.initProvingContext(inputs, seeds).generateExecutionTrace().toValues()
    stark = Stark(air, EC_OPTIONS)
    assertions = [{'step': 256 * i + 255, 'register': 2, 'value': ec_multiply(MODULUS_224, EC_POINT, ks[i])[0]} for i in (0, 9, 15)]
    data = stark.serialize(stark.prove(assertions, inputs, seeds))
    assert stark.verify(assertions, stark.parse(data))
    return trace, data


@pytest.mark.gpu
def test_compiled_point_multiplication_equals_interpreted():
    """gs_air_jit in the 224-bit flavour (products are calls there, csrc/air_jit.hip): same trace, same proof bytes, and the compiled
    programs really ran."""
    plain, compiled = hip_for('p224'), hip_for('p224').jit()
    try:
        want = point_mul16(plain)
        assert compiled.jit_launches == 0
        assert point_mul16(compiled) == want
        assert compiled.jit_launches >= 2 and plain.jit_launches == 0
    finally:
        compiled.close()
        plain.close()
