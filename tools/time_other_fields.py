"""tools/time_other_fields.py — prove() / verify() wall-clock of the reference's examples over fields other than the 128-bit one, on
the HIP build flavours (markdown table).  usage: python tools/time_other_fields.py > profiles/xxx.md"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from genstark_amd import lib224
from genstark_amd._abi import MODULUS_64, MODULUS_224, MODULUS_256, Backend
from genstark_amd.air import MimcAir, runMimc
from genstark_amd.field import PrimeField
from genstark_amd.hostfield import HostField
from genstark_amd.pointmul import point_mul_air, to_bits
from genstark_amd.rescue import rescue2x64_air
from genstark_amd._mirror.stark import Stark

rows = []


def run(name, make_air, field, options, assertions, inputs, seed, reps=5):
    stark = Stark(make_air(field), options)
    for _ in range(2):
        proof = stark.prove(assertions, inputs, seed)
    gc.collect()
    gc.freeze()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        proof = stark.prove(assertions, inputs, seed)
        field.backend.sync()
        t.append((time.perf_counter() - t0) * 1e3)
    data = stark.serialize(proof)
    hv = Stark(make_air(HostField(field.modulus)), options)
    hv.verify(assertions, hv.parse(data))
    t0 = time.perf_counter()
    assert hv.verify(assertions, hv.parse(data))
    tv = (time.perf_counter() - t0) * 1e3
    air = stark.air
    rows.append(f'| {name} | {field.modulus.bit_length()} | {air.steps} x {air.traceRegisterCount} | {options["extensionFactor"]} | {min(t):.2f} | '
                f'{sum(t) / len(t):.2f} | {tv:.1f} | {len(data)} |')


f256, f224, f64 = (PrimeField(backend=Backend(device=0, modulus=m)) for m in (MODULUS_256, MODULUS_224, MODULUS_64))
# examples/mimc/mimc256.ts:13-28: 2^13 steps, E = 16, 40 / 24 queries
steps = 2**13
opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 40, 'friQueryCount': 24}
rc = MimcAir(steps, 16, f256).roundConstants
res = runMimc(f256, steps, rc, 3)
run('MiMC-256 (mimc256.ts)', lambda f: MimcAir(steps, 16, f), f256, opts, [{'step': 0, 'register': 0, 'value': 3}, {'step': steps - 1, 'register': 0, 'value': res[-1]}], [], [3])
steps = 2**17
res = runMimc(f256, steps, rc, 3)
run('MiMC-256, 2^17 steps', lambda f: MimcAir(steps, 16, f), f256, opts, [{'step': 0, 'register': 0, 'value': 3}, {'step': steps - 1, 'register': 0, 'value': res[-1]}], [], [3], reps=3)
# examples/elliptic/pointMul.ts
from test_wide_fields import EC_OPTIONS, EC_POINT, EC_PRODUCT, EC_SCALAR
air = point_mul_air(f224)
raw = [[EC_POINT[0]], [EC_POINT[1]], [to_bits(EC_SCALAR)]]
run('EC point multiplication (pointMul.ts)', point_mul_air, f224, EC_OPTIONS, [{'step': 255, 'register': 2, 'value': EC_PRODUCT[0]}, {'step': 255, 'register': 3, 'value': EC_PRODUCT[1]}],
    air.expandInputs(raw), air.segmentSeeds(raw))
# examples/assembly/lib224.ts
from test_lib224 import OPTS, SIG_G, SIG_H, SIG_P, SIG_R, SIG_S, merkle_case
air = lib224.compute_poseidon_hash_air(f224, 1)
raw = [[42], [43]]
d = lib224.poseidon_hash(f224, [42, 43])
run('lib224 ComputePoseidonHash', lambda f: lib224.compute_poseidon_hash_air(f, 1), f224, OPTS, [{'step': 63, 'register': 0, 'value': d[0]}, {'step': 63, 'register': 1, 'value': d[1]}],
    air.expandInputs(raw), air.segmentSeeds(raw))
tree, leaf, nodes, bits = merkle_case(f224, 8, 42)
inputs, first = lib224.merkle_inputs(f224, leaf, nodes)
run('lib224 ComputeMerkleRoot, depth 8', lambda f: lib224.compute_merkle_root_air(f, bits), f224, OPTS, [{'step': 511, 'register': 0, 'value': tree.root}], inputs, first)
air = lib224.verify_schnorr_signature_air(f224)
raw = [[SIG_G[0]], [SIG_G[1]], [to_bits(SIG_S)], [SIG_P[0]], [SIG_P[1]], [to_bits(SIG_H)], [SIG_R[0]], [SIG_R[1]]]
run('lib224 VerifySchnorrSignature', lambda f: lib224.verify_schnorr_signature_air(f), f224, OPTS, [{'step': 255, 'register': 13, 'value': SIG_H}], air.expandInputs(raw), air.segmentSeeds(raw))
# examples/rescue/hash2x64.ts
opts64 = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24}
run('Rescue 2x64 (hash2x64.ts)', lambda f: rescue2x64_air(32, 16, f), f64, opts64, [{'step': 31, 'register': 0, 'value': 14354339131598895532}], [], [42])

print('# prove() over the other fields of the reference\'s examples (HIP build flavours, one MI355X; host = Python mirror)\n')
print('| example | field bits | steps x registers | E | prove ms (best) | prove ms (mean) | verify ms (GPU-free HostField) | proof bytes |')
print('|---|---|---|---|---|---|---|---|')
print('\n'.join(rows))
print('\nThe reference runs these fields on galois\' generic BigInt code (its wasm path is 128-bit only); it publishes no timings for them.')
