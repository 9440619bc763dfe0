"""The reference's AirAssembly library assembly/lib128.aa — ComputePoseidonHash and ComputeMerkleRoot (the Poseidon Merkle-path
STARK of BASELINE configs[3]) — as GenericAirs with secret / public INPUT registers and masks, driven like
examples/assembly/lib128.ts:55-117 and checked against that example's own control computations (createHash, MerkleTree of
examples/poseidon/utils.ts)."""
import pytest

import genstark_amd as ga
from genstark_amd import lib128
from genstark_amd.errors import StarkError
from genstark_amd.field import PrimeField
from genstark_amd._mirror.stark import Stark

OPTS = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 32, 'exeQueryCount': 44, 'friQueryCount': 20}       # lib128.ts:33-39


def check_hash(backend, hashes):
    f = PrimeField(backend=backend)
    air = lib128.compute_poseidon_hash_air(f, hashes)
    raw = [[42 + 10 * s for s in range(hashes)], [43 + s for s in range(hashes)], [44] * hashes, [45 + s * s for s in range(hashes)]]   # lib128.ts:61
    inputs, seed = air.expandInputs(raw), air.segmentSeeds(raw)
    stark = Stark(air, OPTS)
    trace = air.hostTrace(seed, inputs=inputs)
    assertions = []
    for s in range(hashes):
        digest = lib128.poseidon_hash(f, [col[s] for col in raw])            # lib128.ts:57-58: the example's control value
        assert trace[64 * s + 63][:2] == digest
        assertions += [{'step': 64 * s + 63, 'register': 0, 'value': digest[0]}, {'step': 64 * s + 63, 'register': 1, 'value': digest[1]}]
    device_trace = air.initProvingContext(inputs, seed).generateExecutionTrace().toValues()
    assert device_trace == [list(r) for r in zip(*trace)]
    proof = stark.prove(assertions, inputs, seed)
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof)
    assert len(proof['evProof']['values'][0]) == (6 + 4) * 16               # six trace registers + four secret input registers
    assert stark.verify(assertions, stark.parse(data))
    with pytest.raises(StarkError):
        stark.verify([dict(assertions[0], value=assertions[0]['value'] ^ 1)] + assertions[1:], stark.parse(data))
    return data


def merkle_case(f, depth, index):
    leaves = [(ga.sha256_prng(b'\x2a', 1 << depth, f)[i], ga.sha256_prng(b'\x2b', 1 << depth, f)[i]) for i in range(1 << depth)]
    tree = lib128.PoseidonMerkleTree(f, leaves)
    path = tree.prove(index)
    bits = [(index >> j) & 1 for j in range(depth)]          # toBinaryArray (lib128.ts:170-177): least significant bit first
    bits = [0] + bits[:-1]                                   # lib128.ts:92-95: shifted to line up with the end of the first loop
    return tree, path[0], path[1:], bits


def check_merkle(backend, depth, index):
    f = PrimeField(backend=backend)
    tree, leaf, nodes, bits = merkle_case(f, depth, index)
    air = lib128.compute_merkle_root_air(f, bits)
    inputs, first = lib128.merkle_inputs(f, leaf, nodes)
    trace = air.hostTrace(first, inputs=inputs)
    last = 64 * depth - 1
    # the last level's bit never enters the registers (lib128.ts:95 pops it): the root is H(p, v) = registers 0, 1 when the top bit
    # of the index is 0 (the example's index 42 of 256), and H(v, p) = registers 6, 7 when it is 1
    base = 6 if (index >> (depth - 1)) & 1 else 0
    assert tuple(trace[last][base:base + 2]) == tree.root     # lib128.ts:101-105: the root the example's MerkleTree computes
    stark = Stark(air, OPTS)
    assertions = [{'step': last, 'register': base, 'value': tree.root[0]}, {'step': last, 'register': base + 1, 'value': tree.root[1]}]
    proof = stark.prove(assertions, inputs, first)
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof) and stark.verify(assertions, stark.parse(data))
    # a verifier holding other index bits (the public input) builds another AIR and must reject the proof
    other = Stark(lib128.compute_merkle_root_air(f, [bits[0], 1 - bits[1]] + bits[2:]), OPTS)
    with pytest.raises(StarkError):
        other.verify(assertions, other.parse(data))
    return data


@pytest.mark.parametrize('hashes', [1, 2])
def test_compute_poseidon_hash_oracle(oracle_backend, hashes):
    check_hash(oracle_backend, hashes)


def test_compute_merkle_root_oracle(oracle_backend):
    check_merkle(oracle_backend, 4, 11)


@pytest.mark.gpu
def test_lib128_hip_equals_oracle(hip_backend, oracle_backend):
    assert check_hash(hip_backend, 16) == check_hash(oracle_backend, 16)      # 16 segments: the device trace generator (8 or fewer run on the host)
    assert check_merkle(hip_backend, 4, 5) == check_merkle(oracle_backend, 4, 5)


@pytest.mark.gpu
def test_compute_merkle_root_depth8_hip(hip_backend):
    """lib128.ts:77-117 as it stands: tree depth 8, index 42 -> 512 steps, 12 registers, extension factor 32."""
    check_merkle(hip_backend, 8, 42)


def check_merkle_update(backend, depth, index):
    """lib128.ts:119-167: the same leaf position before and after an update; both roots come out of one 24-register trace."""
    f = PrimeField(backend=backend)
    old_value, new_value = (9, 10), (11, 12)                                    # lib128.ts:125
    tree1, _, nodes, bits = merkle_case(f, depth, index)
    leaves = [tree1.nodes[(1 << depth) + i] for i in range(1 << depth)]
    leaves[index] = old_value
    tree1 = lib128.PoseidonMerkleTree(f, leaves)
    leaves2 = list(leaves)
    leaves2[index] = new_value
    tree2 = lib128.PoseidonMerkleTree(f, leaves2)
    path1, path2 = tree1.prove(index), tree2.prove(index)
    assert path1[1:] == path2[1:] and path1[0] == old_value and path2[0] == new_value
    air = lib128.compute_merkle_update_air(f, depth)
    inputs, first = lib128.merkle_update_inputs(f, old_value, new_value, path1[1:], bits)
    trace = air.hostTrace(first, inputs=inputs)
    last = 64 * depth - 1
    assert tuple(trace[last][0:2]) == tree1.root and tuple(trace[last][12:14]) == tree2.root
    stark = Stark(air, OPTS)
    assertions = [{'step': last, 'register': 0, 'value': tree1.root[0]}, {'step': last, 'register': 1, 'value': tree1.root[1]},
                  {'step': last, 'register': 12, 'value': tree2.root[0]}, {'step': last, 'register': 13, 'value': tree2.root[1]}]   # lib128.ts:150-155
    proof = stark.prove(assertions, inputs, first)
    data = stark.serialize(proof)
    assert len(data) == stark.sizeOf(proof) and stark.verify(assertions, stark.parse(data))
    assert len(proof['evProof']['values'][0]) == (24 + 7) * 16
    # an index "bit" that is not binary breaks the 25th constraint: the prover's own remainder check refuses the proof
    bad = [list(c) for c in inputs]
    bad[6] = [2 if v else 0 for v in bad[6]]
    with pytest.raises(StarkError, match='Low degree proof failed'):
        stark.prove([{'step': 0, 'register': 0, 'value': first[0]}], bad, first)      # the trace itself is consistent with `bad`
    return data


def test_compute_merkle_update_oracle(oracle_backend):
    check_merkle_update(oracle_backend, 4, 5)


@pytest.mark.gpu
def test_compute_merkle_update_hip_equals_oracle(hip_backend, oracle_backend):
    assert check_merkle_update(hip_backend, 4, 6) == check_merkle_update(oracle_backend, 4, 6)
