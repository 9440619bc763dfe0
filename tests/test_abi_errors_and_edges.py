"""Error behaviour and edge cases of the C ABI, identical on both implementations: the CPU oracle (-m "not gpu") and
the HIP library (-m gpu).  Mirrors the reference's habit of throwing on malformed input (e.g. "Number of roots of unity
cannot be smaller than number of values", power-of-two requirements of MerkleTree.create / getRootOfUnity)."""
import ctypes as C

import pytest

from conftest import P, rand_elements, to_bytes
from genstark_amd._abi import GstarkError
from genstark_amd.field import PrimeField
from genstark_amd.merkle import MerkleTree, createHash


def run_cases(backend, rng):
    f = PrimeField(backend=backend)
    h = createHash('blake2s256', backend)
    # ---- length-1 and length-2 objects
    v1 = f.newVectorFrom([5])
    assert f.addVectorElements(v1, v1).toValues() == [10]
    assert f.invVectorElements(v1).toValues() == [pow(5, P - 2, P)]
    assert f.getPowerSeries(7, 1).toValues() == [1]
    assert f.evalPolyAtRoots(f.newVectorFrom([9]), f.getPowerSeries(1, 1)).toValues() == [9]
    w2 = f.getRootOfUnity(2)
    assert w2 == P - 1
    ev = f.evalPolyAtRoots(f.newVectorFrom([3, 4]), f.getPowerSeries(w2, 2)).toValues()
    assert ev == [7, (3 - 4) % P]
    assert f.interpolateRoots(f.getPowerSeries(w2, 2), f.newVectorFrom(ev)).toValues() == [3, 4]
    # ---- evalPolyAtRoots with a single coefficient over a large domain (constant polynomial)
    const = f.evalPolyAtRoots(f.newVectorFrom([11]), f.getPowerSeries(f.getRootOfUnity(1024), 1024)).toValues()
    assert const == [11] * 1024
    # ---- malformed requests raise
    with pytest.raises(GstarkError):
        f.evalPolyAtRoots(f.newVectorFrom([1, 2, 3, 4, 5]), f.getPowerSeries(f.getRootOfUnity(4), 4))   # more values than roots
    with pytest.raises(GstarkError):
        f.addVectorElements(f.newVectorFrom([1, 2]), f.newVectorFrom([1, 2, 3]))                        # length mismatch
    with pytest.raises(GstarkError):
        f.getRootOfUnity(3)                                                                             # not a power of two
    with pytest.raises(GstarkError):
        MerkleTree.create(h.mergeVectorRows([f.newVectorFrom([1, 2, 3])]), h)                           # 3 leaves
    bad_roots = f.newVectorFrom([1, 5, 25, 125, 625, 3125, 15625, 78125])                              # not roots of unity
    with pytest.raises(GstarkError):
        f.evalPolyAtRoots(f.newVectorFrom(list(range(8))), bad_roots)
    with pytest.raises(GstarkError):
        f.transposeVector(f.newVectorFrom(list(range(10))), 4)                                          # 10 % 4 != 0
    tree = MerkleTree.create(h.mergeVectorRows([f.newVectorFrom(rand_elements(rng, 16))]), h)
    with pytest.raises(GstarkError):
        tree.proveBatch([1, 1])                                                                         # repeating indexes
    with pytest.raises(GstarkError):
        tree.proveBatch([16])                                                                           # out of range
    with pytest.raises(GstarkError):
        f.combineManyVectors([f.newVectorFrom([1])] * 2, [1])                                           # coefficient count
    # raw ABI: in-place batch inversion is rejected / handled identically (outputs never alias inputs in the mirror)
    v = f.newVectorFrom(rand_elements(rng, 64))
    out = f.newVector(64)
    backend.call('gs_vec_inv', C.c_void_p(v.ptr), 64, C.c_void_p(out.ptr))
    assert out.toValues() == [pow(x, P - 2, P) if x else 0 for x in v.toValues()]
    # ---- the error state is per call: the context keeps working after a failure
    assert f.mulVectorElements(v1, v1).toValues() == [25]


def test_edges_and_errors_oracle(oracle_backend, rng):
    run_cases(oracle_backend, rng)


@pytest.mark.gpu
def test_edges_and_errors_hip(hip_backend, rng):
    run_cases(hip_backend, rng)


@pytest.mark.gpu
def test_poseidon_and_rescue_shaped_commitments_equal_oracle(hip_backend, oracle_backend):
    """BASELINE configs[2]/[3] shapes with synthetic traces: 4 registers (Rescue 4x128) and 6 registers (Poseidon 6x128),
    T = 2^14, E = 16: iNTT + LDE of every register, row hashing over all registers, Merkle tree — bytes equal to the oracle."""
    import random
    for registers in (4, 6):
        rng = random.Random(registers)
        t, ef = 1 << 14, 16
        n = t * ef
        raw = to_bytes([rng.randrange(P) for _ in range(registers * t)])
        roots = []
        for be in (hip_backend, oracle_backend):
            f = PrimeField(backend=be)
            h = createHash('blake2s256', be)
            w = f.getRootOfUnity(n)
            trace = f.newMatrix(registers, t)
            be.upload(trace.ptr, raw)
            polys = f.interpolateRoots(f.getPowerSeries(f.exp(w, ef), t), trace)
            ev = f.evalPolysAtRoots(polys, f.getPowerSeries(w, n))
            leaves = h.mergeVectorRows(f.matrixRowsToVectors(ev))
            tree = MerkleTree.create(leaves, h)
            proof = tree.proveBatch([1, 2, n // 2, n - 1, 12345])
            roots.append((tree.root, leaves.toBuffer(), proof['values'], proof['nodes']))
        assert roots[0] == roots[1]
