"""Pins the CPU oracle (oracle/oracle_abi.c, oracle/hashes.c) — through the same C ABI the HIP library
exports — against Python big-int arithmetic, hashlib and oracle/pyref.py, plus the in-tree known-answer
values of the reference."""
import pytest

import abi_cases as cases
from oracle import pyref


@pytest.mark.parametrize('n', [1, 2, 7, 64, 1000])
def test_pointwise(oracle_backend, rng, n):
    cases.check_pointwise(oracle_backend, rng, n)


@pytest.mark.parametrize('logn,logsteps', [(4, 2), (8, 4), (10, 5)])
def test_domain_divisions(oracle_backend, rng, logn, logsteps):
    cases.check_domain_divisions(oracle_backend, rng, logn, logsteps)


@pytest.mark.parametrize('logn,logsteps,nroots', [(5, 2, 1), (8, 4, 2), (10, 6, 4)])
def test_mimc_composition(oracle_backend, rng, logn, logsteps, nroots):
    cases.check_mimc_composition(oracle_backend, rng, logn, logsteps, nroots)


@pytest.mark.parametrize('logn,depth', [(4, 0), (8, 0), (8, 1), (10, 2)])
def test_fri_fold(oracle_backend, rng, logn, depth):
    cases.check_fri_fold(oracle_backend, rng, logn, depth)


@pytest.mark.parametrize('logm,depth,nlayers,alg', [(5, 0, 1, 'blake2s256'), (9, 1, 3, 'blake2s256'), (12, 0, 4, 'sha256')])
def test_fri_layers(oracle_backend, rng, logm, depth, nlayers, alg):
    cases.check_fri_layers(oracle_backend, rng, logm, depth, nlayers, alg)


@pytest.mark.parametrize('logn,logsteps,ranks', [(8, 4, 1), (8, 4, 4), (10, 6, 8), (9, 4, 2)])
def test_coset_divisions(oracle_backend, rng, logn, logsteps, ranks):
    cases.check_coset_divisions(oracle_backend, rng, logn, logsteps, ranks)


def test_deferred_readbacks(oracle_backend, rng):
    cases.check_deferred_readbacks(oracle_backend, rng, 256)


def test_inverse_with_zeros(oracle_backend, rng):
    cases.check_inverse_with_zeros(oracle_backend, rng, 257)


@pytest.mark.parametrize('n', [1, 8, 64, 1024])
def test_power_series_and_shuffles(oracle_backend, rng, n):
    cases.check_power_series_and_shuffles(oracle_backend, rng, n)


def test_combine_many(oracle_backend, rng):
    cases.check_combine_many(oracle_backend, rng, 100, 5)


@pytest.mark.parametrize('logn', [0, 1, 2, 3, 5, 8, 10])
def test_ntt_full(oracle_backend, rng, logn):
    cases.check_ntt(oracle_backend, rng, logn, full=True)


@pytest.mark.parametrize('logn,plen,rows', [(6, 3, 2), (10, 64, 3), (12, 256, 1), (13, 8192, 1)])
def test_ntt_zero_extended(oracle_backend, rng, logn, plen, rows):
    cases.check_ntt(oracle_backend, rng, logn, poly_len=plen, rows=rows)


def test_small_polys(oracle_backend, rng):
    cases.check_small_polys(oracle_backend, rng)


@pytest.mark.parametrize('logn,depth', [(8, 0), (10, 1), (12, 2)])
def test_quartic(oracle_backend, rng, logn, depth):
    cases.check_quartic(oracle_backend, rng, logn, depth)


@pytest.mark.parametrize('alg', ['sha256', 'blake2s256'])
def test_hashing(oracle_backend, rng, alg):
    cases.check_hashing(oracle_backend, rng, alg, 64)


@pytest.mark.parametrize('alg,logn', [('sha256', 1), ('blake2s256', 1), ('blake2s256', 4), ('sha256', 7), ('blake2s256', 10)])
def test_merkle(oracle_backend, rng, alg, logn):
    cases.check_merkle(oracle_backend, rng, alg, logn)


@pytest.mark.parametrize('alg,logn,count', [('sha256', 1, 1), ('blake2s256', 1, 4), ('blake2s256', 6, 2), ('sha256', 8, 6), ('blake2s256', 11, 1)])
def test_merkle_commit_rows(oracle_backend, rng, alg, logn, count):
    cases.check_merkle_commit(oracle_backend, rng, alg, logn, count)


def test_device_record_ops(oracle_backend, rng):
    cases.check_device_record_ops(oracle_backend, rng)
    cases.check_combine_adjusted(oracle_backend, rng)


def test_mimc_air(oracle_backend, rng):
    cases.check_mimc_air(oracle_backend, rng, 128)


@pytest.mark.parametrize('logn,logsteps,per_row,lcount,adjusted', [(8, 4, [1], 0, False), (9, 5, [2, 1], 3, True), (10, 6, [4, 1, 3], 7, True),
                                                                   (8, 5, [1, 1], 6, False)])
def test_composition_tail(oracle_backend, rng, logn, logsteps, per_row, lcount, adjusted):
    cases.check_composition_tail(oracle_backend, rng, logn, logsteps, per_row, lcount, adjusted)
    cases.check_composition_tail(oracle_backend, rng, logn, logsteps, per_row, lcount, adjusted, made=True)
    cases.check_composition_tail_limits(oracle_backend)


@pytest.mark.parametrize('logn,logsteps,per_row,lcount,ranks', [(8, 4, [1], 2, 2), (10, 5, [2, 1], 3, 4), (11, 6, [4, 1, 3], 7, 8)])
def test_composition_tail_over_a_rank_s_coset(oracle_backend, rng, logn, logsteps, per_row, lcount, ranks):
    cases.check_composition_tail_coset(oracle_backend, rng, logn, logsteps, per_row, lcount, ranks)


def test_constraints_read_in_place_from_the_evaluation_domain(oracle_backend):
    from genstark_amd.field import PrimeField
    from genstark_amd.poseidon import poseidon6x128_air
    f = PrimeField(backend=oracle_backend)
    cases.check_constraints_strided(oracle_backend, poseidon6x128_air(128, 16, f, segmented=True), [[1, 2, 3, 4], [5, 6, 7, 8]])


# ---- known-answer values held by the reference's own example programs --------------------------------
def test_kat_foo_and_fibonacci():
    """README.md:23,42-45 (Foo: 1 -> 127 after 64 steps of +2) and examples/demo/fibonacci.ts:9-11
    over p = 2^32 - 3*2^25 + 1: pins add for a second modulus (generic big-int field of the oracle)."""
    f = pyref.Field(2**32 - 3 * 2**25 + 1)
    x = 1
    for _ in range(63):
        x = f.add(x, 2)
    assert x == 127
    for steps, result in ((2**6, 1783540607), (2**13, 203257732), (2**17, 2391373091)):
        r0 = r1 = 1
        for _ in range(steps - 1):
            a0 = f.add(r0, r1)
            r0, r1 = a0, f.add(a0, r1)
        assert r1 == result


def test_kat_rescue_4x128():
    """examples/rescue/hash4x128.ts:115-118 — registers 0 and 1 after 32 steps for inputs (42, 43):
    a known answer over THE 128-bit field computed by the reference authors (exp with a 127-bit exponent,
    MDS multiplication, additions).  The permutation is restated from examples/rescue/utils.ts:96-124,150-227;
    the constants are data copied from hash4x128.ts:17-34."""
    import rescue_kat
    assert rescue_kat.run(pyref.Field()) == (302524937772545017647250309501879538110, 205025454306577433144586673939030012640)


def test_kat_rescue_4x128_through_oracle_kernels(oracle_backend):
    cases.check_rescue_kat(oracle_backend)


@pytest.mark.parametrize('n,count', [(33, 65), (1000, 150)])
def test_more_vectors_than_one_launch_carries(oracle_backend, rng, n, count):
    cases.check_many_vectors(oracle_backend, rng, n, count)
