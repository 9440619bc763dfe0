"""ProverPool (genstark_amd/pipeline.py): several prover lanes, each with its own library context, must produce exactly
the proofs the sequential prove() produces (golden proofs of tests/golden/oracle_proofs.json), in job order, and must
surface errors of a lane in the caller.  CPU: the oracle's implementation of the C ABI; GPU: the HIP library."""
import hashlib
import json
import os

import pytest

import genstark_amd as ga
from genstark_amd._abi import Backend
from genstark_amd.errors import StarkError
from genstark_amd.pipeline import ProverPool
from conftest import ORACLE_LIB

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'oracle_proofs.json')) as f:
    GOLDEN = json.load(f)
CASE = GOLDEN[2]          # 256 steps, extension factor 16


def factory(case):
    options = {'hashAlgorithm': case['hash_algorithm'], 'extensionFactor': case['extension_factor'],
               'exeQueryCount': case['exe_query_count'], 'friQueryCount': case['fri_query_count']}
    return lambda backend: ga.instantiateMimc(case['steps'], options, None, backend=backend)


def jobs_for(stark, steps, seeds):
    out = []
    for seed in seeds:
        trace = stark.generateExecutionTrace([], [seed])['dTrace']
        a = [{'step': 0, 'register': 0, 'value': trace.getValue(0, 0)},
             {'step': steps - 1, 'register': 0, 'value': trace.getValue(0, steps - 1)}]
        out.append((a, [], [seed]))
    return out


def check_pool(make_backend, lanes):
    case = CASE
    seq = factory(case)(make_backend())
    seeds = [case['seed'], 5, 7, 11, 13, 17, 19]
    jobs = jobs_for(seq, case['steps'], seeds)
    expected = [seq.serialize(seq.prove(*j)) for j in jobs]
    assert hashlib.sha256(expected[0]).hexdigest() == case['proofSha256']     # job 0 is the golden case itself
    with ProverPool(factory(case), lanes=lanes, backend_factory=make_backend) as pool:
        assert len({id(s.air.field.backend) for s in pool.starks}) == lanes   # one context per lane
        pool.on_every_lane(lambda s: s.prove(*jobs[0]))
        for _ in range(2):
            proofs = pool.prove_many(jobs)
            assert [seq.serialize(p) for p in proofs] == expected
        assert seq.verify(jobs[3][0], proofs[3])
        bad = (jobs[1][0][:1] + [{'step': 1, 'register': 0, 'value': 1}], [], [5])  # assertion the trace does not satisfy
        with pytest.raises(StarkError):
            pool.prove_many([jobs[0], bad, jobs[2]])
        assert seq.serialize(pool.prove_many(jobs[:1])[0]) == expected[0]      # the pool survives a failed job


def check_native_pool(make_backend, lanes):
    case = CASE
    seq = factory(case)(make_backend())
    jobs = jobs_for(seq, case['steps'], [case['seed'], 5, 7, 11, 13])
    expected = [seq.serialize(seq.prove(*j)) for j in jobs]
    with ProverPool(factory(case), lanes=lanes, backend_factory=make_backend, native=True) as pool:
        pool.on_every_lane(lambda s: s.prove_bytes(*jobs[0]))
        assert pool.prove_many_bytes(jobs) == expected
        assert [seq.serialize(p) for p in pool.prove_many(jobs)] == expected      # NativeProver.prove parses its own bytes


def test_native_pool_matches_sequential_proofs_oracle_backend(oracle_backend):
    check_native_pool(lambda: Backend(lib_path=ORACLE_LIB, allow_test_double=True), 3)


@pytest.mark.gpu
def test_native_pool_matches_sequential_proofs_hip():
    check_native_pool(lambda: Backend(device=0), 4)


@pytest.mark.parametrize('lanes', [1, 3])
def test_pool_matches_sequential_proofs_oracle_backend(oracle_backend, lanes):
    check_pool(lambda: Backend(lib_path=ORACLE_LIB, allow_test_double=True), lanes)


def test_pool_reports_missing_library():
    with pytest.raises(Exception):
        ProverPool(factory(CASE), lanes=2, backend_factory=lambda: Backend(lib_path='/nonexistent/libgstark_hip.so'))


@pytest.mark.gpu
def test_pool_matches_sequential_proofs_hip():
    check_pool(lambda: Backend(device=0), 3)


@pytest.mark.gpu
def test_pool_compiles_air_programs_and_matches_sequential_proofs_hip():
    """A pool's lanes have AIR programs compiled (gs_air_jit, once per process): same bytes as the sequential, interpreted prover."""
    from genstark_amd.field import PrimeField
    from genstark_amd.poseidon import poseidon6x128_air
    from genstark_amd._mirror.stark import Stark
    opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 40, 'friQueryCount': 16}
    make = lambda backend: Stark(poseidon6x128_air(1024, 16, PrimeField(backend=backend), segmented=True), opts)
    seeds = [[1 + s, 2, 3 + s, 4] for s in range(16)]
    seq_backend = Backend(device=0)
    seq = make(seq_backend)
    trace = seq.air.initProvingContext([], seeds).generateExecutionTrace()
    assertions = [{'step': 63, 'register': 0, 'value': trace.getValue(0, 63)}, {'step': 1023, 'register': 1, 'value': trace.getValue(1, 1023)}]
    expected = seq.serialize(seq.prove(assertions, [], seeds))
    assert seq_backend.jit_launches == 0
    jobs = [(assertions, [], seeds)] * 6
    for native in (False, True):
        with ProverPool(make, lanes=2, backend_factory=lambda: Backend(device=0), native=native) as pool:
            out = pool.prove_many_bytes(jobs) if native else [seq.serialize(p) for p in pool.prove_many(jobs)]
            assert out == [expected] * 6
            launches = pool.on_every_lane(lambda s: (s.stark if native else s).air.field.backend.jit_launches)
            assert all(n > 0 for n in launches)


def test_pool_constructor_failure_releases_the_healthy_lanes(oracle_backend):
    """One lane failing to come up must not leave the others blocked on the job queue forever."""
    import threading
    calls, lock = [], threading.Lock()
    make = factory(CASE)

    def flaky(backend):
        with lock:
            calls.append(1)
            n = len(calls)
        if n == 2:
            raise RuntimeError('lane construction failed')
        return make(backend)
    before = {t.ident for t in threading.enumerate()}
    with pytest.raises(RuntimeError, match='lane construction failed'):
        ProverPool(flaky, lanes=3, backend_factory=lambda: Backend(lib_path=ORACLE_LIB, allow_test_double=True), native=True)
    assert len(calls) == 3
    assert {t.ident for t in threading.enumerate()} <= before       # every lane thread was joined
