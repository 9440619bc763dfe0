"""Helpers of the distributed-prover tests: the thread communicator (oracle/comm_threads.c, a test double) and a runner that proves
one statement on G ranks = G threads, each with its own context of the given ABI library."""
import ctypes as C
import os
import threading

from conftest import ROOT
from genstark_amd._abi import Backend
from genstark_amd.native import GsComm, NativeProver

COMM_LIB = os.path.join(ROOT, 'oracle', 'libcomm_threads.so')


def thread_comms(backend, size):
    lib = C.CDLL(COMM_LIB)
    lib.gs_threads_comm_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(GsComm)]
    lib.gs_threads_comm_create.restype = C.c_int
    arr = (GsComm * size)()
    rc = lib.gs_threads_comm_create(size, C.c_void_p(backend.lib._handle), arr)
    assert rc == 0, rc
    return arr, lib


def prove_on_ranks(make_backend, make_stark, size, assertions, inputs, seed, fri_gather_below=0, solo_below=1):
    """-> (list of proof bytes per rank, list of collectives of rank 0).  make_backend() -> Backend; make_stark(backend) -> Stark.
    solo_below = 1: the statement is sharded whatever its size (the tests' statements are small; the driver's default, 0, hands a
    statement with fewer than 2^20 points per rank to rank 0 alone)."""
    backends = [make_backend() for _ in range(size)]
    provers = [NativeProver(make_stark(be)) for be in backends]
    comms, keep = thread_comms(backends[0], size)
    for r in range(size):
        comms[r].fri_gather_below = fri_gather_below          # 0: the driver's default; 1: every FRI layer stays on strided shares
        comms[r].solo_below = solo_below
    out, errs, colls = [None] * size, [None] * size, [None]

    def run(r):
        try:
            out[r] = provers[r].prove_bytes(assertions, inputs, seed, comm=comms[r])
            if r == 0:
                colls[0] = provers[r].last_collectives()
        except BaseException as e:      # noqa: BLE001
            errs[r] = e
    threads = [threading.Thread(target=run, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads), 'a rank is stuck'
    for e in errs:
        if e is not None:
            raise e
    for be in backends:
        be.close()
    return out, colls[0]
