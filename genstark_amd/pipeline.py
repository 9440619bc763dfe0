"""Throughput mode for a proving service: several independent prover lanes on ONE GPU.

A single prove() (lib/Stark.ts:81-163) alternates between one host core and the device: the MiMC execution trace is a serial
x -> x^3 + k recurrence that only a CPU core can run (9 ms of a 21 ms proof at 2^20 steps), and between the device phases
the host hashes Merkle roots into the next Fiat-Shamir seed while the GPU drains.  Proofs are independent of each other
(SURVEY.md section 8e), so a service that proves many statements hides both gaps by keeping several proofs in flight:
every lane owns its own library context (HIP stream, block cache, NTT plans) and Stark instance and runs the unmodified
prove(); lane A's trace generation (ctypes releases the GIL for the whole C call) overlaps lane B's kernels, and the
hardware scheduler interleaves the streams.  Nothing is shared between lanes, proofs are byte-identical to the
sequential ones (tests/test_pipeline.py).
"""
import queue
import threading

from ._abi import Backend


class ProverPool:
    def __init__(self, stark_factory, lanes=2, backend_factory=None, native=False, jit=True):
        """stark_factory(backend) -> genstark_amd.prover.Prover (or the mirror's Stark); backend_factory() -> Backend, called once inside every lane's thread (the HIP
        current device is per-thread state, gs_ctx_create sets it for the calling thread).  jit: a pool is a long-lived service,
        so its lanes have AIR programs compiled (gs_air_jit: once per process, shared by the lanes) instead of interpreted."""
        if lanes < 1:
            raise ValueError('lanes must be >= 1')
        self.lanes = lanes
        self.jit = jit
        self.native = native        # lanes prove through the native driver (genstark_amd/native.py): the GIL is released for a whole proof
        self._jobs = queue.Queue()
        self._errors = []
        self._ready = threading.Barrier(lanes + 1)
        self.starks = [None] * lanes
        self._threads = [threading.Thread(target=self._lane, args=(i, stark_factory, backend_factory or Backend), daemon=True)
                         for i in range(lanes)]
        for t in self._threads:
            t.start()
        self._ready.wait()
        if self._errors:
            self.close()                # the lanes that did come up are waiting for jobs: release and join them
            raise self._errors[0]

    def _lane(self, index, stark_factory, backend_factory):
        try:
            backend = backend_factory()
            if self.jit and hasattr(backend, 'jit'):
                backend.jit()
            stark = stark_factory(backend)
            if self.native and not hasattr(stark, 'prove_bytes'):      # a mirror Stark: wrap it; a Prover already is the native driver
                from .native import NativeProver
                stark = NativeProver(stark)
            self.starks[index] = stark
        except BaseException as e:      # surface construction failures (e.g. no HIP library) in the caller
            self._errors.append(e)
            stark = None
        self._ready.wait()
        while stark is not None:
            job = self._jobs.get()
            if job is None:
                return
            fn, slot, results, done = job
            try:
                results[slot] = fn(stark)
            except BaseException as e:
                results[slot] = e
            done.release()

    def _run(self, fns):
        results = [None] * len(fns)
        done = threading.Semaphore(0)
        for slot, fn in enumerate(fns):
            self._jobs.put((fn, slot, results, done))
        for _ in fns:
            done.acquire()
        for r in results:
            if isinstance(r, BaseException):
                raise r
        return results

    def prove_many(self, jobs):
        """jobs: iterable of (assertions, inputs, seed) as for Stark.prove; returns the proofs in job order."""
        return self._run([(lambda s, j=j: s.prove(*j)) for j in jobs])

    def prove_many_bytes(self, jobs):
        """Serialized proofs in job order (native lanes: NativeProver.prove_bytes)."""
        return self._run([(lambda s, j=j: s.prove_bytes(*j)) for j in jobs])

    def on_every_lane(self, fn):
        """Run fn(stark) once on EVERY lane (warm-up of plans and block caches, synchronisation before timing)."""
        gate = threading.Barrier(self.lanes)

        def wrapped(stark):
            gate.wait()                 # a lane that already took one task cannot take a second one
            return fn(stark)
        return self._run([wrapped] * self.lanes)

    def close(self):
        for _ in self._threads:
            self._jobs.put(None)
        for t in self._threads:
            t.join()
        self._threads = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
