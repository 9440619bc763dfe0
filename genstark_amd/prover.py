"""genstark_amd.prover — the product's prove() for Python callers: an AIR + StarkOptions -> serialized proofs through the NATIVE driver
(csrc/prover.cc; csrc/prover_dist.h when a communicator is given), with nothing of the Python mirror in between.

    air    = MimcAir(steps, extensionFactor, PrimeField(backend=Backend()))        # or GenericAir / airassembly.instantiate(...).air
    prover = Prover(air, {'hashAlgorithm': 'blake2s256', 'exeQueryCount': 48, 'friQueryCount': 24})
    data   = prover.prove_bytes(assertions, inputs, seed)                          # Serializer.serializeProof bytes (lib/Serializer.ts:35-79)
    data   = prover.prove_bytes(assertions, inputs, seed, comm=RcclComm(...).comm)  # ONE proof across the communicator's ranks

Options follow lib/Stark.ts:13-20, 318-344 (defaults 80 / 40 queries, sha256; limits 128 / 64).  `verify` is the CPU verifier of
SURVEY 8f-1: it runs the restated lib/Stark.ts:167-248 (genstark_amd/_mirror), imported only when called.
"""
from ._abi import HASH_ALGS
from .native import NativeProver
from .serializer import Serializer

DEFAULT_EXE_QUERY_COUNT, DEFAULT_FRI_QUERY_COUNT = 80, 40          # lib/Stark.ts:13-17
MAX_EXE_QUERY_COUNT, MAX_FRI_QUERY_COUNT = 128, 64
DEFAULT_HASH_ALGORITHM = 'sha256'                                  # lib/Stark.ts:20
DIGEST_SIZE = 32


def security_options(options, extension_factor):
    """StarkOptions -> the four values a proof depends on, with the reference's defaults, limits and messages (lib/Stark.ts:318-344)."""
    options = dict(options or {})
    exe = options.get('exeQueryCount') or DEFAULT_EXE_QUERY_COUNT
    if not isinstance(exe, int) or isinstance(exe, bool) or not 1 <= exe <= MAX_EXE_QUERY_COUNT:
        raise TypeError(f'Execution sample size must be an integer between 1 and {MAX_EXE_QUERY_COUNT}')
    fri = options.get('friQueryCount') or DEFAULT_FRI_QUERY_COUNT
    if not isinstance(fri, int) or isinstance(fri, bool) or not 1 <= fri <= MAX_FRI_QUERY_COUNT:
        raise TypeError(f'FRI sample size must be an integer between 1 and {MAX_FRI_QUERY_COUNT}')
    alg = options.get('hashAlgorithm') or DEFAULT_HASH_ALGORITHM
    if alg not in HASH_ALGS:
        raise TypeError(f'Hash algorithm {alg} is not supported')
    if not extension_factor:
        raise TypeError('Extension factor is undefined')
    return {'extensionFactor': extension_factor, 'exeQueryCount': exe, 'friQueryCount': fri, 'hashAlgorithm': alg}


class Prover:
    def __init__(self, air, options=None):
        self.air = air
        self.options = security_options(options, air.extensionFactor)
        self.exeQueryCount, self.friQueryCount = self.options['exeQueryCount'], self.options['friQueryCount']
        self.hashAlg = HASH_ALGS[self.options['hashAlgorithm']]
        self.serializer = Serializer(air, DIGEST_SIZE)
        self._native = NativeProver(self)

    def prove_bytes(self, assertions, inputs=None, seed=None, comm=None):
        return self._native.prove_bytes(assertions, inputs, seed, comm=comm)

    def prove(self, assertions, inputs=None, seed=None, comm=None):
        """The proof object of lib/Stark.ts:157-162 (parsed from the driver's bytes)."""
        return self.serializer.parseProof(self.prove_bytes(assertions, inputs, seed, comm=comm))

    def pack_seed(self, seed):
        """The statement's first rows in the driver's wire form, packed once for many proofs (native.PackedSeed; generic AIRs)."""
        return self._native.pack_seed(seed)

    def parse(self, data):
        return self.serializer.parseProof(data)

    def last_stats(self):
        return self._native.last_stats()

    def sync_phases(self, on=True):
        """Measuring mode (this thread's next proofs): last_stats()['phases_readme'] = the reference's own phase log with device-synchronised times."""
        self._native.sync_phases(on)
        return self

    def last_collectives(self):
        return self._native.last_collectives()

    def verify_native(self, assertions, data, publicInputs=None):
        """Stark.verify of serialized proof bytes by the NATIVE verifier (csrc/verifier.h; CPU only, ~100x the Python verifier's speed):
        True or StarkError with the reference's message.  publicInputs: the values of the public input registers of an air-assembly
        component (lib/Stark.ts:167); the trace is sized from the shapes the proof carries."""
        return self._native.verify_bytes(assertions, data, publicInputs)

    def verify(self, assertions, proof, publicInputs=None):
        """lib/Stark.ts:167-248 on the CPU side of the same backend (the restated caller in genstark_amd/_mirror; a verifier that
        needs no device at all is Stark(air over HostField(), options).verify, tests/test_host_verifier.py)."""
        from ._mirror.stark import Stark
        stark = Stark(self.air, self.options)
        if isinstance(proof, (bytes, bytearray, memoryview)):
            proof = stark.parse(bytes(proof))
        return stark.verify(assertions, proof, publicInputs) if publicInputs is not None else stark.verify(assertions, proof)
