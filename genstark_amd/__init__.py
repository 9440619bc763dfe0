"""genstark_amd — MI355X-native prime-field polynomial + Merkle backend behind genSTARK's prove().

Layout (only what the hot path needs):
  csrc/        hand-written HIP kernels for gfx950 + the C ABI (include/gstark.h) -> libgstark_hip.so (+ one flavour per other field);
               prover.cc: the native prove() driver above the ABI -> libgstark_prover.so
  _abi.py      ctypes binding of the C ABI (loads the HIP library; no CPU fallback)
  field.py     galois FiniteField / Vector / Matrix surface       (device-resident data)
  merkle.py    merkle Hash / MerkleTree surface
  air.py       air-assembly AirModule / ProvingContext surface for the MiMC AIR (dedicated kernels)
  air_generic.py  the same surface for AIRs given as expressions (register-machine programs, secret registers, segments);
               rescue.py, poseidon.py: the example hash AIRs; lib128.py, lib224.py: the exports of assembly/lib128.aa / lib224.aa;
               pointmul.py: examples/elliptic/pointmul.aa; airassembly.py: AirAssembly source -> AirModule (index.ts:18-33 instantiate)
  prover.py    THE PRODUCT'S prove(): Prover(air, options) -> the native driver (csrc/prover.cc, csrc/prover_dist.h); native.py: its binding
  comm.py      communicators for one proof across several GPUs (RCCL: csrc/comm_rccl.cc)
  pipeline.py  several proofs in flight on one GPU (throughput mode)
  serializer.py, utils.py, hostfield.py   wire format and the CPU verifier's field
  _mirror/     CHECKER ONLY: lib/Stark.ts and lib/components/*.ts restated line by line (stark.py, components/), and the Python
               SPMD forms of the multi-GPU prover (distributed.py, sharded.py) — what the native drivers are compared with
"""
from ._abi import Backend, GstarkError, HIP_LIB_PATH
from .air import MimcAir, runMimc, sha256_prng
from .air_generic import GenericAir
from .errors import StarkError
from .field import MODULUS, Matrix, PrimeField, Vector, createPrimeField
from .merkle import Hash, MerkleTree, createHash
from .prover import Prover
from .utils import Logger, NoopLogger


def mimcProver(steps, options=None, backend=None):
    """The MiMC STARK of examples/mimc/mimc128Assembly.ts as a product prover: MimcAir + Prover (native driver)."""
    options = dict(options or {})
    return Prover(MimcAir(steps, options.get('extensionFactor'), PrimeField(MODULUS, backend)), options)


def instantiateMimc(steps, options=None, logger=None, backend=None):
    """index.ts:18-33 `instantiate(source, component, options, logger)` for the MiMC AirAssembly source of
    examples/mimc/mimc128Assembly.ts, returning the MIRROR's Stark object (prove / verify / serialize as lib/Stark.ts has them):
    what the tests and bench.py's cross-check drive.  The product path is mimcProver / Prover."""
    from ._mirror.stark import Stark
    options = dict(options or {})
    field = PrimeField(MODULUS, backend)
    air = MimcAir(steps, options.get('extensionFactor'), field)
    return Stark(air, options, logger)


def __getattr__(name):          # `genstark_amd.Stark`: the mirror, loaded on first use only
    if name == 'Stark':
        from ._mirror.stark import Stark
        return Stark
    raise AttributeError(name)
