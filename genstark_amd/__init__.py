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
  native.py    binding of the native prove() driver (same bytes as stark.py's prove + serialize)
  pipeline.py  several proofs in flight on one GPU (throughput mode)
  distributed.py  ONE proof across the GPUs of a node (distributed vectors under the unchanged prover); sharded.py: commitments
  components/  mirrors of lib/components/*.ts (the callers of the surface)
  stark.py     mirror of lib/Stark.ts (prove / verify / serialize / parse)
"""
from ._abi import Backend, GstarkError, HIP_LIB_PATH
from .air import MimcAir, runMimc, sha256_prng
from .air_generic import GenericAir
from .errors import StarkError
from .field import MODULUS, Matrix, PrimeField, Vector, createPrimeField
from .merkle import Hash, MerkleTree, createHash
from .stark import Stark
from .utils import Logger, NoopLogger


def instantiateMimc(steps, options=None, logger=None, backend=None):
    """index.ts:18-33 `instantiate(source, component, options, logger)` for the MiMC AirAssembly source
    of examples/mimc/mimc128Assembly.ts (the AirAssembly compiler is out of scope)."""
    options = dict(options or {})
    field = PrimeField(MODULUS, backend)
    air = MimcAir(steps, options.get('extensionFactor'), field)
    return Stark(air, options, logger)
