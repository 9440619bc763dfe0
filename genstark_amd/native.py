"""Binding of the native prove() driver (genstark_amd/csrc/prover.cc -> libgstark_prover.so).

`NativeProver(stark)` proves the same statements as the mirror's `stark.prove()` — same AIR, same options, same bytes
(tests/test_native_prover.py) — with the whole call sequence of lib/Stark.ts:81-163 issued by native host code: no interpreter
between two launches.  The product entry is genstark_amd.prover.Prover (an AIR + options, no mirror object involved).  The driver is bound to the ABI library the Stark's backend loaded
(the HIP library in the product; the oracle's implementation only when a test injected it), it never falls back to anything.
"""
import ctypes as C
import os
import threading

from ._abi import GstarkError
from .air import MimcAir
from .air_generic import GenericAir
from .errors import StarkError
from .field import _le

_HERE = os.path.dirname(os.path.abspath(__file__))
PROVER_LIB_PATH = os.path.join(_HERE, 'csrc', 'libgstark_prover.so')
# one build of the driver per field flavour, like the ABI library (csrc/build.sh): its host-side scalars are computed in that field
from ._abi import MODULUS_17, MODULUS_32, MODULUS_64, MODULUS_128, MODULUS_224, MODULUS_256   # noqa: E402
PROVER_LIB_PATHS = {MODULUS_128: PROVER_LIB_PATH, MODULUS_64: os.path.join(_HERE, 'csrc', 'libgstark_prover_q64.so'),
                    MODULUS_32: os.path.join(_HERE, 'csrc', 'libgstark_prover_q32.so'), MODULUS_17: os.path.join(_HERE, 'csrc', 'libgstark_prover_q17.so'),
                    MODULUS_256: os.path.join(_HERE, 'csrc', 'libgstark_prover_p256.so'), MODULUS_224: os.path.join(_HERE, 'csrc', 'libgstark_prover_p224.so')}
PROVER_LIB_PATH_RUNTIME = os.path.join(_HERE, 'csrc', 'libgstark_prover_rt.so')
ELT_MAX = 32          # GS_PROVER_ELT_MAX: the job's scalar fields (include/gstark_prover.h)


class _Assertion(C.Structure):
    _fields_ = [('step', C.c_uint64), ('reg', C.c_uint32), ('value', C.c_uint8 * ELT_MAX)]


class _InputRegister(C.Structure):      # struct gs_input_register
    _fields_ = [('parent', C.c_int32), ('peer', C.c_int32), ('steps', C.c_uint32), ('shift', C.c_int32), ('secret', C.c_uint32)]


class _StaticSource(C.Structure):       # struct gs_static_source
    _fields_ = [('kind', C.c_uint32), ('index', C.c_uint32)]


class _Air(C.Structure):
    _fields_ = [('kind', C.c_uint32), ('registers', C.c_uint32), ('nconstraints', C.c_uint32), ('degrees', C.POINTER(C.c_uint32)),
                ('seed', C.c_uint8 * ELT_MAX), ('round_constants', C.c_char_p), ('nrc', C.c_uint32), ('k_table', C.c_void_p), ('k_len', C.c_uint64),
                ('t_code', C.POINTER(C.c_uint32)), ('t_ninstr', C.c_uint32), ('i_code', C.POINTER(C.c_uint32)), ('i_ninstr', C.c_uint32),
                ('e_code', C.POINTER(C.c_uint32)), ('e_ninstr', C.c_uint32), ('consts', C.c_char_p), ('nconsts', C.c_uint32),
                ('vm_regs', C.c_uint32), ('static_values', C.c_char_p), ('static_periods', C.POINTER(C.c_uint32)), ('nstatic', C.c_uint32),
                ('static_tables', C.c_void_p), ('static_lens', C.POINTER(C.c_uint64)), ('first_rows', C.c_char_p), ('segments', C.c_uint64),
                ('segment_len', C.c_uint64), ('secret_traces', C.POINTER(C.c_void_p)), ('nsecret', C.c_uint32),
                ('inputs', C.POINTER(_InputRegister)), ('ninputs', C.c_uint32), ('input_shapes', C.POINTER(C.c_uint32)),
                ('static_sources', C.POINTER(_StaticSource)), ('public_inputs', C.c_char_p), ('public_input_counts', C.POINTER(C.c_uint64)),
                ('npublic_inputs', C.c_uint32)]


class _Job(C.Structure):
    _fields_ = [('steps', C.c_uint64), ('extension_factor', C.c_uint32), ('exe_query_count', C.c_uint32), ('fri_query_count', C.c_uint32),
                ('hash_alg', C.c_int32), ('root_of_unity', C.c_uint8 * ELT_MAX), ('assertions', C.POINTER(_Assertion)), ('nassertions', C.c_uint32),
                ('root_of_unity_log2', C.c_uint32), ('air', _Air)]


class _Stats(C.Structure):
    _fields_ = [('ntt_points', C.c_uint64), ('ntt_transforms', C.c_uint64), ('horner_points', C.c_uint64), ('nphases', C.c_uint32),
                ('total_ms', C.c_double), ('phase_ms', C.c_double * 16), ('phase_label', (C.c_char * 48) * 16),
                ('nreadme', C.c_uint32), ('readme_ms', C.c_double * 16), ('readme_label', (C.c_char * 160) * 16)]


class GsComm(C.Structure):
    """struct gs_comm of include/gstark_comm.h: the table of collectives gs_prover_prove_dist is handed (genstark_amd/comm.py builds them)."""
    _fields_ = [('self', C.c_void_p), ('rank', C.c_int32), ('size', C.c_int32),
                ('all_gather', C.c_void_p), ('all_to_all', C.c_void_p), ('take_timings', C.c_void_p), ('name', C.c_char_p),
                ('fri_gather_below', C.c_uint64), ('solo_below', C.c_uint64), ('fork', C.c_void_p), ('join', C.c_void_p)]


class _Collective(C.Structure):
    _fields_ = [('label', C.c_char * 40), ('kind', C.c_uint32), ('bytes', C.c_uint64), ('ms', C.c_double)]


_libs = {}           # driver library path -> CDLL (one image per field flavour)
_bindings = {}       # (driver path, ABI library handle) -> gs_prover_binding*
_bound_lock = threading.Lock()      # ProverPool lanes construct their NativeProver concurrently


def _driver(backend):
    """(driver library, binding) for the ABI library of `backend`: the build of libgstark_prover*.so for the backend's field, and a
    binding of its own for that ABI library (gs_prover_open) — any number of ABI libraries (the HIP library, a test double) share one
    driver image."""
    path = PROVER_LIB_PATHS.get(backend.modulus)
    if path is None:
        if backend.element_size != 32:
            raise GstarkError(f'no build of the native driver for the field of {backend.modulus} elements')
        path = PROVER_LIB_PATH_RUNTIME         # the runtime-modulus build: it adopts the modulus of the library it is bound to
    if os.environ.get('GSTARK_PROVER_LIB_DIR'):       # instrumented builds of the same driver (tools/build_sanitized.sh): same file names, another directory
        path = os.path.join(os.environ['GSTARK_PROVER_LIB_DIR'], os.path.basename(path))
    with _bound_lock:
        lib = _libs.get(path)
        if lib is None:
            if not os.path.exists(path):
                raise GstarkError(f'{path} is missing: run `python -c "import __graft_entry__ as g; g.build()"`')
            lib = C.CDLL(path, mode=getattr(os, 'RTLD_LOCAL', 0) | getattr(os, 'RTLD_NOW', 2))
            lib.gs_prover_open.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
            lib.gs_prover_open.restype = C.c_int
            lib.gs_prover_element_size.argtypes = []
            lib.gs_prover_element_size.restype = C.c_int
            lib.gs_prover_prove_on.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Job), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64]
            lib.gs_prover_prove_on.restype = C.c_int
            lib.gs_prover_verify_on.argtypes = [C.c_void_p, C.POINTER(_Job), C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
            lib.gs_prover_verify_on.restype = C.c_int
            lib.gs_prover_last_stats.argtypes = [C.POINTER(_Stats)]
            lib.gs_prover_last_stats.restype = C.c_int
            lib.gs_prover_sync_phases.argtypes = [C.c_int]
            lib.gs_prover_sync_phases.restype = None
            lib.gs_prover_member_sequence.argtypes = [C.c_int]
            lib.gs_prover_member_sequence.restype = None
            lib.gs_prover_prove_dist_on.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Job), C.POINTER(GsComm), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64]
            lib.gs_prover_prove_dist_on.restype = C.c_int
            lib.gs_prover_last_collectives.argtypes = [C.POINTER(_Collective), C.c_uint32, C.POINTER(C.c_uint32)]
            lib.gs_prover_last_collectives.restype = C.c_int
            lib.gs_prover_remainder_check_on.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_char_p, C.c_int]
            lib.gs_prover_remainder_check_on.restype = C.c_int
            lib.gs_prover_abi_version.argtypes, lib.gs_prover_abi_version.restype = [], C.c_int
            if lib.gs_prover_abi_version() != 2:          # GS_PROVER_ABI_VERSION: the struct layouts this module mirrors (_Job, _Air, GsComm)
                raise GstarkError(f'{path} speaks struct layout {lib.gs_prover_abi_version()}, this binding 2: rebuild (genstark_amd/csrc/build.sh)')
            if lib.gs_prover_element_size() != backend.element_size:
                raise GstarkError(f'{path} is built for {lib.gs_prover_element_size()}-byte elements')
            _libs[path] = lib
        key = (path, backend.lib._handle)
        if key not in _bindings:
            b = C.c_void_p()
            rc = lib.gs_prover_open(C.c_void_p(backend.lib._handle), C.byref(b))
            if rc:
                raise GstarkError(f'gs_prover_open failed ({rc}): the ABI library lacks an entry point the driver needs, or computes in another field')
            _bindings[key] = b
        return lib, _bindings[key]


def _is_assembly(air):
    from .airassembly import AssemblyAir        # (airassembly imports air_generic, which this module imports too: resolved at call time)
    return isinstance(air, AssemblyAir)


class _Shim:
    """what NativeProver reads of a `stark`: the inner GenericAir of an AssemblyAir for one shape, with the outer prover's options"""

    def __init__(self, air, exe, fri, alg):
        self.air, self.exeQueryCount, self.friQueryCount, self.hashAlg = air, exe, fri, alg


class PackedSeed:
    """The first rows of a statement already in the driver's wire form (registers x 16 bytes per row): `NativeProver.pack_seed(seed)`.
    Passing it as `seed` keeps the per-proof job packing out of prove_bytes — what a caller whose inputs already are bytes (a node
    Buffer, a C array) never pays, and what costs ~0.1 us per value when they are Python integers (10 ms for 16 384 Poseidon chains)."""

    def __init__(self, first, rows):
        self.first, self.rows = first, rows


class NativeProver:
    def __init__(self, stark):
        """stark: anything that names the statement's AIR and security options — genstark_amd.prover.Prover (the product entry), or
        the mirror's Stark (tests compare the two drivers on one object)."""
        self.stark = stark
        air = stark.air
        if hasattr(stark, 'indexGenerator'):         # the mirror's Stark
            self._exe, self._fri, self._alg = stark.indexGenerator.exeQueryCount, stark.indexGenerator.friQueryCount, stark.hash.alg
        else:
            self._exe, self._fri, self._alg = stark.exeQueryCount, stark.friQueryCount, stark.hashAlg
        self.field = air.field
        self.backend = self.field.backend
        self.lib, self.binding = _driver(self.backend)       # the driver build for this field, bound to this backend's ABI library
        self._static_pack = None
        self._keep = []
        self._out = C.create_string_buffer(1 << 22)
        # AIR-instance constants the device routines need, computed once (they depend on the AIR only, like the static register
        # polynomials an air-assembly module holds)
        if isinstance(air, MimcAir):
            ctx = air.initProvingContext([], [0])
            self.kind, self.degrees = 0, [3]
            self._kTable = ctx._kTable
            self._rc = b''.join(self.field.le(k) for k in air.roundConstants)
            self.rootOfUnity = ctx.rootOfUnity
        elif isinstance(air, GenericAir):
            rows = [[0] * air.traceRegisterCount]
            from .air_generic import GenericProvingContext
            self.kind, self.degrees = 1, list(air.constraintDegrees)
            if not air.secretInputCount:      # public static registers only: the tables are constants of the AIR
                ctx = GenericProvingContext(air, rows * (air.steps // air.segmentLength if air.segmentLength else 1))
                self._tables, self._lens = ctx._staticTables, ctx._staticLens
            self.rootOfUnity = air.rootOfUnity
        elif _is_assembly(air):
            # an air-assembly component (genstark_amd/airassembly.py): shape-agnostic — the trace is sized from the inputs when they
            # arrive (prove) or from the shapes the proof carries (verify: read by the native verifier itself, csrc/verifier.h)
            self.kind, self.degrees = 1, list(air.constraintDegrees)
            self._inner = {}                             # id(inner GenericAir of one shape) -> its NativeProver
            regs = air.inputRegisters
            self._input_decl = (_InputRegister * max(len(regs), 1))()
            for j, d in enumerate(regs):
                self._input_decl[j].parent = -1 if d['parent'] is None else d['parent']
                self._input_decl[j].peer = -1 if d['peer'] is None else d['peer']
                self._input_decl[j].steps = d['steps'] or 0
                self._input_decl[j].shift = d['shift']
                self._input_decl[j].secret = 1 if d['secret'] else 0
            self._ninputs = len(regs)
            self.rootOfUnity = None
        else:
            raise GstarkError('the native driver knows the MiMC AIR, GenericAir and AssemblyAir')
        self._shape_args = None      # (declarations, count, shapes) of the AssemblyAir this prover is the inner prover of

    def prove_bytes(self, assertions, inputs=None, seed=None, comm=None):
        """The serialized proof of Stark.prove(assertions, inputs, seed): stark.serialize(stark.prove(...)) byte for byte.
        comm (a GsComm, genstark_amd/comm.py): ONE proof across the communicator's ranks (gs_prover_prove_dist; every rank calls this
        with the same statement and receives the same bytes)."""
        stark, air, f = self.stark, self.stark.air, self.field
        if not isinstance(assertions, list):
            raise TypeError('Assertions parameter must be an array')
        if len(assertions) == 0:
            raise TypeError('At least one assertion must be provided')
        if _is_assembly(air):
            # initProvingContext(inputs, seed) (lib/Stark.ts:90): the loader lays the inputs out — the inner AIR of this shape, the secret
            # registers' columns, the first row(s) — and the native driver proves that AIR, writing the inputs' shapes into the proof
            inner, packed, firsts, shapes = air.plan(inputs, seed)
            nat = self._inner.get(id(inner))
            if nat is None:
                if len(self._inner) > 8:
                    self._inner.clear()
                nat = self._inner[id(inner)] = NativeProver(_Shim(inner, self._exe, self._fri, self._alg))
            flat = [w for sh in shapes for w in [len(sh)] + list(sh)]
            nat._shape_args = (self._input_decl, self._ninputs, (C.c_uint32 * max(len(flat), 1))(*flat))
            return nat.prove_bytes(assertions, packed, firsts, comm=comm)
        job = _Job()
        job.steps, job.extension_factor = air.steps, air.extensionFactor
        job.exe_query_count, job.fri_query_count = self._exe, self._fri
        job.hash_alg = self._alg
        es = f.elementSize
        job.root_of_unity[:es] = f.le(self.rootOfUnity)
        arr = (_Assertion * len(assertions))()
        for i, a in enumerate(assertions):
            if a['register'] < 0 or a['step'] < 0:
                raise ValueError('Invalid assertion')
            arr[i].step, arr[i].reg = a['step'], a['register']
            arr[i].value[:es] = f.le(a['value'] % f.modulus)
        job.assertions, job.nassertions = arr, len(assertions)
        ja = job.air
        ja.kind, ja.registers, ja.nconstraints = self.kind, air.traceRegisterCount, len(self.degrees)
        degrees = (C.c_uint32 * len(self.degrees))(*self.degrees)
        ja.degrees = degrees
        keep = [arr, degrees]
        if self._shape_args is not None:
            ja.inputs, ja.ninputs, ja.input_shapes = self._shape_args
        if self.kind == 0:
            ja.seed[:es] = f.le((seed or [0])[0] % f.modulus)
            ja.round_constants, ja.nrc = self._rc, len(air.roundConstants)
            ja.k_table, ja.k_len = self._kTable.ptr, self._kTable.length
        else:
            t_code, t_n, consts, nconsts, nregs = air.transitionProgram.abi_args(es)
            e_code, e_n, consts, nconsts, _ = air.evaluationProgram.abi_args(es) if air.evaluationProgram.consts is air.transitionProgram.consts \
                else (None, 0, None, 0, 0)
            if e_code is None:
                # transition and evaluator have separate constant pools: concatenate and rebase the evaluator's constant indexes
                e_prog, t_prog = air.evaluationProgram, air.transitionProgram
                base = len(t_prog.consts)
                from .air_generic import OP_LOADC, OP_POWC
                code = []
                for op, d, a, b in e_prog.code:
                    if op == OP_LOADC:
                        a += base
                    elif op == OP_POWC:
                        b += base
                    code.extend((op, d, a, b))
                e_code, e_n = (C.c_uint32 * len(code))(*code), len(e_prog.code)
                pool = list(t_prog.consts) + list(e_prog.consts)
                consts, nconsts = b''.join(int(v).to_bytes(es, 'little') for v in pool), len(pool)
                nregs = max(t_prog.nregs, e_prog.nregs)
            ja.t_code, ja.t_ninstr, ja.e_code, ja.e_ninstr = t_code, t_n, e_code, e_n
            ja.consts, ja.nconsts, ja.vm_regs = consts, nconsts, nregs
            if air.initProgram is not None:
                i_code, i_n = air.initProgram.abi_args(es)[:2]
                ja.i_code, ja.i_ninstr = i_code, i_n
                keep.append(i_code)
                ja.vm_regs = max(ja.vm_regs, air.initProgram.nregs)
            packed = seed if isinstance(seed, PackedSeed) else None
            rows = air.firstRows(seed) if packed is None else None
            if air.secretInputCount:
                if packed is not None:
                    raise GstarkError('a packed seed cannot be combined with secret input registers (their tables are built from the rows)')
                # secret registers come with the proof's inputs: their tables and low-degree extensions are per-proof device data
                from .air_generic import GenericProvingContext
                ctx = GenericProvingContext(air, rows, inputs)
                svals, plist = ctx.staticValuesPacked()
                tables, table_lens = ctx._staticTables, ctx._staticLens
                straces = (C.c_void_p * air.secretInputCount)(*[v.ptr for v in ctx.secretRegisterTraces])
                ja.secret_traces, ja.nsecret = straces, air.secretInputCount
                keep += [ctx, straces]
            else:
                # public static registers are constants of the AIR: packed once per prover, not per proof (~500 values for Poseidon)
                if self._static_pack is None:
                    self._static_pack = (b''.join(f.le(v % f.modulus) for values in air.staticRegisters for v in values) or bytes(es),
                                         [len(v) for v in air.staticRegisters])
                tables, table_lens = self._tables, self._lens
                svals, plist = self._static_pack
            periods = (C.c_uint32 * max(len(plist), 1))(*plist)
            lens = (C.c_uint64 * max(len(table_lens), 1))(*table_lens)
            ja.static_values, ja.static_periods, ja.nstatic = svals, periods, len(plist)
            ja.static_tables, ja.static_lens = tables.ptr, lens
            # thousands of values per proof for a segmented AIR: no per-value function call, no reduction of values already in range
            # (exact ints take the fast path; integer-likes such as numpy.int64 go through int())
            first, nrows = (packed.first, packed.rows) if packed is not None else (self._pack_rows(rows), len(rows))
            ja.first_rows = first
            ja.segments, ja.segment_len = (nrows, air.segmentLength) if air.segmentLength else (0, 0)
            keep += [t_code, e_code, consts, svals, periods, lens, first]
        cap = 1 << 22
        out = self._out        # one output buffer per prover (a lane of a pool has its own prover): no 4 MB allocation + copy per proof
        n = C.c_uint64()
        err = C.create_string_buffer(512)
        if comm is None:
            rc = self.lib.gs_prover_prove_on(self.binding, self.backend.ctx, C.byref(job), C.cast(out, C.c_void_p), cap, C.byref(n), err, 512)
        else:
            rc = self.lib.gs_prover_prove_dist_on(self.binding, self.backend.ctx, C.byref(job), C.byref(comm), C.cast(out, C.c_void_p), cap, C.byref(n), err, 512)
        if rc:
            raise StarkError(f'native prove() failed ({rc}): {err.value.decode(errors="replace")}')
        return C.string_at(out, n.value)

    def verify_bytes(self, assertions, data, publicInputs=None):
        """Stark.verify(assertions, stark.parse(data), publicInputs) natively (csrc/verifier.h: lib/Stark.ts:167-248 +
        LowDegreeProver.ts:70-172 on host scalars, no device work): True, or StarkError with the reference's message.  For an
        air-assembly component with input registers the trace length is not part of the job: the verifier reads the inputs' shapes
        from the proof, as the reference does (lib/Stark.ts:176), and lays the PUBLIC registers' values (publicInputs) out itself."""
        air, f = self.stark.air, self.field
        if not isinstance(assertions, list) or len(assertions) == 0:
            raise TypeError('At least one assertion must be provided')
        es = f.elementSize
        job = _Job()
        assembly = _is_assembly(air)
        job.extension_factor = air.extensionFactor
        job.exe_query_count, job.fri_query_count, job.hash_alg = self._exe, self._fri, self._alg
        if assembly:
            # steps = 0: whatever the proof's shapes lay out; the root of unity of the field's largest power-of-two order, squared
            # down by the driver to the evaluation domain's (root_of_unity_log2)
            job.steps = 0
            adicity = ((f.modulus - 1) & -(f.modulus - 1)).bit_length() - 1
            log2 = min(adicity, 32)
            job.root_of_unity[:es] = f.le(f.getRootOfUnity(1 << log2))
            job.root_of_unity_log2 = log2
        else:
            job.steps = air.steps
            job.root_of_unity[:es] = f.le(self.rootOfUnity)
        arr = (_Assertion * len(assertions))()
        for i, a in enumerate(assertions):
            if a['register'] < 0 or a['step'] < 0:
                raise ValueError('Invalid assertion')
            arr[i].step, arr[i].reg = a['step'], a['register']
            arr[i].value[:es] = f.le(a['value'] % f.modulus)
        job.assertions, job.nassertions = arr, len(assertions)
        ja = job.air
        ja.kind, ja.registers, ja.nconstraints = self.kind, air.traceRegisterCount, len(self.degrees)
        degrees = (C.c_uint32 * len(self.degrees))(*self.degrees)
        ja.degrees = degrees
        keep = [arr, degrees]
        if self.kind == 0:
            ja.round_constants, ja.nrc = self._rc, len(air.roundConstants)
        elif assembly:
            nsec = air.secretInputCount
            e_code, e_n, consts, nconsts, nregs = air.evaluationProgram.abi_args(es)
            sources, cycles = air.staticSources()
            src = (_StaticSource * max(len(sources), 1))()
            for i, (kind, index) in enumerate(sources):
                src[i].kind, src[i].index = kind, index
            pub = b''.join(f.le(v % f.modulus) for values in cycles for v in values) or bytes(es)
            periods = (C.c_uint32 * max(len(cycles), 1))(*[len(v) for v in cycles])
            ja.e_code, ja.e_ninstr, ja.consts, ja.nconsts, ja.vm_regs = e_code, e_n, consts, nconsts, nregs
            ja.static_values, ja.static_periods, ja.nstatic, ja.nsecret = pub, periods, len(sources), nsec
            ja.static_sources = src
            ja.inputs, ja.ninputs = self._input_decl, self._ninputs
            # publicInputs: the values of the public input registers in declaration order (lib/Stark.ts:167), flattened row-major
            given = list(publicInputs or [])
            flat_all, counts = [], []
            for values in given:
                flat = values
                while flat and isinstance(flat[0], (list, tuple)):
                    flat = [v for group in flat for v in group]
                if any(isinstance(v, (list, tuple)) for v in flat):
                    raise GstarkError('input register: ragged values')
                counts.append(len(flat))
                flat_all.extend(flat)
            p_ = f.modulus
            pvals = b''.join([(v if type(v) is int and 0 <= v < p_ else int(v) % p_).to_bytes(es, 'little') for v in flat_all]) or bytes(es)
            pcounts = (C.c_uint64 * max(len(counts), 1))(*counts)
            ja.public_inputs, ja.public_input_counts, ja.npublic_inputs = pvals, pcounts, len(counts)
            keep += [e_code, consts, pub, periods, src, pvals, pcounts]
        else:
            nsec = air.secretInputCount
            e_code, e_n, consts, nconsts, nregs = air.evaluationProgram.abi_args(es)
            pub = b''.join(f.le(v % f.modulus) for values in air.staticRegisters for v in values) or bytes(es)
            plist = [len(v) for v in air.staticRegisters] + [1] * nsec          # the secret registers' values come with the proof's leaves
            periods = (C.c_uint32 * max(len(plist), 1))(*plist)
            ja.e_code, ja.e_ninstr, ja.consts, ja.nconsts, ja.vm_regs = e_code, e_n, consts, nconsts, nregs
            ja.static_values, ja.static_periods, ja.nstatic, ja.nsecret = pub, periods, len(plist), nsec
            keep += [e_code, consts, pub, periods]
        err = C.create_string_buffer(512)
        data = bytes(data)
        rc = self.lib.gs_prover_verify_on(self.binding, C.byref(job), data, len(data), err, 512)
        if rc == -3:      # GS_ERR_UNSUPPORTED
            raise GstarkError(err.value.decode(errors='replace') or 'unsupported by the native verifier')
        if rc:
            raise StarkError(err.value.decode(errors='replace') or f'verification failed ({rc})')
        return True

    def _pack_rows(self, rows):
        f = self.field
        p_, tb, es = f.modulus, int.to_bytes, f.elementSize
        return b''.join([tb(v, es, 'little') if type(v) is int and 0 <= v < p_ else tb(int(v) % p_, es, 'little') for row in rows for v in row])

    def pack_seed(self, seed):
        """seed (as prove_bytes takes it) -> PackedSeed: the AIR's init() applied and the rows packed once, for many proofs."""
        rows = self.stark.air.firstRows(seed)
        return PackedSeed(self._pack_rows(rows), len(rows))

    def last_stats(self):
        """What the last prove_bytes() on this thread did, from the driver's own clock and counters (gs_prover_last_stats):
        {'total_ms', 'phases_ms': {label: ms}, 'ntt_points', 'ntt_transforms', 'horner_points'}."""
        st = _Stats()
        rc = self.lib.gs_prover_last_stats(C.byref(st))
        if rc:
            raise GstarkError(f'gs_prover_last_stats failed ({rc})')
        phases = {}
        for i in range(st.nphases):
            phases[bytes(st.phase_label[i]).split(b'\0', 1)[0].decode()] = round(st.phase_ms[i], 4)
        out = {'total_ms': round(st.total_ms, 4), 'phases_ms': phases, 'ntt_points': int(st.ntt_points),
               'ntt_transforms': int(st.ntt_transforms), 'horner_points': int(st.horner_points)}
        if st.nreadme:       # a proof that ran under sync_phases(True): the reference's own phase log (lib/Stark.ts:92-152; README.md:62-73)
            out['phases_readme'] = [(bytes(st.readme_label[i]).split(b'\0', 1)[0].decode(), round(st.readme_ms[i], 4)) for i in range(st.nreadme)]
        return out

    def member_sequence(self, on=True):
        """Checking mode of the calling thread's next proofs: the composition tail as the member-by-member sequence of entries instead of
        gs_composition_tail (gs_prover_member_sequence); same bytes."""
        self.lib.gs_prover_member_sequence(1 if on else 0)

    def sync_phases(self, on=True):
        """Measuring mode for the proofs this THREAD issues next: the device is synchronised at each of the reference's log points, so
        last_stats()['phases_readme'] lists every phase of README.md:62-73 with its own wall-clock.  Same proof bytes."""
        self.lib.gs_prover_sync_phases(1 if on else 0)
        return self

    def last_collectives(self):
        """The collectives the last distributed prove_bytes() on this thread issued: [{'label', 'kind', 'bytes', 'ms'}] (ms: device
        time when the communicator measures it, else None)."""
        arr = (_Collective * 256)()
        n = C.c_uint32()
        rc = self.lib.gs_prover_last_collectives(arr, 256, C.byref(n))
        if rc:
            raise GstarkError(f'gs_prover_last_collectives failed ({rc})')
        return [{'label': arr[i].label.decode(), 'kind': 'all_to_all' if arr[i].kind else 'all_gather', 'bytes': int(arr[i].bytes),
                 'ms': (round(arr[i].ms, 4) if arr[i].ms >= 0 else None)} for i in range(min(n.value, 256))]

    def prove(self, assertions, inputs=None, seed=None):
        return self.stark.parse(self.prove_bytes(assertions, inputs, seed))
