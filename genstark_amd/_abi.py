"""ctypes binding of include/gstark.h.

The product path loads exactly one library: genstark_amd/csrc/libgstark_hip.so (hand-written HIP for
gfx950).  There is no CPU fallback: if the library is missing, is not the HIP build, or no MI355X is
visible, construction raises.  (tests/ may pass an explicit `lib_path` with `allow_test_double=True`
to run the host logic against the CPU oracle's implementation of the same ABI; nothing in this
package ever does.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, 'csrc', 'libgstark_hip.so')
# one build flavour of the library per field (csrc/build.sh): the 128-bit field of the hot path and the small prime fields of the
# reference's examples (csrc/gf_small.h: same kernels, same 16-byte element layout, plain arithmetic), and the two multi-limb
# primes of its examples (csrc/gf_wide.h: same kernels, 32-byte elements)
MODULUS_128 = 2**128 - 9 * 2**32 + 1
MODULUS_64 = 2**64 - 21 * 2**30 + 1        # examples/rescue/hash2x64.ts:10
MODULUS_32 = 2**32 - 3 * 2**25 + 1         # examples/demo/fibonacci.ts:14, README.md:23 (Foo)
MODULUS_17 = 96769                          # examples/demo/staticVariables.ts:11
MODULUS_256 = 2**256 - 351 * 2**32 + 1      # examples/mimc/mimc256.ts:13
MODULUS_224 = 2**224 - 2**96 + 1            # assembly/lib224.aa:3
HIP_LIB_PATHS = {MODULUS_128: HIP_LIB_PATH, MODULUS_64: os.path.join(_HERE, 'csrc', 'libgstark_hip_q64.so'),
                 MODULUS_32: os.path.join(_HERE, 'csrc', 'libgstark_hip_q32.so'),
                 MODULUS_17: os.path.join(_HERE, 'csrc', 'libgstark_hip_q17.so'),
                 MODULUS_256: os.path.join(_HERE, 'csrc', 'libgstark_hip_p256.so'),
                 MODULUS_224: os.path.join(_HERE, 'csrc', 'libgstark_hip_p224.so')}
# any other (odd) modulus below 2^256: the runtime-modulus flavour (gs_set_modulus; one modulus per process; generic kernels)
HIP_LIB_PATH_RUNTIME = os.path.join(_HERE, 'csrc', 'libgstark_hip_rt.so')

GS_OK = 0


class FriLayer(C.Structure):
    """struct gs_fri_layer of include/gstark.h (gs_fri_layers): one layer's outputs."""
    _fields_ = [('next', C.c_void_p), ('leaves', C.c_void_p), ('nodes', C.c_void_p), ('point_out', C.c_void_p), ('ticket', C.c_uint64)]

HASH_ALGS = {'sha256': 0, 'blake2s256': 1}  # gs_hash_alg; lib/Stark.ts:19

_vp, _u64, _u32, _int = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
_pvp = C.POINTER(C.c_void_p)
_bytes = C.c_char_p  # host byte strings (const uint8_t *)

_SIGNATURES = {
    'gs_abi_version': (_int, []),
    'gs_backend_name': (C.c_char_p, []),
    'gs_ctx_create': (_int, [_int, _vp, _pvp]),
    'gs_ctx_destroy': (None, [_vp]),
    'gs_last_error': (C.c_char_p, [_vp]),
    'gs_sync': (_int, [_vp]),
    'gs_stream': (_vp, [_vp]),
    'gs_element_size': (_int, []),
    'gs_field_modulus': (_int, [_vp]),
    'gs_set_modulus': (_int, [_bytes, _u32]),
    'gs_alloc': (_int, [_vp, _u64, _pvp]),
    'gs_free': (_int, [_vp, _vp]),
    'gs_cache_trim': (_int, [_vp]),
    'gs_upload': (_int, [_vp, _vp, _bytes, _u64]),
    'gs_download': (_int, [_vp, _vp, _vp, _u64]),
    'gs_copy': (_int, [_vp, _vp, _vp, _u64]),
    'gs_air_jit': (_int, [_vp, _int]),
    'gs_air_jit_launches': (_u64, [_vp]),
    'gs_traffic_enable': (_int, [_vp, _int]),
    'gs_traffic_read': (_int, [_vp, _vp, _u32, _vp]),
    'gs_air_jit_check': (_int, [_int, _vp, _u32, _vp, _u32, _bytes, _u32, _u32, _u32, _vp, _u32, _vp, _u64]),
    'gs_gather_words': (_int, [_vp, _vp, _u64, _vp]),
    'gs_transpose_records': (_int, [_vp, _vp, _u64, _u64, _u64, _vp]),
    'gs_defer_begin': (_int, [_vp]),
    'gs_defer_end': (_int, [_vp]),
    'gs_readback_post': (_int, [_vp, _vp, _u32, C.POINTER(C.c_uint64)]),
    'gs_readback_wait': (_int, [_vp, _u64, _vp]),
    'gs_gather': (_int, [_vp, _vp, _u64, C.POINTER(_u64), _u64, _vp]),
    'gs_power_series': (_int, [_vp, _bytes, _u64, _vp]),
    'gs_vec_add': (_int, [_vp, _vp, _vp, _u64, _vp]),
    'gs_vec_sub': (_int, [_vp, _vp, _vp, _u64, _vp]),
    'gs_vec_mul': (_int, [_vp, _vp, _vp, _u64, _vp]),
    'gs_vec_add_scalar': (_int, [_vp, _vp, _bytes, _u64, _vp]),
    'gs_vec_sub_scalar': (_int, [_vp, _vp, _bytes, _u64, _vp]),
    'gs_vec_mul_scalar': (_int, [_vp, _vp, _bytes, _u64, _vp]),
    'gs_vec_inv': (_int, [_vp, _vp, _u64, _vp]),
    'gs_vec_div': (_int, [_vp, _vp, _vp, _u64, _vp]),
    'gs_vec_exp': (_int, [_vp, _vp, _bytes, _u64, _vp]),
    'gs_combine_many': (_int, [_vp, _pvp, _bytes, _u32, _u64, _vp]),
    'gs_combine_adjusted': (_int, [_vp, _pvp, _bytes, _bytes, _u32, _vp, _vp, _u64, _vp]),
    'gs_combine': (_int, [_vp, _vp, _vp, _u64, _vp]),
    'gs_pluck': (_int, [_vp, _vp, _u64, _u64, _u64, _vp]),
    'gs_zero_poly_inverses': (_int, [_vp, _bytes, _u64, _u64, _bytes, _vp]),
    'gs_div_by_domain_roots': (_int, [_vp, _vp, _u32, _u64, _bytes, C.POINTER(_u64), C.POINTER(_u32), _u32, _vp]),
    'gs_zero_poly_inverses_coset': (_int, [_vp, _bytes, _u64, _bytes, _u64, _bytes, _vp]),
    'gs_div_by_domain_roots_coset': (_int, [_vp, _vp, _u32, _u64, _bytes, _bytes, C.POINTER(_u64), C.POINTER(_u32), _u32, _vp]),
    'gs_transpose_vector': (_int, [_vp, _vp, _u64, _u32, _u64, _vp]),
    'gs_transpose_matrix': (_int, [_vp, _vp, _u64, _u64, _vp]),
    'gs_sub_matrix_from_vectors': (_int, [_vp, _pvp, _vp, _u32, _u64, _vp]),
    'gs_eval_polys_at_roots': (_int, [_vp, _vp, _u32, _u64, _bytes, _u64, _vp]),
    'gs_interpolate_roots': (_int, [_vp, _vp, _u32, _bytes, _u64, _vp]),
    'gs_eval_poly_at': (_int, [_vp, _vp, _u64, _bytes, _vp]),
    'gs_interpolate_quartic_batch': (_int, [_vp, _vp, _vp, _u64, _vp]),
    'gs_interpolate_quartic_domain': (_int, [_vp, _bytes, _u64, _u64, _vp, _u64, _vp]),
    'gs_fri_fold': (_int, [_vp, _bytes, _u64, _u64, _vp, _u64, _bytes, _vp]),
    'gs_fri_fold_seeded': (_int, [_vp, _bytes, _u64, _u64, _vp, _u64, _vp, _vp]),
    'gs_fri_fold_seeded_scaled': (_int, [_vp, _bytes, _u64, _u64, _vp, _u64, _vp, _bytes, _vp]),
    'gs_merkle_commit_rows_seed': (_int, [_vp, _int, _pvp, _u32, _u64, _vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    'gs_fri_fold_at': (_int, [_vp, _bytes, _u64, _u64, _vp, _u64, _vp, _vp]),
    'gs_fri_layers': (_int, [_vp, _int, _bytes, _u64, _u64, _vp, _u64, _vp, _u32, _vp]),
    'gs_eval_quartic_batch': (_int, [_vp, _vp, _u64, _bytes, _vp]),
    'gs_hash_digest': (_int, [_vp, _int, _bytes, _u64, _vp]),
    'gs_hash_merge_rows': (_int, [_vp, _int, _pvp, _u32, _u64, _vp]),
    'gs_hash_digest_values': (_int, [_vp, _int, _vp, _u64, _u64, _vp]),
    'gs_merkle_build': (_int, [_vp, _int, _vp, _u64, _vp]),
    'gs_merkle_commit_rows': (_int, [_vp, _int, _pvp, _u32, _u64, _vp, _vp]),
    'gs_merkle_prove_batch': (_int, [_vp, _vp, _vp, _u64, C.POINTER(_u64), _u32, _vp, C.POINTER(_u32), C.POINTER(_u32), _vp, _u64]),
    'gs_small_interpolate': (_int, [_bytes, _bytes, _u32, _vp]),
    'gs_small_eval_poly': (_int, [_bytes, _u32, _bytes, _u32, _vp]),
    'gs_mimc_trace': (_int, [_vp, _bytes, _bytes, _u32, _u64, _vp]),
    'gs_mimc_constraints': (_int, [_vp, _vp, _u64, _u64, _vp, _u64, _vp]),
    'gs_mimc_composition': (_int, [_vp, _vp, _u64, _u64, _bytes, _vp, _u64, _bytes, _u64, _u64, _bytes, C.POINTER(_u64), _u32, _bytes, _vp]),
    'gs_pseudorandom_indexes': (_int, [_bytes, _u32, _u32, _u64, _u32, C.POINTER(_u64)]),
    'gs_air_trace_segments': (_int, [_vp, C.POINTER(_u32), _u32, C.POINTER(_u32), _u32, _bytes, _u32, _u32, _u32, _bytes, C.POINTER(_u32), _u32, _bytes, _u64, _u64, _vp]),
    'gs_air_trace': (_int, [_vp, C.POINTER(_u32), _u32, _bytes, _u32, _u32, _u32, _bytes, C.POINTER(_u32), _u32, _bytes, _u64, _vp]),
    'gs_air_constraints': (_int, [_vp, C.POINTER(_u32), _u32, _bytes, _u32, _u32, _u32, _u32, _vp, _u64, _u64, _vp,
                                  C.POINTER(_u64), _u32, _vp]),
    'gs_composition_tail': (_int, [_vp, _u64, _bytes, _vp, _vp, _u64, _bytes, _vp, _u32, _bytes, _u32, C.POINTER(_u64), C.POINTER(_u32), _u32, _bytes, _bytes,
                                   _vp, _u32, _bytes, _bytes, _vp, _u64, _vp, _vp]),
    'gs_composition_tail_coset': (_int, [_vp, _u64, _bytes, _bytes, _vp, _vp, _u64, _bytes, _vp, _u32, _bytes, _u32, C.POINTER(_u64), C.POINTER(_u32), _u32, _bytes,
                                         _bytes, _vp, _u32, _bytes, _bytes, _vp, _u64, _vp, _vp]),
    'gs_air_constraints_strided': (_int, [_vp, C.POINTER(_u32), _u32, _bytes, _u32, _u32, _u32, _u32, _vp, _u64, _u64, _u64, _u64, _vp,
                                          C.POINTER(_u64), _u32, _vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class GstarkError(RuntimeError):
    pass


def load_library(path):
    if not os.path.exists(path):
        raise GstarkError(
            f'{path} not found: the HIP backend is not built (run `python -c "import __graft_entry__ as g; g.build()"`).'
            ' There is no CPU fallback.')
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype, fn.argtypes = res, args
    return lib


class Backend:
    """One gs_ctx (one device, one HIP stream).  All device work of the package goes through it."""

    def __init__(self, device=0, stream=None, lib_path=None, allow_test_double=False, modulus=None):
        """modulus: picks the library flavour built for that field (default: the 128-bit field); lib_path overrides."""
        if lib_path is None:
            # createPrimeField(modulus), index.ts:14: a build of the library per modulus that occurs in the reference tree, and the
            # runtime-modulus build for every other one
            lib_path = HIP_LIB_PATH if modulus is None else HIP_LIB_PATHS.get(modulus, HIP_LIB_PATH_RUNTIME)
        self.lib = load_library(lib_path)
        self.element_size = self.lib.gs_element_size()          # 16, or 32 in the 256- / 224-bit and the runtime-modulus flavours
        if modulus is not None:
            # a fixed build answers 0 for its own modulus; the runtime-modulus build takes the value (once per process)
            if modulus < 3 or modulus >> (8 * self.element_size):
                raise GstarkError(f'a modulus of {modulus.bit_length()} bits does not fit {lib_path}')
            rc = self.lib.gs_set_modulus(modulus.to_bytes(self.element_size, 'little'), self.element_size)
            if rc != GS_OK:
                raise GstarkError(f'{lib_path} does not compute modulo {modulus} (gs_set_modulus: {rc}'
                                  + (': the runtime-modulus library serves ONE modulus per process, and it has been given another' if rc == -3 else '') + ')')
        buf = C.create_string_buffer(self.element_size)
        self.lib.gs_field_modulus(C.cast(buf, C.c_void_p))
        self.modulus = int.from_bytes(buf.raw, 'little')
        if modulus is not None and self.modulus != modulus:
            raise GstarkError(f'{lib_path} is built for the field of {self.modulus} elements, not {modulus}')
        if not self.modulus:
            raise GstarkError(f'{lib_path} takes its modulus at run time: pass modulus=')
        self.name = self.lib.gs_backend_name().decode()
        if self.name != 'hip-gfx950' and not allow_test_double:
            raise GstarkError(f'refusing backend {self.name!r}: the product path runs on hip-gfx950 only')
        if self.lib.gs_abi_version() != 1:
            raise GstarkError('gstark ABI version mismatch')
        ctx = C.c_void_p()
        rc = self.lib.gs_ctx_create(int(device), C.c_void_p(stream), C.byref(ctx))
        if rc != GS_OK or not ctx.value:
            raise GstarkError(f'gs_ctx_create(device={device}) failed with {rc}: no gfx950 device? (no CPU fallback)')
        self.ctx = ctx
        self.device = device
        self.stats = {}      # see call()

    def jit(self, enable=True):
        """AIR programs compiled on first use (True), always interpreted (False) or compiled when already built ('auto', the default of a new context): gs_air_jit."""
        self.call('gs_air_jit', 2 if enable == 'auto' else (1 if enable else 0))
        return self

    @property
    def jit_launches(self):
        return self.lib.gs_air_jit_launches(self.ctx)

    def traffic(self, on=None):
        """gs_traffic_enable / gs_traffic_read (a measurement aid): traffic(True) starts a fresh tally, traffic(False) stops it, traffic()
        returns {kernel name: {'launches', 'bytes', 'units'}} — what every kernel launched since HAD to move (algorithmic bytes)."""
        if on is not None:
            self.call('gs_traffic_enable', 1 if on else 0)
            return self

        class _E(C.Structure):
            _fields_ = [('kernel', C.c_char * 64), ('launches', C.c_uint64), ('bytes', C.c_uint64), ('units', C.c_uint64)]
        n = C.c_uint32()
        arr = (_E * 256)()
        self.call('gs_traffic_read', arr, 256, C.byref(n))
        return {arr[i].kernel.decode(): {'launches': int(arr[i].launches), 'bytes': int(arr[i].bytes), 'units': int(arr[i].units)} for i in range(min(n.value, 256))}

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.gs_ctx_destroy(self.ctx)
            self.ctx = None

    def call(self, name, *args):
        # transform work launched through this backend, counted the way the native driver counts its own (gs_prover_last_stats):
        # rows * n points per call, Horner-served calls (fewer than 256 points, or at most 8 coefficients) apart
        if name == 'gs_eval_polys_at_roots':
            rows, plen, n = int(args[1]), int(args[2]), int(args[4])
            key = 'ntt_points' if n >= 256 and plen > 8 else 'horner_points'
            self.stats[key] = self.stats.get(key, 0) + rows * n
        elif name == 'gs_interpolate_roots':
            rows, n = int(args[1]), int(args[3])
            key = 'ntt_points' if n >= 256 else 'horner_points'
            self.stats[key] = self.stats.get(key, 0) + rows * n
        rc = getattr(self.lib, name)(self.ctx, *args)
        if rc != GS_OK:
            raise GstarkError(f'{name} failed ({rc}): {self.lib.gs_last_error(self.ctx).decode()}')

    # ---- memory
    def alloc(self, nbytes):
        p = C.c_void_p()
        self.call('gs_alloc', int(nbytes), C.byref(p))
        return p.value

    def free(self, ptr):
        if self.ctx:
            self.lib.gs_free(self.ctx, C.c_void_p(ptr))

    def trim(self):
        """Give the cached device blocks back to the driver (long-running services with changing problem sizes)."""
        self.call('gs_cache_trim')

    def upload(self, ptr, data):
        self.call('gs_upload', C.c_void_p(ptr), bytes(data), len(data))

    def download(self, ptr, nbytes, offset=0):
        buf = C.create_string_buffer(int(nbytes))
        self.call('gs_download', C.cast(buf, C.c_void_p), C.c_void_p(ptr + offset), int(nbytes))
        return buf.raw

    def gather(self, ptr, rec_bytes, indexes):
        n = len(indexes)
        if n == 0:
            return []
        idx = (C.c_uint64 * n)(*indexes)
        buf = C.create_string_buffer(n * rec_bytes)
        self.call('gs_gather', C.c_void_p(ptr), rec_bytes, idx, n, C.cast(buf, C.c_void_p))
        raw = buf.raw
        return [raw[i * rec_bytes:(i + 1) * rec_bytes] for i in range(n)]

    def sync(self):
        self.call('gs_sync')

    @property
    def stream(self):
        return self.lib.gs_stream(self.ctx)

    @staticmethod
    def ptr_array(ptrs):
        return (C.c_void_p * len(ptrs))(*ptrs)
