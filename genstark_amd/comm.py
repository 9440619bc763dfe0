"""Communicators for ONE proof across several GPUs (include/gstark_comm.h; the native driver's gs_prover_prove_dist).

`RcclComm` is the product: a communicator owned by csrc/comm_rccl.cc (libgstark_rccl.so) — ncclAllGather and grouped
ncclSend/ncclRecv on the library's own device buffers and stream, one rank per process and GPU, RCCL over xGMI.  The unique id is
created on rank 0 (`RcclComm.unique_id()`) and reaches the other ranks however the launcher likes (bench.py broadcasts it through
torch.distributed).  `TorchComm` runs the same two collectives through torch.distributed from callbacks (gloo in the CPU tests, or
ranks that have to share one GPU): it exists so that the multi-process path is testable without xGMI, not for production use.
"""
import ctypes as C
import os

from ._abi import GstarkError
from .native import GsComm

_HERE = os.path.dirname(os.path.abspath(__file__))
RCCL_LIB_PATH = os.path.join(_HERE, 'csrc', 'libgstark_rccl.so')

_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_TT = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.POINTER(C.c_double), C.c_uint32)


class RcclComm:
    """gs_comm over RCCL (csrc/comm_rccl.cc).  One per rank; `backend` is the rank's Backend (its stream carries the collectives)."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            if not os.path.exists(RCCL_LIB_PATH):
                raise GstarkError(f'{RCCL_LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"`')
            lib = C.CDLL(RCCL_LIB_PATH)
            lib.gs_rccl_unique_id.argtypes = [C.c_void_p]
            lib.gs_rccl_unique_id.restype = C.c_int
            lib.gs_rccl_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(GsComm), C.c_char_p, C.c_uint64]
            lib.gs_rccl_comm_create.restype = C.c_int
            lib.gs_rccl_comm_destroy.argtypes = [C.POINTER(GsComm)]
            lib.gs_rccl_comm_destroy.restype = None
            lib.gs_rccl_comm_timings.argtypes = [C.POINTER(GsComm), C.c_int]
            lib.gs_rccl_comm_timings.restype = None
            cls._lib = lib
        return cls._lib

    @classmethod
    def unique_id(cls):
        buf = C.create_string_buffer(128)
        rc = cls.lib().gs_rccl_unique_id(buf)
        if rc:
            raise GstarkError(f'gs_rccl_unique_id failed ({rc})')
        return buf.raw

    def __init__(self, backend, rank, size, unique_id):
        self.comm = GsComm()
        err = C.create_string_buffer(256)
        rc = self.lib().gs_rccl_comm_create(C.c_void_p(backend.lib._handle), C.c_char_p(unique_id), rank, size, backend.device, C.byref(self.comm), err, 256)
        if rc:
            raise GstarkError(f'gs_rccl_comm_create(rank {rank} of {size}) failed ({rc}): {err.value.decode(errors="replace")}')
        self.rank, self.size = rank, size

    def timings(self, on):
        """Per-collective device times (an event pair around each: `last_collectives()[i]['ms']`) on or off; off keeps the stream free
        of event records — what a latency-bound proof wants."""
        self.lib().gs_rccl_comm_timings(C.byref(self.comm), 1 if on else 0)
        return self

    def close(self):
        if self.comm is not None:
            self.lib().gs_rccl_comm_destroy(C.byref(self.comm))
            self.comm = None


class TorchComm:
    """gs_comm whose collectives are torch.distributed calls made from callbacks (any backend; gloo in the CPU tests).  Buffers of
    the oracle's ABI are host memory and are wrapped in place; device buffers of the HIP library are staged through the host with
    gs_download / gs_upload (ranks sharing one GPU cannot form an RCCL communicator)."""

    def __init__(self, backend, group=None):
        import numpy as np
        import torch
        import torch.distributed as dist
        self.backend, self.group = backend, group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        host = backend.name != 'hip-gfx950'

        def tensor_in(ptr, nbytes):
            if host:
                return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)))
            return torch.frombuffer(bytearray(backend.download(ptr, nbytes)), dtype=torch.uint8)

        def deliver(ptr, t):
            if not host:
                backend.upload(ptr, t.numpy().tobytes())

        def all_gather(_self, _ctx, send, recv, nbytes):
            try:
                if not host:
                    backend.sync()
                src = tensor_in(send, nbytes)
                dst = tensor_in(recv, nbytes * self.size) if host else torch.empty(nbytes * self.size, dtype=torch.uint8)
                dist.all_gather_into_tensor(dst, src.clone() if host else src, group=group)
                deliver(recv, dst)
                return 0
            except Exception as e:      # noqa: BLE001  (a callback must not raise through C)
                self.error = e
                return -3

        def all_to_all(_self, _ctx, send, recv, nbytes):
            try:
                if not host:
                    backend.sync()
                src = tensor_in(send, nbytes * self.size)
                dst = tensor_in(recv, nbytes * self.size) if host else torch.empty(nbytes * self.size, dtype=torch.uint8)
                ins = [src[h * nbytes:(h + 1) * nbytes].clone() for h in range(self.size)]
                outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.size)]
                reqs = []
                for h in range(self.size):          # point-to-point pairs (gloo has no all_to_all)
                    if h == self.rank:
                        outs[h].copy_(ins[h])
                    else:
                        reqs.append(dist.isend(ins[h], h, group=group))
                        reqs.append(dist.irecv(outs[h], h, group=group))
                for r in reqs:
                    r.wait()
                for h in range(self.size):
                    dst[h * nbytes:(h + 1) * nbytes] = outs[h]
                deliver(recv, dst)
                return 0
            except Exception as e:      # noqa: BLE001
                self.error = e
                return -3

        self.error = None
        self._cb = (_AG(all_gather), _AG(all_to_all))
        self.comm = GsComm()
        self.comm.self = None
        self.comm.rank, self.comm.size = self.rank, self.size
        self.comm.all_gather = C.cast(self._cb[0], C.c_void_p)
        self.comm.all_to_all = C.cast(self._cb[1], C.c_void_p)
        self.comm.take_timings = None
        self.comm.name = b'torch.distributed'
