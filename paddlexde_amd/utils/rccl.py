"""The per-attempt all-reduce of the error norm's sums issued DIRECTLY on RCCL (librccl.so, the library behind the ``nccl`` backend
of torch.distributed on ROCm), on the stream the solver's kernels run on.

Why not ``torch.distributed.all_reduce`` for this one call: the process-group layer runs collectives on its own internal stream
and fences the caller's stream with two events around each of them.  For a 256-byte message that choreography is most of the cost —
on one MI355X the step of config 4's per-GPU shard (65536 x 64) grows from 169 to 191 us when the sharded code path is taken with a
group of one (finalize 4 us + all-reduce + controller on finalised sums), and the host pays ~20 us per call.  ``ncclAllReduce`` on
the caller's own stream is one more kernel in that stream: no event, no stream hop, nothing for the host to wait for.

``RcclExchange(group)`` builds its own communicator over the ranks of ``group`` (the 128-byte unique id travels ONCE by
``broadcast_object_list``; one rank per GPU, as RCCL requires) and then stands in for the group's all-reduce in the batch-sharded
solve: ``options={"process_group": pg, "norm_exchange": RcclExchange(pg)}``.  ``torch.distributed`` stays the transport of
everything that happens once per solve.  The reduction (rank-ordered ring / tree inside RCCL) returns bit-identical results on every
rank, which is all the lock-step controllers need.
"""
import ctypes as C
import glob
import os

import torch

from .. import _hip

_NCCL_UNIQUE_ID_BYTES = 128  # rccl.h: NCCL_UNIQUE_ID_BYTES
_NCCL_FLOAT64 = 8  # rccl.h: ncclFloat64
_NCCL_SUM, _NCCL_MAX = 0, 2  # rccl.h: ncclSum, ncclMax


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * _NCCL_UNIQUE_ID_BYTES)]  # (c_ubyte, not c_char: the id is binary, a c_char array reads as a C string)


_lib = None


def _load_rccl():
    """The librccl.so torch itself uses (already mapped into the process by the nccl backend), else ROCm's."""
    global _lib
    if _lib is not None:
        return _lib
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so", "librccl.so"]
    last = None
    for path in cands:
        try:
            lib = C.CDLL(path)
        except OSError as e:
            last = e
            continue
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGetUniqueId.restype = C.c_int
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.restype = C.c_int
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllReduce.restype = C.c_int
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.restype = C.c_int
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclCommGetAsyncError.restype = C.c_int
        lib.ncclCommGetAsyncError.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _lib = lib
        return lib
    raise _hip.XdeError("librccl.so not found ({})".format(last))


class RcclExchange:
    """In-stream RCCL all-reduce of the solver's 2*XDE_MAX_SEG norm sums; same protocol as ``PeerExchange``
    (``exchange(sums, norm_kind)``, ``error()``, ``close()``)."""

    # Whether pipeline="graph" may record the exchange into a hipGraph.  Off: a captured RCCL collective has not been exercised
    # across GPUs here.  XDE_RCCL_CAPTURE=1 (or setting the attribute on an instance) allows it — measured with one rank only.
    capturable = os.environ.get("XDE_RCCL_CAPTURE", "0") == "1"

    def __init__(self, group=None, device=None):
        import torch.distributed as dist

        from .exchange import raise_together

        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = None
        self._comm = None
        # -- rank-local: the library, and on rank 0 the communicator id.  The ranks agree before the id travels: a rank that cannot
        #    load librccl (or rank 0 failing to make an id) is an exception on EVERY rank, with nobody left waiting in a broadcast.
        err, uid = None, _UniqueId()
        try:
            self.lib = _load_rccl()
            self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            if self.rank == 0:
                self._check(self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        except Exception as e:  # noqa: BLE001
            err = e
        raise_together(err, "RcclExchange: library / communicator id", group)
        # -- collective: the 128-byte id, once
        box = [C.string_at(C.addressof(uid), _NCCL_UNIQUE_ID_BYTES) if self.rank == 0 else None]
        src = 0 if group is None else dist.get_global_rank(group, 0)
        dist.broadcast_object_list(box, src=src, group=group)
        err = None
        if not isinstance(box[0], bytes) or len(box[0]) != _NCCL_UNIQUE_ID_BYTES:
            err = _hip.XdeError("the communicator id did not arrive intact")
        raise_together(err, "RcclExchange: communicator id broadcast", group)
        C.memmove(C.addressof(uid), box[0], _NCCL_UNIQUE_ID_BYTES)
        # -- collective inside RCCL: every rank enters ncclCommInitRank together.  A rank that DIES or hangs in here cannot be
        #    rescued by its peers (they wait inside the library): that case is fatal by design and ends in the caller's watchdog.  An
        #    error RETURNED by the call is agreed on like every other step.
        comm = C.c_void_p()
        err = None
        try:
            with torch.cuda.device(self.device):
                self._check(self.lib.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
            self._comm = comm
        except Exception as e:  # noqa: BLE001
            err = e
        try:
            raise_together(err, "RcclExchange: ncclCommInitRank", group)
            # first use outside any timed / captured region: RCCL sets its channels up lazily
            err = None
            try:
                warm = torch.zeros(2 * _hip.XDE_MAX_SEG, dtype=torch.float64, device=self.device)
                self.exchange(warm, _hip.NORM_RMS)
                torch.cuda.synchronize(self.device)
            except Exception as e:  # noqa: BLE001
                err = e
            raise_together(err, "RcclExchange: first all-reduce", group)
        except Exception:
            self.close()
            raise

    def _check(self, rc, who):
        if rc != 0:
            raise _hip.XdeError("{} failed: {}".format(who, self.lib.ncclGetErrorString(rc).decode()))

    def exchange(self, sums, norm_kind):
        """In place: ``sums`` becomes the sum over all ranks (for a linf norm: the max of the first XDE_MAX_SEG values, the sum of
        the non-finite counts behind them).  Enqueued on torch's current stream, nothing else."""
        if not sums.is_cuda or sums.dtype != torch.float64 or sums.numel() != 2 * _hip.XDE_MAX_SEG:
            raise _hip.XdeError("RcclExchange.exchange takes the solver's 2*XDE_MAX_SEG float64 device sums")
        st = _hip.HipBackend._stream(sums)
        p, m = sums.data_ptr(), _hip.XDE_MAX_SEG
        if norm_kind == _hip.NORM_RMS:
            self._check(self.lib.ncclAllReduce(p, p, 2 * m, _NCCL_FLOAT64, _NCCL_SUM, self._comm, st), "ncclAllReduce")
        else:
            self._check(self.lib.ncclAllReduce(p, p, m, _NCCL_FLOAT64, _NCCL_MAX, self._comm, st), "ncclAllReduce(max)")
            self._check(self.lib.ncclAllReduce(p + 8 * m, p + 8 * m, m, _NCCL_FLOAT64, _NCCL_SUM, self._comm, st), "ncclAllReduce(sum)")

    def error(self):
        """Exchange number of a failed exchange: RCCL reports failures as errors of the call or of the communicator, so 0."""
        return 0

    def async_error(self):
        e = C.c_int(0)
        self._check(self.lib.ncclCommGetAsyncError(self._comm, C.byref(e)), "ncclCommGetAsyncError")
        return None if e.value == 0 else self.lib.ncclGetErrorString(e.value).decode()

    def close(self):
        if self._comm is None:
            return
        comm, self._comm = self._comm, None
        try:
            torch.cuda.synchronize(self.device)
        finally:
            self.lib.ncclCommDestroy(comm)

    abandon = close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
