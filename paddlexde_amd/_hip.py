"""ctypes binding of libxde_hip.so (C ABI: include/xde_hip.h).

The library is the product's only compute path.  There is NO CPU fallback: if the shared
object is missing, or a tensor is not on a ROCm device, the calls below raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import torch

XDE_OK, XDE_EBADARG, XDE_EHIP, XDE_ETIMEOUT = 0, 1, 2, 3
XDE_MIRROR_SLOTS = 16
ABI_VERSION = 6
XDE_F32, XDE_F64 = 0, 1
XDE_MAX_K, XDE_MAX_SEG, XDE_MAX_STAGE = 14, 16, 13
XDE_MAX_PACK = 64
XDE_P2P_MAX_RANKS, XDE_P2P_HANDLE_BYTES = 16, 64
COMBINE_RK, COMBINE_FUSE, COMBINE_WFUSE = 0, 1, 2
NORM_RMS, NORM_LINF = 0, 1
HISTORY_METHODS = {"cubic": 0, "linear": 1, "bez": 2}
STATUS_OK, STATUS_DT_UNDERFLOW, STATUS_NONFINITE, STATUS_MAX_STEPS = 0, 1, 2, 3
KID_NAMES = ("combine", "errnorm", "control", "dense", "scalednorm", "finalize", "commit", "combine_fuse", "combine_wfuse")

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libxde_hip.so")

# every symbol include/xde_hip.h declares (tests/test_cabi.py checks the export list against the header)
SYMBOLS = (
    "xde_last_error",
    "xde_abi_version",
    "xde_sizeof_ctrl",
    "xde_sizeof_ctrl_params",
    "xde_sizeof_segments",
    "xde_workspace_bytes",
    "xde_stage_combine",
    "xde_stage_combine_pre",
    "xde_stage_combine_pre_weighted",
    "xde_error_norm_partial",
    "xde_error_norm_control",
    "xde_error_ratio",
    "xde_scaled_norm_partial",
    "xde_norm_finalize",
    "xde_norm_result",
    "xde_rk_control",
    "xde_ctrl_init",
    "xde_ctrl_retarget",
    "xde_initial_step",
    "xde_initial_step_fused",
    "xde_scaled_norm2_partial",
    "xde_initial_step_tail",
    "xde_ctrl_read",
    "xde_host_alloc",
    "xde_host_free",
    "xde_ctrl_wait",
    "xde_dense_eval",
    "xde_commit",
    "xde_pack_segments",
    "xde_dense_commit",
    "xde_hermite_gather",
    "xde_history_gather",
    "xde_lag_grad_workspace_bytes",
    "xde_lag_grad",
    "xde_scale_fanout",
    "xde_graph_replace_memsets",
    "xde_p2p_mailbox_bytes",
    "xde_p2p_alloc",
    "xde_p2p_free",
    "xde_p2p_export",
    "xde_p2p_import",
    "xde_p2p_close",
    "xde_p2p_exchange",
    "xde_p2p_error",
    "xde_p2p_error_info",
    "xde_p2p_rk_control",
    "xde_prof_enable",
    "xde_prof_collect",
)


class XdeCtrl(C.Structure):
    """xde_ctrl_t"""

    _fields_ = [
        ("t0", C.c_double),
        ("t1", C.c_double),
        ("dt", C.c_double),
        ("dt_last", C.c_double),
        ("t_plan", C.c_double),
        ("ratio_prev", C.c_double),
        ("ratio", C.c_double),
        ("ratio_seg", C.c_double * XDE_MAX_SEG),
        ("nonfinite", C.c_double),
        ("n_steps", C.c_int64),
        ("n_accept", C.c_int64),
        ("n_reject", C.c_int64),
        ("steps_in_interval", C.c_int64),
        ("accept", C.c_int32),
        ("sel_used", C.c_int32),
        ("status", C.c_int32),
        ("out_begin", C.c_int32),
        ("out_end", C.c_int32),
        ("next_out", C.c_int32),
        ("n_out", C.c_int32),
        ("done", C.c_int32),
        ("next_step_index", C.c_int32),
        ("on_step_t", C.c_int32),
        ("seq", C.c_int64),
        ("chk", C.c_uint64),
        ("reserved", C.c_int32 * 2),
    ]


class XdeCtrlParams(C.Structure):
    """xde_ctrl_params_t"""

    _fields_ = [
        ("struct_size", C.c_uint32),  # the binding's statement of the layout: checked by the library before anything else is read
        ("abi_version", C.c_uint32),
        ("rtol", C.c_double),
        ("atol", C.c_double),
        ("min_step", C.c_double),
        ("max_step", C.c_double),
        ("safety", C.c_double),
        ("ifactor", C.c_double),
        ("dfactor", C.c_double),
        ("order", C.c_double),
        ("max_num_steps", C.c_int64),
        ("time_dtype", C.c_int32),
        ("state_dtype", C.c_int32),
        ("direction", C.c_int32),
        ("norm_kind", C.c_int32),
        ("n_stage", C.c_int32),
        ("n_seg", C.c_int32),
        ("n_step_t", C.c_int32),
        ("pi_controller", C.c_int32),
        ("pi_beta", C.c_double),
        ("alpha", C.c_double * XDE_MAX_STAGE),
        ("seg_count", C.c_double * XDE_MAX_SEG),
        ("replay", C.c_void_p),
        ("n_replay", C.c_int64),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(type(self))
        self.abi_version = ABI_VERSION


class XdeSegments(C.Structure):
    """xde_segments_t"""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_seg", C.c_int32),
        ("seg_start", C.c_int64 * XDE_MAX_SEG),
        ("seg_len", C.c_int64 * XDE_MAX_SEG),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(type(self))


class XdeError(RuntimeError):
    pass


class _Work:
    """The small per-solve device buffers of an adaptive solver."""

    __slots__ = ("key", "ctrl", "ws", "sums", "t_stage", "t_views")


# torch's C-level accessor of the current stream handle: ~0.3 us against ~4 us for torch.cuda.current_stream().cuda_stream
# (a Stream object is built each time); every launch needs the handle
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen libxde_hip.so and declare prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise XdeError(
                "paddlexde_amd: {} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python -m paddlexde_amd.csrc.build`. There is no CPU fallback.".format(LIB_PATH)
            )
        lib = C.CDLL(LIB_PATH)
        vp, dp, i32, i64, dbl = C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int64, C.c_double
        vpp = C.POINTER(C.c_void_p)
        lib.xde_last_error.restype = C.c_char_p
        lib.xde_last_error.argtypes = []
        lib.xde_abi_version.restype = i32
        lib.xde_sizeof_ctrl.restype = i64
        lib.xde_sizeof_ctrl_params.restype = i64
        lib.xde_sizeof_segments.restype = i64
        lib.xde_workspace_bytes.restype = i64
        lib.xde_stage_combine.restype = i32
        lib.xde_stage_combine.argtypes = [vp, vp, vp, vpp, vp, dp, i32, i32, dbl, dbl, vp, i64, i32, vp, dp, dbl, C.c_uint32, vp]
        lib.xde_stage_combine_pre.restype = i32
        lib.xde_stage_combine_pre.argtypes = [vp, vp, vp, vp, vpp, dp, i32, dbl, vp, i64, i32, C.c_uint32, vp]
        lib.xde_stage_combine_pre_weighted.restype = i32
        lib.xde_stage_combine_pre_weighted.argtypes = [vp, vp, vp, vpp, dp, i32, dbl, dbl, vp, i64, i32, dbl, vp]
        lib.xde_error_norm_partial.restype = i32
        lib.xde_error_norm_partial.argtypes = [vpp, vp, dp, i32, vp, vp, vp, dbl, dbl, dbl, vp, C.POINTER(XdeSegments), i32, i32, vp, vp, vp]
        lib.xde_error_norm_control.restype = i32
        lib.xde_error_norm_control.argtypes = [vpp, vp, dp, i32, vp, vp, vp, C.POINTER(XdeSegments), i32, vp, vp, vp,
                                               C.POINTER(XdeCtrlParams), vp, vp, vp, vp, vp]
        lib.xde_error_ratio.restype = i32
        lib.xde_error_ratio.argtypes = [vp, vpp, vp, dp, i32, vp, vp, vp, dbl, dbl, dbl, vp, i64, i32, vp, vp]
        lib.xde_scaled_norm_partial.restype = i32
        lib.xde_scaled_norm_partial.argtypes = [vp, vp, vp, dbl, dbl, C.POINTER(XdeSegments), i32, i32, vp, i32, vp]
        lib.xde_norm_finalize.restype = i32
        lib.xde_norm_finalize.argtypes = [vp, i32, vp, vp]
        lib.xde_norm_result.restype = i32
        lib.xde_norm_result.argtypes = [vp, dp, i32, i32, i32, vp, vp]
        lib.xde_rk_control.restype = i32
        lib.xde_rk_control.argtypes = [vp, C.POINTER(XdeCtrlParams), vp, vp, vp, vp, vp, vp, vp]
        lib.xde_ctrl_init.restype = i32
        lib.xde_ctrl_init.argtypes = [vp, C.POINTER(XdeCtrlParams), dbl, dbl, C.c_int32, vp, vp, vp, i64, vp, vp]
        lib.xde_ctrl_retarget.restype = i32
        lib.xde_ctrl_retarget.argtypes = [vp, C.POINTER(XdeCtrlParams), vp, C.c_int32, vp, vp]
        lib.xde_initial_step.restype = i32
        lib.xde_initial_step.argtypes = [i32, vp, vp, C.POINTER(XdeCtrlParams), dbl, vp, i32, vp, vp]
        lib.xde_initial_step_fused.restype = i32
        lib.xde_initial_step_fused.argtypes = [i32, vp, vp, vp, C.POINTER(XdeSegments), i32, vp, C.POINTER(XdeCtrlParams), dbl, vp, i32, vp,
                                               C.c_int32, vp, vp, vp, i64, vp]
        lib.xde_scaled_norm2_partial.restype = i32
        lib.xde_scaled_norm2_partial.argtypes = [vp, vp, dbl, dbl, C.POINTER(XdeSegments), i32, i32, vp, vp]
        lib.xde_initial_step_tail.restype = i32
        lib.xde_initial_step_tail.argtypes = [i32, vp, vp, C.POINTER(XdeCtrlParams), dbl, vp, i32, vp, C.c_int32, vp, vp, vp, i64, vp, vp]
        lib.xde_host_alloc.restype = i32
        lib.xde_host_alloc.argtypes = [i64, C.POINTER(C.c_void_p)]
        lib.xde_host_free.restype = i32
        lib.xde_host_free.argtypes = [vp]
        lib.xde_ctrl_wait.restype = i32
        lib.xde_ctrl_wait.argtypes = [vp, i64, dbl, C.POINTER(XdeCtrl)]
        lib.xde_ctrl_read.restype = i32
        lib.xde_ctrl_read.argtypes = [vp, C.POINTER(XdeCtrl), vp]
        lib.xde_dense_eval.restype = i32
        lib.xde_dense_eval.argtypes = [vp, vpp, vp, dp, i32, vp, vp, vp, vp, vp, vp, i32, i64, i32, i64, vp]
        lib.xde_scale_fanout.restype = i32
        lib.xde_scale_fanout.argtypes = [vpp, vp, dp, i32, vp, i64, i32, vp]
        lib.xde_hermite_gather.restype = i32
        lib.xde_hermite_gather.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
        lib.xde_history_gather.restype = i32
        lib.xde_history_gather.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]
        lib.xde_lag_grad_workspace_bytes.restype = i64
        lib.xde_lag_grad_workspace_bytes.argtypes = [i32]
        lib.xde_lag_grad.restype = i32
        lib.xde_lag_grad.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp]
        lib.xde_dense_commit.restype = i32
        lib.xde_dense_commit.argtypes = [vp, vpp, dp, i32, vp, vp, vp, vp, vp, vp, i32, i64, i32, vp]
        lib.xde_commit.restype = i32
        lib.xde_commit.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp]
        lib.xde_graph_replace_memsets.restype = i32
        lib.xde_graph_replace_memsets.argtypes = [vp, C.POINTER(C.c_int)]
        lib.xde_p2p_mailbox_bytes.restype = i64
        lib.xde_p2p_alloc.restype = i32
        lib.xde_p2p_alloc.argtypes = [C.POINTER(C.c_void_p)]
        lib.xde_p2p_free.restype = i32
        lib.xde_p2p_free.argtypes = [vp]
        lib.xde_p2p_export.restype = i32
        lib.xde_p2p_export.argtypes = [vp, vp]
        lib.xde_p2p_import.restype = i32
        lib.xde_p2p_import.argtypes = [vp, C.POINTER(C.c_void_p)]
        lib.xde_p2p_close.restype = i32
        lib.xde_p2p_close.argtypes = [vp]
        lib.xde_p2p_exchange.restype = i32
        lib.xde_p2p_exchange.argtypes = [vp, vp, vpp, i32, i32, i32, i64, vp]
        lib.xde_pack_segments.restype = i32
        lib.xde_pack_segments.argtypes = [vp, vpp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), dp, i32, i64, i32, vp]
        lib.xde_p2p_error.restype = i32
        lib.xde_p2p_error.argtypes = [vp, C.POINTER(C.c_int64), vp]
        lib.xde_p2p_error_info.restype = i32
        lib.xde_p2p_error_info.argtypes = [vp, C.POINTER(C.c_int64), i32, vp]
        lib.xde_p2p_rk_control.restype = i32
        lib.xde_p2p_rk_control.argtypes = [vp, C.POINTER(XdeCtrlParams), vp, vp, vpp, i32, i32, i64, vp, vp, vp, vp, vp]
        lib.xde_prof_enable.restype = i32
        lib.xde_prof_enable.argtypes = [i32]
        lib.xde_prof_collect.restype = i32
        lib.xde_prof_collect.argtypes = [C.POINTER(C.c_int64), dp, dp]
        if lib.xde_abi_version() != ABI_VERSION:
            raise XdeError("libxde_hip.so ABI version mismatch")
        if lib.xde_sizeof_ctrl() != C.sizeof(XdeCtrl):
            raise XdeError("xde_ctrl_t layout mismatch between header and ctypes mirror")
        if lib.xde_sizeof_ctrl_params() != C.sizeof(XdeCtrlParams):
            raise XdeError("xde_ctrl_params_t layout mismatch between header and ctypes mirror")
        if lib.xde_sizeof_segments() != C.sizeof(XdeSegments):
            raise XdeError("xde_segments_t layout mismatch between header and ctypes mirror")
        _lib = lib
    return _lib


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return XDE_F32
    if dt == torch.float64:
        return XDE_F64
    raise TypeError("paddlexde_amd kernels support float32 and float64 states, got {}".format(dt))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _dbl_array(xs):
    if isinstance(xs, C.Array):  # pre-marshalled by the caller (per-stage coefficient rows never change)
        return xs
    return (C.c_double * len(xs))(*[float(x) for x in xs])


dbl_array = _dbl_array


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def make_segments(segs) -> XdeSegments:
    s = XdeSegments()
    s.n_seg = len(segs)
    for i, (start, length) in enumerate(segs):
        s.seg_start[i] = int(start)
        s.seg_len[i] = int(length)
    return s


class HipBackend:
    """The production backend: every method enqueues one kernel of libxde_hip.so on torch's current stream."""

    name = "hip"

    WAIT_TIMEOUT_MS = 600e3

    def __init__(self):
        self.lib = load_library()
        self._mirror_pool = []  # pinned host mirror rings, reused across solver instances
        self._mirrors = {}  # device ctrl pointer -> _Mirror
        self._work_pool = {}  # (device, state dtype, stream) -> free _Work sets
        self._lag_ws = {}  # (device, L, stream) -> workspace of xde_lag_grad (zeroed once; every launch leaves it re-armed)
        self._tls = threading.local()  # .capturing: this THREAD is recording a hipGraph (its launches do not execute)
        # (the init block can ride the mirror only under the checksummed publish protocol, XDE_CTRL_FLAGS bit 8 — the default)
        self._mirror_takes_init = (int(os.environ.get("XDE_CTRL_FLAGS", "15")) & 8) != 0

    # -- host mirror ring of the control block (see xde_rk_control / xde_ctrl_wait) -------------
    class _Mirror:
        __slots__ = ("ptr", "seq", "seq0", "peek", "init_published")

        def __init__(self, ptr):
            self.ptr, self.seq, self.seq0 = ptr, 0, 0
            self.init_published = False  # the block the solve started from is in slot seq0 of the ring (initial_step_tail)
            self.peek = None  # ctrl_peek_async's pool of idle (pinned buffer, event, device) triples

    def _acquire_mirror(self):
        if self._mirror_pool:
            m = self._mirror_pool.pop()
            m.seq0 = m.seq  # nothing published for the new owner yet: ctrl_read falls back to a device copy
            m.init_published = False
            return m
        ptr = C.c_void_p()
        self._check(self.lib.xde_host_alloc(XDE_MIRROR_SLOTS * C.sizeof(XdeCtrl), C.byref(ptr)), "xde_host_alloc")
        return HipBackend._Mirror(ptr.value)

    def _release_mirror(self, key):
        m = self._mirrors.pop(key, None)
        if m is not None:
            self._mirror_pool.append(m)

    # -- helpers -------------------------------------------------------------------------
    def _is_capturing(self):
        """True while THIS thread records a hipGraph through capture(): controller launches recorded there do not run, so
        they must not advance the host's sequence number; other threads' launches do run and are counted as usual."""
        return getattr(self._tls, "capturing", False)

    def _check(self, rc, who):
        if rc != XDE_OK:
            raise XdeError("{} failed (status {}): {}".format(who, rc, self.lib.xde_last_error().decode()))

    def require_device(self, *tensors):
        self._require_device(*tensors)

    @staticmethod
    def _require_device(*tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise XdeError(
                    "paddlexde_amd: tensors must live on a ROCm device (got device={}); there is no CPU path".format(t.device)
                )

    @staticmethod
    def _stream(t):
        """Raw hipStream_t of torch's current stream on the tensor's device (the launch stream of every kernel)."""
        if _raw_stream is not None:
            return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
        return torch.cuda.current_stream(t.device).cuda_stream

    # -- allocation (torch is the allocator; the library never allocates) -------------------
    def new_ctrl(self, device):
        t = torch.zeros(C.sizeof(XdeCtrl), dtype=torch.uint8, device=device)
        key = t.data_ptr()
        self._mirrors[key] = self._acquire_mirror()
        weakref.finalize(t, self._release_mirror, key)
        return t

    def new_workspace(self, device):
        return torch.zeros(int(self.lib.xde_workspace_bytes()), dtype=torch.uint8, device=device)

    # The small per-solve device buffers (control block + mirror, norm workspace, sums, stage times) are recycled between
    # solver instances on the same device and stream: odeint_adjoint's backward builds one solver per output interval, and
    # four allocations + fills each time are a measurable part of a launch-bound solve.  Reuse is stream-ordered.
    def acquire_work(self, device, state_dtype):
        key = (device.index, state_dtype, self._stream_of(device))
        pool = self._work_pool.setdefault(key, [])
        if pool:
            return pool.pop()
        w = _Work()
        w.key = key
        w.ctrl = self.new_ctrl(device)
        w.ws = self.new_workspace(device)
        w.sums = self.new_sums(device)
        w.t_stage = torch.zeros(XDE_MAX_STAGE, dtype=state_dtype, device=device)
        w.t_views = [w.t_stage[i] for i in range(XDE_MAX_STAGE)]  # the 0-dim stage times handed to func (made once per work set)
        return w

    def release_work(self, w):
        pool = self._work_pool.setdefault(w.key, [])
        if len(pool) < 8:
            pool.append(w)

    @staticmethod
    def _stream_of(device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream

    def new_sums(self, device):
        return torch.zeros(2 * XDE_MAX_SEG, dtype=torch.float64, device=device)

    # -- kernels ---------------------------------------------------------------------------
    def stage_combine(self, out, y0, ks, coef, mode, *, scale=1.0, dt_host=0.0, ctrl=None, y0_alt=None, k0_alt=None,
                      out2=None, coef2=None, damping=0.0, nt_mask=0):
        self._require_device(out, y0, out2, *ks)
        if out.numel() == 0:
            return
        rc = self.lib.xde_stage_combine(
            out.data_ptr(), y0.data_ptr(), _ptr(y0_alt), _ptr_array(ks), _ptr(k0_alt), _dbl_array(coef), len(ks),
            mode, float(scale), float(dt_host), _ptr(ctrl), out.numel(), dtype_code(out.dtype), _ptr(out2),
            _dbl_array(coef2) if coef2 is not None else None, float(damping), int(nt_mask) & 0xFFFFFFFF, self._stream(out),
        )
        self._check(rc, "xde_stage_combine")

    def stage_combine_pre(self, out, y0, pre, ks, coef, *, dt_host=0.0, ctrl=None, y0_alt=None, nt_mask=0):
        """out = y0 + ((pre + ks[0] coef[0] dt) + ...): a stage whose earlier operands the previous stage's launch has already summed
        into ``pre`` (its second output)."""
        self._require_device(out, y0, pre, y0_alt, *ks)
        if out.numel() == 0:
            return
        rc = self.lib.xde_stage_combine_pre(out.data_ptr(), y0.data_ptr(), _ptr(y0_alt), pre.data_ptr(), _ptr_array(ks), _dbl_array(coef),
                                            len(ks), float(dt_host), _ptr(ctrl), out.numel(), dtype_code(out.dtype),
                                            int(nt_mask) & 0xFFFFFFFF, self._stream(out))
        self._check(rc, "xde_stage_combine_pre")

    def stage_combine_pre_weighted(self, out, y0, pre, ks, coef, *, scale=1.0, dt_host=0.0, ctrl=None, damping=0.0):
        """out = ((pre + fuse(ks[0]) coef[0]) + ...) * scale: a fixed-step solver's final weighted sum whose leading terms the last
        stage-input launch (mode FUSE, ``out2`` / ``coef2``) has already summed into ``pre``."""
        self._require_device(out, y0, pre, *ks)
        if out.numel() == 0:
            return
        rc = self.lib.xde_stage_combine_pre_weighted(out.data_ptr(), y0.data_ptr(), pre.data_ptr(), _ptr_array(ks), _dbl_array(coef), len(ks),
                                                     float(scale), float(dt_host), _ptr(ctrl), out.numel(), dtype_code(out.dtype),
                                                     float(damping), self._stream(out))
        self._check(rc, "xde_stage_combine_pre_weighted")

    def error_norm_partial(self, ks, c_err, y0, y1, rtol, atol, segs, norm_kind, ws, *, dt_host=0.0, ctrl=None,
                           y0_alt=None, k0_alt=None, e_pre=None):
        self._require_device(y0, y1, ws, e_pre, *ks)
        rc = self.lib.xde_error_norm_partial(
            _ptr_array(ks), _ptr(k0_alt), _dbl_array(c_err), len(ks), y0.data_ptr(), _ptr(y0_alt), y1.data_ptr(),
            float(rtol), float(atol), float(dt_host), _ptr(ctrl), C.byref(segs), norm_kind, dtype_code(y0.dtype),
            ws.data_ptr(), _ptr(e_pre), self._stream(y0),
        )
        self._check(rc, "xde_error_norm_partial")

    def error_norm_control(self, ks, c_err, y0, y1, segs, ws, ctrl, params, t_span_dev, step_t_dev, t_stage, *, y0_alt=None,
                           k0_alt=None, e_pre=None):
        """Error norm + controller in one launch (single GPU, native norm)."""
        self._require_device(y0, y1, ws, ctrl, e_pre, *ks)
        m = self._mirrors.get(ctrl.data_ptr())
        rc = self.lib.xde_error_norm_control(
            _ptr_array(ks), _ptr(k0_alt), _dbl_array(c_err), len(ks), y0.data_ptr(), _ptr(y0_alt), y1.data_ptr(), C.byref(segs),
            dtype_code(y0.dtype), ws.data_ptr(), _ptr(e_pre), ctrl.data_ptr(), C.byref(params), t_span_dev.data_ptr(),
            _ptr(step_t_dev), t_stage.data_ptr(), m.ptr if m is not None else None, self._stream(y0),
        )
        self._check(rc, "xde_error_norm_control")
        if m is not None and not self._is_capturing():
            m.seq += 1

    def error_ratio(self, out, ks, c_err, y0, y1, rtol, atol, *, dt_host=0.0, ctrl=None, y0_alt=None, k0_alt=None, nonfinite_out=None):
        self._require_device(out, y0, y1, y0_alt, k0_alt, nonfinite_out, *ks)
        if out.numel() == 0:
            return
        rc = self.lib.xde_error_ratio(out.data_ptr(), _ptr_array(ks), _ptr(k0_alt), _dbl_array(c_err), len(ks), y0.data_ptr(),
                                      _ptr(y0_alt), y1.data_ptr(), float(rtol), float(atol), float(dt_host), _ptr(ctrl), out.numel(),
                                      dtype_code(out.dtype), _ptr(nonfinite_out), self._stream(out))
        self._check(rc, "xde_error_ratio")

    def scaled_norm_partial(self, a, b, y0, rtol, atol, segs, norm_kind, ws, slot):
        self._require_device(a, b, y0, ws)
        rc = self.lib.xde_scaled_norm_partial(
            a.data_ptr(), _ptr(b), y0.data_ptr(), float(rtol), float(atol), C.byref(segs), norm_kind,
            dtype_code(y0.dtype), ws.data_ptr(), slot, self._stream(y0),
        )
        self._check(rc, "xde_scaled_norm_partial")

    def norm_finalize(self, ws, slot, sums):
        self._require_device(ws, sums)
        self._check(self.lib.xde_norm_finalize(ws.data_ptr(), slot, sums.data_ptr(), self._stream(ws)), "xde_norm_finalize")

    def norm_result(self, sums, seg_count, norm_kind, state_dtype, result):
        self._require_device(sums, result)
        rc = self.lib.xde_norm_result(sums.data_ptr(), _dbl_array(seg_count), len(seg_count), norm_kind, state_dtype,
                                      result.data_ptr(), self._stream(sums))
        self._check(rc, "xde_norm_result")

    def rk_control(self, ctrl, params, ws, sums, t_span_dev, step_t_dev, t_stage):
        self._require_device(ctrl, t_span_dev, t_stage)
        m = self._mirrors.get(ctrl.data_ptr())
        rc = self.lib.xde_rk_control(ctrl.data_ptr(), C.byref(params), _ptr(ws), _ptr(sums), t_span_dev.data_ptr(),
                                     _ptr(step_t_dev), t_stage.data_ptr(), m.ptr if m is not None else None,
                                     self._stream(ctrl))
        self._check(rc, "xde_rk_control")
        if m is not None and not self._is_capturing():
            m.seq += 1

    def p2p_rk_control(self, ctrl, params, ws, exchange, t_span_dev, step_t_dev, t_stage):
        """finalize -> peer-to-peer exchange -> controller of a sharded attempt as ONE launch (``exchange``: a utils.PeerExchange)."""
        self._require_device(ctrl, ws, t_span_dev, t_stage)
        m = self._mirrors.get(ctrl.data_ptr())
        rc = self.lib.xde_p2p_rk_control(ctrl.data_ptr(), C.byref(params), ws.data_ptr(), exchange._local, exchange._peers,
                                         exchange.world, exchange.rank, exchange.SPIN_LIMIT, t_span_dev.data_ptr(), _ptr(step_t_dev),
                                         t_stage.data_ptr(), m.ptr if m is not None else None, self._stream(ctrl))
        self._check(rc, "xde_p2p_rk_control")
        if m is not None and not self._is_capturing():
            m.seq += 1

    def initial_step(self, phase, res, hs, params, t_start, t_probe, ctrl):
        """Scalar part of select_initial_step on the device (phase 0: h0; phase 1: the first step; phase 2: phase 0 with the start
        time in ``res[2]``)."""
        self._require_device(res, hs, ctrl, t_probe)
        rc = self.lib.xde_initial_step(int(phase), res.data_ptr(), hs.data_ptr(), C.byref(params), float(t_start), _ptr(t_probe),
                                       dtype_code(t_probe.dtype) if t_probe is not None else XDE_F32, ctrl.data_ptr(), self._stream(ctrl))
        self._check(rc, "xde_initial_step")

    def initial_step_fused(self, phase, a, b, y0, segs, hs, params, t_start, t_probe, ctrl, n_out=0, t_span_dev=None, step_t_dev=None,
                           t_stage=None, keep_seq=False):
        """The initial-step heuristic of a small state, one workgroup per phase (phase 1 also constructs the control block, like
        ctrl_init with the device-resident first step).  ``t_start = nan``: the start time is ``t_span_dev[0]``; ``keep_seq``: the
        block goes on counting its controller launches (a launch recorded in a graph and replayed per output interval)."""
        self._require_device(a, b, y0, hs, ctrl, t_probe, t_span_dev, t_stage)
        m = self._mirrors.get(ctrl.data_ptr()) if (phase == 1 and not keep_seq) else None
        seq0 = -1 if keep_seq else (m.seq if m is not None else 0)
        rc = self.lib.xde_initial_step_fused(int(phase), a.data_ptr(), _ptr(b), y0.data_ptr(), C.byref(segs), dtype_code(y0.dtype), hs.data_ptr(),
                                             C.byref(params), float(t_start), _ptr(t_probe),
                                             dtype_code(t_probe.dtype) if t_probe is not None else XDE_F32, ctrl.data_ptr(), int(n_out),
                                             _ptr(t_span_dev), _ptr(step_t_dev), _ptr(t_stage), seq0, self._stream(y0))
        self._check(rc, "xde_initial_step_fused")
        if m is not None:
            m.seq0 = seq0
            m.init_published = False

    def scaled_norm2_partial(self, f0, y0, rtol, atol, segs, norm_kind, ws):
        """Partials of norm(y0 / scale) (slot 0) and norm(f0 / scale) (slot 1) in one pass over (y0, f0)."""
        self._require_device(f0, y0, ws)
        rc = self.lib.xde_scaled_norm2_partial(f0.data_ptr(), y0.data_ptr(), float(rtol), float(atol), C.byref(segs), norm_kind,
                                               dtype_code(y0.dtype), ws.data_ptr(), self._stream(y0))
        self._check(rc, "xde_scaled_norm2_partial")

    def initial_step_tail(self, phase, ws, hs, params, t_start, t_probe, ctrl, n_out=0, t_span_dev=None, step_t_dev=None, t_stage=None,
                          keep_seq=False):
        """What followed a norm pass of the initial-step heuristic as launches of its own — finalize, result, the scalar phase, and in
        phase 1 the control block's construction — as ONE one-workgroup launch (states above initial_step_fused's reach).
        ``t_start = nan`` / ``keep_seq``: as initial_step_fused.  Phase 1 (not ``keep_seq``) also publishes the constructed block to the
        control block's host mirror: ``ctrl_init_handle(ctrl)`` is then the handle ``ctrl_wait`` takes — where the first attempt lands,
        without a copy command on the stream (the publication takes a sequence number of its own: the slot it lands in may still
        hold the previous owner's last block under the old number)."""
        self._require_device(ws, hs, ctrl, t_probe, t_span_dev, t_stage)
        m = self._mirrors.get(ctrl.data_ptr()) if (phase == 1 and not keep_seq) else None
        publish = m is not None and not self._is_capturing() and self._mirror_takes_init
        if publish:
            m.seq += 1
        seq0 = -1 if keep_seq else (m.seq if m is not None else 0)
        rc = self.lib.xde_initial_step_tail(int(phase), ws.data_ptr(), hs.data_ptr(), C.byref(params), float(t_start), _ptr(t_probe),
                                            dtype_code(t_probe.dtype) if t_probe is not None else XDE_F32, ctrl.data_ptr(), int(n_out),
                                            _ptr(t_span_dev), _ptr(step_t_dev), _ptr(t_stage), seq0, m.ptr if publish else None,
                                            self._stream(ctrl))
        self._check(rc, "xde_initial_step_tail")
        if m is not None:
            m.seq0 = seq0
            m.init_published = publish

    def ctrl_init_handle(self, ctrl):
        """Handle (for ``ctrl_wait``) of the freshly constructed block when the launch that constructed it published it to the host
        mirror (initial_step_tail); else None — ``ctrl_peek_async`` enqueues a copy instead."""
        m = self._mirrors.get(ctrl.data_ptr())
        if m is not None and m.init_published and m.seq == m.seq0:
            return (m, m.seq0)
        return None

    def ctrl_init(self, ctrl, params, t_start, first_step, n_out, t_span_dev, step_t_dev, t_stage, first_step_dev=None, keep_seq=False):
        """``t_start = nan``: the start time is ``t_span_dev[0]``; ``keep_seq``: the block goes on counting its controller launches
        (both for a launch recorded in a graph and replayed per output interval)."""
        self._require_device(ctrl, t_span_dev, t_stage)
        m = None if keep_seq else self._mirrors.get(ctrl.data_ptr())
        seq0 = -1 if keep_seq else (m.seq if m is not None else 0)
        rc = self.lib.xde_ctrl_init(ctrl.data_ptr(), C.byref(params), float(t_start), float(first_step), int(n_out),
                                    t_span_dev.data_ptr(), _ptr(step_t_dev), t_stage.data_ptr(), seq0, _ptr(first_step_dev),
                                    self._stream(ctrl))
        self._check(rc, "xde_ctrl_init")
        if m is not None:
            m.seq0 = seq0
            m.init_published = False

    def ctrl_retarget(self, ctrl, params, t_span_dev, n_out):
        """New output list for a running solve (the device side of AdaptiveRKSolver.step(next_t))."""
        self._require_device(ctrl, t_span_dev)
        m = self._mirrors.get(ctrl.data_ptr())
        rc = self.lib.xde_ctrl_retarget(ctrl.data_ptr(), C.byref(params), t_span_dev.data_ptr(), int(n_out),
                                        m.ptr if m is not None else None, self._stream(ctrl))
        self._check(rc, "xde_ctrl_retarget")
        if m is not None and not self._is_capturing():
            m.seq += 1

    def ctrl_read(self, ctrl) -> XdeCtrl:
        """The newest control block.  With a host mirror: poll the pinned ring (no HIP call); else a blocking copy."""
        self._require_device(ctrl)
        m = self._mirrors.get(ctrl.data_ptr())
        if m is not None and m.seq > m.seq0:
            return self.ctrl_wait((m, m.seq))
        host = XdeCtrl()
        self._check(self.lib.xde_ctrl_read(ctrl.data_ptr(), C.byref(host), self._stream(ctrl)), "xde_ctrl_read")
        return host

    PEEK_POOL_MAX = 4  # idle (buffer, event) pairs kept per control block's mirror

    class _Peek:
        """One enqueued copy of a control block: OWNS its pinned buffer and its event until ``ctrl_peek_result`` has consumed it (or the
        handle is dropped); only then do they go back to the mirror's pool.  Any number of peeks may be pending on one block."""

        __slots__ = ("host", "ev", "device", "pool", "__weakref__")

        def __init__(self, host, ev, device, pool):
            self.host, self.ev, self.device, self.pool = host, ev, device, pool

        def _recycle(self):
            host, ev, pool = self.host, self.ev, self.pool
            self.host = self.ev = self.pool = None
            if host is None:
                return
            if pool is not None and len(pool) < HipBackend.PEEK_POOL_MAX:
                pool.append((host, ev, self.device))
            else:
                # not kept: releasing a pinned buffer records an event on the stream its copy ran on — which may be recording a
                # hipGraph right now (this runs from __del__, at any allocation): utils.graphed parks the pair until none is
                from .utils.graphed import release_when_idle

                release_when_idle((host, ev))

        def __del__(self):  # a handle nobody read (a solve that ended before its second attempt): the pair is reusable all the same —
            try:  # copies are stream-ordered, a later peek's copy and event land after this one's
                self._recycle()
            except Exception:  # interpreter shutdown
                pass

    def ctrl_peek_async(self, ctrl):
        """Enqueue a copy of the control block AS IT IS AT THIS POINT OF THE STREAM into pinned host memory (a freshly constructed block
        has no mirror slot); ``ctrl_peek_result(handle)`` waits for that copy only — not for anything enqueued after it.  Re-entrant:
        every handle has a buffer and an event of its own until it is consumed; they are pooled with the block's mirror (control
        blocks are recycled between solves), so a steady state of one peek per solve allocates nothing."""
        self._require_device(ctrl)
        m = self._mirrors.get(ctrl.data_ptr())
        pool = None
        if m is not None:
            if m.peek is None:
                m.peek = []
            pool = m.peek
        kept = None
        while pool:
            kept = pool.pop()
            if kept[2] == ctrl.device:
                break
            kept = None
        if kept is None:
            kept = (torch.empty(C.sizeof(XdeCtrl), dtype=torch.uint8).pin_memory(), torch.cuda.Event(), ctrl.device)
        host, ev = kept[0], kept[1]
        with torch.cuda.device(ctrl.device):
            host.copy_(ctrl, non_blocking=True)
            ev.record()
        return HipBackend._Peek(host, ev, ctrl.device, pool)

    def ctrl_peek_result(self, handle) -> XdeCtrl:
        if isinstance(handle, tuple):  # ctrl_init_handle: the block is in the mirror ring
            return self.ctrl_wait(handle)
        if handle.host is None:
            raise XdeError("ctrl_peek_result: this handle has been consumed already")
        handle.ev.synchronize()
        out = XdeCtrl.from_buffer_copy(handle.host.numpy().tobytes())
        handle._recycle()
        return out

    def ctrl_read_async(self, ctrl):
        """Handle for the control block of the newest controller launch; nothing is enqueued on the stream."""
        self._require_device(ctrl)
        m = self._mirrors.get(ctrl.data_ptr())
        if m is None or m.seq == m.seq0:
            raise XdeError("ctrl_read_async needs a control block from new_ctrl() with at least one controller launch")
        return (m, m.seq)

    def ctrl_wait(self, handle) -> XdeCtrl:
        m, seq = handle
        host = XdeCtrl()
        self._check(self.lib.xde_ctrl_wait(m.ptr, seq, self.WAIT_TIMEOUT_MS, C.byref(host)), "xde_ctrl_wait")
        return host

    def dense_eval(self, out_base, ks, mid, y0, y1, f1, ctrl, t_span_dev, time_dtype, *, y0_alt=None, k0_alt=None,
                   expect_step=-1):
        self._require_device(out_base, y0, y1, f1, ctrl, t_span_dev, *ks)
        if y0.numel() == 0:
            return
        rc = self.lib.xde_dense_eval(
            out_base.data_ptr(), _ptr_array(ks), _ptr(k0_alt), _dbl_array(mid), len(ks), y0.data_ptr(), _ptr(y0_alt),
            y1.data_ptr(), f1.data_ptr(), ctrl.data_ptr(), t_span_dev.data_ptr(), time_dtype, y0.numel(),
            dtype_code(y0.dtype), int(expect_step), self._stream(y0),
        )
        self._check(rc, "xde_dense_eval")

    def scale_fanout(self, outs, g, factors, dt_dev=None):
        self._require_device(g, *outs)
        if g.numel() == 0:
            return
        rc = self.lib.xde_scale_fanout(_ptr_array(outs), g.data_ptr(), _dbl_array(factors), len(outs), _ptr(dt_dev), g.numel(),
                                       dtype_code(g.dtype), self._stream(g))
        self._check(rc, "xde_scale_fanout")

    def hermite_gather(self, val, der, his, his_t, lags):
        """his [..., T, D] (contiguous), his_t [T], lags [L]  ->  val, der [..., L, D]."""
        self._require_device(val, der, his, his_t, lags)
        T, D = his.shape[-2], his.shape[-1]
        outer = his.numel() // (T * D) if T * D else 0
        rc = self.lib.xde_hermite_gather(val.data_ptr(), der.data_ptr(), his.data_ptr(), his_t.data_ptr(), lags.data_ptr(), outer,
                                         T, D, lags.numel(), dtype_code(his.dtype), self._stream(his))
        self._check(rc, "xde_hermite_gather")

    def history_gather(self, val, der, his, his_t, lags, method):
        """his [..., T, D] (contiguous), his_t [T], lags [L]  ->  val, der [..., L, D] of the history spline `method`
        ("cubic" | "linear" | "bez")."""
        self._require_device(val, der, his, his_t, lags)
        T, D = his.shape[-2], his.shape[-1]
        outer = his.numel() // (T * D) if T * D else 0
        rc = self.lib.xde_history_gather(val.data_ptr(), der.data_ptr(), his.data_ptr(), his_t.data_ptr(), lags.data_ptr(), outer, T, D,
                                         lags.numel(), dtype_code(his.dtype), HISTORY_METHODS[method], self._stream(his))
        self._check(rc, "xde_history_gather")

    def lag_grad(self, grad_y, der):
        """sum over every axis but the lag axis of grad_y * der ([..., L, D] both, contiguous) -> [L], one launch."""
        self._require_device(grad_y, der)
        L, D = der.shape[-2], der.shape[-1]
        outer = der.numel() // (L * D) if L * D else 0
        if L == 0 or outer == 0:
            return torch.zeros(L, dtype=der.dtype, device=der.device)
        out = torch.empty(L, dtype=der.dtype, device=der.device)
        # per STREAM: the workspace holds the arrival counters and the partials of a launch in flight — two launches of the same L on
        # different streams (autograd on a side stream, two models) must not meet on them; same-stream launches are ordered
        key = (der.device.index, L, self._stream(der))
        ws = self._lag_ws.get(key)
        if ws is None:
            ws = self._lag_ws[key] = torch.zeros(int(self.lib.xde_lag_grad_workspace_bytes(L)), dtype=torch.uint8, device=der.device)
        rc = self.lib.xde_lag_grad(out.data_ptr(), grad_y.data_ptr(), der.data_ptr(), outer, D, L, dtype_code(der.dtype), ws.data_ptr(),
                                   self._stream(der))
        self._check(rc, "xde_lag_grad")
        return out

    def pack_segments(self, flat, tensors, segs, scales=None):
        """``flat`` (16-byte-aligned segments, pads zero) <- the contiguous device tensors ``tensors`` at ``segs`` = [(start, len)], one
        launch.  Returns False (nothing done) when the call does not fit the kernel: the caller packs with framework ops then."""
        n = len(tensors)
        if n < 1 or n > XDE_MAX_PACK or not flat.is_cuda or flat.dtype not in (torch.float32, torch.float64):
            return False
        if len(segs) != n:
            return False
        for x, (_, length) in zip(tensors, segs):
            # (a member whose element count is not its segment's — a func that returned a wrong-shaped or broadcastable member —
            # must not reach the kernel, which reads `length` elements from it: the framework-op pack raises, or broadcasts, as before)
            if x.dtype != flat.dtype or x.device != flat.device or not x.is_contiguous() or x.numel() != int(length):
                return False
        srcs = (C.c_void_p * n)(*[x.data_ptr() if x.numel() else None for x in tensors])
        starts = (C.c_int64 * n)(*[int(s) for s, _ in segs])
        lens = (C.c_int64 * n)(*[int(l) for _, l in segs])
        sc = _dbl_array(scales) if scales is not None else None
        rc = self.lib.xde_pack_segments(flat.data_ptr(), srcs, starts, lens, sc, n, flat.numel(), dtype_code(flat.dtype), self._stream(flat))
        self._check(rc, "xde_pack_segments")
        return True

    def dense_commit(self, out_base, ks, mid, y0, y1, f1, ctrl, t_span_dev, time_dtype):
        """dense_eval + commit in one launch (graph pipeline): rows of the last accepted step, then (y0, ks[0]) <- (y1, f1)."""
        self._require_device(out_base, y0, y1, f1, ctrl, t_span_dev, *ks)
        if y0.numel() == 0:
            return
        rc = self.lib.xde_dense_commit(out_base.data_ptr(), _ptr_array(ks), _dbl_array(mid), len(ks), y0.data_ptr(), y1.data_ptr(),
                                       ks[0].data_ptr(), f1.data_ptr(), ctrl.data_ptr(), t_span_dev.data_ptr(), time_dtype,
                                       y0.numel(), dtype_code(y0.dtype), self._stream(y0))
        self._check(rc, "xde_dense_commit")

    def commit(self, ctrl, y0_dst, y1_src, f0_dst, f1_src):
        self._require_device(ctrl, y0_dst, y1_src, f0_dst, f1_src)
        rc = self.lib.xde_commit(ctrl.data_ptr(), y0_dst.data_ptr(), y1_src.data_ptr(), f0_dst.data_ptr(), f1_src.data_ptr(),
                                 y0_dst.numel(), dtype_code(y0_dst.dtype), self._stream(y0_dst))
        self._check(rc, "xde_commit")

    # -- hipGraph capture of one attempted step ---------------------------------------------------
    class _Graph:
        def __init__(self, backend, graph, ctrl, launches=1):
            self.backend, self.graph, self.ctrl, self.launches = backend, graph, ctrl, int(launches)

        def replay(self):
            """Launch the graph; returns one read handle per controller launch it holds, in order."""
            self.graph.replay()
            m = self.backend._mirrors.get(self.ctrl.data_ptr())
            if m is None:
                return []
            first = m.seq + 1
            m.seq += self.launches
            return [(m, first + i) for i in range(self.launches)]

    def capture(self, body, ctrl, launches=1):
        """Record ``body()`` (kernels of this library + the framework ops of the user's func) into a hipGraph; ``launches`` =
        controller launches ``body`` makes (attempted steps per replay)."""
        from .utils.graphed import CapturedGraph

        g = CapturedGraph()  # (replays of a graph that holds memset nodes are synchronised: see its docstring)
        self._tls.capturing = True
        try:
            with g.capture(capture_error_mode="thread_local"):  # other host threads may keep using the device meanwhile
                body()
        finally:
            self._tls.capturing = False
        g.finish()
        return HipBackend._Graph(self, g, ctrl, launches)

    # -- profiling ---------------------------------------------------------------------------
    def prof_enable(self, period=1):
        """0/False: off; p >= 1: time every p-th launch of each kernel with dispatch-stamped HIP events."""
        self._check(self.lib.xde_prof_enable(int(period)), "xde_prof_enable")

    def prof_collect(self):
        n = len(KID_NAMES)
        counts = (C.c_int64 * n)()
        ms = (C.c_double * n)()
        by = (C.c_double * n)()
        self._check(self.lib.xde_prof_collect(counts, ms, by), "xde_prof_collect")
        return {KID_NAMES[i]: {"launches": int(counts[i]), "ms": float(ms[i]), "bytes": float(by[i])} for i in range(n)}


_backend = None


def get_backend():
    """The HIP backend (created on first use).  Raises loudly when the library is not built."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def _set_backend_for_testing(backend):
    """TEST HOOK ONLY (tests/ inject a CPU double to exercise host logic without a GPU).

    Product code never calls this; passing None restores the HIP backend.
    """
    global _backend
    _backend = backend
