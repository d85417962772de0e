"""Host-side helpers shared by the solvers: operand hygiene and time tables."""
import collections

import numpy as np
import torch

_NP = {torch.float32: np.float32, torch.float64: np.float64}


def np_dtype(dt):
    try:
        return _NP[dt]
    except KeyError:
        raise TypeError("paddlexde_amd supports float32 / float64, got {}".format(dt))


def as_operand(x, like=None):
    """Contiguous, 16-byte aligned tensor the kernels can consume (clone only when needed)."""
    if like is not None:
        if x.dtype != like.dtype:
            x = x.to(like.dtype)
        if x.shape != like.shape:
            x = x.expand_as(like)
    if not x.is_contiguous():
        x = x.contiguous()
    if x.data_ptr() % 16:
        y = torch.empty_like(x)
        y.copy_(x)
        x = y
    return x


def storage_ptr(x):
    return x.untyped_storage().data_ptr()


def t_span_to_host(t_span, np_time_dtype):
    """One device->host copy of the output times (the only place a device t_span is read)."""
    if torch.is_tensor(t_span):
        arr = t_span.detach().to("cpu").numpy()
    else:
        arr = np.asarray(t_span)
    return arr.astype(np_time_dtype)


def upload(arr, device):
    """Host numpy array -> device tensor without blocking the host: staged through torch's caching pinned allocator and
    copied asynchronously on the current stream (a pageable H2D copy would wait for everything enqueued before it)."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


_CONST_TABLES = collections.OrderedDict()  # (dtype, bytes, device, stream) -> device tensor; see upload_const
_CONST_TABLES_MAX = 16


def upload_const(arr, device, stream_key=None):
    """``upload`` for a small table the kernels only READ (a solve's output times): the device copy of an identical table uploaded
    before on the same device and stream is handed out again — a training loop that integrates over the same output times every
    iteration pays the pinned staging and the copy (~25 us of host time, profiles/r06_odeint_tail.txt) once.  The caller must not
    write into the result.  Nothing is cached while a stream capture is in progress (such a tensor would live in the graph's pool)."""
    if device.type != "cuda" or arr.nbytes > 512 or torch.cuda.is_current_stream_capturing():
        return upload(arr, device)
    key = (arr.dtype.str, arr.shape, arr.tobytes(), device.index, stream_key)
    t = _CONST_TABLES.get(key)
    if t is None:
        t = _CONST_TABLES[key] = upload(arr, device)
        while len(_CONST_TABLES) > _CONST_TABLES_MAX:
            _CONST_TABLES.popitem(last=False)
    else:
        _CONST_TABLES.move_to_end(key)
    return t


def scalar_const(value, dtype, device, stream_key=None):
    """``scalar`` for a value func only READS (a solve's start time): the 0-dim tensor made for the same value, dtype, device and
    stream before is handed out again (a fill launch and ~10 us of host time per solve otherwise).  As with the reference, which hands
    func views of the caller's own ``t_span``, a func must not write into its time argument."""
    if torch.device(device).type != "cuda" or torch.cuda.is_current_stream_capturing():
        return scalar(value, dtype, device)
    key = ("scalar", float(value), dtype, torch.device(device).index, stream_key)
    t = _CONST_TABLES.get(key)
    if t is None:
        t = _CONST_TABLES[key] = scalar(value, dtype, device)
        while len(_CONST_TABLES) > _CONST_TABLES_MAX:
            _CONST_TABLES.popitem(last=False)
    else:
        _CONST_TABLES.move_to_end(key)
    return t


def scalar(value, dtype, device):
    """0-dim device tensor holding ``value``: a fill kernel with the value as argument (no host-to-device copy)."""
    return torch.full((), float(value), dtype=dtype, device=device)


def direction_of(t_host):
    """+1 / -1: the direction of integration of an output grid, decided by its first two entries exactly as the oracle's time flip
    decides it (`t_span[1] < t_span[0]`): a grid that repeats its start time counts as forward, so `[0, 0, -1]` is a forward grid
    with an output time behind the start — refused like every other out-of-order time."""
    return -1 if (len(t_host) > 1 and t_host[1] < t_host[0]) else 1
