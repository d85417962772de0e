"""Norms of the reference's ``paddlexde/utils/ode_utils.py`` on torch tensors.

They are real callables (users may call them), and they carry a marker that lets the solvers
map them onto the native norm kernels of libxde_hip.so instead of running them as eager ops:
``_rms_norm`` -> XDE_NORM_RMS, ``_linf_norm`` -> XDE_NORM_LINF (reference lines: ode_utils.py:4-19).
"""
import torch


def _linf_norm(tensor):
    """ode_utils.py:4-5"""
    return tensor.abs().max()


def _rms_norm(tensor):
    """ode_utils.py:8-9"""
    return tensor.abs().pow(2).mean().sqrt()


def _zero_norm(tensor):
    """ode_utils.py:12-13"""
    return 0.0


def _mixed_norm(tensor_tuple):
    """ode_utils.py:16-19"""
    if len(tensor_tuple) == 0:
        return 0.0
    return max([_rms_norm(tensor) for tensor in tensor_tuple])


_rms_norm._xde_native = ("rms",)
_linf_norm._xde_native = ("linf",)


def native_norm_spec(norm):
    """Return the native description of ``norm`` or None for an arbitrary callable.

    ("rms",) / ("linf",)                 single-tensor state
    ("mixed", n_norm_segments | None)    tuple state: max over segments of the per-segment RMS
    """
    return getattr(norm, "_xde_native", None)


def sort_tvals(tvals, t0):
    """ode_utils.py:22-25"""
    tvals = tvals[tvals >= t0]
    return torch.sort(tvals).values
