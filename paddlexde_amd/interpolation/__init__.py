"""The history splines as objects — the reference's ``paddlexde.interpolation`` classes (interpolation/__init__.py:1:
``LinearInterpolation``, ``CubicHermiteSpline``, ``BezierSpline`` over interpolation/interpolate_base.py:7-107) on the history kernels.

``HistoryIndex`` (xde/base_dde.py:104-121) builds one of these per call and asks it for ``evaluate(lags)`` and ``derivative(lags)``; the
kernels (xde_history_gather / xde_hermite_gather) compute both in one pass over the series, so here the classes are a thin front on that
launch: same constructor (``series [..., T, D]``, ``t [T]`` or None), same ``evaluate(t)`` / ``derivative(t)`` ``-> [..., len(t), D]``,
same conventions as written in the reference (``index = clip(bucketize(t) - 1, 0, T - 1)``, one-sided node derivatives for the cubic,
row scales ``scale1..4``), in the package's default dtype float32 (interpolate_base.py:18,27-28) unless the series is float64.  Held to
the reference's own tests for these classes (tests/interpolation/test_interpolation.py:13-85) in tests/_dde_cases.py.
"""
import torch

from .. import _hip

__all__ = ["LinearInterpolation", "CubicHermiteSpline", "BezierSpline"]


class InterpolationBase:
    method = None  # "linear" | "cubic" | "bez": the kernels' name of the spline
    min_times = 2

    def __init__(self, series, t=None, **kwargs):
        be = _hip.get_backend()
        series = torch.as_tensor(series)
        be.require_device(series)
        dtype = series.dtype if series.dtype == torch.float64 else torch.float32  # (`default_type`: float32)
        self._series = series.detach().to(dtype).contiguous()
        n_times = self._series.shape[-2]
        if n_times < self.min_times:
            raise ValueError("{} needs at least {} time points".format(type(self).__name__, self.min_times))
        if t is None:
            # the reference's default grid is linspace(0, T, T + 1) (:20-25): its first T points are the series' times, the extra one is
            # never indexed (`index` is clipped to T - 1) and its spacing repeats the last one — the unit grid 0..T-1 is the same spline
            t = torch.arange(n_times, dtype=dtype, device=self._series.device)
        self._t = torch.as_tensor(t).detach().to(device=self._series.device, dtype=dtype).contiguous().reshape(-1)
        if self._t.numel() != n_times:
            raise ValueError("t must have one entry per time row of the series ({} != {})".format(self._t.numel(), n_times))
        self._backend = be

    @property
    def grid_points(self):
        """The time points (interpolate_base.py:39-42)."""
        return self._t

    @property
    def interval(self):
        """The time interval between the first and the last time point (:44-47)."""
        return torch.stack([self._t[0], self._t[-1]])

    def _gather(self, t):
        tq = torch.as_tensor(t).detach().to(device=self._series.device, dtype=self._series.dtype).contiguous().reshape(-1)
        shape = tuple(self._series.shape[:-2]) + (tq.numel(), self._series.shape[-1])
        val = torch.empty(shape, dtype=self._series.dtype, device=self._series.device)
        der = torch.empty_like(val)
        self._backend.history_gather(val, der, self._series, self._t, tq, self.method)
        return val, der

    def evaluate(self, t):
        """The value at the time points ``t`` (:72-90): ``[..., len(t), D]``."""
        return self._gather(t)[0]

    def derivative(self, t):
        """The time derivative at ``t`` (:92-107)."""
        return self._gather(t)[1]


class LinearInterpolation(InterpolationBase):
    """interpolation/interpolate.py:6-99."""

    method = "linear"
    min_times = 2


class CubicHermiteSpline(InterpolationBase):
    """interpolation/interpolate.py:100-204 (node derivatives: one-sided differences, the last one repeated)."""

    method = "cubic"
    min_times = 2


class BezierSpline(InterpolationBase):
    """interpolation/interpolate.py:207-298 (four rows per interval, cubic Bernstein weights)."""

    method = "bez"
    min_times = 4
