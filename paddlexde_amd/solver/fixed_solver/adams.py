"""Adams-Bashforth(-Moulton) on the HIP combine kernel (reference: paddlexde/solver/fixed_solver/adams.py:457-547).

Step logic of the reference: every step evaluates f0 = f(t0, y0) and pushes it onto a newest-first history of at
most ``max_order - 1`` derivatives; with fewer than 3 entries the step is the reference RK4 variant
(``rk4_alt_step_func``); otherwise the explicit predictor ``dy = sum_j bashforth[order][j] * prev_f[j]`` and
``y1 = fuse(dy, dt, y0)``; with ``implicit=True`` up to ``max_iters`` Adams-Moulton corrector iterations
``dy = moulton[order+1][0] * f(t1, fuse(dy)) + sum_j moulton[order+1][j+1] * prev_f[j]`` follow, stopped by the
element-wise ``linf`` convergence test of ``_has_converged`` (:500-505).

As written the reference concatenates the history along the batch axis and feeds a 1-D coefficient vector and a 2-D
history to ``paddle.dot`` (:511-527), which cannot run; the intent (torchdiffeq's solver, and what the coefficient
tables mean) is the linear combination above, computed here by ONE xde_stage_combine launch over the history
tensors ``func`` returned (zero copy).  Coefficients come from ``_adams_coeffs`` (generated exactly; equal to the
reference's tables).
"""
import collections
import warnings

import torch

from ... import _hip
from ..base_fixed_solver import FixedSolver, _one_third, _two_thirds
from ._adams_coeffs import bashforth, moulton

_MIN_ORDER = 4
_MAX_ORDER = 12
_MAX_ITERS = 4


class AdamsBashforthMoulton(FixedSolver):
    graphable = False  # the corrector iterates until a data-dependent convergence test passes
    order = 4

    def __init__(self, xde, y0, rtol=1e-3, atol=1e-4, implicit=False, max_iters=_MAX_ITERS, max_order=_MAX_ORDER, **kwargs):
        super().__init__(xde, y0, rtol=rtol, atol=rtol, **kwargs)  # (atol=rtol: adams.py:470-472, kept)
        assert max_order <= _MAX_ORDER, "max_order must be at most {}".format(_MAX_ORDER)
        if max_order < _MIN_ORDER:
            warnings.warn("max_order is below {}, so the solver reduces to `rk4`.".format(_MIN_ORDER))
        self.rtol = float(rtol)
        self.atol = float(atol)
        self.implicit = implicit
        self.max_iters = max_iters
        self.max_order = int(max_order)
        self.prev_f = collections.deque(maxlen=self.max_order - 1)  # newest first
        self._ws = None
        self._sums = None
        self._zeros = None

    # same time table as RK4 (the bootstrap steps are rk4_alt steps)
    @staticmethod
    def _time_values(dt):
        return (dt, dt * _one_third, dt * _one_third, dt * _two_thirds)

    def _time_values_tagged(self, dt):
        v = self._time_values(dt)
        return [(v[0], False), (v[1], False), (v[2], True), (v[3], True)]

    def _times(self, t0, dt):
        if self._row is not None:
            return [self._row[j : j + 1] for j in range(4)]
        v = self._time_values(dt)
        t0h = type(dt)(t0.item())
        return [self._tdev(v[0], t0), self._tdev(v[1], t0), self._tdev(t0h + v[2], t0), self._tdev(t0h + v[3], t0)]

    def _has_converged(self, dy_old, dy):
        """linf(|dy_old - dy| / (atol + rtol * max(|dy_old|, |dy|))) < 1 (adams.py:500-505) — one fused norm launch."""
        be = self.backend
        if self._ws is None:
            self._ws = be.new_workspace(dy.device)
            self._sums = be.new_sums(dy.device)
            self._res = torch.zeros(1, dtype=torch.float64, device=dy.device)
        n = dy.numel()
        segs = _hip.make_segments([(0, n)])
        be.error_norm_partial([dy_old, dy], [1.0, -1.0], dy_old, dy, self.rtol, self.atol, segs, _hip.NORM_LINF, self._ws, dt_host=1.0)
        be.norm_finalize(self._ws, 0, self._sums)
        be.norm_result(self._sums, [float(n)], _hip.NORM_LINF, _hip.dtype_code(dy.dtype), self._res)
        return bool(self._res.item() < 1)

    def step(self, t0, t1, y0):
        dt = self._host_dt(t0, t1)
        dtt = self._times(t0, dt)[0]
        f0 = self._f(t0, dtt, y0)
        self.prev_f.appendleft(f0)
        order = min(len(self.prev_f), self.max_order - 1)
        if order < _MIN_ORDER - 1:
            return self.rk4_alt_step_func(t0, t1, y0, f0=f0), f0
        hist = list(self.prev_f)[:order]
        b = [float(c) for c in bashforth(order)]
        if not self.implicit:
            # y1 = fuse(sum_j b_j f_j, dt, y0): one launch
            return self._combine(y0, hist, b, _hip.COMBINE_FUSE, dt, out=self._y1_out), f0
        if self._zeros is None or self._zeros.shape != y0.shape:
            self._zeros = torch.zeros_like(y0)
        dy = self._combine(self._zeros, hist, b, _hip.COMBINE_RK, 1.0)
        m = [float(c) for c in moulton(order + 1)]
        converged = False
        f = None
        for _ in range(self.max_iters):
            dy_old = dy
            f = self._f(t1, dtt, self._combine(y0, [dy], [1.0], _hip.COMBINE_FUSE, dt))
            dy = self._combine(self._zeros, [f] + hist, m, _hip.COMBINE_RK, 1.0)
            converged = self._has_converged(dy_old, dy)
            if converged:
                break
        if not converged:
            warnings.warn("Functional iteration did not converge. Solution may be incorrect.")
            self.prev_f.pop()
        self.prev_f.appendleft(f)
        return self._combine(y0, [dy], [1.0], _hip.COMBINE_FUSE, dt, out=self._y1_out), f0
