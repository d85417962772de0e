"""Adaptive-step solver base: output loop and initial-step heuristic.

Reference: paddlexde/solver/base_adaptive_solver.py:6-72.  ``integrate`` keeps the reference's contract
(``solution[T, *y0.shape]``, ``solution[0] = y0``, ``t_span.astype(dtype)``); the norms inside
``select_initial_step`` (``norm(y0/scale)``, ``norm(f0/scale)``, ``norm((f1-f0)/scale)``, :50-64) are
xde_scaled_norm_partial launches, the Euler probe ``y0 + h0*f0`` (:59) is one xde_stage_combine.
"""
import abc
import warnings

import numpy as np
import torch

from .. import _hip
from ._common import as_operand, direction_of, np_dtype, t_span_to_host


class AdaptiveSolver(metaclass=abc.ABCMeta):
    def __init__(self, xde, dtype, y0, norm, **unused_kwargs):
        self.dtype = dtype
        self.y0 = y0
        self.norm = norm

        self.xde = xde
        self.move = self.xde.move
        self.fuse = self.xde.fuse

    @abc.abstractmethod
    def _before_integrate(self, t_span):
        raise NotImplementedError

    @abc.abstractmethod
    def step(self, next_t):
        raise NotImplementedError

    @abc.abstractmethod
    def _run(self, solution):
        raise NotImplementedError

    def integrate(self, t_span):
        """base_adaptive_solver.py:24-31.  The per-output ``step(t_span[i])`` loop of the reference is
        driven by the device controller (ctrl.next_out) inside ``_run``."""
        y0 = self.y0
        self.backend.require_device(y0)
        if torch.is_grad_enabled():
            func = getattr(self.xde, "func", None)
            trainable = isinstance(func, torch.nn.Module) and any(p.requires_grad for p in func.parameters())
            if y0.requires_grad or trainable:
                warnings.warn(
                    "paddlexde_amd: adaptive solvers do not record an autograd graph (the step kernels read dt from device "
                    "memory); the result is detached. Use odeint_adjoint for gradients, or torch.no_grad() to silence this.",
                    stacklevel=3,
                )
        self.y0 = y0 = as_operand(y0.detach())
        t_host = t_span_to_host(t_span, np_dtype(self.dtype))  # t_span.astype(self.dtype)
        solution = torch.empty((len(t_host),) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
        solution[0] = y0
        if len(t_host) < 2 or y0.numel() == 0:  # (an empty state has nothing to integrate: every row is the empty tensor)
            return solution
        # The reference walks the output times in order and evaluates its interpolant on the step that has just reached each of
        # them; a time that lies BEHIND the previous one is outside that step and fails `interp_evaluate`'s assertion
        # (utils/ode_utils.py:65-67).  The device controller would extrapolate the last step's quartic instead — so the same
        # condition is checked here, before anything is launched (same exception type and message shape).
        # Stricter than the reference in one corner, deliberately: an out-of-order time that still lies inside the step the
        # reference happens to be in passes its assertion; which times those are depends on the step sequence, so here EVERY
        # out-of-order time is refused.  (Raised explicitly: an `assert` statement would vanish under `python -O`.)
        d0 = direction_of(t_host)
        for i in range(2, len(t_host)):
            if not d0 * t_host[i] >= d0 * t_host[i - 1]:
                raise AssertionError("invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}".format(
                    t_host[i - 1], t_host[i], t_host[i - 1]))
        try:
            self._before_integrate(t_host)
            # rows whose time equals the start time need no step (`while next_t > rk_state.t1` is false at once,
            # base_adaptive_solver_rk.py:119); the device controller starts its row counter after them
            d = direction_of(t_host)
            e = 1
            while e < len(t_host) and d * t_host[e] <= d * t_host[0]:
                solution[e] = y0
                e += 1
            if e < len(t_host):
                self._run(solution)
        finally:
            self._after_integrate()
        return solution

    def _after_integrate(self):
        """Hook: the solve has ended (normally or by an assertion); solvers are single-use, as in the reference."""

    # base_adaptive_solver.py:33-72
    def select_initial_step(self, t0, y0, order, rtol, atol, f0=None):
        """Hairer's initial-step heuristic.  Host arithmetic is done in the state dtype with the
        reference's op order; the three norms are device reductions (one 3-scalar read in total)."""
        be = self.backend
        yt = np_dtype(y0.dtype)
        tt = np_dtype(self.dtype)
        t0h = tt(t0)
        if f0 is None:
            f0 = self._eval(self._scalar_t(t0h, self.dtype), y0)
        d = self._scaled_norms([(y0, None), (f0, None)], y0, rtol, atol)
        d0, d1 = yt(abs(d[0])), yt(abs(d[1]))
        if d0 < 1e-5 or d1 < 1e-5:
            h0 = yt(1e-6)
        else:
            h0 = 0.01 * d0 / d1
        h0 = abs(h0)
        # the Euler probe goes in the direction of integration (the reference integrates forward only; in reverse time this
        # is its heuristic on the flipped problem t -> -t, f -> -f)
        hs0 = -h0 if getattr(self, "_direction", 1) < 0 else h0
        y1 = torch.empty_like(y0)
        be.stage_combine(y1, y0, [f0], [1.0], _hip.COMBINE_FUSE, dt_host=float(hs0))  # fuse(f0, h0, y0)
        f1 = self._eval(self._scalar_t(t0h + hs0, torch.promote_types(self.dtype, y0.dtype)), y1)
        (n2,) = self._scaled_norms([(f1, f0)], y0, rtol, atol)
        with np.errstate(all="ignore"):
            d2 = abs(yt(n2) / h0)
            if d1 <= 1e-15 and d2 <= 1e-15:
                h1 = max(yt(1e-6), h0 * 1e-3)
            else:
                h1 = (0.01 / max(d1, d2)) ** (1.0 / float(order + 1))
            h1 = abs(h1)
            return tt(np.fmin(100.0 * h0, h1))
