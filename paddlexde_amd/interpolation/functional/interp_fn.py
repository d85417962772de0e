"""The fixed solvers' per-step interpolants, restated for callers (reference: paddlexde/interpolation/functional/interp_fn.py:4-20).

OFF the hot path: ``FixedSolver.integrate`` evaluates its interpolant at ``t == t1`` only (its grid IS ``t_span``; the reference's
``step_size`` / ``grid_constructor`` sub-stepping never worked), where ``linear_interp`` returns ``y1`` and the Hermite cubic
reduces to ``y1`` — the solvers therefore never launch these (solver/base_fixed_solver.py).  They are here, as plain framework ops in the
reference's op order, for scripts that call them directly."""
import torch


def _same_time(a, b):
    r = a == b  # (tensors on one device, or a tensor and a number: the reference's `if t == t0`)
    return bool(r.all()) if torch.is_tensor(r) else bool(r)


def linear_interp(t0, t1, y0, y1, t):
    """interp_fn.py:4-10"""
    if _same_time(t, t0):
        return y0
    if _same_time(t, t1):
        return y1
    slope = (t - t0) / (t1 - t0)
    return y0 + slope * (y1 - y0)


def cubic_hermite_interp(t0, y0, dy0, t1, y1, dy1, t):
    """interp_fn.py:13-20"""
    h = (t - t0) / (t1 - t0)
    h00 = (1 + 2 * h) * (1 - h) * (1 - h)
    h10 = h * (1 - h) * (1 - h)
    h01 = h * h * (3 - 2 * h)
    h11 = h * h * (h - 1)
    dt = t1 - t0
    return h00 * y0 + h10 * dt * dy0 + h01 * y1 + h11 * dt * dy1
