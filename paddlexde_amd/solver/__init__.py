"""Solver classes, passed as objects to ``odeint(..., solver=Cls)`` (the reference's paddlexde/solver/__init__.py:1-6
exports the same names, minus the SciPy wrapper which is out of scope)."""
from . import adaptive_solver as _ad
from . import fixed_solver as _fx
from .base_adaptive_solver import AdaptiveSolver
from .base_adaptive_solver_rk import AdaptiveRKSolver
from .base_fixed_solver import FixedSolver

AdaptiveHeun, Bosh3, Dopri5, Dopri8, Fehlberg2 = _ad.AdaptiveHeun, _ad.Bosh3, _ad.Dopri5, _ad.Dopri8, _ad.Fehlberg2
RK4, Euler, Midpoint, AdamsBashforthMoulton = _fx.RK4, _fx.Euler, _fx.Midpoint, _fx.AdamsBashforthMoulton

__all__ = ["AdaptiveSolver", "AdaptiveRKSolver", "FixedSolver", "AdaptiveHeun", "Bosh3", "Dopri5", "Dopri8", "Fehlberg2", "RK4", "Euler",
           "Midpoint", "AdamsBashforthMoulton"]
