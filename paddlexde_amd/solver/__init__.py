"""Solver classes, passed as objects to ``odeint(..., solver=Cls)`` (reference: paddlexde/solver/__init__.py:1-6)."""
from .adaptive_solver import AdaptiveHeun, Bosh3, Dopri5, Dopri8, Fehlberg2  # noqa: F401
from .base_adaptive_solver import AdaptiveSolver  # noqa: F401
from .base_adaptive_solver_rk import AdaptiveRKSolver  # noqa: F401
from .base_fixed_solver import FixedSolver  # noqa: F401
from .fixed_solver import RK4, AdamsBashforthMoulton, Euler, Midpoint  # noqa: F401
