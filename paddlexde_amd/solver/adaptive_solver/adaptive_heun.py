"""Adaptive Heun 2(1) (reference: paddlexde/solver/adaptive_solver/adaptive_heun.py:5-26)."""
from ..base_adaptive_solver_rk import AdaptiveRKSolver, _ButcherTableau

_ADAPTIVE_HEUN_TABLEAU = _ButcherTableau(
    alpha=[1.0],
    beta=[[1.0]],
    c_sol=[0.5, 0.5],
    c_error=[0.5, -0.5],
)
_AH_C_MID = [0.5, 0.0]


class AdaptiveHeun(AdaptiveRKSolver):
    order = 2
    tableau = _ADAPTIVE_HEUN_TABLEAU
    mid = _AH_C_MID
