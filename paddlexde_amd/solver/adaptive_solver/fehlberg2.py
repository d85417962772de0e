"""Fehlberg 2(1) (reference: paddlexde/solver/adaptive_solver/fehlberg2.py:5-21)."""
from ..base_adaptive_solver_rk import AdaptiveRKSolver, _ButcherTableau

_FEHLBERG2_TABLEAU = _ButcherTableau(
    alpha=[1 / 2, 1.0],
    beta=[[1 / 2], [1 / 256, 255 / 256]],
    c_sol=[1 / 512, 255 / 256, 1 / 512],
    c_error=[-1 / 512, 0, 1 / 512],
)
_FE_C_MID = [0.0, 0.5, 0.0]


class Fehlberg2(AdaptiveRKSolver):
    order = 2
    tableau = _FEHLBERG2_TABLEAU
    mid = _FE_C_MID
