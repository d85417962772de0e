"""Bogacki-Shampine 3(2) (reference: paddlexde/solver/adaptive_solver/bosh3.py:5-24)."""
from ..base_adaptive_solver_rk import AdaptiveRKSolver, _ButcherTableau

_BOGACKI_SHAMPINE_TABLEAU = _ButcherTableau(
    alpha=[1 / 2, 3 / 4, 1.0],
    beta=[[1 / 2], [0.0, 3 / 4], [2 / 9, 1 / 3, 4 / 9]],
    c_sol=[2 / 9, 1 / 3, 4 / 9, 0.0],
    c_error=[2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8],
)
_BS_C_MID = [0.0, 0.5, 0.0, 0.0]


class Bosh3(AdaptiveRKSolver):
    order = 3
    tableau = _BOGACKI_SHAMPINE_TABLEAU
    mid = _BS_C_MID
