"""Embedded Runge-Kutta pairs driven by the generic stepper (``AdaptiveRKSolver``) — one table per pair.

Each entry: order, nodes (alpha), stage rows (beta), solution weights (c_sol), error weights (c_error) and the
half-step weights of the dense output (mid).  Sources: Dormand & Prince 5(4) with Shampine's mid-point weights
(reference: paddlexde/solver/adaptive_solver/dopri5.py:5-61), Bogacki & Shampine 3(2) (bosh3.py:5-24), Fehlberg 2(1)
(fehlberg2.py:5-21), Heun-Euler 2(1) (adaptive_heun.py:5-26).  Coefficients are Python floats (double); the kernels
round them to the state dtype at use, which is what the reference's ``.astype(y0.dtype)`` does
(solver/base_adaptive_solver_rk.py:73-79).  Dopri8 lives in dopri8.py (hex doubles).
"""
from fractions import Fraction as F

from ..base_adaptive_solver_rk import AdaptiveRKSolver, _ButcherTableau


def _f(*xs):
    return [float(x) for x in xs]


PAIRS = {
    "Dopri5": dict(
        order=5,
        alpha=_f(F(1, 5), F(3, 10), F(4, 5), F(8, 9), 1, 1),
        beta=[
            _f(F(1, 5)),
            _f(F(3, 40), F(9, 40)),
            _f(F(44, 45), F(-56, 15), F(32, 9)),
            _f(F(19372, 6561), F(-25360, 2187), F(64448, 6561), F(-212, 729)),
            _f(F(9017, 3168), F(-355, 33), F(46732, 5247), F(49, 176), F(-5103, 18656)),
            _f(F(35, 384), 0, F(500, 1113), F(125, 192), F(-2187, 6784), F(11, 84)),
        ],
        c_sol=_f(F(35, 384), 0, F(500, 1113), F(125, 192), F(-2187, 6784), F(11, 84), 0),
        # 5th-order weights minus the embedded 4th-order ones, evaluated in double like the reference does
        c_error=[
            35 / 384 - 1951 / 21600,
            0.0,
            500 / 1113 - 22642 / 50085,
            125 / 192 - 451 / 720,
            -2187 / 6784 - -12231 / 42400,
            11 / 84 - 649 / 6300,
            -1.0 / 60.0,
        ],
        mid=[
            6025192743 / 30085553152 / 2,
            0.0,
            51252292925 / 65400821598 / 2,
            -2691868925 / 45128329728 / 2,
            187940372067 / 1594534317056 / 2,
            -1776094331 / 19743644256 / 2,
            11237099 / 235043384 / 2,
        ],
    ),
    "Bosh3": dict(
        order=3,
        alpha=_f(F(1, 2), F(3, 4), 1),
        beta=[_f(F(1, 2)), _f(0, F(3, 4)), [2 / 9, 1 / 3, 4 / 9]],
        c_sol=[2 / 9, 1 / 3, 4 / 9, 0.0],
        c_error=[2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8],
        mid=_f(0, F(1, 2), 0, 0),
    ),
    "Fehlberg2": dict(
        order=2,
        alpha=_f(F(1, 2), 1),
        beta=[_f(F(1, 2)), _f(F(1, 256), F(255, 256))],
        c_sol=_f(F(1, 512), F(255, 256), F(1, 512)),
        c_error=_f(F(-1, 512), 0, F(1, 512)),
        mid=_f(0, F(1, 2), 0),
    ),
    "AdaptiveHeun": dict(
        order=2,
        alpha=_f(1),
        beta=[_f(1)],
        c_sol=_f(F(1, 2), F(1, 2)),
        c_error=_f(F(1, 2), F(-1, 2)),
        mid=_f(F(1, 2), 0),
    ),
}


def _make(name):
    spec = PAIRS[name]
    tab = _ButcherTableau(alpha=spec["alpha"], beta=spec["beta"], c_sol=spec["c_sol"], c_error=spec["c_error"])
    return type(name, (AdaptiveRKSolver,), {"order": spec["order"], "tableau": tab, "mid": spec["mid"], "__module__": __name__,
                                             "__doc__": "Embedded pair {} on the generic HIP stepper.".format(name)})


Dopri5 = _make("Dopri5")
Bosh3 = _make("Bosh3")
Fehlberg2 = _make("Fehlberg2")
AdaptiveHeun = _make("AdaptiveHeun")
