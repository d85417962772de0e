"""Dormand-Prince 8(7) pair with 13 stages — "RK8(7)13M" of P.J. Prince & J.R. Dormand, "High order embedded
Runge-Kutta formulae", J. Comput. Appl. Math. 7 (1981) — the tableau behind the reference's ``Dopri8``
(paddlexde/solver/adaptive_solver/dopri8.py:5-252: nodes ``A``, rows ``B``, ``C_sol``, ``C_err`` and the half-step
weights ``C_mid`` obtained from the continuous extension evaluated at h = 1/2).

The coefficients are stored as exact IEEE-754 doubles (hex), i.e. the values the reference's rational literals
evaluate to, so both implementations feed identical numbers to the stage combines.  They are validated
independently of the reference by tests/test_oracle_pinning.py: row sums equal the nodes, sum(c_sol) = 1,
sum(c_error) = 0, sum(mid) = 1/2, and measured order of convergence 8.
"""
from ..base_adaptive_solver_rk import AdaptiveRKSolver, _ButcherTableau

_H = float.fromhex

_ALPHA = [_H("0x1.c71c71c71c71cp-5"), _H("0x1.5555555555555p-4"), _H("0x1.0000000000000p-3"), _H("0x1.4000000000000p-2"), _H("0x1.8000000000000p-2"), _H("0x1.2e147ae147ae1p-3"), _H("0x1.dc28f5c28f5c3p-2"), _H("0x1.21360b60a7776p-1"), _H("0x1.4cccccccccccdp-1"), _H("0x1.d96c8c31039dbp-1"), _H("0x1.0000000000000p+0"), _H("0x1.0000000000000p+0"), _H("0x1.0000000000000p+0")]

_BETA = [
    [_H("0x1.c71c71c71c71cp-5")],
    [_H("0x1.5555555555555p-6"), _H("0x1.0000000000000p-4")],
    [_H("0x1.0000000000000p-5"), 0.0, _H("0x1.8000000000000p-4")],
    [_H("0x1.4000000000000p-2"), 0.0, _H("-0x1.2c00000000000p+0"), _H("0x1.2c00000000000p+0")],
    [_H("0x1.3333333333333p-5"), 0.0, 0.0, _H("0x1.8000000000000p-3"), _H("0x1.3333333333333p-3")],
    [_H("0x1.887ad701404acp-5"), 0.0, 0.0, _H("0x1.cbc54e6660e1dp-4"), _H("-0x1.a1e28caf3b65cp-6"), _H("0x1.a4f6f83ae9731p-7")],
    [_H("0x1.152f31366e4d8p-6"), 0.0, 0.0, _H("0x1.8d28195fa13c2p-2"), _H("0x1.26ba035d10b6dp-5"), _H("0x1.93651ea2bd3c4p-3"), _H("-0x1.61b7ccdaf2f38p-3")],
    [_H("0x1.1b04260f85fe2p-4"), 0.0, 0.0, _H("-0x1.44bc269b358ddp-1"), _H("-0x1.4a21f44e45fd3p-3"), _H("0x1.1bf4b185a5c0bp-3"), _H("0x1.e1c165324ef0ap-1"), _H("0x1.b16e62e7158fcp-3")],
    [_H("0x1.77ecbb1301621p-3"), 0.0, 0.0, _H("-0x1.3c0097b3c5a32p+1"), _H("-0x1.2a471c23b2d29p-2"), _H("-0x1.b1bbe5082a5c1p-6"), _H("0x1.6c85fb0a3e9bfp+1"), _H("0x1.20240028afd67p-2"), _H("0x1.fadbee9f5b0f4p-4")],
    [_H("-0x1.372614b1764cfp+0"), 0.0, 0.0, _H("0x1.0ac3014df3e48p+4"), _H("0x1.d4dc1ce9424acp-1"), _H("-0x1.839f6df39ea9cp+2"), _H("-0x1.000ea32f607acp+4"), _H("0x1.db2d7daa814a6p+3"), _H("-0x1.abe3f2cbe1d36p+3"), _H("0x1.489672d167d27p+2")],
    [_H("0x1.0912d609427e0p-2"), 0.0, 0.0, _H("-0x1.31912cd3f9270p+2"), _H("-0x1.bd8905e38fcd7p-2"), _H("-0x1.865578467943fp+1"), _H("0x1.64fca455cea0cp+2"), _H("0x1.89f9250f88c23p+2"), _H("-0x1.43f985843ddf3p+2"), _H("0x1.18d292a5d3212p+1"), _H("0x1.13b7d81af1344p-3")],
    [_H("0x1.a5153af7727fdp-1"), 0.0, 0.0, _H("-0x1.7513d9f0583c5p+3"), _H("-0x1.83e70bcbd3e65p-1"), _H("0x1.6d8df236b4d37p-1"), _H("0x1.826cbfaa51862p+3"), _H("-0x1.10572243a9883p+1"), _H("0x1.fd7b8854e12f5p+0"), _H("-0x1.dfd195e96a441p-3"), _H("0x1.683d837559248p-3"), 0.0],
    [_H("0x1.55fed5a492d16p-5"), 0.0, 0.0, 0.0, 0.0, _H("-0x1.c643f63bea075p-5"), _H("0x1.ea1cd5438b4f0p-3"), _H("0x1.68328ceaf3204p-1"), _H("-0x1.84ff364c4f34cp-1"), _H("0x1.5235514d8405cp-1"), _H("0x1.43f7cc8023f22p-3"), _H("-0x1.e7a5f94e7938dp-3"), _H("0x1.0000000000000p-2")],
]

_C_SOL = [_H("0x1.55fed5a492d16p-5"), 0.0, 0.0, 0.0, 0.0, _H("-0x1.c643f63bea075p-5"), _H("0x1.ea1cd5438b4f0p-3"), _H("0x1.68328ceaf3204p-1"), _H("-0x1.84ff364c4f34cp-1"), _H("0x1.5235514d8405cp-1"), _H("0x1.43f7cc8023f22p-3"), _H("-0x1.e7a5f94e7938dp-3"), _H("0x1.0000000000000p-2"), 0.0]

_C_ERR = [_H("0x1.8f950374a4f36p-7"), 0.0, 0.0, 0.0, 0.0, _H("0x1.8bdad591ce63bp-1"), _H("-0x1.269e126744244p-4"), _H("-0x1.c38aa8c018342p+0"), _H("0x1.c984c3155396ap+0"), _H("-0x1.90e37b7ca45bcp-1"), _H("0x1.42a64f4ea6f18p-4"), _H("-0x1.2155d4d4bf749p-2"), _H("0x1.0000000000000p-2"), 0.0]

_C_MID = [_H("0x1.5020e16364620p-5"), 0.0, 0.0, 0.0, 0.0, _H("0x1.a1d68e083d340p-5"), _H("0x1.e940c4c984690p-3"), _H("0x1.3d1e97bb5e6bap-2"), _H("-0x1.bce46aedf7320p-3"), _H("0x1.40db364171ec0p-4"), _H("-0x1.e7f6ebaa66900p-10"), _H("0x1.cf86a28a4b200p-8"), _H("0x1.c4c0f4be86800p-8"), _H("-0x1.c574f30dd2600p-7")]

_DOPRI8_TABLEAU = _ButcherTableau(alpha=_ALPHA, beta=_BETA, c_sol=_C_SOL, c_error=_C_ERR)


class Dopri8(AdaptiveRKSolver):
    order = 8
    tableau = _DOPRI8_TABLEAU
    mid = _C_MID
