"""``ddeint`` — delay differential equations (reference: paddlexde/functional/ddeint.py:9-47).

Same signature and return value ``(solution, y_lags)`` as the reference.  The delayed states are gathered once by
xde_hermite_gather (``BaseDDE`` / ``HistoryIndex``), the integration is the fixed-step loop on the combine kernel
with the damped ``fuse``.  Gradients (the reference trains D3STN by back-propagating through this call) flow through
the combine autograd node into ``func``'s parameters and, through ``HistoryIndex.backward``, into the lags.
"""
import torch

from ..utils.ode_utils import _rms_norm
from ..xde.base_dde import BaseDDE


def ddeint(
    func,
    y0,
    t_span,
    lags,
    his,
    his_span,
    solver,
    his_processed=False,
    rtol=1e-7,
    atol=1e-9,
    options: object = {"norm": _rms_norm},
    fixed_solver_interp="linear",
):
    if not torch.is_tensor(t_span):
        t_span = torch.as_tensor(t_span)
    xde = BaseDDE(func, y0=y0, t_span=t_span, lags=lags, his=his, his_span=his_span, his_processed=his_processed)

    s = solver(xde=xde, y0=xde.y0, rtol=rtol, atol=atol, interp=fixed_solver_interp, **options)
    solution = s.integrate(t_span)

    return solution, xde.y_lags
