"""Delay-equation problem wrapper (reference: paddlexde/xde/base_dde.py:14-127).

``BaseDDE`` evaluates the history at the learned lags ONCE, at construction (``HistoryIndex``: cubic-Hermite spline of
``his`` sampled at ``his_span`` by default; ``interp_method="linear"`` / ``"bez"`` as in the reference, :104-109), and then behaves like an ODE wrapper whose ``move`` calls ``func(y_lags, y0)`` and
whose ``fuse`` is damped: ``y = dy*dt + y0; (dy - 0.001*y)*dt + y0`` (base_dde.py:47-58).  The fixed-step solvers map
that ``fuse`` onto xde_stage_combine with ``damping=0.001``; the history gather is xde_history_gather (value and time
derivative in one pass), with the reference's backward: d loss / d lags = sum over every axis but the lag axis of
``grad_y * derivative`` (base_dde.py:123-127) as ONE reduction launch (xde_lag_grad); the history itself receives no gradient.
"""
import torch

from .. import _hip
from .base_xde import BaseXDE

DDE_DAMPING = 0.001  # `_lambda` of BaseDDE.fuse


class HistoryIndex(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lags, his, his_span, interp_method="cubic"):
        if interp_method not in _hip.HISTORY_METHODS:
            raise NotImplementedError  # (as the reference, base_dde.py:110-111)
        be = _hip.get_backend()
        be.require_device(his)
        dtype = his.dtype if his.dtype in (torch.float32, torch.float64) else torch.float32
        his_c = his.detach().to(dtype).contiguous()
        t_c = his_span.detach().to(device=his.device, dtype=dtype).contiguous()
        lags_c = lags.detach().to(device=his.device, dtype=dtype).contiguous().reshape(-1)
        out_shape = tuple(his_c.shape[:-2]) + (lags_c.numel(), his_c.shape[-1])
        y_lags = torch.empty(out_shape, dtype=dtype, device=his.device)
        derivative_lags = torch.empty_like(y_lags)
        if interp_method == "cubic" or not hasattr(be, "history_gather"):
            if interp_method != "cubic":
                raise NotImplementedError("this backend serves the cubic history spline only")
            be.hermite_gather(y_lags, derivative_lags, his_c, t_c, lags_c)
        else:
            need = 2 if interp_method == "linear" else 4
            if his_c.shape[-2] < need:
                raise ValueError("HistoryIndex(interp_method={!r}) needs at least {} history times".format(interp_method, need))
            be.history_gather(y_lags, derivative_lags, his_c, t_c, lags_c, interp_method)
        ctx.save_for_backward(derivative_lags)
        ctx.lags_shape = tuple(lags.shape)
        ctx.lags_dtype = lags.dtype
        return y_lags

    @staticmethod
    def backward(ctx, grad_y):
        (derivative_lags,) = ctx.saved_tensors
        be = _hip.get_backend()
        if (hasattr(be, "lag_grad") and grad_y.dtype == derivative_lags.dtype and grad_y.shape == derivative_lags.shape
                and derivative_lags.shape[-2] <= 2048):
            # one launch: sum over every axis but the lag axis (xde_lag_grad serves every row length D and alignment; up to 2048 lags)
            grad = be.lag_grad(grad_y.contiguous(), derivative_lags)
        else:
            grad = grad_y * derivative_lags
            dims = [d for d in range(grad.dim()) if d != grad.dim() - 2]  # every axis but the lag axis (reference: [0, 1, 3])
            grad = grad.sum(dim=dims)
        grad = grad.reshape(ctx.lags_shape).to(ctx.lags_dtype)
        return grad, None, None, None


class BaseDDE(BaseXDE):
    def __init__(self, func, y0, t_span, lags, his, his_span, his_processed=False):
        super().__init__(name="DDE", var_nums=1, y0=y0, t_span=t_span)
        self.func = func
        self.lags = lags
        if not his_processed:
            self.y_lags = HistoryIndex.apply(lags, his, his_span)
        else:
            self.y_lags = his
        self.his = his
        self.his_span = his_span
        self.init_y0(y0)

    def init_y0(self, input):
        self.y0 = input

    def handle(self, h, ts):
        pass

    def move(self, t0, dt, y0):
        """base_dde.py:47-53 — the delayed states are an argument of func; t0/dt are not."""
        return self.func(self.y_lags, y0)

    def fuse(self, dy, dt, y0):
        """base_dde.py:55-58"""
        y = dy * dt + y0
        return (dy - DDE_DAMPING * y) * dt + y0

    def call_func(self, t, y0, lags, y_lags):
        return self.func(t, y0, lags, y_lags)

    def init_lags(self):
        pass

    def flatten(self, input):
        return input

    def unflatten(self, input, length):
        return input
