"""Autograd node around xde_stage_combine for discretise-then-optimise training.

The reference trains by back-propagating through its eager solver ops (example/ode_demo.py:51-53:
``pred_y = odeint(func, batch_y0, t_span, solver=RK4); loss.backward()``).  Here the forward is one combine launch
and the backward one fan-out launch (every input gradient is a scalar multiple of the output gradient);
``func`` itself is differentiated by the framework as usual.
"""
import torch
from torch.autograd.function import once_differentiable

from .. import _hip


class CombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, backend, coef, mode, scale, dt, damp, y0, *ks):
        out = torch.empty_like(y0)
        backend.stage_combine(out, y0.detach(), [k.detach() for k in ks], coef, mode, scale=scale, dt_host=float(dt),
                              damping=float(damp))
        ctx.backend = backend
        ctx.meta = (tuple(float(c) for c in coef), mode, float(scale), float(dt), float(damp))
        ctx.needs = (y0.requires_grad,) + tuple(k.requires_grad for k in ks)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        coef, mode, scale, dt, damp = ctx.meta
        # damped fuse: (dy - damp*(dy*dt + y0))*dt + y0 = dy*dt*(1 - damp*dt) + y0*(1 - damp*dt)
        g_damp = 1.0 - damp * dt
        if mode == _hip.COMBINE_RK or mode == _hip.COMBINE_FUSE:
            # out = y0 + sum_j k_j (c_j dt)   |   out = fuse(sum_j c_j k_j, dt, y0)
            fy0 = g_damp
            fk = [c * dt * g_damp for c in coef]
        else:
            # out = scale * sum_j w_j fuse(k_j, dt, y0)
            fy0 = scale * sum(coef) * g_damp
            fk = [scale * w * dt * g_damp for w in coef]
        g = g.contiguous()
        if g.data_ptr() % 16:
            g = g.clone()
        factors = [fy0] + fk
        outs, todo_o, todo_f = [], [], []
        for need, f in zip(ctx.needs, factors):
            if not need:
                outs.append(None)
            elif f == 1.0:
                outs.append(g)
            else:
                o = torch.empty_like(g)
                outs.append(o)
                todo_o.append(o)
                todo_f.append(f)
        if todo_o:
            ctx.backend.scale_fanout(todo_o, g, todo_f)
        return (None, None, None, None, None, None) + tuple(outs)
