"""Shared pieces of the end-to-end cases (tests/_forward_cases.py, _options_cases.py, _pipeline_cases.py, _adjoint_cases.py): imports,
the solver tables, the funcs and fixtures more than one of them uses."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import xde_oracle as O
from paddlexde_amd import RK4, AdamsBashforthMoulton, AdaptiveHeun, Bosh3, Dopri5, Dopri8, Euler, Fehlberg2, Midpoint, _hip, odeint, odeint_adjoint
from paddlexde_amd.utils import _linf_norm, _rms_norm

from . import problems as P


def _blocks(n):
    """Number of seeded blocks of a randomised sweep; XDE_SWEEP_SCALE=k runs k times as many (a soak, not the default)."""
    import os

    return n * int(os.environ.get("XDE_SWEEP_SCALE", "1"))


FIXED = {"euler": Euler, "midpoint": Midpoint, "rk4": RK4, "adams": AdamsBashforthMoulton}
ADAPTIVE = {"dopri5": Dopri5, "bosh3": Bosh3, "fehlberg2": Fehlberg2, "adaptive_heun": AdaptiveHeun, "dopri8": Dopri8}


class ConstantLayer(nn.Module):
    """The reference's ConstantXDE as the Layer it is there (tests/testing_utils.py:8-26: parameters a = 0.2, b = 3.0,
    `a + (y - (a t + b))^5`) — what `odeint_adjoint` needs to find adjoint parameters on."""

    def __init__(self):
        super().__init__()
        self.a = nn.Parameter(torch.tensor([P.Constant.a], dtype=torch.float32))
        self.b = nn.Parameter(torch.tensor([P.Constant.b], dtype=torch.float32))

    def forward(self, t, y):
        d = y - (self.a * t + self.b).to(y.dtype)
        return self.a + d * d * d * d * d


# ----------------------------------------------------------------------------------------------
# config 2 at oracle-sized batches: linear ODE, Dopri5, rtol 1e-5 / atol 1e-7
# ----------------------------------------------------------------------------------------------
def _linear(B, D, dtype):
    A = P.skew_matrix(D).to(dtype)
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(0)).to(dtype)
    return A, y0


# ----------------------------------------------------------------------------------------------
# config 3: neural-ODE adjoint (2-layer MLP on y**3), gradients for 252 params
# ----------------------------------------------------------------------------------------------
class ODEFunc(nn.Module):
    """example/ode_demo.py:17-33: Linear(2,50) -> Tanh -> Linear(50,2) on y**3, weights 0.1*randn, biases 0."""

    def __init__(self, dtype):
        super().__init__()
        g = torch.Generator().manual_seed(42)
        self.W1 = nn.Parameter(0.1 * torch.randn(2, 50, generator=g, dtype=dtype))
        self.b1 = nn.Parameter(torch.zeros(50, dtype=dtype))
        self.W2 = nn.Parameter(0.1 * torch.randn(50, 2, generator=g, dtype=dtype))
        self.b2 = nn.Parameter(torch.zeros(2, dtype=dtype))

    def forward(self, t, y):
        return torch.tanh((y * y * y) @ self.W1 + self.b1) @ self.W2 + self.b2


def _mlp_numpy(m):
    W1, b1, W2, b2 = [p.detach().cpu().numpy() for p in m.parameters()]

    def fn(t, y):
        return np.tanh((y * y * y) @ W1 + b1) @ W2 + b2

    def vjp(t, y, cot):
        u = y * y * y
        a = np.tanh(u @ W1 + b1)
        gh = (cot @ W2.T) * (1 - a * a)
        return (gh @ W1.T) * 3 * y * y, [u.T @ gh, gh.sum(0), a.T @ cot, cot.sum(0)]

    return fn, vjp, [W1, b1, W2, b2]


def _mlp_foreign(m):
    """The spiral MLP (ODEFunc) as a layer of the stand-in framework, and its HAND-WRITTEN vector-Jacobian product — no autograd of any
    framework.  The arithmetic is the chain rule written out with the framework's primitives in the order a reverse sweep meets them
    (`c @ W2^T`, `a^T @ c`, tanh's derivative, the cube's three product-rule terms added left to right): the sequence of kernels
    torch's autograd runs for ODEFunc, so that the two routes can be compared bit for bit."""
    W1, b1, W2, b2 = [p.detach() for p in m.parameters()]
    F = P.Foreign

    def func(t, y):
        assert isinstance(t, F) and isinstance(y, F)
        y_ = y.raw
        return F(torch.tanh((y_ * y_ * y_) @ W1 + b1) @ W2 + b2)

    def vjp(t, y, cotangent):
        assert isinstance(t, F) and isinstance(y, F) and isinstance(cotangent, F)
        y_, c = y.raw, cotangent.raw
        yy = y_ * y_
        u = yy * y_
        a = torch.tanh(u @ W1 + b1)
        f = a @ W2 + b2
        g_b2 = c.sum(0, keepdim=True).view(b2.shape)
        g_a = c.mm(W2.t())
        g_W2 = a.t().mm(c)
        g_h = torch.ops.aten.tanh_backward(g_a, a)  # g_a (1 - a^2): the framework's fused primitive
        g_b1 = g_h.sum(0, keepdim=True).view(b1.shape)
        g_u = g_h.mm(W1.t())
        g_W1 = u.t().mm(g_h)
        g_yy = g_u * y_
        g_y = g_u * yy + g_yy * y_ + g_yy * y_
        return F(f), None, F(g_y), F(g_W1), F(g_b1), F(g_W2), F(g_b2)  # (autonomous: no time gradient)

    return func, vjp, [F(p) for p in (W1, b1, W2, b2)]


# ----------------------------------------------------------------------------------------------
# committed golden vectors (tests/golden/*.npz, generated from the oracle by tests/golden/make_golden.py)
# ----------------------------------------------------------------------------------------------
def _golden(name):
    import os

    return np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))


class _SmallMLP(nn.Module):
    def __init__(self, d, h, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.W1 = nn.Parameter(0.3 * torch.randn(d, h, generator=g, dtype=torch.float64))
        self.b1 = nn.Parameter(0.1 * torch.randn(h, generator=g, dtype=torch.float64))
        self.W2 = nn.Parameter(0.3 * torch.randn(h, d, generator=g, dtype=torch.float64))
        self.b2 = nn.Parameter(0.1 * torch.randn(d, generator=g, dtype=torch.float64))

    def forward(self, t, y):
        return torch.tanh((y * y * y) @ self.W1 + self.b1) @ self.W2 + self.b2 + 0.1 * t


# ----------------------------------------------------------------------------------------------
# the adjoint of a module with MANY parameter tensors (> XDE_MAX_SEG norm segments)
# ----------------------------------------------------------------------------------------------
class DeepFunc(nn.Module):
    """n_layers Linear layers with tanh between them: 2 * n_layers parameter tensors."""

    def __init__(self, n_layers, width, dtype):
        super().__init__()
        g = torch.Generator().manual_seed(7)
        dims = [2] + [width] * (n_layers - 1) + [2]
        self.layers = nn.ModuleList([nn.Linear(a, b, dtype=dtype) for a, b in zip(dims[:-1], dims[1:])])
        for lin in self.layers:
            lin.weight.data = 0.3 * torch.randn(lin.weight.shape, generator=g, dtype=dtype)
            lin.bias.data = 0.05 * torch.randn(lin.bias.shape, generator=g, dtype=dtype)

    def forward(self, t, y):
        for i, lin in enumerate(self.layers):
            y = lin(y)
            if i + 1 < len(self.layers):
                y = torch.tanh(y)
        return y


def P_rms():
    from paddlexde_amd.utils import _rms_norm

    return _rms_norm
