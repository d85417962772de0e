"""odeint_adjoint's backward where the speculative ("lag") pipeline runs its interval solves — states above 8 MiB per operand, eager
dynamics — with and without the wait for every interval's first verdict (XDE_SHORT_SOLVES=0 = before), alternating; median of 5 passes.
`python3 profiles/tools/adjoint_lag.py`"""
import os
import statistics
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddlexde_amd import Dopri5, odeint_adjoint  # noqa: E402
from paddlexde_amd.utils import _rms_norm  # noqa: E402

dev = torch.device("cuda:0")


class Lin(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.A = nn.Parameter(-0.5 * torch.eye(dim) + 0.05 * torch.randn(dim, dim, generator=torch.Generator().manual_seed(1)))

    def forward(self, t, y):
        return torch.tanh(y @ self.A)


def run(batch, dim, n_out, wait, t_end=1.0):
    os.environ["XDE_SHORT_SOLVES"] = "1" if wait else "0"
    func = Lin(dim).to(dev)
    y0 = (torch.rand(batch, dim, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    t = torch.linspace(0.0, t_end, n_out).to(dev)
    ms = []
    for call in range(6):
        func.A.grad = None
        pred = odeint_adjoint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm})
        loss = pred.abs().mean()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        if call >= 1:
            ms.append(1e3 * (time.perf_counter() - t0))
    return statistics.median(ms), float(func.A.grad.double().norm())


run(4096, 16, 4, True)
for batch, dim, n_out, t_end in ((65536, 128, 16, 1.0), (65536, 128, 16, 0.05), (65536, 128, 32, 0.05), (524288, 64, 16, 0.05)):
    rows = []
    for rep in range(2):
        a, ga = run(batch, dim, n_out, True, t_end)
        b, gb = run(batch, dim, n_out, False, t_end)
        assert ga == gb, (ga, gb)
        rows.append("%.1f vs %.1f ms" % (a, b))
    print("batch %6d x dim %3d, %2d output times over [0, %g]: first verdict awaited vs not: %s; same gradient" % (
        batch, dim, n_out, t_end, "; ".join(rows)), flush=True)
