"""odeint_adjoint's backward (Dopri5, 16 output times, MLP dim-4*dim-dim on y^3) over state sizes, with the captured interval solves
(default) and without (XDE_INTERVAL_GRAPH=0: each evaluation of the captured dynamics a replay of its own), alternating, a fresh module
per run; median of 5 backward passes after 2 warm-up calls.  `python3 profiles/tools/adjoint_sizes.py`"""
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


class F(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, 4 * dim), nn.Tanh(), nn.Linear(4 * dim, dim))
        for m in self.net:
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.1 / dim ** 0.5)
                nn.init.zeros_(m.bias)

    def forward(self, t, y):
        return self.net(y * y * y)


def run(batch, dim, interval_graph):
    os.environ["XDE_INTERVAL_GRAPH"] = "1" if interval_graph else "0"
    torch.manual_seed(0)
    func = F(dim).to(dev)
    y0 = (torch.rand(batch, dim) * 2 - 1).to(dev)
    t = torch.linspace(0.0, 1.0, 16).to(dev)
    ms = []
    for call in range(7):
        for p in func.parameters():
            p.grad = None
        pred = odeint_adjoint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm})
        loss = pred.abs().mean()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        if call >= 2:
            ms.append(1e3 * (time.perf_counter() - t0))
    g = float(sum(p.grad.double().pow(2).sum() for p in func.parameters()).sqrt())
    return statistics.median(ms), g


run(1024, 2, True)  # the process's own first-use costs
for batch, dim in ((8192, 2), (8192, 16), (8192, 64), (32768, 64)):
    rows = []
    for rep in range(2):
        a, ga = run(batch, dim, True)
        b, gb = run(batch, dim, False)
        rows.append("%.2f vs %.2f ms" % (a, b))
        assert ga == gb, (ga, gb)
    print("batch %6d x dim %3d (%8d state elements): captured interval solves vs per-evaluation replays: %s; same gradients" % (
        batch, dim, batch * dim, "; ".join(rows)), flush=True)
