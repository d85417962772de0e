#!/usr/bin/env python3
"""cProfile of config 3's adjoint backward (host side): python3 profiles/tools/c3_host_profile.py"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from paddlexde_amd import Dopri5, odeint_adjoint  # noqa: E402
from paddlexde_amd.utils import _rms_norm  # noqa: E402


class ODEFunc(nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(42)
        self.net = nn.Sequential(nn.Linear(2, 50), nn.Tanh(), nn.Linear(50, 2))
        for m in self.net:
            if isinstance(m, nn.Linear):
                with torch.no_grad():
                    m.weight.copy_(0.1 * torch.randn(m.weight.shape, generator=g))
                    m.bias.zero_()

    def forward(self, t, y):
        return self.net(y**3)


dev = torch.device("cuda", 0)
func = ODEFunc().to(dev)
y0 = (torch.rand(8192, 2, generator=torch.Generator().manual_seed(0)) * 4 - 2).to(dev)
t = torch.linspace(0.0, 25.0, 1000)[:32].to(dev)


def run(profile=None):
    for p in func.parameters():
        p.grad = None
    pred = odeint_adjoint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm})
    loss = pred.abs().mean()
    torch.cuda.synchronize()
    if profile is not None:
        profile.enable()
    loss.backward()
    torch.cuda.synchronize()
    if profile is not None:
        profile.disable()


for _ in range(3):
    run()
import time  # noqa: E402

t0 = time.perf_counter()
run()
print("forward+backward %.2f ms" % (1e3 * (time.perf_counter() - t0)))
# the backward runs on the autograd engine's worker thread: profile the sweep itself
import importlib  # noqa: E402

OA = importlib.import_module("paddlexde_amd.functional.odeint_adjoint")
pr = cProfile.Profile()
_sweep = OA._sweep


def profiled(*a, **k):
    pr.enable()
    try:
        return _sweep(*a, **k)
    finally:
        torch.cuda.synchronize()
        pr.disable()


OA._sweep = profiled
run()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
