#!/usr/bin/env python3
"""Spiral neural-ODE demo on paddlexde_amd — this package's counterpart of the reference's example/ode_demo.py
(workload definition of BASELINE.json configs 1 and 3; data generation as in example/demo_utils.py:136-176).

Ground truth: y' = (y^3) A, A = [[-0.1, 2], [-2, -0.1]], y0 = [2, 0], t = linspace(0, 25, data_len), integrated with the
reference's RK4.  Model: Linear(2,50) -> Tanh -> Linear(50,2) applied to y^3, weights 0.1*randn, biases 0.  Training:
mini-batches of (y0, t[:pred_len], y[:pred_len]) windows, loss = mean |pred - true|, RMSprop(lr=1e-3) — by back-propagating
through odeint(RK4) (as the reference does) or with --adjoint through odeint_adjoint.

    python examples/ode_demo.py --max-steps 200 [--adjoint] [--solver dopri5]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from paddlexde_amd import RK4, Dopri5, odeint, odeint_adjoint  # noqa: E402
from paddlexde_amd.utils import _rms_norm  # noqa: E402


class ODEFunc(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(2, 50), nn.Tanh(), nn.Linear(50, 2))
        for m in self.net:
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=0.1)
                nn.init.zeros_(m.bias)

    def forward(self, t, y):
        return self.net(y**3)


def make_data(device, data_len=1000):
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device=device)
    t = torch.linspace(0.0, 25.0, data_len, device=device)
    y0 = torch.tensor([[2.0, 0.0]], device=device)
    with torch.no_grad():
        true_y = odeint(lambda t_, y: (y**3) @ A, y0, t, solver=RK4)  # [data_len, 2]
    return t, true_y


def get_batch(t, true_y, batch_size, pred_len, gen):
    idx = torch.randint(0, len(t) - pred_len, (batch_size,), generator=gen).to(true_y.device)
    y0 = true_y[idx]  # [B, 2]
    win = idx[:, None] + torch.arange(pred_len, device=true_y.device)[None, :]
    return y0, t[:pred_len], true_y[win]  # [B, 2], [T], [B, T, 2]


def train(max_steps=200, batch_size=20, pred_len=10, adjoint=False, solver="rk4", seed=42, device="cuda:0", log_every=50):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    t, true_y = make_data(device)
    func = ODEFunc().to(device)
    opt = torch.optim.RMSprop(func.parameters(), lr=1e-3)
    S = {"rk4": RK4, "dopri5": Dopri5}[solver]
    losses = []
    t0 = time.perf_counter()
    for step in range(1, max_steps + 1):
        y0, bt, by = get_batch(t, true_y, batch_size, pred_len, gen)
        if S is RK4:
            # fixed solvers concatenate time on axis -2: give the state a length-1 time axis -> [B, T, 2]
            args = (func, y0[:, None, :], bt)
            pred = (odeint_adjoint if adjoint else odeint)(*args, solver=RK4)
        else:
            pred = odeint_adjoint(func, y0, bt, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm})  # [T, B, 2]
            pred = pred.permute(1, 0, 2)
        loss = torch.mean(torch.abs(pred - by))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if log_every and step % log_every == 0:
            print("Iter {:04d} | Total Loss {:.6f} | {:.1f} it/s".format(step, losses[-1], step / (time.perf_counter() - t0)), flush=True)
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--batch-size", type=int, default=20)
    ap.add_argument("--pred-len", type=int, default=10)
    ap.add_argument("--adjoint", action="store_true")
    ap.add_argument("--solver", default="rk4", choices=["rk4", "dopri5"])
    a = ap.parse_args()
    ls = train(a.max_steps, a.batch_size, a.pred_len, a.adjoint, a.solver)
    print("first-10 mean loss {:.4f} -> last-10 mean loss {:.4f}".format(sum(ls[:10]) / 10, sum(ls[-10:]) / 10))
