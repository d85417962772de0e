#!/usr/bin/env python3
"""Spiral neural-DDE demo on paddlexde_amd — counterpart of the reference's example/dde_demo.py, written against the
protocol the library really has (xde/base_dde.py:47-52: ``move`` calls ``func(y_lags, y0)``; the demo in the reference still
carries an older ``(t, y0, lags, y_lags)`` signature).

Data: the spiral y' = (y^3) A of example/demo_utils.py:136-176.  A sample is a history window ``his [his_len, 2]`` on
``his_span = arange(his_len)``, the state after it ``y0`` and the next ``pred_len`` states.  Model (example/dde_demo.py:32-62):
Linear(2,128) on y0^3, a 2-layer GRU over the delayed states, averaged, tanh, Linear(128,2).  The delays ``lags`` (32 real
numbers in [0, his_len)) are trained together with the weights: ``ddeint`` gathers the delayed states with the cubic-Hermite
history spline (xde_hermite_gather) whose backward is d loss / d lags.

    python examples/dde_demo.py --max-steps 200
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from paddlexde_amd import RK4, ddeint, odeint  # noqa: E402


class DDEFunc(nn.Module):
    def __init__(self):
        super().__init__()
        self.linear1 = nn.Linear(2, 128)
        self.linear2 = nn.Linear(128, 2)
        self.gru = nn.GRU(2, 128, 2, batch_first=True)
        for lin in (self.linear1, self.linear2):
            nn.init.normal_(lin.weight, mean=0.0, std=0.1)
            nn.init.zeros_(lin.bias)

    def forward(self, y_lags, y0):
        """y_lags [B, n_lags, D] (delayed states, fixed during the solve), y0 [B, 1, D] -> [B, 1, D]"""
        h = self.linear1(y0**3)
        h_lags = self.gru(y_lags)[0][:, -1:, :]
        return self.linear2(torch.tanh((h + h_lags) / 2))


def make_data(device, data_len=1000):
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device=device)
    t = torch.linspace(0.0, 25.0, data_len, device=device)
    with torch.no_grad():
        true_y = odeint(lambda t_, y: (y**3) @ A, torch.tensor([[2.0, 0.0]], device=device), t, solver=RK4)  # [data_len, 2]
    return t, true_y


def get_batch(true_y, batch_size, his_len, pred_len, gen):
    dev = true_y.device
    idx = torch.randint(0, len(true_y) - his_len - pred_len, (batch_size,), generator=gen).to(dev)
    his = true_y[idx[:, None] + torch.arange(his_len, device=dev)[None, :]]  # [B, his_len, 2]
    win = true_y[idx[:, None] + his_len + torch.arange(pred_len, device=dev)[None, :]]  # [B, pred_len, 2]
    return win[:, :1, :].contiguous(), win, his.contiguous()


def train(max_steps=200, batch_size=20, his_len=20, pred_len=10, n_lags=32, seed=42, device="cuda:0", log_every=50):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    t, true_y = make_data(device)
    func = DDEFunc().to(device)
    lags = (torch.randint(0, his_len, (n_lags,), generator=gen).float()).to(device).requires_grad_(True)
    opt = torch.optim.RMSprop(list(func.parameters()) + [lags], lr=1e-3)
    t_span = t[:pred_len]
    his_span = torch.arange(his_len, dtype=torch.float32, device=device)
    losses, lag_grad = [], 0.0
    t0 = time.perf_counter()
    for step in range(1, max_steps + 1):
        y0, by, his = get_batch(true_y, batch_size, his_len, pred_len, gen)
        pred, _ = ddeint(func, y0, t_span, lags, his, his_span, solver=RK4)  # [B, pred_len, 2]
        loss = torch.mean(torch.abs(pred - by))
        opt.zero_grad()
        loss.backward()
        lag_grad = max(lag_grad, float(lags.grad.abs().max()))
        opt.step()
        losses.append(loss.item())
        if log_every and step % log_every == 0:
            print("Iter {:04d} | Total Loss {:.6f} | {:.1f} it/s".format(step, losses[-1], step / (time.perf_counter() - t0)), flush=True)
    return losses, lag_grad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--batch-size", type=int, default=20)
    ap.add_argument("--his-len", type=int, default=20)
    ap.add_argument("--pred-len", type=int, default=10)
    a = ap.parse_args()
    ls, g = train(a.max_steps, a.batch_size, a.his_len, a.pred_len)
    print("first-10 mean loss {:.4f} -> last-10 mean loss {:.4f}; max |d loss / d lags| seen {:.3e}".format(
        sum(ls[:10]) / 10, sum(ls[-10:]) / 10, g))
