#!/usr/bin/env python3
"""The spiral neural-ODE demo trained from ANOTHER framework's side of the boundary — what a PaddleXDE user's script does, with a
stand-in for Paddle (PaddlePaddle is not installed in the build image).

The reference's demo trains a `paddle.nn.Layer` through `odeint_adjoint` (example/ode_demo.py:51,67), whose backward differentiates the
layer with Paddle itself (`paddle.autograd.grad(..., grad_outputs=-adj_y)`, functional/odeint_adjoint.py:108-114).  Here the same division
of labour, framework-neutral:

  * `MiniTensor` is "the caller's framework": it owns device memory and exposes `__dlpack__` / `__dlpack_device__` and nothing else the
    solver could use; its arithmetic (what its kernels would do) is written against its own raw handle;
  * the caller's layer `func(t, y)` and its vector-Jacobian product `vjp(t, y, cotangent)` are written by hand on MiniTensors — the
    chain rule of Linear(2,50) -> Tanh -> Linear(50,2) applied to y^3 — no autograd of any framework runs anywhere in this script;
  * `paddlexde_amd.AdjointProblem(...).forward / .backward` integrate; a few lines of RMSprop on the caller's parameter tensors train.

With Paddle the two hand-written functions are `layer(t, y)` and the 10-line `paddle.autograd.grad` hook of INTEGRATION.md section B.1,
and `MiniTensor.from_dlpack` is `paddle.utils.dlpack.from_dlpack`.

    python examples/foreign_adjoint_demo.py --max-steps 200 [--solver dopri5]
"""
import argparse
import os
import sys
import time

import torch  # (the stand-in framework computes with torch kernels under the hood; the SOLVER never sees that)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from paddlexde_amd import RK4, AdjointProblem, Dopri5, odeint  # noqa: E402
from paddlexde_amd.utils import _rms_norm  # noqa: E402


class MiniTensor:
    """A tensor of 'another framework': device memory + the DLPack protocol."""

    def __init__(self, raw):
        self.raw = raw
        self.shape, self.dtype = tuple(raw.shape), str(raw.dtype)

    def __dlpack__(self, stream=None, **kw):
        return self.raw.__dlpack__(stream=stream) if stream is not None else self.raw.__dlpack__()

    def __dlpack_device__(self):
        return self.raw.__dlpack_device__()

    @staticmethod
    def from_dlpack(x):  # the framework's importer (consumes any __dlpack__ producer: here, the solver's buffers)
        return MiniTensor(torch.from_dlpack(x))


class SpiralLayer:
    """Linear(2,50) -> Tanh -> Linear(50,2) on y^3 (example/ode_demo.py:17-33: weights 0.1*randn, biases 0), forward AND vjp by hand."""

    def __init__(self, device, seed=42):
        g = torch.Generator().manual_seed(seed)
        mk = lambda *s: MiniTensor((0.1 * torch.randn(*s, generator=g)).to(device))  # noqa: E731
        self.W1, self.b1, self.W2, self.b2 = mk(2, 50), MiniTensor(torch.zeros(50, device=device)), mk(50, 2), MiniTensor(torch.zeros(2, device=device))

    def parameters(self):
        return [self.W1, self.b1, self.W2, self.b2]

    def __call__(self, t, y):
        y_ = y.raw
        return MiniTensor(torch.tanh((y_ * y_ * y_) @ self.W1.raw + self.b1.raw) @ self.W2.raw + self.b2.raw)

    def vjp(self, t, y, cotangent):
        """(f, c^T df/dt, c^T df/dy, c^T df/dW1, c^T df/db1, c^T df/dW2, c^T df/db2) — the state may carry leading batch axes."""
        y_, c = y.raw, cotangent.raw
        yy = y_ * y_
        u = yy * y_
        a = torch.tanh(u @ self.W1.raw + self.b1.raw)
        f = a @ self.W2.raw + self.b2.raw
        c2, a2, u2 = c.reshape(-1, 2), a.reshape(-1, 50), u.reshape(-1, 2)
        g_h = (c @ self.W2.raw.t()) * (1.0 - a * a)
        g_y = (g_h @ self.W1.raw.t()) * (3.0 * yy)
        gh2 = g_h.reshape(-1, 50)
        return (MiniTensor(f), None, MiniTensor(g_y), MiniTensor(u2.t() @ gh2), MiniTensor(gh2.sum(0)), MiniTensor(a2.t() @ c2),
                MiniTensor(c2.sum(0)))


def make_data(device, data_len=1000):
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device=device)
    t = torch.linspace(0.0, 25.0, data_len, device=device)
    with torch.no_grad():
        true_y = odeint(lambda t_, y: (y**3) @ A, torch.tensor([[2.0, 0.0]], device=device), t, solver=RK4)  # [data_len, 2]
    return t, true_y


def train(max_steps=200, batch_size=20, pred_len=10, solver="rk4", seed=42, device="cuda:0", log_every=50, lr=1e-3):
    gen = torch.Generator().manual_seed(seed)
    t, true_y = make_data(device)
    layer = SpiralLayer(device, seed)
    S = {"rk4": RK4, "dopri5": Dopri5}[solver]
    tol = dict(rtol=1e-5, atol=1e-7) if S is Dopri5 else {}
    problem = AdjointProblem(layer, vjp=layer.vjp, adjoint_params=layer.parameters(), solver=S, options={"norm": _rms_norm},
                             from_dlpack=MiniTensor.from_dlpack, **tol)
    square_avg = [torch.zeros_like(p.raw) for p in layer.parameters()]  # RMSprop state (alpha 0.99, eps 1e-8: the reference demo's optimiser)
    losses, t0 = [], time.perf_counter()
    for step in range(1, max_steps + 1):
        idx = torch.randint(0, len(t) - pred_len, (batch_size,), generator=gen).to(device)
        y0 = true_y[idx]  # [B, 2]
        target = true_y[idx[:, None] + torch.arange(pred_len, device=device)[None, :]]  # [B, T, 2]
        bt = t[:pred_len]
        if S is RK4:  # fixed solvers concatenate time on axis -2: a [B, 1, 2] state gives [B, T, 2]
            pred = problem.forward(MiniTensor(y0[:, None, :].contiguous()), MiniTensor(bt))
            diff = pred.raw - target
        else:  # adaptive solvers put time first: [T, B, 2]
            pred = problem.forward(MiniTensor(y0.contiguous()), MiniTensor(bt))
            diff = pred.raw - target.permute(1, 0, 2)
        losses.append(float(diff.abs().mean()))
        grad_pred = MiniTensor(torch.sign(diff) / diff.numel())  # d mean|pred - true| / d pred, formed by the caller's framework
        _, _, grads = problem.backward(MiniTensor(bt), pred, grad_pred)
        for p, g, v in zip(layer.parameters(), grads, square_avg):  # RMSprop, in place on the caller's own parameter storage
            v.mul_(0.99).addcmul_(g.raw.reshape(v.shape), g.raw.reshape(v.shape), value=0.01)
            p.raw.addcdiv_(g.raw.reshape(v.shape), v.sqrt().add_(1e-8), value=-lr)
        if log_every and step % log_every == 0:
            print("Iter {:04d} | Total Loss {:.6f} | {:.1f} it/s".format(step, losses[-1], step / (time.perf_counter() - t0)), flush=True)
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--solver", default="rk4", choices=["rk4", "dopri5"])
    a = ap.parse_args()
    ls = train(max_steps=a.max_steps, solver=a.solver)
    print("first 10: {:.4f}   last 10: {:.4f}".format(sum(ls[:10]) / 10, sum(ls[-10:]) / 10))
