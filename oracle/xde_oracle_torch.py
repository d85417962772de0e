"""Torch-CPU twin of the oracle's adaptive Runge-Kutta step — TEST INFRASTRUCTURE (CPU baseline only).

Same op sequence as oracle/xde_oracle.py (hence as the reference: stage-innermost ``k[..., S+1]`` buffer,
``k[..., :i+1] * (beta_i * dt)`` -> ``sum(-1)`` -> add per stage, unfused error-ratio chain, ``isfinite`` pass and
interpolation refit on every accepted step; solver/base_adaptive_solver_rk.py:129-292, utils/ode_utils.py:28-97) but on
torch CPU tensors, so that element-wise ops use every host core like the reference's Paddle CPU build would.
It exists to give bench.py's ``cpu_baseline`` a multi-threaded "reference-equivalent eager CPU" number (BASELINE.md
B1); tests/test_oracle_pinning.py checks it against the numpy oracle step for step.
"""
import collections

import torch

from . import xde_oracle as O

RKState = collections.namedtuple("RKState", "y1 f1 t0 t1 dt interp_coeff")


def _rms_norm(x):
    return x.abs().pow(2).mean().sqrt()


class TorchAdaptiveStepper:
    def __init__(self, func, y0, rtol, atol, method="dopri5", safety=0.9, ifactor=10.0, dfactor=0.2):
        self.func = func
        self.y0 = y0
        self.order, tab, mid = O.ADAPTIVE[method]
        dt_ = y0.dtype
        self.alpha = [float(a) for a in tab.alpha]
        self.beta = [torch.tensor(b, dtype=dt_) for b in tab.beta]
        self.c_sol = torch.tensor(tab.c_sol, dtype=dt_)
        self.c_error = torch.tensor(tab.c_error, dtype=dt_)
        self.mid = torch.tensor(mid, dtype=dt_)
        self.fsal = bool(tab.c_sol[-1] == 0 and (tab.c_sol[:-1] == tab.beta[-1]).all())
        self.rtol = torch.tensor(rtol, dtype=torch.float32)
        self.atol = torch.tensor(atol, dtype=torch.float32)
        self.safety = torch.tensor(safety, dtype=torch.float32)
        self.ifactor = torch.tensor(ifactor, dtype=torch.float32)
        self.dfactor = torch.tensor(dfactor, dtype=torch.float32)
        self.nfe = 0
        self.trace = []

    def move(self, t, y):
        self.nfe += 1
        return self.func(t, y)

    def start(self, t0, first_step):
        t0 = torch.tensor(t0, dtype=torch.float32)
        f0 = self.move(t0, self.y0)
        self.state = RKState(self.y0, f0, t0, t0, torch.tensor(first_step, dtype=torch.float32), [self.y0] * 5)

    def _rk_step(self, y0, f0, t0, dt, t1):
        tdt = y0.dtype
        t0, dt, t1 = t0.to(tdt), dt.to(tdt), t1.to(tdt)
        k = torch.empty(f0.shape + (len(self.alpha) + 1,), dtype=y0.dtype)
        k[..., 0] = f0
        yi = None
        for i, (alpha_i, beta_i) in enumerate(zip(self.alpha, self.beta)):
            ti = t1 if alpha_i == 1.0 else t0 + alpha_i * dt
            yi = y0 + torch.sum(k[..., : i + 1] * (beta_i * dt), dim=-1).reshape(y0.shape)
            k[..., i + 1] = self.move(ti, yi)
        if not self.fsal:
            yi = y0 + torch.sum(k * (dt * self.c_sol), dim=-1).reshape(y0.shape)
        return yi, k[..., -1], torch.sum(k * (dt * self.c_error), dim=-1), k

    def step(self):
        y0, f0, _, t0, dt, interp = self.state
        t1 = t0 + dt
        assert t0 + dt > t0, "underflow in dt {}".format(dt.item())
        assert torch.isfinite(y0).all(), "non-finite values in state `y`"
        y1, f1, err, k = self._rk_step(y0, f0, t0, dt, t1)
        tol = self.atol + self.rtol * torch.fmax(y0.abs(), y1.abs())
        ratio = _rms_norm(err / tol).abs()
        accept = bool(ratio <= 1)
        self.trace.append((float(t0), float(dt), float(ratio), accept))
        if accept:
            dty = dt.to(y0.dtype)
            y_mid = y0 + torch.sum(k * (dty * self.mid), dim=-1).reshape(y0.shape)
            f0k, f1k = k[..., 0], k[..., -1]
            a = 2 * dty * (f1k - f0k) - 8 * (y1 + y0) + 16 * y_mid
            b = dty * (5 * f0k - 3 * f1k) + 18 * y0 + 14 * y1 - 32 * y_mid
            c = dty * (f1k - 4 * f0k) - 11 * y0 - 5 * y1 + 16 * y_mid
            interp = [y0, dty * f0k, c, b, a]
            t_next, y_next, f_next = t1, y1, f1
        else:
            t_next, y_next, f_next = t0, y0, f0
        if ratio == 0:
            dt_next = dt * self.ifactor
        else:
            dfactor = torch.tensor(1.0) if ratio < 1 else self.dfactor
            exponent = torch.tensor(float(self.order), dtype=dt.dtype).reciprocal()
            factor = torch.fmin(self.ifactor, torch.fmax(self.safety / ratio.to(dt.dtype) ** exponent, dfactor))
            dt_next = dt * factor
        self.state = RKState(y_next, f_next, t0, t_next, dt_next, interp)
