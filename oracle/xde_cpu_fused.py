"""ctypes view of oracle/build/libxde_cpu.so (oracle/xde_cpu_kernels.c) and a Dopri5 attempt stepper built on it —
TEST / BASELINE INFRASTRUCTURE (SURVEY.md section 8(d) baseline "B2": a careful fused CPU implementation of the reference's
step: one pass per stage, the error estimate fused into the last stage and the norm pass, as the HIP path does).  Only
bench.py's cpu_baseline leg and tests/ import this; the product never does.

The controller and the initial step are the numpy oracle's (oracle/xde_oracle.py); func stays the caller's (torch-CPU matmul
in bench.py, i.e. the framework's multi-threaded GEMM)."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

from . import xde_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "build", "libxde_cpu.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        lib = C.CDLL(LIB)
        fp, dp = C.c_void_p, C.POINTER(C.c_double)
        lib.xde_cpu_stage_combine.restype = None
        lib.xde_cpu_stage_combine.argtypes = [fp, fp, C.POINTER(C.c_void_p), dp, C.c_int, C.c_double, C.c_int64, fp, dp]
        lib.xde_cpu_error_norm.restype = C.c_double
        lib.xde_cpu_error_norm.argtypes = [fp, fp, C.c_double, fp, fp, C.c_double, C.c_double, C.c_double, C.c_int64, C.POINTER(C.c_int64)]
        _lib = lib
    return _lib


def _dbl(xs):
    return (C.c_double * len(xs))(*[float(x) for x in xs])


def stage_combine(out, y0, ks, coef, dt, out2=None, coef2=None):
    """out = y0 + sum_j ks[j] * (coef[j] * dt)  [+ out2 = sum_j ks[j] * (dt * coef2[j])]; float32 contiguous torch tensors."""
    lib = load()
    ptrs = (C.c_void_p * len(ks))(*[k.data_ptr() for k in ks])
    lib.xde_cpu_stage_combine(out.data_ptr(), y0.data_ptr(), ptrs, _dbl(coef), len(ks), float(dt), out.numel(),
                              None if out2 is None else out2.data_ptr(), None if coef2 is None else _dbl(coef2))


def error_norm(e_pre, k_last, c_last, y0, y1, rtol, atol, dt):
    lib = load()
    nf = C.c_int64(0)
    s = lib.xde_cpu_error_norm(e_pre.data_ptr(), k_last.data_ptr(), float(c_last), y0.data_ptr(), y1.data_ptr(), float(rtol), float(atol),
                               float(dt), y0.numel(), C.byref(nf))
    return s, int(nf.value)


class FusedDopri5Stepper:
    """Attempted Dopri5 steps (solver/base_adaptive_solver_rk.py:129-284, the reference's I-controller) on the fused kernels."""

    def __init__(self, func, y0, rtol, atol):
        order, tab, _ = O.ADAPTIVE["dopri5"]
        self.func, self.order = func, order
        self.beta = [[float(b) for b in row] for row in tab.beta]
        self.c_err = [float(c) for c in tab.c_error]
        self.alpha = [float(a) for a in tab.alpha]
        self.rtol, self.atol = np.float32(rtol), np.float32(atol)
        self.y0 = y0.contiguous()
        self.n_accept = self.n_reject = 0
        self.trace = []

    def start(self, t0, first_step):
        self.t = np.float32(t0)
        self.dt = np.float32(first_step)
        self.f0 = self.func(torch.tensor(float(self.t)), self.y0).contiguous()
        self.scratch = torch.empty_like(self.y0)
        self.ebuf = torch.empty_like(self.y0)

    def step(self):
        y0, f0, dt, t0 = self.y0, self.f0, self.dt, self.t
        ks = [f0]
        y1 = None
        for i in range(6):
            idx = [0] + [j for j in range(1, i + 1) if self.beta[i][j] != 0.0]
            last = i == 5
            out = torch.empty_like(y0) if last else self.scratch
            stage_combine(out, y0, [ks[j] for j in idx], [self.beta[i][j] for j in idx], dt,
                          out2=self.ebuf if last else None, coef2=[self.c_err[j] for j in idx] if last else None)
            ti = t0 + dt if self.alpha[i] == 1.0 else t0 + np.float32(self.alpha[i]) * dt
            ks.append(self.func(torch.tensor(float(ti)), out).contiguous())
            y1 = out
        s, nf = error_norm(self.ebuf, ks[6], self.c_err[6], y0, y1, self.rtol, self.atol, dt)
        assert nf == 0, "non-finite values in state `y`"
        ratio = np.float32(np.sqrt(np.float32(s / y0.numel())))
        accept = bool(ratio <= 1)
        self.trace.append((float(t0), float(dt), float(ratio), accept))
        if accept:
            self.y0, self.f0, self.t = y1, ks[6], np.float32(t0 + dt)
            self.n_accept += 1
        else:
            self.n_reject += 1
        self.dt = np.float32(O.optimal_step_size(dt, ratio, np.float32(0.9), np.float32(10.0), np.float32(0.2), self.order))
