"""Decompose the controller launch: A = normal (partials + host mirror), B = no mirror, C = finalised sums + no mirror,
D = xde_norm_finalize alone.  HIP-event durations per launch (dispatch-stamped)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from paddlexde_amd import _hip, Dopri5
from paddlexde_amd.utils import _rms_norm
from paddlexde_amd.xde import BaseODE

dev = torch.device("cuda:0")
be = _hip.get_backend()
B, D = 65536, 128
y0 = torch.randn(B, D, device=dev)
A = torch.randn(D, D, device=dev) * 0.05
s = Dopri5(xde=BaseODE(lambda t, y: y @ A, y0=y0, t_span=torch.tensor([0.0, 1e9])), y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm, pipeline="lag")
s._before_integrate(np.asarray([0.0, 1e9], dtype=np.float32))
s.advance(5)
torch.cuda.synchronize()
ctrl_nomirror = torch.zeros(_hip.C.sizeof(_hip.XdeCtrl), dtype=torch.uint8, device=dev)
ctrl_nomirror.copy_(s._ctrl)
sums = torch.zeros(32, dtype=torch.float64, device=dev); sums[0] = 1e6


def run(name, fn, kid, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    be.prof_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    r = be.prof_collect()[kid]
    be.prof_enable(False)
    print("%-40s %7.2f us  (%d launches)" % (name, 1e3 * r["ms"] / max(r["launches"], 1), r["launches"]), flush=True)


run("A control: partials + mirror", lambda: be.rk_control(s._ctrl, s._params, s._ws, None, s._t_span_dev, None, s._t_stage), "control")
run("B control: partials, no mirror", lambda: be.rk_control(ctrl_nomirror, s._params, s._ws, None, s._t_span_dev, None, s._t_stage), "control")
run("C control: sums, no mirror", lambda: be.rk_control(ctrl_nomirror, s._params, None, sums, s._t_span_dev, None, s._t_stage), "control")
run("C' control: sums + mirror", lambda: be.rk_control(s._ctrl, s._params, None, sums, s._t_span_dev, None, s._t_stage), "control")
run("D finalize alone", lambda: be.norm_finalize(s._ws, 0, sums), "finalize")
# the same back to back with another kernel in between (cache state as in a real step)
junk = torch.empty(1 << 24, device=dev)
def step_like():
    junk.mul_(1.0001)
    be.rk_control(s._ctrl, s._params, s._ws, None, s._t_span_dev, None, s._t_stage)
run("A' control after a streaming kernel", step_like, "control")
