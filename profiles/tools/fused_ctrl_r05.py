"""Round 5: error norm + controller as ONE ticketed launch (xde_error_norm_control above the one-workgroup size) against the two launches
of the default path, at config 4's shard and config 2 — re-measured because the controller lost 1-2 us since round 2's break-even.
Whole-sequence time by torch events around 300 repetitions (both variants end in the same state: control block published)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from paddlexde_amd import Dopri5, _hip
from paddlexde_amd.utils import _rms_norm
from paddlexde_amd.xde import BaseODE

dev = torch.device("cuda:0")
be = _hip.get_backend()
for name, (B, D) in (("c4 shard", (65536, 64)), ("c2", (65536, 128))):
    y0 = torch.randn(B, D, device=dev)
    A = torch.randn(D, D, device=dev) * 0.05
    s = Dopri5(xde=BaseODE(lambda t, y: y @ A, y0=y0, t_span=torch.tensor([0.0, 1e9])), y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm, pipeline="lag")
    s.y0 = y0
    s._before_integrate(np.asarray([0.0, 1e9], dtype=np.float32))
    s.advance(5)
    torch.cuda.synchronize()
    k6, e_pre, y1 = torch.randn_like(y0), torch.randn_like(y0) * 1e-6, y0 + 1e-3
    coef = [float(s.tableau.c_error[-1])]
    junk = torch.empty(1 << 24, device=dev)

    def two():
        be.error_norm_partial([k6], coef, y0, y1, float(s.rtol), float(s.atol), s._xsegs, s._norm_kind, s._ws, ctrl=s._ctrl, e_pre=e_pre)
        be.rk_control(s._ctrl, s._params, s._ws, None, s._t_span_dev, None, s._t_stage)

    def one():
        be.error_norm_control([k6], coef, y0, y1, s._xsegs, s._ws, s._ctrl, s._params, s._t_span_dev, None, s._t_stage, e_pre=e_pre)

    for label, fn in (("two launches (error norm, controller)", two), ("one ticketed launch", one), ("two launches", two), ("one ticketed launch", one)):
        for _ in range(30):
            junk.mul_(1.0001)
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # (a streaming kernel in front of each repetition, as in a real step; its own time is measured apart and subtracted)
        e0.record()
        for _ in range(300):
            junk.mul_(1.0001)
            fn()
        e1.record()
        e1.synchronize()
        total = e0.elapsed_time(e1)
        e0.record()
        for _ in range(300):
            junk.mul_(1.0001)
        e1.record()
        e1.synchronize()
        print("%-9s %-40s %7.2f us per repetition" % (name, label, 1e3 * (total - e0.elapsed_time(e1)) / 300), flush=True)
