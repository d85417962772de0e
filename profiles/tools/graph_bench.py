"""Is hipGraph replay of one attempted step host-bound or GPU-bound?  Config 5 (VdP 4096 x 2)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from paddlexde_amd import _hip, Dopri5
from paddlexde_amd.utils import _rms_norm
from paddlexde_amd.xde import BaseODE

dev = torch.device("cuda:0")
mu = 1000.0
def vdp(t, y):
    x, v = y[..., 0], y[..., 1]
    return torch.stack([v, mu * (1 - x * x) * v - x], dim=-1)
def cheap(t, y):
    return y * -0.5
for name, f in (("vdp (6 torch kernels)", vdp), ("y*-0.5 (1 torch kernel)", cheap)):
    y0 = (torch.tensor([2.0, 0.0]) + 0.01 * torch.randn(4096, 2)).to(dev)
    s = Dopri5(xde=BaseODE(f, y0=y0, t_span=torch.tensor([0.0, 1e9])), y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm, pipeline="graph", max_num_steps=10**9)
    s._before_integrate(np.asarray([0.0, 1e9], dtype=np.float32))
    s.advance(50)
    torch.cuda.synchronize()
    g = s._graphs[max(s._graphs)]
    types = g.graph.node_types
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-26s nodes=%d  host issue %.1f us/replay, wall %.1f us/replay" % (name, len(types), 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n), flush=True)
    # the solver's own loop
    t0 = time.perf_counter(); c = s.advance(300); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("   solver.advance: %.1f us/attempt" % (1e6 * (t1 - t0) / 300), flush=True)
    s._after_integrate()
