"""What the FIRST odeint_adjoint call of a module pays for the captures (the augmented dynamics per time signature, the two interval
graphs) and what later calls cost: config 3's problem, forward + backward, call by call.  `python3 profiles/tools/adjoint_first_call.py`"""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddlexde_amd import Dopri5, odeint_adjoint  # noqa: E402
from paddlexde_amd.utils import _rms_norm  # noqa: E402


class ODEFunc(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(2, 50), nn.Tanh(), nn.Linear(50, 2))

    def forward(self, t, y):
        return self.net(y * y * y)


dev = torch.device("cuda:0")
torch.manual_seed(0)
y0 = (torch.rand(8192, 2) * 4 - 2).to(dev)
t = torch.linspace(0.0, 25.0, 1000)[:32].to(dev)
for mode in ("graph_func=False (first in the process: its call 0 also pays the runtime's own first-use costs)", "default", "XDE_INTERVAL_GRAPH=0",
             "graph_func=False"):
    func = ODEFunc().to(dev)
    adj = {"graph_func": False} if mode.startswith("graph_func=False") else {}
    os.environ["XDE_INTERVAL_GRAPH"] = "0" if mode == "XDE_INTERVAL_GRAPH=0" else "1"
    rows = []
    for call in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pred = odeint_adjoint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm}, adjoint_options=dict(adj))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pred.abs().mean().backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append("call %d: forward %.1f ms, backward %.1f ms" % (call, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    print(mode + ": " + "; ".join(rows))
