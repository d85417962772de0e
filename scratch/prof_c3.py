import cProfile, pstats, sys, time, io
import torch, torch.nn as nn
sys.path.insert(0, ".")
from paddlexde_amd import Dopri5, odeint_adjoint
from paddlexde_amd.utils import _rms_norm
dev = torch.device("cuda", 0)
class ODEFunc(nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(42)
        self.net = nn.Sequential(nn.Linear(2, 50), nn.Tanh(), nn.Linear(50, 2))
        for m in self.net:
            if isinstance(m, nn.Linear):
                with torch.no_grad():
                    m.weight.copy_(0.1 * torch.randn(m.weight.shape, generator=g)); m.bias.zero_()
    def forward(self, t, y):
        return self.net(y**3)
func = ODEFunc().to(dev)
y0 = (torch.rand(8192, 2, generator=torch.Generator().manual_seed(0)) * 4 - 2).to(dev)
t = torch.linspace(0.0, 25.0, 1000)[:32].to(dev)
import paddlexde_amd.functional
OA = sys.modules["paddlexde_amd.functional.odeint_adjoint"]
_orig_bwd = OA.OdeintAdjointMethod.backward
PROF = {"on": False}
def _bwd(ctx, *g):
    if not PROF["on"]:
        return _orig_bwd(ctx, *g)
    pr = cProfile.Profile(); pr.enable()
    out = _orig_bwd(ctx, *g)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60); print(s.getvalue())
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue())
    return out
OA.OdeintAdjointMethod.backward = staticmethod(_bwd)
def run(profile=False):
    PROF["on"] = profile
    profile = False
    for p in func.parameters(): p.grad = None
    pred = odeint_adjoint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm, "pipeline": "lag"},
                          adjoint_options={"pipeline": "sync", "graph_func": True})
    loss = torch.mean(torch.abs(pred))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if profile:
        pr = cProfile.Profile(); pr.enable()
    loss.backward()
    torch.cuda.synchronize()
    if profile:
        pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue())
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue())
    return time.perf_counter() - t0
run(); print("bwd", run()); print("bwd", run()); run(True)
