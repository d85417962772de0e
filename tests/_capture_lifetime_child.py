"""Child process of tests/test_gpu_capture_lifetime.py — the round-5 abort (gpurun_out/r05f/suite.log) made deterministic.

Round 5's record: inside the stream capture of one module's augmented dynamics, a cyclic collection reaped ANOTHER, dropped module's
entry of the per-module capture cache (functional/_adjoint_capture.py::_GRAPH_CACHE, weakly keyed) and destroyed its captured graphs
and their private pools in the middle of the recording: `Fatal Python error: Aborted` under `weakref.remove`.

Here: module A (victim) gets cached captures (dynamics graphs + interval-solve graphs) and sits inside a reference cycle; module B
holds the ONLY reference to it and, from inside its own `forward` WHILE A RECORDING IS OPEN, drops that reference and calls
`gc.collect()` itself — which `gc.disable()` does not prevent.  A second victim without a cycle dies by a plain reference-count drop
at the same place.  The run must end with B's gradients bit-equal to the eager route's.

    python tests/_capture_lifetime_child.py                 # the product as it is: prints "OK ..." and exits 0
    python tests/_capture_lifetime_child.py --no-deferral   # the deferred release switched off by hand: what round 5 hit
"""
import gc
import os
import sys
import weakref

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from paddlexde_amd.functional import _adjoint_capture as AC  # noqa: E402
from paddlexde_amd.functional import odeint_adjoint  # noqa: E402
from paddlexde_amd.solver import Dopri5  # noqa: E402
from paddlexde_amd.utils import _rms_norm, graphed  # noqa: E402


class Field(nn.Module):
    """The spiral demo's func (example/ode_demo.py:21-33 in the reference): Linear(2,50) -> Tanh -> Linear(50,2) on y**3."""

    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.net = nn.Sequential(nn.Linear(2, 50), nn.Tanh(), nn.Linear(50, 2))
        for m in self.net:
            if isinstance(m, nn.Linear):
                with torch.no_grad():
                    m.weight.copy_(0.1 * torch.randn(m.weight.shape, generator=g))
                    m.bias.zero_()
        object.__setattr__(self, "victims", [])
        object.__setattr__(self, "dropped_inside_a_recording", 0)

    def forward(self, t, y):
        if self.victims and graphed.recordings_open():
            self.victims.clear()  # the acyclic victim dies here, by reference count
            gc.collect()  # ... and the one inside a cycle here: an explicit collection inside the recorded body
            gc.collect()  # (the victims' interval solvers sit in cycles of their own: garbage only once the first pass has run)
            object.__setattr__(self, "dropped_inside_a_recording", self.dropped_inside_a_recording + 1)
        return self.net(y ** 3)


def grads(m, y0, t, **adj):
    for p in m.parameters():
        p.grad = None
    y = y0.clone().requires_grad_(True)
    sol = odeint_adjoint(m, y, t, solver=Dopri5, rtol=1e-6, atol=1e-8, options={"norm": _rms_norm, "pipeline": "sync"}, adjoint_options=adj)
    (sol * sol).mean().backward()
    return [y.grad.clone()] + [p.grad.clone() for p in m.parameters()]


def main():
    if "--no-deferral" in sys.argv:
        del graphed.CapturedGraph.__del__  # (round 5's library: a dying owner destroys its graph wherever it dies)
    dev = "cuda:0"
    y0 = (torch.rand(64, 2, generator=torch.Generator().manual_seed(5)) * 4 - 2).to(dev)
    t = torch.linspace(0.0, 2.0, 6).to(dev)
    gc.disable()  # nothing dies by an automatic collection before the point this script chooses

    refs = []
    b = Field(1).to(dev)
    for cyclic in (True, False):
        a = Field(2 + cyclic).to(dev)
        grads(a, y0, t, graph_func=True)  # captures cached on the module
        cache = AC._GRAPH_CACHE[a]
        n_graphs = sum(g.captures for g in cache.values() if hasattr(g, "captures"))
        n_intervals = sum(len(getattr(g, "_intervals", {})) for g in cache.values())
        assert n_graphs >= 1 and n_intervals >= 1, (n_graphs, n_intervals)
        if cyclic:
            object.__setattr__(a, "_cycle", [a])
        for p in a.parameters():
            p.grad = None
        refs.append(weakref.ref(a))
        b.victims.append(a)
        del a, cache
    if not all(r() is not None for r in refs):
        raise SystemExit("the victims died before the recording opened")

    want = grads(b, y0, t, graph_func=False)
    assert b.dropped_inside_a_recording == 0 and len(b.victims) == 2
    before = graphed._REC["deferred_total"]
    got = grads(b, y0, t, graph_func=True)  # captures B: its forward drops A and collects INSIDE the recording
    deferred = graphed._REC["deferred_total"] - before
    assert b.dropped_inside_a_recording == 1, b.dropped_inside_a_recording
    assert all(r() is None for r in refs), "the victims are still alive"
    assert graphed.recordings_open() == 0 and not graphed._DEFERRED
    if "--no-deferral" not in sys.argv:
        assert deferred >= 4, deferred  # two victims x (dynamics graph(s) + interval graphs)
    assert all(torch.equal(x, y) for x, y in zip(got, want)), "captured gradients differ from the eager route's"
    again = grads(b, y0, t, graph_func=True)  # replays
    assert all(torch.equal(x, y) for x, y in zip(again, want))
    torch.cuda.synchronize()
    print("OK deferred={} recordings_open={}".format(deferred, graphed.recordings_open()))


if __name__ == "__main__":
    main()
