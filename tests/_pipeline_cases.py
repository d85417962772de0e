"""End-to-end cases, run on the GPU (tests/test_gpu_odeint.py) and — host logic only — on the CPU double (tests/test_host_logic.py).
Pipelines (sync / lag / graph / auto), captured funcs, threads, the public step() API: bit-identical results whichever runs."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import xde_oracle as O
from paddlexde_amd import RK4, AdamsBashforthMoulton, AdaptiveHeun, Bosh3, Dopri5, Dopri8, Euler, Fehlberg2, Midpoint, _hip, odeint, odeint_adjoint
from paddlexde_amd.utils import _linf_norm, _rms_norm

from . import problems as P

from ._e2e_common import ADAPTIVE, ConstantLayer, DeepFunc, FIXED, ODEFunc, P_rms, _SmallMLP, _blocks, _golden, _linear, _mlp_foreign, _mlp_numpy  # noqa: F401


def test_native_library_is_the_one_running(dev):
    be = _hip.get_backend()
    if str(dev).startswith("cuda"):
        assert be.name == "hip" and isinstance(be, _hip.HipBackend)
    else:
        assert be.name.startswith("numpy-double")


def test_lag_pipeline_is_bitwise_equal_to_sync(dev):
    A, y0 = _linear(4096, 64, torch.float32)
    t = torch.linspace(0.0, 2.0, 7).to(dev)
    Ad = A.to(dev)
    f = lambda t_, y: y @ Ad.T  # noqa: E731
    a = odeint(f, y0.to(dev), t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm, "pipeline": "sync"})
    b = odeint(f, y0.to(dev), t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm, "pipeline": "lag"})
    assert torch.equal(a, b)
    c = odeint(f, y0.to(dev), t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm, "pipeline": "graph"})
    assert torch.equal(a, c)


def test_graph_pipeline_vdp_rejections(dev):
    """hipGraph replay with rejected steps: the predicated commit must leave (y0, f0) untouched on reject."""
    from paddlexde_amd.xde import BaseODE

    z = _golden("vdp_dopri5_f64")
    y0 = torch.from_numpy(z["y0"]).to(dev)
    t = torch.from_numpy(z["t"])
    xde = BaseODE(P.vdp_torch(float(z["mu"])), y0=y0, t_span=t)
    s = Dopri5(xde=xde, y0=y0, rtol=1e-6, atol=1e-8, norm=_rms_norm, dtype=torch.float64, pipeline="graph", record_trace=True)
    got = s.integrate(t)
    assert [s.stats["n_accept"], s.stats["n_reject"], s.stats["nfe"]] == list(z["counts"])
    assert P.parity_ok(got.cpu().numpy(), z["sol"], rtol=1e-8, atol=1e-10)
    assert len(s.trace) == len(z["trace"])


def test_graphed_func_forward(dev):
    from paddlexde_amd.utils import GraphedFunc

    z = _golden("vdp_dopri5_f64")
    y0 = torch.from_numpy(z["y0"]).to(dev)
    t = torch.from_numpy(z["t"])
    gf = GraphedFunc(P.vdp_torch(float(z["mu"])))
    got = odeint(gf, y0, t, solver=Dopri5, rtol=1e-6, atol=1e-8, options={"norm": _rms_norm, "dtype": torch.float64})
    assert P.parity_ok(got.cpu().numpy(), z["sol"], rtol=1e-8, atol=1e-10)
    if str(dev).startswith("cuda"):
        assert gf.captures >= 1 and gf.replays == int(z["counts"][2])  # one replay per function evaluation


# ----------------------------------------------------------------------------------------------
# fixed solvers: hipGraph pipeline (one captured step replayed over the grid)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,variant", [("euler", None), ("midpoint", None), ("rk4", "alt"), ("rk4", "classic"), ("adams", None)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_fixed_graph_pipeline_is_bitwise_equal_to_eager(dev, name, variant, dtype):
    """options={"pipeline": "graph"}: same kernels on the same operands, dt and the stage times read from device memory
    — bit-identical trajectory and the same NFE (Adams, whose step is data-dependent, silently stays eager; so does the
    CPU double, which has nothing to capture)."""
    from paddlexde_amd.xde import BaseODE

    rng = np.random.RandomState(12)
    y0 = torch.from_numpy(rng.uniform(-1, 1, size=(3, 2, 5))).to(dtype).to(dev)
    t = torch.from_numpy(np.cumsum(rng.uniform(0.01, 0.04, size=37))).to(dtype)
    w = torch.from_numpy(rng.uniform(-1, 1, size=(5,))).to(dtype).to(dev)

    def f(t_, y):
        return -0.5 * y - 0.1 * (y * y * y) + w * t_ + 0.25 * (y * w)

    runs = {}
    for pipeline in ("sync", "graph"):
        kw = {"variant": variant} if variant else {}
        with torch.no_grad():
            s = FIXED[name](xde=BaseODE(f, y0=y0, t_span=t), y0=y0, rtol=1e-3, atol=1e-4, norm=_rms_norm, pipeline=pipeline, **kw)
            runs[pipeline] = (s.integrate(t), s.nfe)
    assert runs["graph"][0].shape == (3, 37 * 2, 5)
    assert torch.allclose(runs["sync"][0], runs["graph"][0], rtol=0, atol=0, equal_nan=True)  # bit for bit
    assert torch.isfinite(runs["sync"][0]).all() or name == "adams"  # (high-order Adams on this uneven grid may blow up)
    assert runs["sync"][1] == runs["graph"][1]


def test_fixed_graph_pipeline_records_the_step_sizes_of_every_combine(dev):
    from paddlexde_amd.xde import BaseODE

    y0 = torch.zeros(1, 2, device=dev)
    t = np.cumsum(np.random.RandomState(0).uniform(0.01, 0.04, size=9)).astype(np.float32)
    dts = t[1:] - t[:-1]
    expect = {
        ("euler", None): [dts],
        ("midpoint", None): [np.float32(0.5) * dts, dts],
        ("rk4", "alt"): [dts * (1 / 3), dts, dts, dts],
    }
    for (name, variant), cols in expect.items():
        kw = {"variant": variant} if variant else {}
        s = FIXED[name](xde=BaseODE(lambda t_, y: y, y0=y0, t_span=torch.from_numpy(t)), y0=y0, rtol=1e-3, atol=1e-4, norm=_rms_norm, **kw)
        got = s._record_combine_dts(dts)
        assert len(got) == len(cols)
        for g, c in zip(got, cols):
            assert g.dtype == np.float64 and np.array_equal(g, c.astype(np.float64)), (name, variant)
        assert s._rec is None and s._dt is None and s.nfe == 0


def test_graphed_func_with_a_memset_node_replays_correctly(dev):
    """ROCm 7.2: a hipGraph MEMSET node (PyTorch's multi-block reductions zero a semaphore with hipMemsetAsync) does not hold
    its place in the graph — replayed among ordinary stream work, [memset, reduce kernel] returns the previous replay's
    result in a large fraction of launches.  GraphedFunc reads the node types back after capture and replaces the memset
    nodes by fill-kernel nodes (xde_graph_replace_memsets) before instantiating the graph."""
    from paddlexde_amd.utils import GraphedFunc

    w = torch.linspace(-1.0, 1.0, 50, device=dev)

    def reducing(t, y):  # y [8192, 50]: the column sum is a two-stage reduction
        return y * 0.5 + y.sum(0) * w

    def plain(t, y):
        return (y * 0.5 + w).tanh()

    t = torch.zeros((), device=dev)
    junk = torch.randn(8192, 64, device=dev)
    for func, n_memsets in ((reducing, 1), (plain, 0)):
        gf = GraphedFunc(func)
        for i in range(200):
            y = torch.randn(8192, 50, generator=torch.Generator().manual_seed(i)).to(dev)
            got = gf(t, y)
            junk.sum(1)  # ordinary stream work between replays
            assert torch.equal(got, func(t, y)), (func.__name__, i)
        if str(dev).startswith("cuda"):
            cap = list(gf._captures.values())[0].graph
            assert gf.replays >= 199 and not gf.safe_mode and cap.memsets_replaced == n_memsets and 2 not in cap.node_types


def test_graph_pipelines_with_a_reducing_func(dev):
    """A func with a multi-block reduction inside (its captured MEMSET node is replaced by a fill kernel, see
    utils/graphed.py::CapturedGraph): the adaptive and the fixed-step graph pipelines stay bit-identical to eager over
    hundreds of replays."""
    from paddlexde_amd.xde import BaseODE

    w = torch.linspace(-1.0, 1.0, 50, device=dev)
    y0 = torch.randn(8192, 50, generator=torch.Generator().manual_seed(3)).to(dev)

    def f(t_, y):
        return -0.3 * y + 1e-4 * y.sum(0) * w + 0.1 * t_

    t = torch.linspace(0.0, 20.0, 5)
    outs = {}
    for pipeline in ("sync", "graph"):
        s = Dopri5(xde=BaseODE(f, y0=y0, t_span=t), y0=y0, rtol=1e-7, atol=1e-9, norm=_rms_norm, pipeline=pipeline)
        outs[pipeline] = (s.integrate(t), s.stats["n_steps"])
    assert outs["sync"][1] == outs["graph"][1] and outs["sync"][1] > 20
    assert torch.equal(outs["sync"][0], outs["graph"][0])
    y0f = y0[:, None, :].contiguous()
    tf = torch.linspace(0.0, 1.0, 120)
    with torch.no_grad():
        a = odeint(f, y0f, tf.to(dev), solver=RK4, options={"norm": _rms_norm})
        b = odeint(f, y0f, tf.to(dev), solver=RK4, options={"norm": _rms_norm, "pipeline": "graph"})
    assert torch.equal(a, b)


def test_two_threads_two_streams_run_independent_solves(dev):
    """Thread-safety as INTEGRATION.md states it: distinct streams + distinct solver instances.  Two host threads, each on its
    own stream, integrate different problems concurrently ("sync" and "lag"; a hipGraph capture needs the device to itself —
    HIP refuses other threads' stream operations meanwhile — so "graph" is not a concurrent pipeline); each result is bit for
    bit its sequential one."""
    import threading

    from paddlexde_amd.xde import BaseODE

    if not str(dev).startswith("cuda"):
        pytest.skip("streams are a device notion")
    problems = []
    for k in range(2):
        A = P.skew_matrix(24, seed=10 + k).to(dev)
        y0 = torch.randn(512, 24, generator=torch.Generator().manual_seed(k)).to(dev)
        problems.append((A, y0, torch.linspace(0.0, 2.0 + k, 7)))

    def solve(k, pipeline):
        A, y0, t = problems[k]
        s = Dopri5(xde=BaseODE(lambda t_, y: y @ A.T - 0.01 * y * y * y, y0=y0, t_span=t), y0=y0, rtol=1e-6, atol=1e-8, norm=_rms_norm,
                   pipeline=pipeline)
        return s.integrate(t)

    for pipeline in ("sync", "lag"):
        ref = [solve(k, pipeline) for k in range(2)]
        torch.cuda.synchronize()
        out, err = [None, None], []

        def worker(k):
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    for _ in range(3):
                        out[k] = solve(k, pipeline)
                    st.synchronize()
            except Exception as e:  # noqa: BLE001
                err.append(e)

        th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        [x.start() for x in th]
        [x.join() for x in th]
        assert not err, err
        for k in range(2):
            assert torch.equal(out[k], ref[k]), (pipeline, k)


# ----------------------------------------------------------------------------------------------
# AdaptiveRKSolver.step(next_t) — the reference's public per-solver method (base_adaptive_solver_rk.py:116-127)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["dopri5", "bosh3", "dopri8"])
@pytest.mark.parametrize("reverse", [False, True])
def test_manual_step_loop_equals_integrate(dev, name, reverse):
    """`_before_integrate(t)` + `step(t_i)` for every output time gives the rows of `integrate(t)` bit for bit (same attempts,
    same dense-output arithmetic), with the same NFE and counts."""
    from paddlexde_amd.xde import BaseODE

    A, y0 = _linear(48, 24, torch.float32)
    Ad = A.to(dev)
    t = torch.linspace(0.0, 2.0, 9)
    if reverse:
        t = t.flip(0).contiguous()

    def make(**kw):
        y0d = y0.to(dev)
        return ADAPTIVE[name](xde=BaseODE(lambda t_, y: y @ Ad.T, y0=y0d, t_span=t), y0=y0d, rtol=1e-5, atol=1e-7, norm=_rms_norm, **kw)

    s1 = make()
    want = s1.integrate(t)
    s2 = make(record_trace=True)
    s2._before_integrate(t)
    rows = [y0.to(dev)] + [s2.step(ti) for ti in t[1:]]
    s2._after_integrate()
    got = torch.stack(rows)
    assert torch.equal(got, want)
    assert (s2.stats["n_accept"], s2.stats["n_reject"], s2.stats["nfe"]) == (s1.stats["n_accept"], s1.stats["n_reject"], s1.stats["nfe"])


def test_step_at_arbitrary_times_vs_oracle(dev):
    """step() takes any time at or after the start of the last accepted step: times that are not in t_span, several of them
    inside one accepted step (no new attempt then), and the oracle's step() agrees; going back before the retained step is the
    reference's `invalid interpolation` assertion (ode_utils.py:65-67)."""
    from paddlexde_amd.xde import BaseODE

    A, y0 = _linear(16, 8, torch.float64)
    An, Ad = A.numpy(), A.to(dev)
    t = np.array([0.0, 3.0])
    so = O.AdaptiveRKSolver(lambda t_, y: y @ An.T, y0.numpy(), 1e-6, 1e-8, method="dopri5", norm=O._rms_norm, dtype=np.float64)
    so._before_integrate(t)
    y0d = y0.to(dev)
    s = Dopri5(xde=BaseODE(lambda t_, y: y @ Ad.T, y0=y0d, t_span=torch.from_numpy(t)), y0=y0d, rtol=1e-6, atol=1e-8, norm=_rms_norm,
               dtype=torch.float64)
    s._before_integrate(torch.from_numpy(t))
    times = [0.3, 0.31, 0.32, 1.234, 2.5, 2.5, 2.9999]
    for x in times:
        n0 = s.stats.get("n_steps", 0)
        got = s.step(x).cpu().numpy()
        ref = so.step(np.float64(x))
        assert P.parity_ok(got, ref, rtol=1e-9, atol=1e-11), (x, P.worst(got, ref, 1e-9, 1e-11))
        assert s.stats["n_steps"] == len(so.trace)  # attempts were made exactly when the oracle made them
    assert s.stats["n_steps"] < len(times) + 10
    with pytest.raises(AssertionError, match="invalid interpolation"):
        s.step(0.1)
    s._after_integrate()


def test_auto_pipeline_survives_a_func_that_cannot_be_captured(dev):
    """pipeline="auto" (the default) tries a hipGraph capture of the attempted step once a small-state solve has run 16
    attempts.  A func that synchronises with the host (here: it reads a tensor value) makes that capture fail; the solve must
    carry on eagerly and return what pipeline="sync" returns, bit for bit, with the same counts."""
    from paddlexde_amd.xde import BaseODE

    A, y0 = _linear(32, 16, torch.float32)
    Ad = A.to(dev)
    t = torch.linspace(0.0, 12.0, 5)
    calls = []

    def func(t_, y):
        calls.append(float(y.abs().max()))  # device -> host read: illegal inside a stream capture
        return y @ Ad.T

    def run(pipeline):
        y0d = y0.to(dev)
        s = Dopri5(xde=BaseODE(func, y0=y0d, t_span=t), y0=y0d, rtol=1e-6, atol=1e-8, norm=_rms_norm, pipeline=pipeline)
        return s.integrate(t), s

    want, s1 = run("sync")
    got, s2 = run("auto")
    assert s1.stats["n_steps"] > 40  # long enough for auto to attempt its capture
    assert torch.equal(got, want)
    assert (s2.stats["n_accept"], s2.stats["n_reject"], s2.stats["nfe"]) == (s1.stats["n_accept"], s1.stats["n_reject"], s1.stats["nfe"])
    if str(dev).startswith("cuda"):
        assert s2._auto_state == "sync"  # the capture was attempted and abandoned
    # and the device is still usable for a capture that can succeed
    plain, s3 = (lambda: (lambda s: (s.integrate(t), s))(Dopri5(xde=BaseODE(lambda t_, y: y @ Ad.T, y0=y0.to(dev), t_span=t), y0=y0.to(dev),
                                                            rtol=1e-6, atol=1e-8, norm=_rms_norm)))()
    assert torch.equal(plain, want)
    if str(dev).startswith("cuda"):
        assert s3._auto_state == "graph"


@pytest.mark.parametrize("name", ["euler", "midpoint", "rk4"])
def test_fixed_auto_pipeline(dev, name):
    """Fixed solvers default to pipeline="auto": an inference-style call (grad mode off) on a small state with >= 24 steps
    replays ONE captured step over the grid — same trajectory bit for bit, same NFE — and a func that cannot be captured
    (it reads a tensor value on the host) makes the call fall back to the eager loop."""
    from paddlexde_amd.xde import BaseODE

    y0 = torch.tensor([[2.0, 0.0], [1.0, -1.0]]).to(dev)
    t = torch.linspace(0.0, 0.5, 60).to(dev)  # (short enough for first-order Euler to stay bounded on the cubic spiral)

    def run(func, pipeline):
        s = FIXED[name](xde=BaseODE(func, y0=y0, t_span=t), y0=y0, rtol=1e-3, atol=1e-4, norm=_rms_norm, pipeline=pipeline)
        with torch.no_grad():
            return s.integrate(t), s

    want, s1 = run(P.spiral_torch, "sync")
    got, s2 = run(P.spiral_torch, "auto")
    assert torch.isfinite(want).all() and torch.equal(got, want) and s2.nfe == s1.nfe
    seen = []

    def syncing(t_, y):
        seen.append(float(y.abs().max()))
        return P.spiral_torch(t_, y)

    got2, s3 = run(syncing, "auto")
    assert torch.equal(got2, want) and s3.nfe == s1.nfe
    # training-style call (grad mode on): the eager loop with its autograd graph, as before
    y0g = y0.clone().requires_grad_(True)
    sol = odeint(P.spiral_torch, y0g, t[:8], solver=FIXED[name])
    sol.sum().backward()
    assert y0g.grad is not None and torch.isfinite(y0g.grad).all()


def test_lag_pipeline_does_not_speculate_past_the_end_of_a_solve(dev):
    """The speculative pipeline enqueues attempt n+1 before it knows attempt n's verdict — except at the end: the predecessor's block
    says where attempt n lands if accepted (`t_plan`), and when that is the last output time the host waits for the verdict first.
    So a solve of several attempts calls func exactly as often under "lag" as under "sync" (no discarded attempt), with the same
    rows bit for bit; so does a solve that ends with its FIRST attempt, whose step size was chosen on the device: a copy of the
    freshly constructed block tells the host where it lands."""
    from paddlexde_amd.xde import BaseODE

    A = P.skew_matrix(8).double().to(dev)
    y0 = torch.randn(32, 8, generator=torch.Generator().manual_seed(3), dtype=torch.float64).to(dev)
    calls = [0]

    def f(t_, y):
        calls[0] += 1
        return y @ A.T

    for t in (torch.linspace(0.0, 2.0, 5, dtype=torch.float64), torch.tensor([0.0, 3.0], dtype=torch.float64), torch.tensor([1.0, -1.5], dtype=torch.float64)):
        got = {}
        for pipeline in ("sync", "lag"):
            calls[0] = 0
            s = Dopri5(xde=BaseODE(f, y0=y0, t_span=t), y0=y0, rtol=1e-8, atol=1e-10, norm=_rms_norm, dtype=torch.float64, pipeline=pipeline)
            with torch.no_grad():
                sol = s.integrate(t)
            got[pipeline] = (sol.clone(), calls[0], s.stats["nfe"], s.stats["n_steps"])
        assert got["sync"][3] >= 3  # (several attempts: the end is predictable)
        assert torch.equal(got["sync"][0], got["lag"][0])
        assert got["lag"][1] == got["sync"][1] == got["sync"][2] == got["lag"][2], got  # calls == nfe, identical under both pipelines
    # a one-attempt solve: its only attempt's step size was chosen on the device
    t = torch.tensor([0.0, 1e-6], dtype=torch.float64)
    calls[0] = 0
    s = Dopri5(xde=BaseODE(f, y0=y0, t_span=t), y0=y0, rtol=1e-8, atol=1e-10, norm=_rms_norm, dtype=torch.float64, pipeline="lag")
    with torch.no_grad():
        s.integrate(t)
    # (before round 4's end one speculative attempt — six evaluations — ran for nothing here: profiles/r04_adjoint_lag.txt)
    assert s.stats["n_steps"] == 1 and calls[0] == s.stats["nfe"], (calls[0], s.stats)
