"""End-to-end cases, run on the GPU (tests/test_gpu_odeint.py) and — host logic only — on the CPU double (tests/test_host_logic.py).
SURVEY 8(f2): the options surface — norms, mixed precision, PI controller, step_t, error conventions, degenerate inputs, reuse_f0, callbacks."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import xde_oracle as O
from paddlexde_amd import RK4, AdamsBashforthMoulton, AdaptiveHeun, Bosh3, Dopri5, Dopri8, Euler, Fehlberg2, Midpoint, _hip, odeint, odeint_adjoint
from paddlexde_amd.utils import _linf_norm, _rms_norm

from . import problems as P

from ._e2e_common import ADAPTIVE, ConstantLayer, DeepFunc, FIXED, ODEFunc, P_rms, _SmallMLP, _blocks, _golden, _linear, _mlp_foreign, _mlp_numpy  # noqa: F401


def test_linf_norm_and_options(dev):
    A, y0 = _linear(64, 32, torch.float64)
    t = torch.linspace(0.0, 1.0, 4, dtype=torch.float64)
    An = A.numpy()
    opts_o = {"norm": O._linf_norm, "dtype": np.float64, "first_step": 0.01, "max_step": 0.2, "safety": 0.8}
    ref = O.odeint(lambda t_, y: y @ An.T, y0.numpy(), t.numpy(), "dopri5", rtol=1e-6, atol=1e-8, options=opts_o)
    Ad = A.to(dev)
    opts = {"norm": _linf_norm, "dtype": torch.float64, "first_step": 0.01, "max_step": 0.2, "safety": 0.8}
    got = odeint(lambda t_, y: y @ Ad.T, y0.to(dev), t, solver=Dopri5, rtol=1e-6, atol=1e-8, options=opts)
    assert P.parity_ok(got.cpu().numpy(), ref, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("pipeline", ["sync", "lag", "graph", "auto"])
def test_custom_norm_callable(dev, pipeline):
    """A user-supplied norm: err/tol is materialised by xde_error_ratio and the callable runs on it as framework
    ops; its scalar feeds the device controller without visiting the host, so every pipeline serves it (speculative enqueue:
    the kernel takes the operand select; graph: the callable's ops are captured with the step)."""
    A, y0 = _linear(64, 32, torch.float64)
    t = torch.linspace(0.0, 4.0, 4, dtype=torch.float64)
    An = A.numpy()
    ref, so = O.odeint(lambda t_, y: y @ An.T, y0.numpy(), t.numpy(), "dopri5", rtol=1e-6, atol=1e-8,
                       options={"norm": lambda x: 0.5 * np.abs(x).max() + 0.5 * np.sqrt(np.mean(x * x)), "dtype": np.float64},
                       return_solver=True)
    Ad = A.to(dev)
    calls = []

    def my_norm(x):
        calls.append(tuple(x.shape))
        return 0.5 * x.abs().max() + 0.5 * x.pow(2).mean().sqrt()

    from paddlexde_amd.xde import BaseODE

    y0d = y0.to(dev)
    s = Dopri5(xde=BaseODE(lambda t_, y: y @ Ad.T, y0=y0d, t_span=t), y0=y0d, rtol=1e-6, atol=1e-8, norm=my_norm, dtype=torch.float64,
               pipeline=pipeline)
    got = s.integrate(t)
    assert P.parity_ok(got.cpu().numpy(), ref, rtol=1e-9, atol=1e-11)
    assert (s.stats["n_accept"], s.stats["n_reject"], s.stats["nfe"]) == (so.n_accept, so.n_reject, so.nfe)
    assert calls and all(c == (64, 32) for c in calls)
    if pipeline == "sync":
        assert len(calls) == 3 + so.n_accept + so.n_reject  # 3 in select_initial_step + one per attempted step
    # non-finite state: the count rides along in the same kernel and raises the reference's assertion
    bad = y0d.clone()
    bad[3, 5] = float("nan")
    s2 = Dopri5(xde=BaseODE(lambda t_, y: y @ Ad.T, y0=bad, t_span=t), y0=bad, rtol=1e-6, atol=1e-8, norm=my_norm, dtype=torch.float64,
                pipeline=pipeline, first_step=0.01)
    with pytest.raises(AssertionError, match="non-finite"):
        s2.integrate(t)


@pytest.mark.parametrize("pipeline", ["sync", "lag"])
def test_mixed_precision_fp32_state_fp64_time(dev, pipeline):
    """State in float32, time-like scalars in float64 (options["dtype"], base_adaptive_solver_rk.py:47-69): the
    controller runs in double while stage times and the ratio are rounded to the state dtype."""
    A, y0 = _linear(256, 32, torch.float32)
    t = torch.linspace(0.0, 2.0, 9, dtype=torch.float64)
    An = A.numpy()
    ref, so = O.odeint(lambda t_, y: y @ An.T, y0.numpy(), t.numpy(), "dopri5", rtol=1e-4, atol=1e-6,
                       options={"norm": O._rms_norm, "dtype": np.float64}, return_solver=True)
    Ad = A.to(dev)
    from paddlexde_amd.xde import BaseODE

    xde = BaseODE(lambda t_, y: y @ Ad.T, y0=y0.to(dev), t_span=t)
    s = Dopri5(xde=xde, y0=xde.y0, rtol=1e-4, atol=1e-6, norm=_rms_norm, dtype=torch.float64, pipeline=pipeline, record_trace=True)
    got = s.integrate(t)
    assert got.dtype == torch.float32 and ref.dtype == np.float32
    assert P.rel_err(got.cpu().numpy(), ref) <= 1e-5
    assert (s.stats["n_accept"], s.stats["n_reject"]) == (so.n_accept, so.n_reject)
    # dt is a genuine double (not float32-representable) from the second step on
    assert any(float(np.float32(d)) != d for _, d, _, _ in s.trace[1:])


def test_pi_controller_is_opt_in(dev):
    """north_star mentions a PI controller; the reference has a plain I-controller.  Default = reference;
    controller="PI" (Hairer's dopri5 form) changes the step sequence and still meets the tolerance."""
    import scipy.linalg

    from paddlexde_amd.xde import BaseODE

    mu = 5.0
    y0 = torch.tensor([[2.0, 0.0]], dtype=torch.float64).repeat(16, 1).to(dev)
    t = torch.tensor([0.0, 6.0], dtype=torch.float64)
    runs = {}
    for ctl in ("I", "PI"):
        xde = BaseODE(P.vdp_torch(mu), y0=y0, t_span=t)
        s = Dopri5(xde=xde, y0=y0, rtol=1e-7, atol=1e-9, norm=_rms_norm, dtype=torch.float64, controller=ctl, record_trace=True)
        runs[ctl] = (s.integrate(t), s)
    assert Dopri5(xde=BaseODE(P.vdp_torch(mu), y0=y0, t_span=t), y0=y0, rtol=1e-7, atol=1e-9, norm=_rms_norm).controller == "I"
    ref = O.odeint(P.vdp_np(mu), y0.cpu().numpy(), t.numpy(), "dopri5", rtol=1e-10, atol=1e-12, options={"norm": O._rms_norm, "dtype": np.float64})
    for ctl in ("I", "PI"):
        assert P.rel_err(runs[ctl][0].cpu().numpy(), ref) <= 1e-5, ctl
    dts_i = [d for _, d, _, _ in runs["I"][1].trace]
    dts_pi = [d for _, d, _, _ in runs["PI"][1].trace]
    assert dts_i != dts_pi  # the PI law really is active
    assert runs["PI"][1].stats["n_reject"] <= runs["I"][1].stats["n_reject"]  # smoother step-size sequence


def test_repeated_start_time_rows(dev):
    """t_span = [t0, t0, t1]: the reference's loop takes no step for the second row (base_adaptive_solver_rk.py:119) and
    then evaluates its interpolant on the empty interval [t0, t0] — 0/0, a NaN row (ode_utils.py:65-68; torchdiffeq, whose
    loop this is, rejects such grids up front).  Deliberate deviation: the row is y0, the only value it can mean."""
    A, y0 = _linear(8, 4, torch.float64)
    Ad = A.to(dev)
    t = torch.tensor([0.0, 0.0, 0.5], dtype=torch.float64)
    got = odeint(lambda t_, y: y @ Ad.T, y0.to(dev), t, solver=Dopri5, rtol=1e-8, atol=1e-10, options={"norm": _rms_norm, "dtype": torch.float64})
    ref = O.odeint(lambda t_, y: y @ A.numpy().T, y0.numpy(), np.array([0.0, 0.5]), "dopri5", rtol=1e-8, atol=1e-10,
                   options={"norm": O._rms_norm, "dtype": np.float64})
    assert torch.equal(got[0].cpu(), y0) and torch.equal(got[1].cpu(), y0)
    assert P.parity_ok(got[2].cpu().numpy(), ref[1], 1e-9, 1e-11)
    same = odeint(lambda t_, y: y @ Ad.T, y0.to(dev), torch.zeros(3, dtype=torch.float64), solver=Dopri5, options={"norm": _rms_norm})
    assert all(torch.equal(same[i].cpu(), y0) for i in range(3))


def test_step_t_option(dev):
    A, y0 = _linear(16, 8, torch.float64)
    t = torch.linspace(0.0, 1.0, 3, dtype=torch.float64)
    An = A.numpy()
    st = [0.13, 0.61, 0.4]
    ref, so = O.odeint(lambda t_, y: y @ An.T, y0.numpy(), t.numpy(), "dopri5", rtol=1e-6, atol=1e-8,
                       options={"norm": O._rms_norm, "dtype": np.float64, "step_t": st}, return_solver=True)
    Ad = A.to(dev)
    got = odeint(lambda t_, y: y @ Ad.T, y0.to(dev), t, solver=Dopri5, rtol=1e-6, atol=1e-8,
                 options={"norm": _rms_norm, "dtype": torch.float64, "step_t": st})
    assert P.parity_ok(got.cpu().numpy(), ref, rtol=1e-9, atol=1e-11)


# ----------------------------------------------------------------------------------------------
# error conventions
# ----------------------------------------------------------------------------------------------
def test_assertion_messages(dev):
    y0 = torch.ones(4, 2, device=dev)
    t = torch.tensor([0.0, 1.0], device=dev)
    with pytest.raises(AssertionError, match="max_num_steps exceeded"):
        odeint(P.vdp_torch(1000.0), y0 * 2, t, solver=Dopri5, rtol=1e-6, atol=1e-8, options={"norm": _rms_norm, "max_num_steps": 5})
    bad = y0.clone()
    bad[1, 1] = float("nan")
    with pytest.raises(AssertionError, match="non-finite values in state"):
        odeint(lambda t_, y: -y, bad, t, solver=Dopri5, options={"norm": _rms_norm, "first_step": 0.1})
    # without first_step the NaN reaches dt through select_initial_step and the reference's FIRST assertion fires
    with pytest.raises(AssertionError, match="underflow in dt nan"):
        odeint(lambda t_, y: -y, bad, t, solver=Dopri5, options={"norm": _rms_norm})
    with pytest.raises(AssertionError, match="underflow in dt"):
        odeint(lambda t_, y: y * float("inf"), y0, t, solver=Dopri5, options={"norm": _rms_norm})
    with pytest.raises(KeyError):
        RK4(xde=__import__("paddlexde_amd").BaseODE(lambda t_, y: y, y0=y0, t_span=t), y0=y0)
    if str(dev).startswith("cuda"):
        with pytest.raises(_hip.XdeError):
            odeint(lambda t_, y: -y, torch.ones(4, 2), torch.tensor([0.0, 1.0]), solver=Dopri5)  # CPU tensors: no fallback


def test_empty_batch_and_single_output_time(dev):
    """Degenerate inputs.  Fixed solvers: as the reference (an empty batch gives an empty solution, one output time gives the
    state back — oracle checked).  Adaptive solvers: the reference trips over itself there (the RMS norm of an empty state is NaN ->
    "underflow in dt nan"; one output time -> IndexError on t_span[1], base_adaptive_solver.py) — here both are served: no launch
    has a zero-sized grid (every C entry point returns XDE_OK for n == 0), the solution is the empty / one-row tensor."""
    f = lambda t_, y: -y  # noqa: E731
    t = torch.linspace(0.0, 1.0, 5, device=dev)
    tn = t.cpu().numpy()
    for name in ("euler", "rk4"):
        got = odeint(f, torch.zeros(0, 3, device=dev), t, solver=FIXED[name])
        want = O.odeint(f, np.zeros((0, 3), np.float32), tn, name)
        assert tuple(got.shape) == want.shape == (0, 3)
        got = odeint(f, torch.ones(2, 3, device=dev), t[:1], solver=FIXED[name])
        want = O.odeint(f, np.ones((2, 3), np.float32), tn[:1], name)
        assert np.array_equal(got.cpu().numpy(), want)
    for pipeline in ("sync", "lag", "graph", "auto"):
        opts = {"norm": _rms_norm, "pipeline": pipeline}
        got = odeint(f, torch.zeros(0, 3, device=dev), t, solver=Dopri5, options=dict(opts))
        assert tuple(got.shape) == (5, 0, 3) and got.dtype == torch.float32
        y0 = torch.full((2, 3), 1.5, device=dev)
        got = odeint(f, y0, t[:1], solver=Dopri5, options=dict(opts))
        assert tuple(got.shape) == (1, 2, 3) and torch.equal(got[0], y0)
    with pytest.raises(AssertionError, match="underflow in dt nan"):  # (what the reference does with the empty batch)
        O.odeint(f, np.zeros((0, 3), np.float32), tn, "dopri5")
    # ... and through the adjoint: nothing flows back from an empty batch or from the initial state alone
    for solver, y_shape, tt in ((Dopri5, (0, 3), t), (Dopri5, (2, 3), t[:1]), (RK4, (0, 1, 3), t), (RK4, (2, 1, 3), t[:1])):
        layer = nn.Linear(3, 3).to(dev)
        y0 = torch.ones(y_shape, device=dev, requires_grad=True)
        sol = odeint_adjoint(lambda t_, y: layer(y), y0, tt, solver=solver, adjoint_params=tuple(layer.parameters()), options={"norm": _rms_norm})
        sol.sum().backward()
        assert tuple(y0.grad.shape) == y_shape and all(float(p.grad.abs().sum()) == 0.0 for p in layer.parameters())
        if 0 not in y_shape:
            assert torch.equal(y0.grad, torch.ones_like(y0))  # d sum(y0) / d y0


@pytest.mark.parametrize("pipeline", ["sync", "lag", "graph"])
def test_pi_controller_vs_its_cpu_statement(dev, pipeline):
    """The opt-in PI controller (not in the reference) against oracle.optimal_step_size_pi: same step sequence — every
    (t0, dt, ratio, accept) — and the same solution, on a problem with rejections, fp64."""
    from paddlexde_amd.xde import BaseODE

    mu = 30.0
    y0 = (torch.tensor([[2.0, 0.0]], dtype=torch.float64) + 0.01 * torch.randn(16, 2, generator=torch.Generator().manual_seed(0), dtype=torch.float64))
    t = torch.linspace(0.0, 3.0, 5, dtype=torch.float64)
    for beta in (0.04, 0.08):
        ref, so = O.odeint(P.vdp_np(mu), y0.numpy(), t.numpy(), "dopri5", rtol=1e-7, atol=1e-9,
                           options={"norm": O._rms_norm, "dtype": np.float64, "controller": "PI", "pi_beta": beta}, return_solver=True)
        s = Dopri5(xde=BaseODE(P.vdp_torch(mu), y0=y0.to(dev), t_span=t), y0=y0.to(dev), rtol=1e-7, atol=1e-9, norm=_rms_norm, dtype=torch.float64,
                   controller="PI", pi_beta=beta, pipeline=pipeline, record_trace=True)
        got = s.integrate(t).cpu().numpy()
        assert so.n_reject > 0
        assert (s.stats["n_accept"], s.stats["n_reject"]) == (so.n_accept, so.n_reject)
        assert P.parity_ok(got, ref, rtol=1e-9, atol=1e-11), P.worst(got, ref, 1e-9, 1e-11)
        mine = np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace])
        theirs = np.asarray([[r.t0, r.dt, r.ratio, float(r.accept)] for r in so.trace])
        assert mine.shape == theirs.shape
        assert np.array_equal(mine[:, 3], theirs[:, 3])
        assert np.allclose(mine[:, :2], theirs[:, :2], rtol=1e-8, atol=1e-12)
        # (the error ratio is a cancellation: ulp differences between numpy's and the device's func show up at 1e-7)
        assert np.allclose(mine[:, 2], theirs[:, 2], rtol=1e-6, atol=1e-12)


def test_output_times_behind_the_previous_one_are_refused_like_the_reference(dev):
    """The reference evaluates each output time on the step that has just reached it; a time BEHIND the previous one lies outside
    that step and trips `interp_evaluate`'s assertion (utils/ode_utils.py:65-67) — so does the oracle.  The device controller could
    extrapolate instead; the host refuses first, with the same exception type and message shape.  Equal consecutive times are fine,
    and an empty state integrates to empty rows."""
    y0 = torch.ones(3, 2, device=dev)
    f = lambda t, y: -y  # noqa: E731
    for bad in ([0.0, 1.0, 0.5], [1.0, 0.2, 0.6], [0.0, 0.0, -1.0]):
        with pytest.raises(AssertionError, match="invalid interpolation"):
            odeint(f, y0, torch.tensor(bad), solver=Dopri5, rtol=1e-5, atol=1e-7)
        with pytest.raises(AssertionError, match="invalid interpolation"):
            O.odeint(lambda t, y: -y, np.ones((3, 2), dtype=np.float32), np.asarray(bad, dtype=np.float32), "dopri5", rtol=1e-5, atol=1e-7)
    ok = odeint(f, y0, torch.tensor([0.0, 0.5, 0.5, 1.0]), solver=Dopri5, rtol=1e-6, atol=1e-8)
    assert torch.equal(ok[1], ok[2]) and abs(float(ok[3, 0, 0]) - np.exp(-1.0)) < 1e-5
    empty = odeint(f, torch.zeros(0, 2, device=dev), torch.tensor([0.0, 0.5, 1.0]), solver=Dopri5)
    assert tuple(empty.shape) == (3, 0, 2)


def test_reuse_f0_calls_func_once_less_and_changes_nothing_else(dev):
    """The reference evaluates func(t0, y0) twice before the first attempt (base_adaptive_solver_rk.py:83 and, with f0=None, :84-87).
    `reuse_f0=True` hands the first value to the initial-step heuristic: one call less, the same solution bit for bit, the same step
    trace; `nfe` is the number of calls func really received, `nfe_reference` what the reference would report.  odeint_adjoint
    switches it on for its backward intervals (and lets the caller switch it off); plain odeint leaves it off."""
    from paddlexde_amd.xde import BaseODE

    A = P.skew_matrix(8).double().to(dev)
    y0 = torch.randn(16, 8, generator=torch.Generator().manual_seed(2), dtype=torch.float64).to(dev)
    t = torch.linspace(0.0, 1.0, 4, dtype=torch.float64)
    calls = [0]

    def f(t_, y):
        calls[0] += 1
        return y @ A.T

    out = {}
    for reuse in (False, True):
        calls[0] = 0
        s = Dopri5(xde=BaseODE(f, y0=y0, t_span=t), y0=y0, rtol=1e-7, atol=1e-9, norm=_rms_norm, dtype=torch.float64, record_trace=True,
                   pipeline="sync", reuse_f0=reuse)
        sol = s.integrate(t)
        out[reuse] = (sol.clone(), list(s.trace), s.stats["nfe"], calls[0], s.stats["nfe_reference"])
    assert torch.equal(out[False][0], out[True][0]) and out[False][1] == out[True][1]
    assert out[False][2] == out[False][3] and out[True][2] == out[True][3]  # nfe = the calls func received, either way
    assert out[True][3] == out[False][3] - 1
    assert out[False][4] == out[True][4] == out[False][3]  # nfe_reference = the calls the reference makes

    m = ODEFunc(torch.float64).to(dev)
    counted = [0]
    hook = m.register_forward_hook(lambda *_: counted.__setitem__(0, counted[0] + 1))
    yg = (torch.rand(32, 2, generator=torch.Generator().manual_seed(0), dtype=torch.float64) * 4 - 2).to(dev)
    tt = torch.linspace(0.0, 1.0, 5, dtype=torch.float64).to(dev)
    grads, n_calls = {}, {}
    for reuse in (None, False):
        for p_ in m.parameters():
            p_.grad = None
        y = yg.clone().requires_grad_(True)
        adj = {"dtype": torch.float64, "graph_func": False}
        if reuse is not None:
            adj["reuse_f0"] = reuse
        sol = odeint_adjoint(m, y, tt, solver=Dopri5, rtol=1e-7, atol=1e-9, options={"norm": _rms_norm, "dtype": torch.float64}, adjoint_options=adj)
        counted[0] = 0
        sol.abs().mean().backward()
        grads[reuse] = [y.grad.clone()] + [p_.grad.clone() for p_ in m.parameters()]
        n_calls[reuse] = counted[0]
    hook.remove()
    for a, b in zip(grads[None], grads[False]):
        assert torch.equal(a, b)
    assert n_calls[None] == n_calls[False] - (len(tt) - 1)  # one evaluation of the augmented dynamics less per output interval


def test_step_callbacks_adaptive_and_fixed(dev):
    """The reference names three callbacks in its adaptive stepper and leaves the calls commented out
    (`self.func.callback_step(t0, y0, dt)` / `callback_accept_step` / `callback_reject_step`, base_adaptive_solver_rk.py:186,259,275), and
    binds `xde.on_integrate_step_end` in the fixed-step loop without calling it (base_fixed_solver.py:64, xde/base_xde.py:102-103).
    Here they are live: as options, or as methods of the user's func; every attempt announces itself with the (t0, dt) it runs
    with — equal to the recorded trace — and exactly one of accept / reject follows; the results are those of a solve without
    callbacks, bit for bit; pipelines that enqueue ahead of the verdicts refuse them."""
    from paddlexde_amd import Dopri5, RK4
    from paddlexde_amd.xde import BaseODE

    mu = 30.0
    f = P.vdp_torch(mu)
    y0 = (torch.tensor([2.0, 0.0], dtype=torch.float64) + 0.01 * torch.randn(64, 2, generator=torch.Generator().manual_seed(0), dtype=torch.float64)).to(dev)
    t = torch.tensor([0.0, 0.4, 1.1], dtype=torch.float64)
    log = []
    opts = {"norm": P_rms(), "dtype": torch.float64, "record_trace": True,
            "callback_step": lambda t0, y, dt: log.append(("step", float(t0), float(dt), tuple(y.shape), t0.dtype, t0.device.type)),
            "callback_accept_step": lambda t0, y, dt: log.append(("accept", float(t0), float(dt))),
            "callback_reject": lambda t0, y, dt: log.append(("reject", float(t0), float(dt)))}
    xde = BaseODE(f, y0=y0, t_span=t)
    s = Dopri5(xde=xde, y0=y0, rtol=1e-6, atol=1e-8, **opts)
    sol = s.integrate(t)
    assert s.pipeline == "sync"  # "auto" resolved to the pipeline that can call back
    steps = [e for e in log if e[0] == "step"]
    verdicts = [e for e in log if e[0] != "step"]
    assert len(steps) == len(verdicts) == len(s.trace) == s.stats["n_steps"] and s.stats["n_reject"] > 0
    for st, vd, (t0, dt, _ratio, acc) in zip(steps, verdicts, s.trace):
        assert st[1] == vd[1] == t0 and st[2] == vd[2] == dt and vd[0] == ("accept" if acc else "reject")
        assert st[3] == (64, 2) and st[4] == torch.float64 and st[5] == "cpu"
    assert [e[0] for e in log[:2]] == ["step", log[1][0]] and log[1][0] in ("accept", "reject")  # interleaved: step, verdict, step, ...
    plain = Dopri5(xde=BaseODE(f, y0=y0, t_span=t), y0=y0, rtol=1e-6, atol=1e-8, norm=P_rms(), dtype=torch.float64, pipeline="sync").integrate(t)
    assert torch.equal(sol, plain)
    # methods of the user's func, as the reference's comments spell them
    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.seen = []

        def forward(self, t_, y):
            return f(t_, y)

        def callback_step(self, t0, y, dt):
            self.seen.append(float(t0))

    fm = F()
    s2 = Dopri5(xde=BaseODE(fm, y0=y0, t_span=t), y0=y0, rtol=1e-6, atol=1e-8, norm=P_rms(), dtype=torch.float64)
    with torch.no_grad():
        sol2 = s2.integrate(t)
    assert torch.equal(sol2, plain) and len(fm.seen) == s2.stats["n_steps"]
    for pipeline in ("lag", "graph"):
        with pytest.raises(NotImplementedError, match="callbacks"):
            Dopri5(xde=BaseODE(f, y0=y0, t_span=t), y0=y0, rtol=1e-6, atol=1e-8, norm=P_rms(), pipeline=pipeline, callback_step=lambda *a: None)

    # fixed-step: xde.on_integrate_step_end(y0, y1, t0, t1) after every step of the eager loop
    class Watched(BaseODE):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.ends = []

        def on_integrate_step_end(self, y0=None, y1=None, t0=None, t1=None):
            self.ends.append((float(t0), float(t1), y0.clone(), y1.clone()))

    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device=dev)
    g = lambda t_, y: (y * y * y) @ A  # noqa: E731
    ys = torch.tensor([[2.0, 0.0]], device=dev)
    tf = torch.linspace(0.0, 1.0, 41)
    xw = Watched(g, y0=ys, t_span=tf)
    with torch.no_grad():
        out = RK4(xde=xw, y0=ys, rtol=1e-7, atol=1e-9, norm=P_rms()).integrate(tf)  # 40 steps: "auto" would capture; the hook keeps it eager
        ref = RK4(xde=BaseODE(g, y0=ys, t_span=tf), y0=ys, rtol=1e-7, atol=1e-9, norm=P_rms()).integrate(tf)
    assert torch.equal(out, ref) and len(xw.ends) == 40
    for i, (t0, t1, a, b) in enumerate(xw.ends):
        assert t0 == float(tf[i]) and t1 == float(tf[i + 1]) and torch.equal(a, out[i : i + 1]) and torch.equal(b, out[i + 1 : i + 2])
    with pytest.raises(NotImplementedError, match="on_integrate_step_end"):
        RK4(xde=xw, y0=ys, rtol=1e-7, atol=1e-9, norm=P_rms(), pipeline="graph")
