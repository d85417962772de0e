"""Replay-mode parity (SURVEY section 7 "hard parts" (i)): the device controller is made to take the ORACLE's exact
(dt, accept) sequence (xde_ctrl_params_t.replay), so that accept/reject chaos is out of the picture and fp32
ARITHMETIC is held to north_star's element-wise bar `|got - ref| <= 1e-7 + 1e-5 |ref|` on every attempted step's
y1, with the error ratio of every attempt compared as well.

Replayed logic: AdaptiveRKSolver._adaptive_step / _runge_kutta_step (solver/base_adaptive_solver_rk.py:129-284),
compute_error_ratio (utils/ode_utils.py:80-82).  Collected by test_gpu_odeint.py (HIP) and test_host_logic.py (CPU double).

What the ratio bar can be.  The error estimate `sum_j k_j (dt c_err_j)` is a cancellation: its terms are O(dt |f|) and
the sum is O(rtol |y|), so rounding differences of one ulp in the k_j (a different GEMM summation order in `func`)
appear in the estimate amplified by dt|f| / (rtol |y|) ~ 10^2..10^3.  The per-test `ratio_rtol` states what that
allows; where `func` is element-wise (+, -, * only: Van der Pol) both sides compute bit-identical k_j and the ratio
bar is the reduction-order bar (1e-5).
"""
import numpy as np
import pytest
import torch

from oracle import xde_oracle as O
from paddlexde_amd import Dopri5
from paddlexde_amd.utils import _rms_norm
from paddlexde_amd.xde import BaseODE

from . import problems as P
from ._e2e_common import ODEFunc, _mlp_numpy


def _oracle_run(f_np, y0, t, *, rtol, atol, options=None):
    """Free-running oracle in the default precision, recording every attempt's y1."""
    states = []
    opts = dict({"norm": O._rms_norm}, **(options or {}))
    opts["step_hook"] = lambda i, y0_, y1_, ratio, acc: states.append(np.array(y1_, copy=True))
    ref, so = O.odeint(f_np, y0, t, "dopri5", rtol=rtol, atol=atol, options=opts, return_solver=True)
    return ref, so, states


def _replay_run(f_t, y0, t, so, dev, *, rtol, atol, pipeline="sync", hook=True, **opts):
    states = []

    def on_attempt(i, y0_, y1_, ks, c):
        states.append(y1_.detach().cpu().numpy().copy())

    y0d = torch.from_numpy(y0).to(dev)
    tt = torch.from_numpy(t)
    s = Dopri5(xde=BaseODE(f_t, y0=y0d, t_span=tt), y0=y0d, rtol=rtol, atol=atol, norm=_rms_norm, pipeline=pipeline,
               record_trace=True, _replay=[(r.dt, r.accept) for r in so.trace],
               _step_hook=on_attempt if (hook and pipeline == "sync") else None, **opts)
    got = s.integrate(tt).cpu().numpy()
    return got, s, states


def _check(got, s, states, ref, so, ref_states, *, ratio_rtol, pipeline, ratio_atol=0.0, y_atol=1e-7, dense_atol=None, label=None):
    theirs = np.asarray([[r.t0, r.dt, r.ratio, float(r.accept)] for r in so.trace])
    mine = np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace])
    assert mine.shape == theirs.shape, (mine.shape, theirs.shape)
    assert np.array_equal(mine[:, 3], theirs[:, 3])  # the prescribed verdicts were taken
    assert np.array_equal(mine[:, 1], theirs[:, 1])  # ... with the prescribed steps, bit for bit
    assert np.allclose(mine[:, 0], theirs[:, 0], rtol=1e-6, atol=0)  # t0: fp32 sums of the same dt's in the same order
    assert (s.stats["n_accept"], s.stats["n_reject"], s.stats["nfe"]) == (so.n_accept, so.n_reject, so.nfe)
    # the error ratio of EVERY attempt (not only the accepted ones)
    rr = np.abs(mine[:, 2] - theirs[:, 2]) / (ratio_atol + ratio_rtol * np.abs(theirs[:, 2]))
    assert rr.max() <= 1.0, ("ratio", float(rr.max()), int(rr.argmax()), mine[rr.argmax(), 2], theirs[rr.argmax(), 2])
    # per-step y1, element-wise at the north-star bar, fp32
    if pipeline == "sync":
        assert len(states) == len(ref_states)
        for i, (a, b) in enumerate(zip(states, ref_states)):
            assert a.dtype == b.dtype == np.float32
            assert P.parity_ok(a, b, rtol=1e-5, atol=y_atol), ("y1 of attempt", i, P.worst(a, b, 1e-5, y_atol))
    # the emitted solution rows (dense output of the same steps), same bar
    assert got.dtype == ref.dtype == np.float32
    dense_atol = y_atol if dense_atol is None else dense_atol
    assert P.parity_ok(got, ref, rtol=1e-5, atol=dense_atol), ("solution", P.worst(got, ref, 1e-5, dense_atol))
    if label is not None:  # what was actually observed, next to the bar (print the worst ulp count)
        P.report(label, {"pipeline": pipeline, "attempts": int(mine.shape[0]),
                         "worst_ulps_near_zero_y1": max([P.worst_ulps(a, b) for a, b in zip(states, ref_states)], default=None) if pipeline == "sync" else None,
                         "worst_ulps_near_zero_rows": P.worst_ulps(got, ref), "bar_fraction_rows": P.worst(got, ref, 1e-5, dense_atol),
                         "allowed_ulps_y1": y_atol / float(np.spacing(np.float32(np.abs(ref).max()))),
                         "allowed_ulps_rows": dense_atol / float(np.spacing(np.float32(np.abs(ref).max()))),
                         "worst_ratio_rel": float((np.abs(mine[:, 2] - theirs[:, 2]) / np.maximum(np.abs(theirs[:, 2]), 1e-300)).max())})


@pytest.mark.parametrize("pipeline", ["sync", "lag", "graph"])
def test_replay_config2_shape_fp32(dev, pipeline):
    """Config-2-shaped: linear ODE dy/dt = A y, 4096 x 128, Dopri5, rtol 1e-5 / atol 1e-7, fp32, 11 output times."""
    A = P.skew_matrix(128).float()
    y0 = torch.randn(4096, 128, generator=torch.Generator().manual_seed(0)).numpy()
    t = np.linspace(0.0, 1.0, 11).astype(np.float32)
    An, Ad = A.numpy(), A.to(dev)
    ref, so, ref_states = _oracle_run(lambda t_, y: y @ An.T, y0, t, rtol=1e-5, atol=1e-7)
    got, s, states = _replay_run(lambda t_, y: y @ Ad.T, y0, t, so, dev, rtol=1e-5, atol=1e-7, pipeline=pipeline)
    # func is a 128-term GEMM: numpy's and the device's summation orders differ by ulps in every k_j.  Relative part of the
    # bar: 1e-5, strictly.  Absolute part (elements passing through zero; the state is O(4)): 2 ulp of the state's scale for
    # every step's y1, 4 ulp for the emitted rows — the dense-output quartic (ode_utils.py:28-49) forms its coefficients from
    # differences like 18 y0 + 14 y1 - 32 y_mid, which amplify a last-bit difference of their inputs (P.ulp_atol).
    _check(got, s, states, ref, so, ref_states, ratio_rtol=2e-2, ratio_atol=1e-4, pipeline=pipeline,
           y_atol=P.ulp_atol(ref, 2), dense_atol=P.ulp_atol(ref, 4), label="replay_config2_shape_fp32")


@pytest.mark.parametrize("controller", ["I", "PI"])
@pytest.mark.parametrize("pipeline", ["sync", "graph"])
def test_replay_config5_vdp_fp32(dev, controller, pipeline):
    """Config 5 at full size in fp32: stiff Van der Pol mu = 1000, 4096 x 2, t in [0, 1], rtol 1e-5 / atol 1e-7, the
    reference's I controller and the PI controller BASELINE.json names — ~1200 attempts, hundreds rejected."""
    mu = 1000.0
    y0 = (np.array([2.0, 0.0]) + 0.01 * torch.randn(4096, 2, generator=torch.Generator().manual_seed(0)).double().numpy()).astype(np.float32)
    t = np.array([0.0, 1.0], dtype=np.float32)
    copts = {"controller": controller, "max_num_steps": 10**6}
    ref, so, ref_states = _oracle_run(P.vdp_np(np.float32(mu)), y0, t, rtol=1e-5, atol=1e-7, options=copts)
    assert so.n_reject > 100 and so.n_accept > 500
    got, s, states = _replay_run(P.vdp_torch(mu), y0, t, so, dev, rtol=1e-5, atol=1e-7, pipeline=pipeline, controller=controller,
                                 max_num_steps=10**6)
    # element-wise func: identical k_j on both sides, the ratio differs only by the reduction order of 8192 squares
    _check(got, s, states, ref, so, ref_states, ratio_rtol=1e-5, pipeline=pipeline, label="replay_config5_vdp_fp32/" + controller)


def test_replay_config3_forward_fp32(dev):
    """Config 3's forward solve (the solve odeint_adjoint saves for its backward): spiral neural ODE, 2-layer MLP on y**3,
    8192 x 2, 32 output times, fp32."""
    m = ODEFunc(torch.float32)
    fn, _, _ = _mlp_numpy(m)
    m = m.to(dev)
    y0 = (torch.rand(8192, 2, generator=torch.Generator().manual_seed(0)) * 4 - 2).numpy()
    t = np.linspace(0.0, 25.0, 1000).astype(np.float32)[:32]
    ref, so, ref_states = _oracle_run(lambda t_, y: fn(t_, y).astype(np.float32), y0, t, rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        got, s, states = _replay_run(lambda t_, y: m(t_, y), y0, t, so, dev, rtol=1e-5, atol=1e-7)
    # tanh and two GEMMs differ by ulps between numpy and the device.  The state is O(2) and many of its 16384 elements pass
    # through zero: there the bar's absolute part, 1e-7, is below ONE fp32 ulp of the quantities the element was summed from
    # (ulp(2) = 2.4e-7), so the absolute part is 2 ulp of the state's scale here; the relative part stays 1e-5.
    _check(got, s, states, ref, so, ref_states, ratio_rtol=2e-2, ratio_atol=1e-4, pipeline="sync", y_atol=P.ulp_atol(ref, 2),
           dense_atol=P.ulp_atol(ref, 4), label="replay_config3_forward_fp32")


def test_replay_table_shorter_than_the_solve(dev):
    """After the table is exhausted the controller decides again (the first attempts are still the prescribed ones)."""
    A = P.skew_matrix(16).double()
    y0 = torch.randn(8, 16, generator=torch.Generator().manual_seed(3)).double().numpy()
    t = np.linspace(0.0, 2.0, 3)
    An, Ad = A.numpy(), A.to(dev)
    ref, so = O.odeint(lambda t_, y: y @ An.T, y0, t, "dopri5", rtol=1e-6, atol=1e-8, options={"norm": O._rms_norm, "dtype": np.float64},
                       return_solver=True)
    y0d = torch.from_numpy(y0).to(dev)
    forced = [(0.01, True), (0.02, False), (0.005, True)]
    s = Dopri5(xde=BaseODE(lambda t_, y: y @ Ad.T, y0=y0d, t_span=torch.from_numpy(t)), y0=y0d, rtol=1e-6, atol=1e-8, norm=_rms_norm,
               dtype=torch.float64, record_trace=True, _replay=forced)
    got = s.integrate(torch.from_numpy(t)).cpu().numpy()
    tr = s.trace
    assert [(round(b, 12), d) for _, b, _, d in tr[:3]] == [(0.01, True), (0.02, False), (0.005, True)]
    assert tr[1][0] == tr[2][0] == 0.01  # the rejected attempt did not advance time
    assert len(tr) > 3 and all(r[2] <= 1.0 for r in tr[3:] if r[3])  # free-running again: accepted <=> ratio <= 1
    assert P.rel_err(got, ref) <= 1e-5  # still a valid integration of the same problem


def test_replay_config3_adjoint_gradients_fp32(dev):
    """Config 3's BACKWARD in fp32, replayed: every interval's augmented solve follows the oracle's (dt, accept) sequence for that
    interval (functional/odeint_adjoint.py:134-159), the forward solve the oracle's forward sequence; d loss / d y0 and the 252
    parameter gradients are then held element-wise to 1e-5 |ref| + 32 ulp of each gradient's scale = 4e-6 of its max-norm (tanh,
    GEMMs and fp32 row sums over 1024 rows x dozens of stage evaluations differ by roundings between numpy and the device; round
    1's bar here was 1e-5 of the max-norm, without the element-wise relative part)."""
    from paddlexde_amd import odeint_adjoint

    dtype = torch.float32
    m = ODEFunc(dtype)
    fn, vjp, params = _mlp_numpy(m)
    m = m.to(dev)
    y0 = torch.rand(1024, 2, generator=torch.Generator().manual_seed(0)) * 4 - 2
    t = torch.linspace(0.0, 25.0, 1000)[:12]
    f32 = lambda f: (lambda *a: np.asarray(f(*a), dtype=np.float32))  # noqa: E731
    vjp32 = lambda t_, y, c: (lambda r: (r[0].astype(np.float32), [g.astype(np.float32) for g in r[1]]))(vjp(t_, y, c))  # noqa: E731
    ans, bw = O.odeint_adjoint(f32(fn), vjp32, params, y0.numpy(), t.numpy(), "dopri5", rtol=1e-5, atol=1e-7,
                               options={"norm": O._rms_norm, "dtype": np.float32})
    # the forward trace of the same solve
    _, so = O.odeint(f32(fn), y0.numpy(), t.numpy(), "dopri5", rtol=1e-5, atol=1e-7, options={"norm": O._rms_norm}, return_solver=True)
    gy0, gps = bw((np.sign(ans) / ans.size).astype(np.float32))
    assert len(bw.traces) == len(t) - 1 and sum(len(tr) for tr in bw.traces) >= len(t) - 1

    y0g = y0.clone().to(dev).requires_grad_(True)
    sol = odeint_adjoint(m, y0g, t.to(dev), solver=Dopri5, rtol=1e-5, atol=1e-7,
                         options={"norm": _rms_norm, "_replay": [(r.dt, r.accept) for r in so.trace]},
                         adjoint_options={"graph_func": False, "_replay_intervals": [[(r.dt, r.accept) for r in tr] for tr in bw.traces]})
    sol.abs().mean().backward()
    assert P.parity_ok(sol.detach().cpu().numpy(), ans, rtol=1e-5, atol=P.ulp_atol(ans, 4)), P.worst(sol.detach().cpu().numpy(), ans, 1e-5, P.ulp_atol(ans, 4))
    pairs = [("dL/dy0", y0g.grad.cpu().numpy(), gy0)] + [("dL/d" + n, p_.grad.cpu().numpy(), g_) for (n, p_), g_ in zip(m.named_parameters(), gps)]
    for name, got, ref in pairs:
        atol = P.ulp_atol(ref, 32, floor=0.0)  # (gradients are O(1e-3): ulps of THEIR scale, no 1e-7 floor)
        assert P.parity_ok(got, ref, rtol=1e-5, atol=atol), (name, float(np.abs(ref).max()), P.worst(got, ref, 1e-5, atol))
