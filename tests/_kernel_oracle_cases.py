"""Kernel level, against the ORACLE's own functions (oracle/xde_oracle.py — not the builder's numpy contract of the kernels):
one embedded Runge-Kutta attempt taken apart.  The oracle's `_runge_kutta_step` supplies the stage derivatives in its
stage-innermost `k[..., S+1]` buffer; every stage input, the error estimate, the error ratio, the controller's next step and
the dense-output rows the kernels produce from the same `k_j` (as separate SoA tensors, zero coefficients skipped, error
estimate split between the last stage's second output and the norm pass) must equal what the oracle's code computes:
element-wise results BIT for bit, the reduction to fp64-accumulation accuracy.

Reference lines restated by the oracle functions used: `_runge_kutta_step` solver/base_adaptive_solver_rk.py:129-181,
`compute_error_ratio` utils/ode_utils.py:80-82, `optimal_step_size` :85-97, `interp_fit` / `interp_evaluate` :28-77.
Collected by test_gpu_odeint.py (HIP kernels) and test_host_logic.py (CPU double)."""
import numpy as np
import pytest
import torch

from oracle import xde_oracle as O
from paddlexde_amd import _hip
from paddlexde_amd.solver.base_adaptive_solver_rk import _build_plans
from paddlexde_amd.solver.adaptive_solver import Bosh3, Dopri5, Dopri8

from . import problems as P

CASES = {"dopri5": Dopri5, "bosh3": Bosh3, "dopri8": Dopri8}


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("name", list(CASES))
def test_one_attempt_taken_apart_vs_oracle(dev, name, dtype):
    be = _hip.get_backend()
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    rng = np.random.RandomState(12)
    n = 4099  # not a multiple of the vector width: the scalar tail runs too
    y0 = rng.uniform(-2, 2, size=(n, 2)).astype(dtype)
    f_np = P.vdp_np(dtype(3.0))  # +, -, * only: bit-identical on both sides
    t0, dt = dtype(0.25), dtype(0.0371)
    rtol, atol = 1e-3, 1e-5
    so = O.AdaptiveRKSolver(f_np, y0, rtol, atol, method=name, norm=O._rms_norm, dtype=dtype)
    f0 = so.move(t0, 0, y0)
    with np.errstate(all="ignore"):
        y1_ref, f1_ref, err_ref, k = so._runge_kutta_step(y0, f0, t0, dt, t0 + dt, so.tableau)
    S = k.shape[-1] - 1
    cls = CASES[name]
    n_stage, stage_plan, fsal, sol_plan, err_plan, mid_plan, fuse_err, err2_coef, _ = _build_plans(cls.tableau, cls.mid)
    assert n_stage == S
    mv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ks = [mv(k[..., j]) for j in range(S + 1)]  # SoA: one tensor per stage derivative
    y0d = mv(y0)

    # ---- every stage input: y_i = y0 + sum_j k_j (beta_ij dt)                                        :166-168
    ebuf = torch.empty_like(y0d)
    y_last = None
    for i in range(S):
        idx, coef = stage_plan[i]
        out = torch.empty_like(y0d)
        last = i == S - 1
        be.stage_combine(out, y0d, [ks[j] for j in idx], coef, _hip.COMBINE_RK, dt_host=float(dt),
                         out2=ebuf if (last and fuse_err) else None, coef2=err2_coef if (last and fuse_err) else None)
        want = y0 + O._sum_last(k[..., : i + 1] * (so.tableau.beta[i] * dt)).reshape(y0.shape)
        assert np.array_equal(out.cpu().numpy(), want), (name, "stage", i)
        y_last = out
    if fsal:
        y1d = y_last
    else:
        idx, coef = sol_plan
        y1d = torch.empty_like(y0d)
        be.stage_combine(y1d, y0d, [ks[j] for j in idx], coef, _hip.COMBINE_RK, dt_host=float(dt))
    assert np.array_equal(y1d.cpu().numpy(), y1_ref), (name, "y1")

    # ---- error ratio: estimate + tolerance + norm                                   :180; ode_utils.py:80-82
    p = _hip.XdeCtrlParams()
    p.rtol, p.atol, p.min_step, p.max_step = float(so.rtol), float(so.atol), 0.0, float("inf")
    p.safety, p.ifactor, p.dfactor, p.order = float(so.safety), float(so.ifactor), float(so.dfactor), float(so.order)
    p.max_num_steps = 2**31 - 1
    p.time_dtype = p.state_dtype = _hip.XDE_F32 if dtype == np.float32 else _hip.XDE_F64
    p.direction, p.norm_kind, p.n_stage, p.n_seg = 1, _hip.NORM_RMS, S, 1
    for i, a in enumerate(cls.tableau.alpha):
        p.alpha[i] = float(a)
    p.seg_count[0] = float(y0.size)
    device = y0d.device
    ctrl, ws = be.new_ctrl(device), be.new_workspace(device)
    ts = torch.zeros(_hip.XDE_MAX_STAGE, dtype=tdt, device=device)
    t_out = dtype(t0 + dtype(0.6) * dt)
    t_span = torch.tensor([float(t0), float(t_out), 100.0], dtype=torch.float64, device=device)
    be.ctrl_init(ctrl, p, float(t0), float(dt), 3, t_span, None, ts)
    segs = _hip.make_segments([(0, y0.size)])
    idx, coef = err_plan
    if fuse_err:
        be.error_norm_partial([ks[S]], [coef[-1]], y0d, y1d, p.rtol, p.atol, segs, _hip.NORM_RMS, ws, ctrl=ctrl, e_pre=ebuf)
    else:
        be.error_norm_partial([ks[j] for j in idx], coef, y0d, y1d, p.rtol, p.atol, segs, _hip.NORM_RMS, ws, ctrl=ctrl)
    be.rk_control(ctrl, p, ws, None, t_span, None, ts)
    c = be.ctrl_read(ctrl)
    with np.errstate(all="ignore"):
        ratio_ref = O.compute_error_ratio(err_ref, so.rtol, so.atol, y0, y1_ref, O._rms_norm)
    assert c.ratio == pytest.approx(float(ratio_ref), rel=3e-6), (name, c.ratio, float(ratio_ref))
    assert bool(c.accept) == bool(ratio_ref <= 1)
    # the controller's next step from the ORACLE's ratio is what optimal_step_size gives               ode_utils.py:85-97
    tt = so.tt
    want_dt = tt(np.clip(O.optimal_step_size(tt(dt), tt(c.ratio), so.safety, so.ifactor, so.dfactor, so.order), so.min_step, so.max_step))
    # (`ratio ** (1/order)` is the one transcendental on the path: the device's pow and numpy's agree to an ulp, not always to the bit)
    assert c.dt == pytest.approx(float(want_dt), rel=2.4e-7 if dtype == np.float32 else 4.5e-16), (name, c.dt, float(want_dt))

    # ---- dense output at a time inside the step                                     :286-292; ode_utils.py:28-77
    assert c.accept  # (every case is built to accept, so that the dense-output part below always runs)
    if c.accept:
        assert (c.out_begin, c.out_end) == (1, 2)
        sol = torch.zeros((3,) + tuple(y0d.shape), dtype=tdt, device=device)
        idx, coef = mid_plan
        be.dense_eval(sol, [ks[j] for j in idx], coef, y0d, y1d, ks[S], ctrl, t_span, p.time_dtype)
        coeffs = so._interp_fit(y0, y1_ref, k, dt)
        want = O.interp_evaluate(coeffs, tt(t0), tt(t0) + tt(dt), tt(t_out))
        assert np.array_equal(sol[1].cpu().numpy(), want), (name, "dense")
