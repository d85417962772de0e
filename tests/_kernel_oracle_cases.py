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
from paddlexde_amd.solver._rk_plans import build_plans as _build_plans
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
    n_stage, stage_plan, fsal, sol_plan, err_plan, mid_plan, fuse_err, err2_coef, _, presum = _build_plans(cls.tableau, cls.mid)
    assert n_stage == S
    if name == "dopri5" and presum:
        assert sorted(presum) == [4]  # stage 5 (y0, k0..k4: six arrays in) reads y0, the partial sum stage 4's launch emitted, and k4
    mv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ks = [mv(k[..., j]) for j in range(S + 1)]  # SoA: one tensor per stage derivative
    y0d = mv(y0)

    # ---- every stage input: y_i = y0 + sum_j k_j (beta_ij dt)                                        :166-168
    ebuf = torch.empty_like(y0d)
    sbuf = torch.empty_like(y0d)
    y_last = None
    for i in range(S):
        idx, coef = stage_plan[i]
        out = torch.empty_like(y0d)
        last = i == S - 1
        if i in presum:  # the previous launch has summed this stage's earlier operands: same association, same bits
            be.stage_combine_pre(out, y0d, sbuf, [ks[j] for j in presum[i][1]], presum[i][2], dt_host=float(dt))
        elif i + 1 in presum:
            be.stage_combine(out, y0d, [ks[j] for j in idx], coef, _hip.COMBINE_RK, dt_host=float(dt), out2=sbuf, coef2=presum[i + 1][0])
        else:
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


# ----------------------------------------------------------------------------------------------
# the norm kernels and the initial-step scalars against the ORACLE's functions (round 3; they used to reach the oracle only
# through end-to-end runs): xde_scaled_norm_partial + xde_initial_step vs AdaptiveRKSolver.select_initial_step
# (solver/base_adaptive_solver.py:33-72); xde_error_norm_partial + xde_rk_control with LINF and with a 5-segment mixed norm vs
# compute_error_ratio over _linf_norm / _mixed_norm (utils/ode_utils.py:4-19,80-82); forward and reverse time.
# ----------------------------------------------------------------------------------------------
def _cubic_np(dtype):
    c = dtype(0.25)
    return lambda t, y: (y - (y * y) * y * c).astype(dtype)  # element-wise, +, -, * only: f(0) = 0 keeps pads at zero


def _params(so, cls, S, dtype, n_seg, seg_counts, norm_kind, direction):
    p = _hip.XdeCtrlParams()
    p.rtol, p.atol, p.min_step, p.max_step = float(so.rtol), float(so.atol), 0.0, float("inf")
    p.safety, p.ifactor, p.dfactor, p.order = float(so.safety), float(so.ifactor), float(so.dfactor), float(so.order)
    p.max_num_steps = 2**31 - 1
    p.time_dtype = p.state_dtype = _hip.XDE_F32 if dtype == np.float32 else _hip.XDE_F64
    p.direction, p.norm_kind, p.n_stage, p.n_seg = direction, norm_kind, S, n_seg
    for i, a in enumerate(cls.tableau.alpha):
        p.alpha[i] = float(a)
    for i, cnt in enumerate(seg_counts):
        p.seg_count[i] = float(cnt)
    return p


@pytest.mark.parametrize("direction", [1, -1])
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("kind", ["linf", "mixed5", "rms"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_error_norm_kinds_vs_oracle(dev, dtype, kind, fused, direction):
    """One attempted Dopri5 step on a ~1 M-element state: the error ratio the norm kernel + controller produce equals
    compute_error_ratio(y1_error, rtol, atol, y0, y1, norm) with the oracle's norm functions — LINF to the bit, RMS / the
    5-segment mixed norm (max of per-segment RMS = the adjoint's default norm shape) to reduction-order accuracy, every
    segment's own value too.  Reverse time: the kernels get (k_j = f, dt < 0), the oracle its flipped problem (f -> -f, dt > 0)."""
    be = _hip.get_backend()
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    width = 4 if dtype == np.float32 else 2
    rng = np.random.RandomState(5)
    lens = [1, 300001, 7, 500000, 252] if kind == "mixed5" else [(1 << 20) + 3]
    starts, pos = [], 0
    for ln in lens:  # tuple state: every segment starts on a 16-byte boundary, pads are zero
        starts.append(pos)
        pos += -(-ln // width) * width
    total = pos
    flat = np.zeros(total, dtype=dtype)
    for s0, ln in zip(starts, lens):
        flat[s0 : s0 + ln] = rng.uniform(-2, 2, size=ln).astype(dtype)
    compact = np.concatenate([flat[s0 : s0 + ln] for s0, ln in zip(starts, lens)])
    f = _cubic_np(dtype)
    rtol, atol = 1e-4, 1e-6
    norm = {"linf": O._linf_norm, "rms": O._rms_norm,
            "mixed5": lambda x: O._mixed_norm(O._unflatten(x, [(ln,) for ln in lens]))}[kind]
    f_or = f if direction > 0 else (lambda t, y: -f(-t, y))  # D5: the oracle integrates the flipped problem
    so = O.AdaptiveRKSolver(f_or, compact, rtol, atol, method="dopri5", norm=norm, dtype=dtype)
    t0, dt = dtype(0.5), dtype(0.11)
    f0 = so.move(t0, 0, compact)
    y1_ref, f1_ref, err_ref, k = so._runge_kutta_step(compact, f0, t0, dt, t0 + dt, so.tableau)
    ratio_ref = O.compute_error_ratio(err_ref, so.rtol, so.atol, compact, y1_ref, norm)
    seg_ref = [float(np.abs((O._linf_norm if kind == "linf" else O._rms_norm)(
        (err_ref / (so.atol + so.rtol * np.fmax(np.abs(compact), np.abs(y1_ref))))[a : a + ln]))) for a, ln in
        zip(np.cumsum([0] + lens[:-1]), lens)]

    S = k.shape[-1] - 1
    n_stage, stage_plan, fsal, sol_plan, err_plan, mid_plan, fuse_err, err2_coef, _, _presum = _build_plans(Dopri5.tableau, Dopri5.mid)

    def pad(compact_arr):  # the oracle's compact vector laid out with the device's pads
        out = np.zeros(total, dtype=dtype)
        a = 0
        for s0, ln in zip(starts, lens):
            out[s0 : s0 + ln] = compact_arr[a : a + ln]
            a += ln
        return torch.from_numpy(out).to(dev)

    sign = dtype(direction)
    ks = [pad(sign * k[..., j]) for j in range(S + 1)]  # the product's k_j = f(t, y) in real time
    y0d, y1d = pad(compact), pad(y1_ref)
    dts = float(sign * dt)
    kind_code = _hip.NORM_LINF if kind == "linf" else _hip.NORM_RMS
    p = _params(so, Dopri5, S, dtype, len(lens), lens, kind_code, direction)
    ctrl, ws = be.new_ctrl(y0d.device), be.new_workspace(y0d.device)
    ts = torch.zeros(_hip.XDE_MAX_STAGE, dtype=tdt, device=y0d.device)
    t_span = torch.tensor([direction * 0.5, direction * 100.0], dtype=torch.float64, device=y0d.device)
    be.ctrl_init(ctrl, p, direction * 0.5, dts, 2, t_span, None, ts)
    segs = _hip.make_segments(list(zip(starts, lens)))
    idx, coef = err_plan
    if fused:
        # the last stage's combine emits the partial error sum from the operands it holds (xde_stage_combine out2) ...
        sidx, scoef = stage_plan[S - 1]
        y_last, ebuf = torch.empty_like(y0d), torch.empty_like(y0d)
        be.stage_combine(y_last, y0d, [ks[j] for j in sidx], scoef, _hip.COMBINE_RK, ctrl=ctrl, out2=ebuf, coef2=err2_coef)
        assert np.array_equal(y_last.cpu().numpy(), y1d.cpu().numpy())  # = the oracle's y1, bit for bit, also with dt < 0
        # ... and the norm pass reads 4 arrays instead of 8
        be.error_norm_partial([ks[S]], [coef[-1]], y0d, y1d, p.rtol, p.atol, segs, kind_code, ws, ctrl=ctrl, e_pre=ebuf)
    else:
        be.error_norm_partial([ks[j] for j in idx], coef, y0d, y1d, p.rtol, p.atol, segs, kind_code, ws, ctrl=ctrl)
    be.rk_control(ctrl, p, ws, None, t_span, None, ts)
    c = be.ctrl_read(ctrl)
    if kind == "linf":
        assert c.ratio == float(ratio_ref), (c.ratio, float(ratio_ref))  # a maximum does not depend on the order it is taken in
    else:
        rel = 3e-6 if dtype == np.float32 else 1e-12  # numpy's pairwise fp32 mean vs fp32 lanes -> fp64 accumulation
        assert c.ratio == pytest.approx(float(ratio_ref), rel=rel), (c.ratio, float(ratio_ref))
        for s_, want in enumerate(seg_ref):
            assert c.ratio_seg[s_] == pytest.approx(want, rel=rel), (s_, c.ratio_seg[s_], want)
        assert int(np.argmax(seg_ref)) == int(np.argmax([c.ratio_seg[s_] for s_ in range(len(lens))]))
    assert bool(c.accept) == bool(ratio_ref <= 1) and c.nonfinite == 0
    tt = so.tt
    want_dt = tt(np.clip(O.optimal_step_size(tt(dt), tt(c.ratio), so.safety, so.ifactor, so.dfactor, so.order), so.min_step, so.max_step))
    assert c.dt == pytest.approx(direction * float(want_dt), rel=2.4e-7 if dtype == np.float32 else 4.5e-16)


@pytest.mark.parametrize("direction", [1, -1])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_initial_step_vs_oracle_select_initial_step(dev, dtype, direction):
    """select_initial_step (solver/base_adaptive_solver.py:33-72) on a 1 M-element state.  The oracle's own run is instrumented
    (its norm and fuse calls are recorded), then (1) the three scaled norms of xde_scaled_norm_partial are held to the oracle's
    norm values; (2) the scalar arithmetic of xde_initial_step, FED the oracle's norm values bit for bit, gives the oracle's h0
    exactly and its first step to one ulp of the transcendental (pow); (3) the solver's own device path (_before_integrate, no
    host read) lands on the oracle's first step, signed by the direction of integration."""
    from paddlexde_amd import Dopri5 as Solver
    from paddlexde_amd.utils import _rms_norm
    from paddlexde_amd.xde import BaseODE

    be = _hip.get_backend()
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    rng = np.random.RandomState(8)
    y0 = rng.uniform(-2, 2, size=((1 << 19) + 3, 2)).astype(dtype)
    mu = 3.0
    f_np, f_t = P.vdp_np(dtype(mu)), P.vdp_torch(mu)
    rtol, atol = 1e-5, 1e-7
    norms, h0s = [], []

    def rec_norm(x):
        v = O._rms_norm(x)
        norms.append(v)
        return v

    f_or = f_np if direction > 0 else (lambda t, y: -f_np(-t, y))
    so = O.AdaptiveRKSolver(f_or, y0, rtol, atol, method="dopri5", norm=rec_norm, dtype=dtype)
    fuse0 = so.fuse
    so.fuse = lambda dy, dt, y: (h0s.append(dt), fuse0(dy, dt, y))[1]
    t0 = so.tt(direction * 0.25)
    # real start time direction * 0.25; in reverse the oracle sees the flipped problem, whose start time is +0.25 again
    first_ref = so.select_initial_step(so.tt(0.25), y0, so.order - 1, so.rtol, so.atol)
    d0_ref, d1_ref, n3_ref = [float(np.abs(v)) for v in norms]
    h0_ref = float(h0s[0])

    y0d = torch.from_numpy(y0).to(dev)
    t_span = np.asarray([direction * 0.25, direction * 5.0])
    s = Solver(xde=BaseODE(f_t, y0=y0d, t_span=torch.from_numpy(t_span)), y0=y0d, rtol=rtol, atol=atol, norm=_rms_norm, dtype=tdt)
    s.y0 = y0d
    s._before_integrate(t_span.astype(np.float32 if dtype == np.float32 else np.float64))
    rel = 3e-6 if dtype == np.float32 else 1e-12
    if hasattr(s, "_first_step_dbg"):  # (absent where the heuristic's scalars go through the host: a custom norm)
        res, hs = s._first_step_dbg
        res, hs = res.cpu().numpy(), hs.cpu().numpy()
        # (1) the norms
        assert hs[0] == pytest.approx(d0_ref, rel=rel) and hs[1] == pytest.approx(d1_ref, rel=rel)
        assert abs(res[0]) == pytest.approx(n3_ref, rel=30 * rel)  # f1 - f0 at h0 ~ 1e-5: a cancellation, and h0 differs in its last bits
        # (3) the whole device path
        assert hs[2] == pytest.approx(h0_ref, rel=2 * rel)
        assert hs[3] == pytest.approx(float(first_ref), rel=30 * rel)
    else:
        assert abs(float(s.rk_state.dt)) == pytest.approx(float(first_ref), rel=30 * rel)

    # (2) the scalar arithmetic alone, on the oracle's norm values
    p = s._params
    ctrl2 = be.new_ctrl(y0d.device)
    r2 = torch.tensor([norms[0], norms[1]], dtype=torch.float64, device=y0d.device)
    h2 = torch.zeros(4, dtype=torch.float64, device=y0d.device)
    t_probe = torch.empty((), dtype=tdt, device=y0d.device)
    be.initial_step(0, r2, h2, p, float(t0), t_probe, ctrl2)
    assert h2.cpu().numpy()[2] == h0_ref  # h0 = 0.01 d0 / d1 in the state dtype, reference op order: exact
    assert float(t_probe.cpu()) == float(so.tt(t0) + dtype(direction * h0_ref) if dtype == np.float32 else t0 + direction * h0_ref)
    r2 = torch.tensor([norms[2], 0.0], dtype=torch.float64, device=y0d.device)
    be.initial_step(1, r2, h2, p, float(t0), None, ctrl2)
    ulp = 1.2e-7 if dtype == np.float32 else 2.3e-16
    assert h2.cpu().numpy()[3] == pytest.approx(float(first_ref), rel=2 * ulp), (h2.cpu().numpy(), first_ref)


@pytest.mark.parametrize("direction", [1, -1])
@pytest.mark.parametrize("norm_name", ["rms", "linf"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_initial_step_equals_the_separate_launches_and_the_oracle(dev, dtype, norm_name, direction, monkeypatch):
    """Small states take select_initial_step (solver/base_adaptive_solver.py:33-72) as TWO one-workgroup launches + the Euler probe
    (xde_initial_step_fused) instead of twelve launches.  Held to (a) the separate launches on the same inputs — d0, d1, h0, the
    first step and the constructed control block (same stage times, same output bookkeeping) — and (b) the oracle's own
    select_initial_step.  4099 x 2 elements: the scalar tail runs too."""
    from paddlexde_amd import Dopri5 as Solver
    from paddlexde_amd.utils import _linf_norm, _rms_norm
    from paddlexde_amd.xde import BaseODE

    be = _hip.get_backend()
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    rng = np.random.RandomState(5)
    y0 = rng.uniform(-2, 2, size=(4099, 2)).astype(dtype)
    f_np, f_t = P.vdp_np(dtype(3.0)), P.vdp_torch(3.0)
    rtol, atol = 1e-5, 1e-7
    y0d = torch.from_numpy(y0).to(dev)
    t_span = np.asarray([direction * 0.25, direction * 1.0, direction * 5.0])
    tsp = t_span.astype(np.float32 if dtype == np.float32 else np.float64)

    def run(fused):
        s = Solver(xde=BaseODE(f_t, y0=y0d, t_span=torch.from_numpy(t_span)), y0=y0d, rtol=rtol, atol=atol, _fused_first_step=fused,
                   norm=_rms_norm if norm_name == "rms" else _linf_norm, dtype=tdt, step_t=torch.tensor([direction * 0.2500001, direction * 2.0]))
        s.y0 = y0d
        s._before_integrate(tsp)
        assert s._small_state
        assert s._fused_first_step() == fused and s._ctrl_ready == fused
        res, hs = s._first_step_dbg
        return s, float(res.cpu().numpy()[0]), hs.cpu().numpy()[:4].copy(), be.ctrl_read(s._ctrl), s._t_stage.cpu().numpy().copy()

    sf, n3f, hf, cf, tf = run(True)
    su, n3u, hu, cu, tu = run(False)
    assert sf.nfe == su.nfe == 3  # f0, f0 again (as the reference counts it), f1
    tight = 1e-6 if dtype == np.float32 else 1e-13  # one workgroup sums in another order than 512 (fp64 sums of state-dtype squares)
    assert hf[0] == pytest.approx(hu[0], rel=tight) and hf[1] == pytest.approx(hu[1], rel=tight) and hf[2] == pytest.approx(hu[2], rel=tight)
    assert n3f == pytest.approx(n3u, rel=1e-3 if dtype == np.float32 else 1e-9)  # (f1 - f0 at h0 ~ 1e-5: a cancellation)
    assert hf[3] == pytest.approx(hu[3], rel=1e-4 if dtype == np.float32 else 1e-10)
    for name in ("t0", "t1"):
        assert getattr(cf, name) == getattr(cu, name)
    for name in ("dt", "t_plan"):
        assert getattr(cf, name) == pytest.approx(getattr(cu, name), rel=1e-4 if dtype == np.float32 else 1e-10)
    for name in ("n_out", "next_out", "out_begin", "out_end", "done", "next_step_index", "on_step_t", "n_steps", "accept", "status"):
        assert getattr(cf, name) == getattr(cu, name), name
    assert cf.next_out == 1 and cf.dt * direction > 0, (cf.next_out, cf.dt, cf.on_step_t, cf.next_step_index)
    assert np.allclose(tf[:6], tu[:6], rtol=1e-4 if dtype == np.float32 else 1e-10)
    # (b) the oracle
    f_or = f_np if direction > 0 else (lambda t, y: -f_np(-t, y))
    so = O.AdaptiveRKSolver(f_or, y0, rtol, atol, method="dopri5", norm=O._rms_norm if norm_name == "rms" else O._linf_norm, dtype=dtype)
    first_ref = so.select_initial_step(so.tt(0.25), y0, so.order - 1, so.rtol, so.atol)
    assert hf[3] == pytest.approx(float(first_ref), rel=1e-4 if dtype == np.float32 else 1e-9)


@pytest.mark.parametrize("direction", [1, -1])
@pytest.mark.parametrize("norm_name", ["rms", "linf"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_large_state_initial_step_in_four_launches_equals_the_twelve(dev, dtype, norm_name, direction):
    """VERDICT r05 (next 5): states above the one-workgroup kernels' reach paid 12 launches for select_initial_step
    (solver/base_adaptive_solver.py:33-72): 3 x (xde_scaled_norm_partial + xde_norm_finalize + xde_norm_result) + 2 x xde_initial_step +
    xde_ctrl_init.  Now 4: d0 and d1 in ONE pass over (y0, f0) (xde_scaled_norm2_partial — SURVEY A6's "two norms in one pass"), and
    everything that followed a norm pass as ONE one-workgroup launch per phase (xde_initial_step_tail).  Same grid, same per-lane order,
    same fixed-order reduction, same scalar code: d0, d1, h0, the third norm, the first step, the constructed control block and the
    stage times are BIT-identical to the twelve launches' — and so to what test_initial_step_vs_oracle_select_initial_step holds to
    the oracle.  (2^17 + 3) x 2 elements: not a multiple of the vector width, several workgroups."""
    from paddlexde_amd import Dopri5 as Solver
    from paddlexde_amd.utils import _linf_norm, _rms_norm
    from paddlexde_amd.xde import BaseODE

    be = _hip.get_backend()
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    rng = np.random.RandomState(6)
    y0 = rng.uniform(-2, 2, size=((1 << 17) + 3, 2)).astype(dtype)
    f_t = P.vdp_torch(3.0)
    y0d = torch.from_numpy(y0).to(dev)
    t_span = np.asarray([direction * 0.25, direction * 1.0, direction * 5.0])
    tsp = t_span.astype(np.float32 if dtype == np.float32 else np.float64)
    heuristic = ("scaled_norm_partial", "scaled_norm2_partial", "norm_finalize", "norm_result", "initial_step", "initial_step_tail", "ctrl_init")

    def run(folded):
        counts, depth = {}, [0]
        s = Solver(xde=BaseODE(f_t, y0=y0d, t_span=torch.from_numpy(t_span)), y0=y0d, rtol=1e-5, atol=1e-7, _fused_first_step=folded,
                   norm=_rms_norm if norm_name == "rms" else _linf_norm, dtype=tdt, step_t=torch.tensor([direction * 0.2500001, direction * 2.0]))
        s.y0 = y0d
        originals = {name: getattr(be, name) for name in heuristic}
        try:
            for name, fn in originals.items():
                def counted(*a, _fn=fn, _name=name, **k):
                    if not depth[0]:  # (the CPU double composes its folded calls from the separate ones: the solver's own calls count)
                        counts[_name] = counts.get(_name, 0) + 1
                    depth[0] += 1
                    try:
                        return _fn(*a, **k)
                    finally:
                        depth[0] -= 1

                setattr(be, name, counted)
            s._before_integrate(tsp)
        finally:
            for name in heuristic:
                delattr(be, name)  # (the instance attributes set above: the class's methods show through again)
        assert not s._small_state and s._tail_first_step() == folded and s._ctrl_ready == folded
        res, hs = s._first_step_dbg
        return s, counts, float(res.cpu().numpy()[0]), hs.cpu().numpy()[:4].copy(), be.ctrl_read(s._ctrl), s._t_stage.cpu().numpy().copy()

    sf, nf, n3f, hf, cf, tf = run(True)
    su, nu, n3u, hu, cu, tu = run(False)
    assert nf == {"scaled_norm2_partial": 1, "scaled_norm_partial": 1, "initial_step_tail": 2}, nf  # 4 launches
    assert nu == {"scaled_norm_partial": 3, "norm_finalize": 3, "norm_result": 3, "initial_step": 2, "ctrl_init": 1}, nu  # 12
    assert sf.nfe == su.nfe == 3  # f0, f0 again (as the reference counts it), f1
    assert np.array_equal(hf, hu) and n3f == n3u, (hf, hu, n3f, n3u)  # d0, d1, h0, first step; the third norm: bit for bit
    for name in ("t0", "t1", "dt", "t_plan", "n_out", "next_out", "out_begin", "out_end", "done", "next_step_index", "on_step_t", "n_steps",
                 "accept", "status", "dt_last", "ratio", "n_accept", "n_reject", "steps_in_interval"):
        assert getattr(cf, name) == getattr(cu, name), name
    assert cf.next_out == 1 and cf.dt * direction > 0
    assert np.array_equal(tf[:6], tu[:6])
    if hasattr(be, "ctrl_init_handle"):
        # the folded heuristic's last launch also PUBLISHED the constructed block to the host mirror (the speculative pipeline reads where
        # the first attempt lands from there: no copy command on the stream); the separate launches leave that to ctrl_peek_async
        h = be.ctrl_init_handle(sf._ctrl)
        assert h is not None and be.ctrl_init_handle(su._ctrl) is None
        pub = be.ctrl_wait(h)
        for name in ("t0", "t1", "dt", "t_plan", "n_out", "next_out", "done", "seq", "n_steps", "status"):
            assert getattr(pub, name) == getattr(cf, name), name
