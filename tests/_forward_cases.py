"""End-to-end cases, run on the GPU (tests/test_gpu_odeint.py) and — host logic only — on the CPU double (tests/test_host_logic.py).
SURVEY 8(a) rows A1-A11, forward: the reference's own problems, fixed and adaptive solvers against the oracle, golden fixtures,
full-size properties, randomised sweeps."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import xde_oracle as O
from paddlexde_amd import RK4, AdamsBashforthMoulton, AdaptiveHeun, Bosh3, Dopri5, Dopri8, Euler, Fehlberg2, Midpoint, _hip, odeint, odeint_adjoint
from paddlexde_amd.utils import _linf_norm, _rms_norm

from . import problems as P

from ._e2e_common import ADAPTIVE, ConstantLayer, DeepFunc, FIXED, ODEFunc, P_rms, _SmallMLP, _blocks, _golden, _linear, _mlp_foreign, _mlp_numpy  # noqa: F401


# ----------------------------------------------------------------------------------------------
# the reference's own acceptance tests (tests/functional/test_fixed_solver.py:26-44,
# tests/functional/test_adaptive_solver.py:32-87): analytic problems at the reference's tolerances
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(FIXED))
def test_reference_fixed_constant(dev, name):
    p, y0, t, sol = P.construct_problem("constant")
    y = odeint(p.f_torch(dev), torch.from_numpy(y0).to(dev), torch.from_numpy(t).to(dev), solver=FIXED[name])
    assert y.shape == (10, 1)  # concat on axis -2 of y0 [1, 1]
    assert np.allclose(sol, y.cpu().numpy(), rtol=1e-2, atol=1e-8)


@pytest.mark.parametrize("name", list(ADAPTIVE))
@pytest.mark.parametrize("ode", ["sine", "linear"])
def test_reference_adaptive(dev, name, ode):
    p, y0, t, sol = P.construct_problem(ode)
    y = odeint(p.f_torch(dev), torch.from_numpy(y0).to(dev), torch.from_numpy(t).to(dev), solver=ADAPTIVE[name])
    assert y.shape == (10,) + y0.shape  # time first
    rtol = 1e-2 if (name == "adaptive_heun" and ode == "linear") else 4e-3
    assert np.allclose(sol[:, None, :], y.cpu().numpy(), rtol=rtol, atol=1e-8)


# ----------------------------------------------------------------------------------------------
# config 1: the spiral demo, RK4 (reference variant), batch=1, dim=2 — against the oracle
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["euler", "midpoint", "rk4"])  # (Adams: test_adams_bashforth_moulton_vs_oracle)
def test_spiral_fixed_vs_oracle(dev, name):
    y0 = np.array([[2.0, 0.0]], dtype=np.float32)
    t = np.linspace(0.0, 25.0, 1000).astype(np.float32)
    if name == "euler":  # first-order: needs a finer grid to stay bounded on the cubic spiral
        t = np.linspace(0.0, 2.0, 400).astype(np.float32)
    ref = O.odeint(P.spiral_np, y0, t, name)
    got = odeint(P.spiral_torch, torch.from_numpy(y0).to(dev), torch.from_numpy(t).to(dev), solver=FIXED[name])
    assert got.shape == ref.shape == (len(t), 2)
    # every kernel is bit-exact and func uses only +,-,*: the whole trajectory is bit-exact
    assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("implicit", [False, True])
def test_adams_bashforth_moulton_vs_oracle(dev, implicit):
    """AdamsBashforthMoulton (fixed_solver/adams.py:457-547): RK4-variant bootstrap, then the explicit predictor of the
    highest available order (<= max_order - 1), optionally the Adams-Moulton corrector iterations — bit-exact."""
    rng = np.random.RandomState(5)
    y0 = rng.uniform(-1.0, 1.0, size=(3, 2)).astype(np.float32)
    t = np.linspace(0.0, 0.2, 41).astype(np.float32)
    name = "adams_implicit" if implicit else "adams"
    for max_order in (4, 6, 12):
        ref = O.odeint(P.spiral_np, y0, t, name, rtol=1e-3, atol=1e-4, options={"norm": O._rms_norm, "max_order": max_order})
        got = odeint(P.spiral_torch, torch.from_numpy(y0).to(dev), torch.from_numpy(t).to(dev), solver=AdamsBashforthMoulton,
                     rtol=1e-3, atol=1e-4, options={"norm": _rms_norm, "implicit": implicit, "max_order": max_order})
        assert got.shape == ref.shape == (41 * 3, 2)
        assert np.isfinite(ref).all()
        assert np.array_equal(got.cpu().numpy(), ref), (implicit, max_order, float(np.abs(got.cpu().numpy() - ref).max()))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_step_interpolants_vs_oracle(dev, dtype):
    """interpolation/functional/interp_fn.py:4-20 (`linear_interp`, `cubic_hermite_interp`) — off the hot path (the solvers only ever
    evaluate them at t == t1, where both are y1), restated as framework ops in the reference's op order: bit-exact with the oracle's,
    inside, at the ends of and beyond the step."""
    from paddlexde_amd.interpolation.functional import cubic_hermite_interp, linear_interp

    rng = np.random.RandomState(0)
    y0, y1, d0, d1 = (rng.randn(5, 7).astype(dtype) for _ in range(4))
    t0, t1 = dtype(0.25), dtype(1.5)
    T = lambda x: torch.from_numpy(np.asarray(x)).to(dev)  # noqa: E731
    for tq in (0.25, 1.5, 0.8, 1.4999, 2.0):
        tq = dtype(tq)
        want = O.linear_interp(t0, t1, y0, y1, tq)
        got = linear_interp(T(t0), T(t1), T(y0), T(y1), T(tq))
        assert np.array_equal(got.cpu().numpy(), want), tq
        want = O.cubic_hermite_interp(t0, y0, d0, t1, y1, d1, tq)
        got = cubic_hermite_interp(T(t0), T(y0), T(d0), T(t1), T(y1), T(d1), T(tq))
        assert np.array_equal(got.cpu().numpy(), want), tq
    assert np.array_equal(cubic_hermite_interp(T(t0), T(y0), T(d0), T(t1), T(y1), T(d1), T(t1)).cpu().numpy(), y1)  # what the solver relies on


def test_fixed_layout_batched(dev):
    """y0 [B, L, D] -> [B, T*L, D] against the oracle."""
    rng = np.random.RandomState(0)
    y0 = rng.uniform(-2, 2, size=(7, 3, 2)).astype(np.float32)
    t = np.linspace(0.0, 0.2, 9).astype(np.float32)
    ref = O.odeint(P.spiral_np, y0, t, "rk4")
    got = odeint(P.spiral_torch, torch.from_numpy(y0).to(dev), torch.from_numpy(t).to(dev), solver=RK4)
    assert got.shape == (7, 27, 2)
    assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("pipeline", ["sync", "lag", "graph"])
@pytest.mark.parametrize("name", list(ADAPTIVE))
def test_linear_adaptive_vs_oracle_fp64(dev, name, pipeline):
    """fp64 state and time: step decisions are robust, so the tight bar applies."""
    A, y0 = _linear(64, 32, torch.float64)
    t = torch.linspace(0.0, 1.0, 6, dtype=torch.float64)
    tol = dict(rtol=1e-6, atol=1e-8) if name != "adaptive_heun" else dict(rtol=1e-4, atol=1e-6)
    An = A.numpy()
    ref, so = O.odeint(lambda t_, y: y @ An.T, y0.numpy(), t.numpy(), name, options={"norm": O._rms_norm, "dtype": np.float64},
                       return_solver=True, **tol)
    Ad = A.to(dev)
    from paddlexde_amd.xde import BaseODE

    xde = BaseODE(lambda t_, y: y @ Ad.T, y0=y0.to(dev), t_span=t)
    s = ADAPTIVE[name](xde=xde, y0=xde.y0, norm=_rms_norm, dtype=torch.float64, pipeline=pipeline, **tol)
    got = s.integrate(t)
    assert P.parity_ok(got.cpu().numpy(), ref, rtol=1e-9, atol=1e-11), P.worst(got.cpu().numpy(), ref, 1e-9, 1e-11)
    assert (s.stats["n_accept"], s.stats["n_reject"], s.stats["nfe"]) == (so.n_accept, so.n_reject, so.nfe)


@pytest.mark.parametrize("pipeline", ["sync", "lag"])
def test_linear_dopri5_vs_oracle_fp32(dev, pipeline):
    A, y0 = _linear(512, 128, torch.float32)
    t = torch.linspace(0.0, 1.0, 11)
    An = A.numpy()
    ref, so = O.odeint(lambda t_, y: y @ An.T, y0.numpy(), t.numpy(), "dopri5", rtol=1e-5, atol=1e-7, return_solver=True)
    Ad = A.to(dev)
    got = odeint(lambda t_, y: y @ Ad.T, y0.to(dev), t.to(dev), solver=Dopri5, rtol=1e-5, atol=1e-7,
                 options={"norm": _rms_norm, "pipeline": pipeline})
    # fp32: two correct implementations differ by an ulp in the error ratio (reduction order), hence in dt, and
    # fp32 cancellation noise in the error estimate amplifies that to ~1e-6 absolute on O(1) values.  The bar is
    # therefore taken against the solution's scale: max|diff| <= 1e-5 * max|ref| (north_star: "<=1e-5 relative").
    assert P.rel_err(got.cpu().numpy(), ref) <= 1e-5, P.rel_err(got.cpu().numpy(), ref)


def test_reverse_time_vs_oracle(dev):
    A, y0 = _linear(64, 32, torch.float64)
    t = torch.linspace(1.0, 0.0, 5, dtype=torch.float64)
    An = A.numpy()
    ref = O.odeint(lambda t_, y: y @ An.T, y0.numpy(), t.numpy(), "dopri5", rtol=1e-7, atol=1e-9,
                   options={"norm": O._rms_norm, "dtype": np.float64})
    Ad = A.to(dev)
    got = odeint(lambda t_, y: y @ Ad.T, y0.to(dev), t, solver=Dopri5, rtol=1e-7, atol=1e-9,
                 options={"norm": _rms_norm, "dtype": torch.float64})
    assert P.parity_ok(got.cpu().numpy(), ref, rtol=1e-9, atol=1e-11)


# ----------------------------------------------------------------------------------------------
# config 5: stiff Van der Pol, step-rejection stress
# ----------------------------------------------------------------------------------------------
def test_vdp_rejections_vs_oracle_fp64(dev):
    mu = 50.0
    y0 = np.array([2.0, 0.0]) + 0.01 * np.random.RandomState(0).randn(64, 2)
    t = np.array([0.0, 1.0])
    ref, so = O.odeint(P.vdp_np(mu), y0, t, "dopri5", rtol=1e-6, atol=1e-8, options={"norm": O._rms_norm, "dtype": np.float64},
                       return_solver=True)
    from paddlexde_amd.xde import BaseODE

    y0t = torch.from_numpy(y0).to(dev)
    xde = BaseODE(P.vdp_torch(mu), y0=y0t, t_span=torch.from_numpy(t))
    s = Dopri5(xde=xde, y0=y0t, rtol=1e-6, atol=1e-8, norm=_rms_norm, dtype=torch.float64)
    got = s.integrate(torch.from_numpy(t))
    assert so.n_reject > 0
    assert (s.stats["n_accept"], s.stats["n_reject"]) == (so.n_accept, so.n_reject)
    assert P.parity_ok(got.cpu().numpy(), ref, rtol=1e-8, atol=1e-10), P.worst(got.cpu().numpy(), ref, 1e-8, 1e-10)


# ----------------------------------------------------------------------------------------------
# BASELINE.json configs 3 and 5 at their FULL sizes against the oracle (it finishes these in seconds)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pipeline", ["sync", "graph"])
def test_config5_full_size_vs_oracle(dev, pipeline):
    """Stiff Van der Pol mu=1000, batch 4096 x 2, t in [0, 1], rtol 1e-5 / atol 1e-7 (fp64 state and time so that the
    ~1200 accept/reject decisions are reproducible): identical step counts, solution to 1e-8."""
    from paddlexde_amd.xde import BaseODE

    mu = 1000.0
    y0 = np.array([2.0, 0.0]) + 0.01 * torch.randn(4096, 2, generator=torch.Generator().manual_seed(0)).double().numpy()
    t = np.array([0.0, 1.0])
    ref, so = O.odeint(P.vdp_np(mu), y0, t, "dopri5", rtol=1e-5, atol=1e-7, options={"norm": O._rms_norm, "dtype": np.float64,
                                                                                      "max_num_steps": 10**6}, return_solver=True)
    y0t = torch.from_numpy(y0).to(dev)
    xde = BaseODE(P.vdp_torch(mu), y0=y0t, t_span=torch.from_numpy(t))
    s = Dopri5(xde=xde, y0=y0t, rtol=1e-5, atol=1e-7, norm=_rms_norm, dtype=torch.float64, max_num_steps=10**6, pipeline=pipeline)
    got = s.integrate(torch.from_numpy(t))
    assert so.n_reject > 100 and so.n_accept > 500
    assert (s.stats["n_accept"], s.stats["n_reject"], s.stats["nfe"]) == (so.n_accept, so.n_reject, so.nfe)
    assert P.rel_err(got.cpu().numpy(), ref) <= 1e-8, P.rel_err(got.cpu().numpy(), ref)


def test_golden_spiral_rk4(dev):
    z = _golden("spiral_rk4")
    got = odeint(P.spiral_torch, torch.from_numpy(z["y0"]).to(dev), torch.from_numpy(z["t"]).to(dev), solver=RK4)
    assert np.array_equal(got.cpu().numpy(), z["sol"])  # bit-exact: 999 steps x 4 combines


def test_golden_fixed_small(dev):
    z = _golden("spiral_fixed_small")
    y0, t = torch.from_numpy(z["y0"]).to(dev), torch.from_numpy(z["t"]).to(dev)
    for name, cls, opts in [("euler", Euler, {}), ("midpoint", Midpoint, {}), ("rk4", RK4, {}), ("rk4_classic", RK4, {"variant": "classic"})]:
        got = odeint(P.spiral_torch, y0, t, solver=cls, options={"norm": _rms_norm, **opts})
        assert np.array_equal(got.cpu().numpy(), z["sol_" + name]), name


@pytest.mark.parametrize("pipeline", ["sync", "lag", "graph"])
def test_golden_linear_dopri5_trace(dev, pipeline):
    """Step-for-step: (t0, dt, ratio, accept) of every attempted step against the oracle's trace."""
    from paddlexde_amd.xde import BaseODE

    z = _golden("linear_dopri5_f64")
    A = torch.from_numpy(z["A"]).to(dev)
    y0 = torch.from_numpy(z["y0"]).to(dev)
    t = torch.from_numpy(z["t"])
    xde = BaseODE(lambda t_, y: y @ A.T, y0=y0, t_span=t)
    s = Dopri5(xde=xde, y0=y0, rtol=1e-7, atol=1e-9, norm=_rms_norm, dtype=torch.float64, pipeline=pipeline, record_trace=True)
    got = s.integrate(t)
    assert P.parity_ok(got.cpu().numpy(), z["sol"], rtol=1e-10, atol=1e-12)
    tr = np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace])
    assert tr.shape == z["trace"].shape
    assert np.array_equal(tr[:, 3], z["trace"][:, 3])  # identical accept/reject decisions
    assert np.allclose(tr[:, :2], z["trace"][:, :2], rtol=1e-9, atol=0)  # t0, dt
    assert np.allclose(tr[:, 2], z["trace"][:, 2], rtol=1e-6)  # error ratio (cancellation-limited)
    assert s.stats["nfe"] == int(z["nfe"])


def test_golden_vdp_counts(dev):
    from paddlexde_amd.xde import BaseODE

    z = _golden("vdp_dopri5_f64")
    y0 = torch.from_numpy(z["y0"]).to(dev)
    t = torch.from_numpy(z["t"])
    xde = BaseODE(P.vdp_torch(float(z["mu"])), y0=y0, t_span=t)
    s = Dopri5(xde=xde, y0=y0, rtol=1e-6, atol=1e-8, norm=_rms_norm, dtype=torch.float64, record_trace=True)
    got = s.integrate(t)
    assert [s.stats["n_accept"], s.stats["n_reject"], s.stats["nfe"]] == list(z["counts"])
    assert int(z["counts"][1]) > 0
    assert P.parity_ok(got.cpu().numpy(), z["sol"], rtol=1e-8, atol=1e-10)


# ----------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties (the oracle cannot run these in seconds)
# ----------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_config4_shard_size_properties(dev):
    """Config 4's per-GPU shard (65536 x 64): exact-solution rows and norm conservation (the 8-GPU coupling itself is
    covered by tests/test_sharded_gloo.py)."""
    import scipy.linalg

    B, D = 65536, 64
    A = P.skew_matrix(D)
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(3))
    Ad, y0d = A.to(dev), y0.to(dev)
    sol = odeint(lambda t_, y: y @ Ad.T, y0d, torch.tensor([0.0, 1.0], device=dev), solver=Dopri5, rtol=1e-5, atol=1e-7,
                 options={"norm": _rms_norm, "pipeline": "lag"})
    rows = [0, 77, 40000, 65535]
    exact = y0[rows].double().numpy() @ scipy.linalg.expm(A.double().numpy()).T
    assert np.allclose(sol[1][rows].cpu().numpy(), exact, rtol=1e-4, atol=2e-5)
    assert torch.allclose(y0d.double().norm(dim=1), sol[1].double().norm(dim=1), rtol=2e-5)


@pytest.mark.gpu
def test_config2_full_size_properties(dev):
    """batch=65536 x dim=128 Dopri5: (i) rows checked against the exact solution expm(tA) y0, (ii) the flow
    of a skew-symmetric A is a rotation: row norms are conserved, (iii) forward-then-backward round trip."""
    import scipy.linalg

    B, D = 65536, 128
    A = P.skew_matrix(D)
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(0))
    Ad, y0d = A.to(dev), y0.to(dev)
    f = lambda t_, y: y @ Ad.T  # noqa: E731
    t = torch.tensor([0.0, 0.5, 1.0], device=dev)
    sol = odeint(f, y0d, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm, "pipeline": "lag"})
    assert sol.shape == (3, B, D)
    rows = [0, 1, 4097, 65535]
    E = scipy.linalg.expm(A.double().numpy() * 1.0)
    exact = y0[rows].double().numpy() @ E.T
    assert np.allclose(sol[2][rows].cpu().numpy(), exact, rtol=1e-4, atol=2e-5)
    n0 = y0d.double().norm(dim=1)
    n1 = sol[2].double().norm(dim=1)
    assert torch.allclose(n0, n1, rtol=2e-5)
    back = odeint(f, sol[2], torch.tensor([1.0, 0.0], device=dev), solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm})
    assert torch.allclose(back[1], y0d, rtol=1e-4, atol=5e-5)


# ----------------------------------------------------------------------------------------------
# seeded randomised sweep over the option space (fp64: tight bar, identical step decisions)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("block", range(_blocks(10)))
def test_randomised_adaptive_sweep_vs_oracle(dev, block):
    """8 random configurations per block: tableau, pipeline, tolerances, number and spacing of output times, direction of
    time, norm, first_step / min_step / max_step / safety / ifactor / dfactor / max_num_steps, step_t, time-dependent cubic dynamics.  Solution to
    1e-9 relative, identical accept / reject / NFE counts; the reference's assertion where the oracle raises it."""
    from paddlexde_amd.xde import BaseODE

    rng = np.random.RandomState(4242 + block)
    for case in range(8):
        name = list(ADAPTIVE)[rng.randint(len(ADAPTIVE))]
        pipeline = ("sync", "lag", "graph")[rng.randint(3)]
        if case % 4 == 3:
            pipeline = "auto"  # the default: resolves per solve (no extra random draw: the other cases stay what they were)
        B, D = int(rng.randint(1, 9)), int(rng.randint(2, 17))
        A = P.skew_matrix(D, seed=int(rng.randint(1, 100))).to(torch.float64)
        y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(int(rng.randint(1 << 30))), dtype=torch.float64)
        T = int(rng.randint(2, 8))
        t = np.sort(rng.uniform(0.0, 1.5, size=T))
        if rng.rand() < 0.3:
            t = t[::-1].copy()
        rtol = float(10 ** rng.uniform(-8, -4))
        atol = rtol * 1e-2
        if name in ("adaptive_heun", "fehlberg2"):
            rtol, atol = max(rtol, 1e-5), max(atol, 1e-7)
        opts = {}
        if rng.rand() < 0.3:
            opts["first_step"] = float(rng.uniform(1e-3, 5e-2))
        if rng.rand() < 0.3:
            opts["max_step"] = float(rng.uniform(0.05, 0.3))
        if rng.rand() < 0.3:
            opts["safety"] = float(rng.uniform(0.7, 0.95))
        if rng.rand() < 0.2:
            opts["ifactor"], opts["dfactor"] = float(rng.uniform(3, 12)), float(rng.uniform(0.1, 0.5))
        if rng.rand() < 0.25:
            lo, hi = min(t[0], t[-1]), max(t[0], t[-1])
            opts["step_t"] = np.sort(rng.uniform(lo, hi, size=int(rng.randint(1, 4))))
        linf = rng.rand() < 0.3
        if rng.rand() < 0.2:
            opts["min_step"] = float(10 ** rng.uniform(-3, -1.3))  # steps at or below it are accepted whatever the error
        if rng.rand() < 0.1:
            opts["max_num_steps"] = int(rng.randint(2, 12))  # "max_num_steps exceeded" where the oracle says so
        An = A.numpy()
        Ad = A.to(dev)

        def f_np(t_, y):
            return y @ An.T - 0.05 * (y * y * y) + 0.3 * t_

        def f_t(t_, y):
            return y @ Ad.T - 0.05 * (y * y * y) + 0.3 * t_

        o_opts = dict(opts, norm=O._linf_norm if linf else O._rms_norm, dtype=np.float64)
        tag = (block, case, name, pipeline, B, D, T, rtol, sorted(opts), linf)
        failure = None
        try:
            ref, so = O.odeint(f_np, y0.numpy(), t, name, rtol=rtol, atol=atol, options=o_opts, return_solver=True)
        except AssertionError as e:  # e.g. a forced grid point a rounding error away from a step end: "underflow in dt"
            failure = str(e).split(" ")[0]
        k_opts = dict(opts)
        if "step_t" in k_opts:
            k_opts["step_t"] = torch.from_numpy(k_opts["step_t"])
        xde = BaseODE(f_t, y0=y0.to(dev), t_span=torch.from_numpy(t))
        s = ADAPTIVE[name](xde=xde, y0=xde.y0, rtol=rtol, atol=atol, norm=_linf_norm if linf else _rms_norm, dtype=torch.float64,
                           pipeline=pipeline, **k_opts)
        if failure is not None:  # the same assertion, as the reference would raise it
            with pytest.raises(AssertionError, match=failure):
                s.integrate(torch.from_numpy(t))
            continue
        got = s.integrate(torch.from_numpy(t)).cpu().numpy()
        if not np.isfinite(ref).all():
            # a forced-accept option (min_step) can drive the cubic problem to overflow: then it overflows here too, at
            # the same entries, after the same number of steps
            assert np.array_equal(np.isfinite(got), np.isfinite(ref)), tag
            assert (s.stats["n_accept"], s.stats["n_reject"], s.stats["nfe"]) == (so.n_accept, so.n_reject, so.nfe), tag
            continue
        if min(rec.ratio for rec in so.trace) < 1e-5:
            # a step whose error estimate is below the round-off of its own terms (err/tol ~ 1e-9: a high-order pair on a
            # short first step): the ratio, hence the next dt, is rounding noise in ANY implementation (oracle 4.01e-9 vs
            # 4.05e-9 here), and the outputs carry the quartic interpolant's own error on steps placed slightly differently:
            # the two solutions agree as two valid integrations do, not to 1e-9
            assert P.rel_err(got, ref) <= 1e-4, (tag, P.rel_err(got, ref))
            continue
        assert P.parity_ok(got, ref, rtol=1e-9, atol=1e-11), (tag, P.worst(got, ref, 1e-9, 1e-11))
        assert (s.stats["n_accept"], s.stats["n_reject"], s.stats["nfe"]) == (so.n_accept, so.n_reject, so.nfe), tag


@pytest.mark.parametrize("block", range(_blocks(4)))
def test_randomised_fixed_sweep_vs_oracle(dev, block):
    """10 random configurations per block: solver, state shape ``[..., L, D]`` (0-3 leading axes), dtype, non-uniform /
    reversed grids, ``interp``, Adams order / corrector, time-dependent cubic dynamics built from +, -, * only — the whole
    trajectory BIT-EXACT and the same number of ``func`` evaluations."""
    rng = np.random.RandomState(9100 + block)
    for case in range(10):
        name = ("euler", "midpoint", "rk4", "adams", "adams_implicit")[rng.randint(5)]
        dtype = (np.float32, np.float64)[rng.randint(2)]
        lead = tuple(int(x) for x in rng.randint(1, 4, size=rng.randint(0, 4)))
        L, D = int(rng.randint(1, 4)), int(rng.randint(1, 6))
        y0 = rng.uniform(-1.0, 1.0, size=lead + (L, D)).astype(dtype)
        T = int(rng.randint(2, 24))
        t = np.cumsum(rng.uniform(0.005, 0.03, size=T)).astype(dtype) if rng.rand() < 0.5 else np.linspace(0.0, 0.4, T).astype(dtype)
        if rng.rand() < 0.25:
            t = t[::-1].copy()
        interp = ("linear", "cubic")[rng.randint(2)] if rng.rand() < 0.5 else "linear"
        w = rng.uniform(-1.0, 1.0, size=(D,)).astype(dtype)
        wt = torch.from_numpy(w).to(dev)
        calls = {"np": 0, "t": 0}

        def f_np(t_, y):
            calls["np"] += 1
            return -0.5 * y - 0.1 * (y * y * y) + w * t_ + 0.25 * (y * w)

        def f_t(t_, y):
            calls["t"] += 1
            return -0.5 * y - 0.1 * (y * y * y) + wt * t_ + 0.25 * (y * wt)

        o_opts = {"norm": O._rms_norm, "interp": interp}
        k_opts = {"norm": _rms_norm, "interp": interp}
        solver = FIXED[name.replace("_implicit", "")]
        if name.startswith("adams"):
            mo = int(rng.choice([3, 4, 6, 12]))
            o_opts["max_order"] = k_opts["max_order"] = mo
            k_opts["implicit"] = name.endswith("implicit")
        ref = O.odeint(f_np, y0, t, name, rtol=1e-3, atol=1e-4, options=o_opts)
        got = odeint(f_t, torch.from_numpy(y0).to(dev), torch.from_numpy(t).to(dev), solver=solver, rtol=1e-3, atol=1e-4, options=k_opts)
        tag = (block, case, name, dtype.__name__, lead, L, D, T, interp)
        assert tuple(got.shape) == ref.shape == lead + (T * L, D), tag
        # a high-order Adams run on a coarse uneven grid may blow up: then it does so here too, at the same entries (past the
        # overflow an entry may read inf on one side and nan on the other — 0 * inf of an operand this package skips)
        g = got.cpu().numpy()
        fin = np.isfinite(ref)
        # (where the reference's arithmetic has already overflowed — 0 * inf in its cubic interpolation, which here is the
        # identity at t == t1 — nothing is compared)
        assert np.array_equal(g[fin], ref[fin]), (tag, float(np.abs(g[fin] - ref[fin]).max(initial=0.0)))
        assert calls["np"] == calls["t"], (tag, calls)


@pytest.mark.parametrize("block", range(_blocks(5)))
def test_randomised_adaptive_sweep_fp32(dev, block):
    """The same option space in the DEFAULT precision (fp32 state, fp32 time-like scalars) against the oracle, REPLAYED: the
    device controller takes the oracle's own (dt, accept) sequence (xde_ctrl_params_t.replay), so a decision flipped by fp32
    round-off cannot hide an arithmetic difference behind "two valid integrations", and every configuration is held to
    north_star's relative bar element-wise, |got - ref| <= 1e-5 |ref| + 16 ulp of the state's scale (P.ulp_atol: func is a GEMM).
    The free-running solve of the same configuration must make the oracle's decisions wherever its error ratios are clear of
    the fp32 noise band around 1."""
    from paddlexde_amd.xde import BaseODE

    rng = np.random.RandomState(7300 + block)
    for case in range(8):
        name = ("dopri5", "bosh3", "fehlberg2", "adaptive_heun")[rng.randint(4)]
        pipeline = ("sync", "lag", "graph")[rng.randint(3)]
        if case % 4 == 3:
            pipeline = "auto"  # the default: resolves per solve (no extra random draw: the other cases stay what they were)
        B, D = int(rng.randint(1, 33)), int(rng.randint(2, 33))
        A = P.skew_matrix(D, seed=int(rng.randint(1, 100))).float()
        y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(int(rng.randint(1 << 30))))
        T = int(rng.randint(2, 8))
        t = np.sort(rng.uniform(0.0, 1.5, size=T)).astype(np.float32)
        if rng.rand() < 0.3:
            t = t[::-1].copy()
        rtol = float(10 ** rng.uniform(-5, -3))
        atol = rtol * 1e-2
        opts = {}
        if rng.rand() < 0.3:
            opts["first_step"] = float(rng.uniform(1e-3, 5e-2))
        if rng.rand() < 0.3:
            opts["max_step"] = float(rng.uniform(0.05, 0.3))
        if rng.rand() < 0.3:
            opts["safety"] = float(rng.uniform(0.7, 0.95))
        linf = rng.rand() < 0.3
        An, Ad = A.numpy(), A.to(dev)

        def f_np(t_, y):
            return y @ An.T - np.float32(0.05) * (y * y * y) + np.float32(0.3) * t_

        def f_t(t_, y):
            return y @ Ad.T - 0.05 * (y * y * y) + 0.3 * t_

        tag = (block, case, name, pipeline, B, D, T, rtol, sorted(opts), linf)
        failure = None
        try:
            ref, so = O.odeint(f_np, y0.numpy(), t, name, rtol=rtol, atol=atol, options=dict(opts, norm=O._linf_norm if linf else O._rms_norm),
                               return_solver=True)
        except AssertionError as e:  # e.g. a step that ends a rounding error short of an output time: "underflow in dt"
            failure = str(e).split(" ")[0]

        def make(**kw):
            return ADAPTIVE[name](xde=BaseODE(f_t, y0=y0.to(dev), t_span=torch.from_numpy(t)), y0=y0.to(dev), rtol=rtol, atol=atol,
                                  norm=_linf_norm if linf else _rms_norm, pipeline=pipeline, record_trace=True, **opts, **kw)

        s = make()
        if failure is not None:
            with pytest.raises(AssertionError, match=failure):
                s.integrate(torch.from_numpy(t))
            continue
        free = s.integrate(torch.from_numpy(t)).cpu().numpy()
        assert free.dtype == ref.dtype == np.float32 and free.shape == ref.shape and np.isfinite(free).all(), tag
        ratios = np.asarray([r.ratio for r in so.trace])
        if np.all(np.abs(ratios - 1.0) > 0.05) and len(s.trace) == len(so.trace):
            assert [a[3] for a in s.trace] == [r.accept for r in so.trace], tag  # same decisions when none is a coin toss
        # the arithmetic, on the oracle's step sequence
        r_ = make(_replay=[(rec.dt, rec.accept) for rec in so.trace])
        got = r_.integrate(torch.from_numpy(t)).cpu().numpy()
        assert [(abs(a[1]), a[3]) for a in r_.trace] == [(rec.dt, rec.accept) for rec in so.trace], tag  # (reverse time: signed dt here)
        # absolute part: 16 ulp of the state's scale (2e-6 max|ref|, a fifth of round 1's max-norm bar): up to a dozen steps of a
        # cubic, GEMM-driven flow carry the per-step last-bit differences of func further than the linear problems do
        # (measured worst over the 50-block soak: 6.8 ulp)
        atol_ulp = P.ulp_atol(ref, 16)
        assert P.parity_ok(got, ref, rtol=1e-5, atol=atol_ulp), (tag, P.worst(got, ref, 1e-5, atol_ulp))
