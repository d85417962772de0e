"""bench.py's side workloads (`python bench.py --workload c1|c3|c5|rk4|dense|dde`): the latency-bound configurations of BASELINE.json, the
fixed-step bandwidth line, dense output and the delay-equation kernels.  Not the driver's line — that is `bench.py`'s `main()`.
`ctx` carries what these share with it: `emit` (the one JSON line, on the process's real stdout), `make_problem`, `HBM_PEAK_GBS`."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def side_workload(args, ctx):
    """Configs 3 and 5 of BASELINE.json: latency-bound (state of 16-64 KB), reported as time per step."""
    import torch.nn as nn

    from paddlexde_amd import Dopri5, odeint_adjoint
    from paddlexde_amd.utils import _rms_norm
    from paddlexde_amd.xde import BaseODE

    dev = torch.device("cuda", 0)
    if args.workload == "c1":
        # BASELINE.json configs[0]: example/ode_demo.py's data generation (demo_utils.py:136-164) — spiral y' = (y^3) A,
        # y0 = [[2, 0]], t = linspace(0, 25, 1000), the reference's RK4.  Plumbing: 999 steps of a 2-element state.
        from oracle import xde_oracle as O  # checker only: the GPU trajectory must equal the oracle's bit for bit
        from paddlexde_amd import RK4, odeint
        from paddlexde_amd.utils import GraphedFunc

        A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]])
        An = A.numpy()
        y0 = torch.tensor([[2.0, 0.0]])
        t = torch.linspace(0.0, 25.0, 1000)
        Ad = A.to(dev)
        res = {}
        from paddlexde_amd.utils import _rms_norm

        plain = lambda t_, y: (y * y * y) @ Ad  # noqa: E731
        for label, func, pipeline in (("eager", plain, "sync"), ("GraphedFunc(func)", GraphedFunc(plain), "sync"),
                                      ("pipeline=graph (one captured step, replayed)", plain, "graph")):
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with torch.no_grad():
                    sol = odeint(func, y0.to(dev), t.to(dev), solver=RK4, options={"norm": _rms_norm, "pipeline": pipeline})
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            res[label] = {"seconds": el, "us_per_step": 1e6 * el / 999, "shape": list(sol.shape)}
            last = sol
        t0 = time.perf_counter()
        ref = O.odeint(lambda t_, y: (y * y * y) @ An, y0.numpy(), t.numpy(), "rk4")
        res["cpu_baseline"] = {"seconds": time.perf_counter() - t0, "kind": "port", "cores": 1, "sample": "the whole trajectory, numpy oracle"}
        res["bit_exact_vs_oracle"] = bool(np.array_equal(last.cpu().numpy(), ref))
        ctx.emit({"metric": "seconds for the 1000-point spiral trajectory (launch-latency-bound plumbing)", "workload": "c1: "
                          "example/ode_demo.py spiral, RK4 (reference variant), batch 1 x dim 2, 999 steps", "results": res})
        return
    if args.workload == "c5":
        mu = 1000.0

        def vdp(t, y):
            x, v = y[..., 0], y[..., 1]
            return torch.stack([v, mu * (1 - x * x) * v - x], dim=-1)

        y0 = (torch.tensor([2.0, 0.0]) + 0.01 * torch.randn(4096, 2, generator=torch.Generator().manual_seed(0))).to(dev)
        t = torch.tensor([0.0, 1.0])
        res = {}
        # "I" is the reference's controller (ode_utils.py:85-97); "PI" is the opt-in one BASELINE.json's config 5 names
        for controller in ("I", "PI"):
            for dtype in (torch.float32, torch.float64):
                y = y0.to(dtype)
                for rep in range(2):  # first repetition warms allocator / kernels up
                    xde = BaseODE(vdp, y0=y, t_span=t)
                    s = Dopri5(xde=xde, y0=y, rtol=1e-5, atol=1e-7, norm=_rms_norm, max_num_steps=10**6, pipeline=args.pipeline,
                               dtype=dtype, controller=controller)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    sol = s.integrate(t)
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t0
                st = s.stats
                res[controller + "/" + str(dtype).split(".")[-1]] = {
                    "n_accept": st["n_accept"], "n_reject": st["n_reject"], "nfe": st["nfe"], "seconds": el,
                    "us_per_attempted_step": 1e6 * el / max(st["n_steps"], 1), "finite": bool(torch.isfinite(sol).all())}
        ctx.emit({"metric": "us per attempted dopri5 step (latency-bound)", "workload": "c5: Van der Pol mu=1000, batch 4096 x 2, "
                          "t in [0,1], rtol 1e-5 atol 1e-7", "pipeline": args.pipeline, "tunable_op": bool(args.tunable_op), "results": res})
        return

    class ODEFunc(nn.Module):  # example/ode_demo.py:17-33
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(42)
            self.net = nn.Sequential(nn.Linear(2, 50), nn.Tanh(), nn.Linear(50, 2))
            for m in self.net:
                if isinstance(m, nn.Linear):
                    with torch.no_grad():
                        m.weight.copy_(0.1 * torch.randn(m.weight.shape, generator=g))
                        m.bias.zero_()

        def forward(self, t, y):
            return self.net(y**3)

    func = ODEFunc().to(dev)
    y0 = (torch.rand(8192, 2, generator=torch.Generator().manual_seed(0)) * 4 - 2).to(dev)
    t = torch.linspace(0.0, 25.0, 1000)[:32].to(dev)
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device=dev)
    res = {}
    from paddlexde_amd import RK4, odeint

    with torch.no_grad():
        y_true = odeint(lambda t_, y: (y**3) @ A, y0, t, solver=RK4)  # [8192*32... fixed layout: [T*B?]
    for name, solver in (("dopri5", Dopri5), ("rk4", RK4)):
        for rep in range(2):
            for p in func.parameters():
                p.grad = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            opts = {"norm": _rms_norm}
            if name == "dopri5":
                opts["pipeline"] = args.pipeline
            aopts = {k: v for k, v in opts.items() if k != "norm"}
            if name == "rk4":
                opts["pipeline"] = "graph"  # forward: one captured RK4 step replayed over the 31 intervals
            if "pipeline" in aopts:
                aopts["pipeline"] = "sync"  # adjoint intervals are 1-3 steps long: every attempt is resolved before the next (what the captured interval solve does anyway)
            if args.graph_func != "auto":
                aopts["graph_func"] = args.graph_func == "on"
            pred = odeint_adjoint(func, y0, t, solver=solver, rtol=1e-5, atol=1e-7, options=opts, adjoint_options=aopts)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            target = y_true if name == "rk4" else pred.detach() * 0.0
            loss = torch.mean(torch.abs(pred - target))
            loss.backward()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        gn = float(sum(p.grad.double().pow(2).sum() for p in func.parameters()).sqrt())
        res[name] = {"forward_s": t1 - t0, "backward_s": t2 - t1, "grad_norm": gn, "n_params": sum(p.numel() for p in func.parameters())}
    ctx.emit({"metric": "seconds per forward / adjoint backward (latency-bound)", "workload": "c3: spiral neural-ODE (2-50-2 MLP on y^3), "
                      "batch 8192, 32 output times, odeint_adjoint", "pipeline": args.pipeline, "graph_func": args.graph_func,
                      "tunable_op": bool(args.tunable_op), "results": res})


def rk4_workload(args, ctx):
    """The bandwidth-bound fixed-step line: the reference's RK4 (`rk4_alt_step_func`, solver/base_fixed_solver.py:166-197) on
    config 2's state (65536 x 128 fp32, func = torch matmul).  One step = 4 func calls + 3 FUSE stage combines (3, 4, 5 + 1
    arrays: the last one also emits the final sum's leading terms) + the final combine (4 arrays, written straight into the
    output slice): 17 N 4 B = 570 MB (18 N before round 5's pre-summing; `--rk4-presum off` measures that form: same bits)."""
    from paddlexde_amd import RK4, _hip, odeint
    from paddlexde_amd.solver.base_fixed_solver import FixedSolver
    from paddlexde_amd.utils import _rms_norm

    if args.rk4_presum == "off":  # (A/B only: the full final combine)
        FixedSolver._presum_ok = lambda self, y0, ks: False

    dev = torch.device("cuda", 0)
    B = 65536 if args.batch is None else args.batch
    D = 128 if args.dim is None else args.dim
    A, y0 = ctx.make_problem(B, D, 0, dev)
    AT = A.T.contiguous()
    func = lambda t, y: y @ AT  # noqa: E731
    K, W = args.steps, args.warmup
    t = torch.linspace(0.0, 0.05 * K, K + 1, device=dev)
    be = _hip.get_backend()
    with torch.no_grad():
        for _ in range(max(1, -(-W // K))):  # >= W untimed steps, with the timed pass's own shapes (allocator warm)
            odeint(func, y0, t, solver=RK4, options={"norm": _rms_norm})
        torch.cuda.synchronize()
        if not args.no_kernel_events:
            be.prof_enable(args.event_period)  # (11: coprime with the 3 FUSE launches per step, all stages sampled)
        t0 = time.perf_counter()
        sol = odeint(func, y0, t, solver=RK4, options={"norm": _rms_norm})
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    prof = None
    if not args.no_kernel_events:
        prof = be.prof_collect()
        be.prof_enable(False)
    N = B * D
    out = {
        "metric": "integrated states/sec (batch*dim/step_time) rk4 (reference variant), fixed step",
        "value": N * K / elapsed, "unit": "states/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "linear ODE dy/dt=Ay, RK4 reference variant (rk4_alt_step_func), batch={} x dim={}, {} fixed steps, "
                               "func = torch matmul; final combine {}".format(B, D, K, "pre-summed (17 N per step)" if args.rk4_presum == "on" else "full (18 N per step)"),
                   "global_batch": B, "dim": D},
        "finite": bool(torch.isfinite(sol[-B:]).all()),
    }
    if prof is not None:
        kern = {}
        for name in ("combine_fuse", "combine_wfuse"):
            rec = prof[name]
            if rec["launches"]:
                us = 1e3 * rec["ms"] / rec["launches"]
                gbs = rec["bytes"] / (rec["ms"] * 1e-3) / 1e9
                kern[name] = {"launches": rec["launches"], "avg_us": us, "algorithmic_GBps": gbs, "frac": gbs / ctx.HBM_PEAK_GBS,
                              "bytes_per_launch": rec["bytes"] / rec["launches"]}
        out["kernels"] = kern
        ms = prof["combine_fuse"]["ms"] + prof["combine_wfuse"]["ms"]
        by = prof["combine_fuse"]["bytes"] + prof["combine_wfuse"]["bytes"]
        if ms > 0:
            a = by / (ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "xde_combine_kernel<float, FUSE|WFUSE, vec> (3 + 1 launches per step)",
                               "achieved": a, "peak": ctx.HBM_PEAK_GBS, "unit": "GB/s", "frac": a / ctx.HBM_PEAK_GBS, "traffic": None}
            if kern:
                per_step = 3 * kern.get("combine_fuse", {}).get("avg_us", 0.0) + kern.get("combine_wfuse", {}).get("avg_us", 0.0)
                out["solver_kernel_ms_per_step"] = per_step * 1e-3
    ctx.emit(out)


def dense_workload(args, ctx):
    """Dense output on config 2's state (A10: `_interp_fit` + `interp_evaluate`, base_adaptive_solver_rk.py:286-292, ode_utils.py:28-77 —
    here ONE lazy launch per accepted step that covers an output time, coefficients never materialised).  The real solve:
    65536 x 128 fp32, Dopri5, t in [0, 1], T = 11 output times, sync pipeline (so that only covering steps launch the kernel).
    Algorithmic bytes per launch (9 N per output row): reads k0,k2..k6,y0,y1 = 8 N, writes `rows` N."""
    from paddlexde_amd import Dopri5, _hip, odeint
    from paddlexde_amd.utils import _rms_norm

    dev = torch.device("cuda", 0)
    B = 65536 if args.batch is None else args.batch
    D = 128 if args.dim is None else args.dim
    T = 11
    A, y0 = ctx.make_problem(B, D, 0, dev)
    AT = A.T.contiguous()
    func = lambda t, y: y @ AT  # noqa: E731
    t = torch.linspace(0.0, 1.0, T)
    be = _hip.get_backend()
    opts = {"norm": _rms_norm, "pipeline": "sync"}
    with torch.no_grad():
        odeint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options=opts)  # warm-up (allocator, GEMM tuning)
        torch.cuda.synchronize()
        reps = max(1, args.steps // 10)
        be.prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(reps):
            sol = odeint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options=opts)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    prof = be.prof_collect()
    be.prof_enable(False)
    rec = prof["dense"]
    N = B * D
    rows = (T - 1) * reps
    by = (8.0 * rec["launches"] + rows) * N * 4.0
    a = by / (rec["ms"] * 1e-3) / 1e9 if rec["ms"] > 0 else 0.0
    # the same solve with 2 output times: what the 9 extra rows cost end to end
    with torch.no_grad():
        t2 = torch.tensor([0.0, 1.0])
        odeint(func, y0, t2, solver=Dopri5, rtol=1e-5, atol=1e-7, options=opts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            odeint(func, y0, t2, solver=Dopri5, rtol=1e-5, atol=1e-7, options=opts)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
    ctx.emit({"metric": "dense-output rows per second (lazy xde_dense_eval inside a Dopri5 solve)", "value": rows / el, "unit": "rows/s", "n_gpus": 1,
          "steps": reps, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
          "config": {"workload": "config 2's state {} x {} fp32, Dopri5 t in [0,1], T = {} output times, sync pipeline".format(B, D, T)},
          "solve_ms_T11": 1e3 * el / reps, "solve_ms_T2": 1e3 * el2 / reps, "finite": bool(torch.isfinite(sol[-1]).all()),
          "roofline": {"bound": "hbm", "kernel": "xde_dense_kernel<float, float, vec> (one launch per accepted step that covers output times)",
                       "achieved": a, "peak": ctx.HBM_PEAK_GBS, "unit": "GB/s", "frac": a / ctx.HBM_PEAK_GBS, "traffic": None,
                       "launches": rec["launches"], "rows": rows, "avg_launch_us": 1e3 * rec["ms"] / max(rec["launches"], 1),
                       "bytes_per_launch": by / max(rec["launches"], 1)}})


def dde_workload(args, ctx):
    """The delay-equation caller's history gather (HistoryIndex, xde/base_dde.py:82-127 over the cubic-Hermite
    spline of interpolation/interpolate.py:100-204) at the reference application's size (D3STN, PeMS04-like: 307 nodes x batch 32 =
    9824 series, 288 history times, 64 channels, 12 learned lags): value AND time derivative at the lags in one pass,
    xde_hermite_gather.  Algorithmic bytes: 3 history rows in + value + derivative out = 5 x (series x lags x channels) x 4 B.
    (Parity of this kernel against the oracle's history_index: tests/_dde_cases.py.)"""
    from paddlexde_amd import _hip

    dev = torch.device("cuda", 0)
    be = _hip.get_backend()
    g = torch.Generator().manual_seed(0)
    S, T, D, L = 9824, 288, 64, 12
    his = torch.randn(S, T, D, generator=g).to(dev)
    his_t = torch.linspace(0.0, 287.0, T).to(dev)
    lags = (torch.rand(L, generator=g) * 287.0).to(dev)
    val, der = torch.empty(S, L, D, device=dev), torch.empty(S, L, D, device=dev)
    for _ in range(args.warmup):
        be.hermite_gather(val, der, his, his_t, lags)
    torch.cuda.synchronize()
    be.prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        be.hermite_gather(val, der, his, his_t, lags)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rec = be.prof_collect()["dense"]
    be.prof_enable(False)
    by = 5.0 * S * L * D * 4.0
    a = by * rec["launches"] / (rec["ms"] * 1e-3) / 1e9 if rec["ms"] > 0 else 0.0
    out = {"metric": "history-spline gathers per second (xde_hermite_gather, value + derivative)", "value": args.steps / el, "unit": "gathers/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "D3STN-sized history: {} series x {} times x {} channels, {} lags".format(S, T, D, L)},
           "roofline": {"bound": "hbm", "kernel": "xde_hermite_vec_kernel<float>", "achieved": a, "peak": ctx.HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": a / ctx.HBM_PEAK_GBS, "traffic": None, "avg_launch_us": 1e3 * rec["ms"] / max(rec["launches"], 1),
                        "bytes_per_launch": by, "history_bytes": float(S) * T * D * 4.0}}
    # the other two history splines (interp_method "linear" / "bez": 2 / 4 rows in, value + derivative out) and HistoryIndex.backward
    # (xde_lag_grad: grad_y and derivative in, L numbers out — one launch instead of a framework multiply + sum)
    extra = {}
    for method, rows in (("linear", 2), ("bez", 4)):
        for _ in range(args.warmup):
            be.history_gather(val, der, his, his_t, lags, method)
        be.prof_enable(1)
        for _ in range(args.steps):
            be.history_gather(val, der, his, his_t, lags, method)
        r2 = be.prof_collect()["dense"]
        be.prof_enable(False)
        b2 = float(rows + 2) * S * L * D * 4.0
        a2 = b2 * r2["launches"] / (r2["ms"] * 1e-3) / 1e9 if r2["ms"] > 0 else 0.0
        extra["gather_" + method] = {"avg_launch_us": 1e3 * r2["ms"] / max(r2["launches"], 1), "bytes_per_launch": b2, "achieved": a2,
                                     "frac": a2 / ctx.HBM_PEAK_GBS}
    gy = torch.randn(S, L, D, generator=g).to(dev)
    for _ in range(args.warmup):
        be.lag_grad(gy, der)
    be.prof_enable(1)
    for _ in range(args.steps):
        be.lag_grad(gy, der)
    r3 = be.prof_collect()["dense"]
    be.prof_enable(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        (gy * der).sum(dim=(0, 2))
    torch.cuda.synchronize()
    fw = (time.perf_counter() - t0) / args.steps
    b3 = 2.0 * S * L * D * 4.0
    a3 = b3 * r3["launches"] / (r3["ms"] * 1e-3) / 1e9 if r3["ms"] > 0 else 0.0
    extra["lag_grad"] = {"kernel": "xde_lag_grad_plane_kernel<float, vec>", "avg_launch_us": 1e3 * r3["ms"] / max(r3["launches"], 1), "bytes_per_launch": b3,
                         "achieved": a3, "frac": a3 / ctx.HBM_PEAK_GBS, "framework_multiply_plus_sum_us": 1e6 * fw}
    out["history_index"] = extra
    ctx.emit(out)
