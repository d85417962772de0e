"""Where a whole `odeint()` call at config 2 (65536 x 128 fp32, Dopri5, t in [0, 1], T = 2) spends what it spends on top of its
attempted steps (`profiles/r06_odeint_tail.txt`).

    python3 profiles/tools/odeint_tail.py                       # phases by host clock and by GPU events, per call
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 profiles/tools/odeint_tail.py --calls 6 --plain
    python3 profiles/tools/odeint_tail.py --trace DIR           # one call's kernels from that trace: busy time, gaps, the tail's launches

Phases (events are recorded on the solver's stream, no synchronisation is added inside a call):
  entry      odeint() entered -> _before_integrate entered: BaseODE, solver construction, result tensor + row 0 copy
  heuristic  _before_integrate: buffers, uploads, f0, f0 again (the reference's NFE), the initial-step heuristic, the control block
  attempts   _run: every attempted step, dense rows, the wait for the last verdict
  exit       _run returned -> odeint() returned
"""
import argparse
import collections
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def problem():
    import torch

    from tests import problems as P

    B, D = 65536, 128
    A = P.skew_matrix(D).float().cuda()
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(0)).cuda()
    return A, y0


def run(calls, plain):
    import torch

    from paddlexde_amd import Dopri5, odeint
    from paddlexde_amd.solver.base_adaptive_solver_rk import AdaptiveRKSolver
    from paddlexde_amd.utils import _rms_norm

    A, y0 = problem()
    func = lambda t, y: y @ A.T  # noqa: E731
    t = torch.linspace(0.0, 1.0, 2)
    marks = []

    def mark(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append((name, time.perf_counter(), ev))

    if not plain:
        before, runner = AdaptiveRKSolver._before_integrate, AdaptiveRKSolver._run

        def _before(self, t_span):
            mark("heuristic>")
            out = before(self, t_span)
            mark("heuristic<")
            return out

        def _run(self, solution):
            mark("attempts>")
            out = runner(self, solution)
            mark("attempts<")
            return out

        AdaptiveRKSolver._before_integrate, AdaptiveRKSolver._run = _before, _run
    rows = []
    for call in range(calls + 2):
        st = {}
        del marks[:]
        torch.cuda.synchronize()
        if plain:
            torch.cuda._sleep(1)  # a launch of its own name (spin_kernel) in front of every call: the trace reduction splits on it
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if not plain:
            mark("entry>")
        with torch.no_grad():
            sol = odeint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm, "stats_out": st})
        if not plain:
            mark("exit<")
        t_ret = time.perf_counter()
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        if call < 2:
            continue  # warm-up
        row = {"total_ms": 1e3 * (t_end - t0), "host_returned_ms": 1e3 * (t_ret - t0), "attempts": st["n_steps"], "nfe": st["nfe"]}
        if not plain:
            m = {n: (h, e) for n, h, e in marks}
            for name, a, b in (("entry", "entry>", "heuristic>"), ("heuristic", "heuristic>", "heuristic<"), ("attempts", "attempts>", "attempts<"),
                               ("exit", "attempts<", "exit<")):
                row[name + "_host_ms"] = 1e3 * (m[b][0] - m[a][0])
                row[name + "_gpu_ms"] = m[a][1].elapsed_time(m[b][1])
            row["between_heuristic_and_attempts_host_ms"] = 1e3 * (m["attempts>"][0] - m["heuristic<"][0])
            row["gpu_first_to_last_event_ms"] = m["entry>"][1].elapsed_time(m["exit<"][1])
        rows.append(row)
        del sol
    keys = list(rows[0])
    print("# odeint(func, y0 [65536, 128] fp32, t = [0, 1], Dopri5, rtol 1e-5 / atol 1e-7): {} calls after 2 warm-up calls{}".format(
        calls, " (no phase marks: the figure bench.py's odeint_ms_T2 measures)" if plain else ""))
    print("# *_host_ms: the host's clock between the marks (enqueue time); *_gpu_ms: HIP events on the stream at the same marks")
    for k in keys:
        vals = sorted(r[k] for r in rows)
        print("{:44s} median {:9.3f}   min {:9.3f}   max {:9.3f}".format(k, vals[len(vals) // 2], vals[0], vals[-1]))


def trace(directory, calls):
    path = glob.glob(directory + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a call = the launches after one marker launch (spin_kernel, enqueued by run(plain=True) in front of every call) up to the next
    groups, cur = [], None
    for row in rows:
        if "spin_kernel" in row[2]:
            if cur:
                groups.append(cur)
            cur = []
        elif cur is not None:
            cur.append(row)
    if cur:
        groups.append(cur)
    groups = groups[-calls:]
    print("# kernels of the last {} odeint() calls in {} (a call: the launches between two marker launches)".format(len(groups), path))
    for gi, g in enumerate(groups):
        span = (g[-1][1] - g[0][0]) / 1e3
        busy = sum(e - s for s, e, _ in g) / 1e3
        print("call {}: {} launches, first start -> last end {:.1f} us, inside kernels {:.1f} us, idle between launches {:.1f} us".format(
            gi, len(g), span, busy, span - busy))
    g = groups[-1]
    print("# the last call, launch by launch up to the first stage combine of the first attempt, and from the last controller on:")
    names = [n for _, _, n in g]

    def short(n):
        return n.split("(")[0][:90]

    first_ctrl = next(i for i, n in enumerate(names) if "xde_control_kernel" in n or "xde_errnorm" in n)
    head_end = max(i for i in range(first_ctrl) if "xde_initial_step" in names[i] or "ctrl_init" in names[i]) + 1  # first launch of attempt 1
    t_first = g[0][0]
    for i in range(head_end + 1):
        s, e, n = g[i]
        gap = (s - g[i - 1][1]) / 1e3 if i else 0.0
        print("  +{:8.1f} us  dur {:7.2f}  gap-before {:6.2f}  {}".format((s - t_first) / 1e3, (e - s) / 1e3, gap, short(n)))
    print("  ... {} launches of the attempted steps ...".format(len(g) - head_end - 1))
    last_ctrl = max(i for i, n in enumerate(names) if "xde_control_kernel" in n)
    for i in range(last_ctrl - 1, len(g)):
        s, e, n = g[i]
        print("  +{:8.1f} us  dur {:7.2f}  gap-before {:6.2f}  {}".format((s - t_first) / 1e3, (e - s) / 1e3, (s - g[i - 1][1]) / 1e3, short(n)))
    head = sum(e - s for s, e, _ in g[: head_end]) / 1e3
    print("# heuristic + set-up launches: {} launches, {:.1f} us inside kernels, {:.1f} us from the call's first launch to the first attempt's first launch".format(
        head_end, head, (g[head_end][0] - t_first) / 1e3))
    per = collections.Counter(short(n) for n in names[:head_end])
    for k, v in per.most_common():
        print("    {:3d} x {}".format(v, k))


def host_profile(calls):
    """The HOST's share of a call: a state just above the one-workgroup kernels' reach (1024 x 128: the same code path as config 2,
    no GPU time to speak of) over an interval of ONE attempted step, under cProfile."""
    import cProfile
    import pstats

    import torch

    from paddlexde_amd import Dopri5, odeint
    from paddlexde_amd.utils import _rms_norm
    from tests import problems as P

    A = P.skew_matrix(128).float().cuda()
    y0 = torch.randn(1024, 128, generator=torch.Generator().manual_seed(0)).cuda()
    func = lambda t, y: y @ A.T  # noqa: E731
    t = torch.tensor([0.0, 1e-3])

    def one():
        st = {}
        with torch.no_grad():
            odeint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm, "pipeline": "lag", "stats_out": st})
        return st

    for _ in range(20):
        st = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        one()
    host = (time.perf_counter() - t0) / calls
    torch.cuda.synchronize()
    print("# host time of a whole odeint() call of {} attempt(s), {} func evaluations (1024 x 128 state, lag pipeline): {:.1f} us".format(
        st["n_steps"], st["nfe"], 1e6 * host))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(calls):
        one()
    pr.disable()
    torch.cuda.synchronize()
    ps = pstats.Stats(pr, stream=sys.stdout)
    ps.sort_stats("cumulative").print_stats(45)


def host_sections(calls):
    """Host time (perf_counter, inclusive) of the pieces of a call's set-up — everything before the first attempted step — at config 2's
    size, where allocations are 32 MiB blocks and the GPU is busy: wrappers around the functions, no profiler."""
    import torch

    from paddlexde_amd import Dopri5, _hip, odeint
    from paddlexde_amd.solver import _common
    from paddlexde_amd.solver.base_adaptive_solver import AdaptiveSolver
    from paddlexde_amd.solver.base_adaptive_solver_rk import AdaptiveRKSolver
    from paddlexde_amd.utils import _rms_norm
    from paddlexde_amd.xde import BaseODE

    A, y0 = problem()
    func = lambda t, y: y @ A.T  # noqa: E731
    t = torch.linspace(0.0, 1.0, 2)
    acc = collections.defaultdict(float)
    cnt = collections.Counter()
    state = {"on": False}

    def timed(owner, name, label=None):
        fn = getattr(owner, name)
        label = label or name

        def wrapper(*a, **k):
            if not state["on"]:
                return fn(*a, **k)
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[label] += time.perf_counter() - t0
                cnt[label] += 1

        setattr(owner, name, wrapper)

    be = _hip.get_backend()
    for name in ("_setup", "_eval", "_select_initial_step_device", "_before_integrate", "__init__"):
        timed(AdaptiveRKSolver, name, "solver." + name)
    timed(AdaptiveSolver, "integrate", "solver.integrate (whole)")
    timed(BaseODE, "__init__", "BaseODE.__init__")
    for name in ("scaled_norm2_partial", "initial_step_tail", "stage_combine", "scaled_norm_partial", "acquire_work"):
        timed(be, name, "backend." + name)
    timed(_common, "upload", "upload")
    import paddlexde_amd.solver.base_adaptive_solver_rk as M

    timed(M, "upload", "upload")
    timed(M, "scalar", "scalar (torch.full)")
    run_ = AdaptiveRKSolver._run

    def _run(self, solution):
        state["on"] = False  # the attempts are not part of the set-up
        return run_(self, solution)

    AdaptiveRKSolver._run = _run
    totals = []
    for call in range(calls + 3):
        torch.cuda.synchronize()
        state["on"] = call >= 3
        t0 = time.perf_counter()
        with torch.no_grad():
            odeint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm})
        state["on"] = False
        torch.cuda.synchronize()
        if call >= 3:
            totals.append(time.perf_counter() - t0)
    print("# host time per call of the pieces of the set-up of odeint() at config 2 (65536 x 128), {} calls; the wrappers cost ~0.3 us each".format(calls))
    for k in sorted(acc, key=lambda k: -acc[k]):
        print("{:42s} {:8.1f} us   ({:.0f} x per call)".format(k, 1e6 * acc[k] / calls, cnt[k] / calls))
    totals.sort()
    print("whole call, median {:.3f} ms".format(1e3 * totals[len(totals) // 2]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=9)
    ap.add_argument("--plain", action="store_true", help="no phase marks (for a rocprofv3 run, and for the unmarked total)")
    ap.add_argument("--trace", default=None, help="directory of a rocprofv3 --kernel-trace run of this script: reduce it")
    ap.add_argument("--host-profile", action="store_true")
    ap.add_argument("--host-sections", action="store_true")
    a = ap.parse_args()
    if a.host_sections:
        host_sections(20)
    elif a.host_profile:
        host_profile(200)
    elif a.trace:
        trace(a.trace, min(a.calls, 3))
    else:
        run(a.calls, a.plain)
