#!/usr/bin/env python3
"""bench.py — headline benchmark (BASELINE.json): integrated states/sec of adaptive Dopri5 steps.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): this process starts the N ranks itself (a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`), relays rank 0's
line and exits non-zero if a rank fails, if the box has fewer than N GPUs (unless XDE_BENCH_REHEARSAL=1: all ranks on cuda:0 over
gloo), if the line does not say n_gpus == N, or — exit code 124, with the stage every rank was in — if the job outlives
XDE_BENCH_TIMEOUT seconds.  Under a launcher (the driver's torch.distributed.run) it is one rank of the job.  Either way every rank
runs each stage (rendezvous, transport probe, communicator, negotiation, timed region, extras, teardown) under a watchdog.

Set-up (untimed, like the framework's GEMM tuning) ends with XDE_BENCH_SETTLE_MS (default 100) milliseconds of attempts — their number is
`solver.settle_steps` on the line — so that a short timed block reads what the same block reads when repeated (profiles/r05_settle.txt);
then W warm-up steps, then EXACTLY K timed attempts; the same K-step block is repeated twice more (`ms_per_step_blocks`).

A "step" is ONE attempted Dopri5 step over the whole batch: 6 stage combines (xde_stage_combine), 6 calls
of the user's func (a framework call: torch matmul ``y @ A^T``), one fused error-norm launch and the device
controller — exactly what ``paddlexde_amd.odeint(..., solver=Dopri5)`` runs per attempt.

Workload (N=1): BASELINE.json configs[1] — linear ODE dy/dt = A y, A = U - U^T (seed 1), batch 65536 x dim
128 fp32, y0 = randn (seed 0), rtol 1e-5 / atol 1e-7, inputs resident in HBM before the timed region.
N>1: BASELINE.json configs[3] — the same ODE at GLOBAL batch 524288 x dim 64, rows split evenly over the N ranks
(rank r owns rows [r*B/N, (r+1)*B/N): 65536 x 64 per GPU at N=8); the total work is the same for every N > 1
("scaling": "strong"); the only exchange on the data path is that of the error norm's partial sums (32 doubles) per attempted
step — by `--exchange auto`: one-shot stores into IPC-mapped peer mailboxes over xGMI fused with the controller launch if that
transport's probe (child processes) succeeds, else ncclAllReduce on the solver's stream, else the nccl group's all_reduce; the
line says which (`norm_exchange`, `norm_exchange_report`), times the others too (`exchange_ab`), and lists each rank's device and
the peer-access matrix (`rccl_ranks` = size of the nccl group the job formed and checked).  Because the driver's `--gpus 1` line is
config 2 — a different amount of work — the N > 1 line carries `n1_same_workload`: rank 0's own single-GPU run of the SAME global 524288 x 64 problem,
taken after the timed region, so a strong-scaling efficiency can be computed from one line.  `--batch` (rows PER GPU) / `--dim` override either default (then "weak").
`--workload rk4`: the bandwidth-bound fixed-step line (reference RK4 variant, 65536 x 128, 18 N 4 B per step).

value = states/sec = (global batch * dim) / (wall time per attempted step), whole job.
roofline: bytes the stage-combine launches really move (SURVEY 8d counts (operands + 2) * N * 4 B per stage = 32 N * 4 B per Dopri5
step; with stage 5 pre-summed — it reads y0, the partial sum stage 4's launch emitted, and k4 — the step moves 30 N, + 1 N for the
partial error estimate) / their launch durations measured with HIP events on the launch stream.
cpu_baseline: "B1", the oracle's torch-CPU twin (op-for-op restatement of the reference's eager op sequence — the reference's
Paddle CPU path itself cannot run: Paddle is not installed and the reference never travels to the GPU box), timed on the host cores
of this box on a bounded sample (~10 s of attempted steps at the SAME batch x dim); cpu_baseline_fused: "B2", the same step on fused
C/OpenMP kernels (oracle/xde_cpu_kernels.c) — the strong CPU baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_launch import Watchdog, _free_port, _stage_report, run_p2p_probe, self_launch  # noqa: E402,F401

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SETTLE_MS = float(os.environ.get("XDE_BENCH_SETTLE_MS", "100"))  # milliseconds of untimed attempts in set-up, before the W warm-up steps (timed_run)
_REAL_STDOUT = None


def emit(obj):
    """Print the result line on the process's real stdout (see main)."""
    sys.stdout.flush()
    if _REAL_STDOUT is not None:
        os.dup2(_REAL_STDOUT, 1)
    print(json.dumps(obj), flush=True)


def make_problem(B, D, rank, device):
    g = torch.Generator().manual_seed(1)
    U = 0.1 * torch.randn(D, D, generator=g)
    A = (U - U.T).contiguous()
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(rank))
    return A.to(device), y0.to(device)


def _cpu_share():
    """Host threads this process may really use: scheduler affinity, capped by the cgroup CPU quota (a GPU box
    exposes all 256 host CPUs but grants a share of ~16 per GPU) and by 32."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(-(-int(txt[0]) // int(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, -(-q // per)))
            break
        except Exception:
            continue
    return max(1, min(n, int(os.environ.get("XDE_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(B, D, budget_s=10.0):
    """The oracle ("port") on a bounded sample of the SAME workload: batch B x dim D (the GPU run's own size), attempted
    Dopri5 steps for ~budget_s, on every granted host core (oracle/xde_oracle_torch.py: the reference's eager op sequence
    on torch-CPU tensors, checked against the numpy oracle by tests/test_oracle_pinning.py).  The first step size is the
    numpy oracle's (Hairer's heuristic on the same inputs)."""
    from oracle import xde_oracle as O
    from oracle import xde_oracle_torch as OT

    g = torch.Generator().manual_seed(1)
    U = 0.1 * torch.randn(D, D, generator=g)
    A = (U - U.T).contiguous()
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(0))
    AT = A.T.contiguous()
    ATn = AT.numpy()
    s = O.AdaptiveRKSolver(lambda t, y: y @ ATn, y0.numpy(), 1e-5, 1e-7, method="dopri5", norm=O._rms_norm)
    s._before_integrate(np.asarray([0.0, 1e9], dtype=np.float32))
    cores = _cpu_share()
    torch.set_num_threads(cores)
    tw = OT.TorchAdaptiveStepper(lambda t, y: y @ AT, y0, 1e-5, 1e-7)
    tw.start(0.0, float(s.rk_state.dt))
    for _ in range(3):
        tw.step()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        tw.step()
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= 8) or n >= 2000:
            break
    eager = {
        "value": B * D * n / el,
        "unit": "states/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": "B1: torch-CPU twin of the oracle (the reference's eager op sequence, ~10 element-wise ops per stage), {} attempted "
                  "dopri5 steps, batch {} x dim {} fp32, {:.1f} s, {} threads".format(n, B, D, el, torch.get_num_threads()),
    }
    # B2: the strong CPU baseline — the same step on fused C + OpenMP kernels (oracle/xde_cpu_kernels.c: one
    # pass per stage, error estimate fused into the last stage and the norm pass, like the HIP path), func = the
    # framework's multi-threaded CPU GEMM.  B1 — the reference's own op sequence, which is what "the reference's CPU path"
    # means — is the reported cpu_baseline; B2 rides along as cpu_baseline_fused.
    try:
        from oracle import xde_cpu_fused as F

        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        fs = F.FusedDopri5Stepper(lambda t, y: y @ AT, y0, 1e-5, 1e-7)
        fs.start(0.0, float(s.rk_state.dt))
        for _ in range(3):
            fs.step()
        n2, t0 = 0, time.perf_counter()
        while True:
            fs.step()
            n2 += 1
            el2 = time.perf_counter() - t0
            if (el2 > budget_s and n2 >= 8) or n2 >= 2000:
                break
        fused = {
            "value": B * D * n2 / el2,
            "unit": "states/s",
            "cores": cores,
            "kind": "port",
            "sample": "B2: fused C/OpenMP statement of the step (oracle/xde_cpu_kernels.c) + torch-CPU GEMM for func, {} attempted dopri5 "
                      "steps, batch {} x dim {} fp32, {:.1f} s, {} threads".format(n2, B, D, el2, cores),
        }
        return eager, fused
    except Exception as e:  # no compiler on the box: the eager port alone
        eager["sample"] += " (B2 unavailable: {})".format(type(e).__name__)
        return eager, None


def enable_tunable_op(on):
    """The user's func is a framework call; its GEMMs are the framework's to pick.  PyTorch's own tuner (TunableOp) times the
    hipBLASLt / rocBLAS candidates for each GEMM shape on first use (inside warm-up) and keeps the fastest: 25.7 -> 21.5 us for
    config 2's [65536,128]x[128,128], 37 -> ~6 us for config 3's skinny weight-gradient products.  Returns whether it is on."""
    if not on:
        return False
    try:
        import torch.cuda.tunable as tunable

        tunable.enable(True)
        tunable.tuning_enable(True)
    except Exception:
        return False
    import tempfile

    for call, arg in (("write_file_on_exit", False),  # (not in every release)
                      ("set_filename", os.path.join(tempfile.gettempdir(), "xde_bench_tunableop.csv")),  # keep the tree clean
                      ("set_max_tuning_duration", 30), ("set_max_tuning_iterations", 50)):
        try:
            getattr(tunable, call)(arg)
        except Exception:
            pass
    return True


def time_unsharded(B, D, dtype, pipeline, device, steps, warmup):
    """One rank, no process group: attempted Dopri5 steps of the linear ODE at batch B x dim D.  Used by the N > 1 line for
    `n1_same_workload` (config 4's GLOBAL problem on one GPU — the N=1 point of the strong-scaling curve)."""
    from paddlexde_amd import Dopri5
    from paddlexde_amd.utils import _rms_norm
    from paddlexde_amd.xde import BaseODE

    A, y0 = make_problem(B, D, 0, device)
    if dtype == "f64":
        A, y0 = A.double(), y0.double()
    AT = A.T.contiguous()
    func = lambda t, y: y @ AT  # noqa: E731
    xde = BaseODE(func, y0=y0, t_span=torch.tensor([0.0, 1.0e9]))
    s = Dopri5(xde=xde, y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm, pipeline=pipeline)
    s.y0 = y0
    s._before_integrate(np.asarray([0.0, 1.0e9], dtype=np.float32))
    settle = 0
    if SETTLE_MS > 0 and os.environ.get("XDE_BENCH_REHEARSAL", "0") != "1":  # (the same settle phase as the main line's: timed_run)
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        s.advance(32)
        torch.cuda.synchronize()
        more = min(4096, max(0, int(SETTLE_MS * 1e-3 * 32 / max(time.perf_counter() - t_s, 1e-6)) - 32))
        s.advance(more)
        settle = 32 + more
    s.advance(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.advance(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"value": B * D * steps / el, "unit": "states/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": warmup,
            "settle_steps": settle, "global_batch": B, "dim": D, "n_gpus": 1}


def pmc_traffic(B, D, dtype):
    """HBM bytes per launch of the stage-combine kernel from the committed rocprofv3 counter passes (`--pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE`, separate runs, reduced by profiles/tools/pmc_summarise.py) -> (bytes or None, where it comes from).
    A counter pass cannot run inside this process, so the figure is a RECORDED measurement of this exact kernel source: the file
    carries the stamp of the kernel's sources (csrc/build.py::kernel_stamp) it was taken with, and a figure whose stamp is not the
    stamp of the sources this library was built from is reported as null, with the reason."""
    tpath = os.path.join(ROOT, "profiles", "traffic_combine.json")
    key = "{}x{}/{}".format(B, D, dtype)
    try:
        from paddlexde_amd.csrc.build import kernel_stamp

        rec = json.load(open(tpath)).get("by_size", {}).get(key)
        if rec is None:
            return None, "no counter pass recorded for {} in profiles/traffic_combine.json".format(key)
        now = kernel_stamp("combine")
        src = {"file": rec.get("source"), "round": rec.get("round"), "kernel_stamp": rec.get("kernel_stamp"), "current_kernel_stamp": now}
        if rec.get("kernel_stamp") != now:
            src["stale"] = "the stage-combine kernel's sources changed after this counter pass was taken: not reported"
            return None, src
        if rec.get("rocprofv3_avg_launch_us"):  # the same kernel's mean launch duration by rocprofv3 (kernel summary of the same profile run)
            src["rocprofv3_avg_launch_us"] = rec["rocprofv3_avg_launch_us"]
            src["rocprofv3_file"] = rec.get("rocprofv3_source")
        return rec.get("hbm_bytes_per_launch"), src
    except Exception as e:
        return None, "{}: {}".format(type(e).__name__, e)


def odeint_calls(func, y0, args, reps=5):
    """What a USER call costs at the headline size (SURVEY 8(d) timing (i); reference functional/odeint.py:9-35 ->
    solver/base_adaptive_solver.py:24-31): `odeint(func, y0, t_span, solver=Dopri5, rtol=1e-5, atol=1e-7)` on t in [0, 1] with T = 2 and
    T = 11 output times, through the public entry point — initial-step selection (3 func evaluations, 3 norm passes), every attempt,
    dense output rows, the speculative pipeline's spare attempt, the result tensor.  Median of `reps` calls after one warm-up call, with
    the attempts and func evaluations each call made.  Outside the headline `value`."""
    from paddlexde_amd import Dopri5, odeint
    from paddlexde_amd.utils import _rms_norm

    out = {}
    for T in (2, 11):
        t = torch.linspace(0.0, 1.0, T)
        times, st = [], None
        for rep in range(reps + 1):
            st = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                sol = odeint(func, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm, "pipeline": args.pipeline, "stats_out": st})
            torch.cuda.synchronize()
            if rep:
                times.append(time.perf_counter() - t0)
        times.sort()
        out["odeint_ms_T{}".format(T)] = 1e3 * times[len(times) // 2]
        out["odeint_T{}".format(T)] = {"calls": reps, "min_ms": 1e3 * times[0], "max_ms": 1e3 * times[-1], "attempts": st["n_steps"],
                                       "accepted": st["n_accept"], "rejected": st["n_reject"], "nfe": st["nfe"],
                                       "states_per_s_per_attempt": y0.numel() * st["n_steps"] / times[len(times) // 2],
                                       "rows": list(sol.shape), "finite": bool(torch.isfinite(sol[-1]).all())}
    return out


def p2p_probe_child():
    """The probe itself (`bench.py --probe-p2p`, started by run_p2p_probe with RANK / WORLD_SIZE / LOCAL_RANK of its parent rank)."""
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("XDE_BENCH_REHEARSAL", "0") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("XDE_BENCH_TEST_PROBE_FAIL") == str(rank):  # test hook: this rank's probe dies the hard way, before anything else
        os.abort()
    wd = Watchdog(rank)
    wd.stage("probe: rendezvous", 45)
    dist.init_process_group("gloo")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from paddlexde_amd import Dopri5
    from paddlexde_amd.utils import PeerExchange, _rms_norm
    from paddlexde_amd.utils import exchange as X
    from paddlexde_amd.xde import BaseODE

    wd.stage("probe: mailboxes", 60)
    ex = PeerExchange(None, dev)
    ex.SPIN_LIMIT = 2_000_000  # ~0.1 s: a probe that cannot see its peers' stores must say so quickly
    wd.stage("probe: exchange self-test", 60)
    ok, why = X.selftest(ex, None, rounds=8)
    if ok:
        # the fused finalize -> exchange -> controller launch inside a short sharded solve: every rank must take the same steps
        wd.stage("probe: sharded solve", 60)
        A, y0 = make_problem(256, 16, rank, dev)
        AT = A.T.contiguous()
        t = torch.tensor([0.0, 0.5])
        s = Dopri5(xde=BaseODE(lambda t_, y: y @ AT, y0=y0, t_span=t), y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm, pipeline="lag",
                   process_group=True, norm_exchange=ex, record_trace=True)
        with torch.no_grad():
            sol = s.integrate(t)
        torch.cuda.synchronize()
        traces = [None] * world
        dist.all_gather_object(traces, [tuple(x) for x in s.trace])
        ok = bool(torch.isfinite(sol).all()) and len(s.trace) > 2 and all(tr == traces[0] for tr in traces) and ex.error() == 0
        why = None if ok else "the ranks of the probe's sharded solve did not stay in lock-step"
        ok = X.agree(ok)
    wd.stage("probe: close", 60)
    ex.close()
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        print("p2p probe failed: {}".format(why), file=sys.stderr)
    return 0 if ok else 1


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed attempted steps (200 x 0.34 ms = 68 ms at the headline size)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="rows PER GPU (default: 65536 at N=1 = config 2; 524288/N at N>1 = config 4)")
    ap.add_argument("--dim", type=int, default=None, help="default: 128 at N=1 (config 2), 64 at N>1 (config 4)")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "sync", "lag", "graph"],
                    help="auto = the library default (resolves to 'lag' at the headline size)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="state dtype (the headline metric is quoted on f32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rccl", "allreduce", "p2p"],
                    help="N>1: how the per-attempt norm sums travel — 'p2p': one-shot stores into IPC-mapped mailboxes over xGMI, "
                         "finalize + exchange + controller as ONE launch (utils.PeerExchange); 'rccl': ncclAllReduce issued directly "
                         "on the solver's stream (utils.RcclExchange); 'allreduce': torch.distributed all_reduce over the nccl (= RCCL) "
                         "backend; 'auto' (default): p2p if its first contact with this machine — made in child processes — "
                         "succeeds, else rccl, else allreduce; every candidate is self-tested and the whole group moves together "
                         "(utils/exchange.py).  The line says what was used and why (norm_exchange, norm_exchange_report)")
    ap.add_argument("--graph-func", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="c3: replay the augmented dynamics from a captured HIP graph (auto = the library default, which captures here)")
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c5", "rk4", "c4-shard", "c4-n1", "dense", "dde"],
                    help="c1: configs[0], the demo's 1000-point spiral with RK4 (plumbing); c2: BASELINE.json configs[1] (headline, default); c3: spiral neural-ODE odeint_adjoint backward, "
                         "batch 8192 (latency-bound, reports ms per fwd+bwd and per attempted step); c5: stiff Van der Pol "
                         "mu=1000 batch 4096 (step-rejection stress, reports accepted/rejected and us per step)")
    ap.add_argument("--solver", default="dopri5", choices=["dopri5", "dopri8", "bosh3", "fehlberg2", "adaptive_heun"],
                    help="embedded pair of the c2-family workloads (the headline metric is quoted on dopri5)")
    ap.add_argument("--no-ab", action="store_true", help="N>1: skip the short runs on the other norm-exchange transports (exchange_ab)")
    ap.add_argument("--no-odeint", action="store_true", help="N=1 headline: skip the whole-odeint() calls (odeint_ms_T2 / odeint_ms_T11)")
    ap.add_argument("--probe-p2p", action="store_true", help=argparse.SUPPRESS)  # internal: the child of a rank, see run_p2p_probe
    ap.add_argument("--rk4-presum", default="on", choices=["on", "off"], help="--workload rk4: A/B of the pre-summed final combine")
    ap.add_argument("--no-n1", action="store_true", help="N>1: skip rank 0's extra single-GPU run of the same global problem (n1_same_workload)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--no-tunable-op", action="store_true",
                    help="leave PyTorch's TunableOp off (by default the framework tunes the GEMMs inside the user's func during "
                         "warm-up: same fp32 arithmetic, a better hipBLASLt/rocBLAS kernel for the shape)")
    ap.add_argument("--event-period", type=int, default=11,
                    help="time every p-th launch of each kernel inside the timed region (a dispatch-stamped launch costs the step ~5 us: "
                         "every 5th one made the step 2.5 %% longer, every 11th ~1 %%; 11 is coprime with the 6 combines per step, so all "
                         "stages are sampled evenly)")
    return ap.parse_args()


def setup_ranks(args):
    """This process's place in the job — rank, device, watchdog — and, for a sharded run, its groups: the gloo control group, the
    peer-to-peer probe (in child processes, BEFORE this process opens its GPU), then the nccl (= RCCL) group.  Nothing is timed here."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE={}: launch one rank per GPU (or run `python bench.py --gpus N` with no "
                         "launcher, which starts the ranks itself)".format(args.gpus, world))
    # rehearsal on a one-GPU box (XDE_BENCH_REHEARSAL=1): all ranks share cuda:0 and gloo carries the collectives
    # (RCCL refuses several ranks on one device); the driver's real runs use one GPU per rank
    rehearsal = os.environ.get("XDE_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("--gpus {} but this box has {} GPU(s) (XDE_BENCH_REHEARSAL=1 rehearses on one GPU over gloo)".format(
            world, torch.cuda.device_count()))
    device = torch.device("cuda", local_rank)
    wd = Watchdog(rank)
    dist = None
    nccl_pg = None
    rccl_ranks = 0
    p2p_probe = None
    # XDE_BENCH_FORCE_DIST=1: take the sharded code path (finalize -> exchange -> controller) even with one
    # rank, to measure its per-step overhead on a one-GPU box
    force_dist = os.environ.get("XDE_BENCH_FORCE_DIST", "0") == "1"
    sharded = world > 1 or force_dist
    if sharded:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # The job's control plane is a gloo group (host memory, no GPU): flags, handles, the timing barrier.  Nothing below has
        # touched the GPU yet, so the first contact of the peer-to-peer transport with this machine can be made in CHILD processes.
        wd.stage("rendezvous (gloo control group)", 300)
        dist.init_process_group("gloo")
        if args.exchange in ("auto", "p2p"):
            wd.stage("peer-to-peer probe (child processes)", 260)
            p2p_probe = run_p2p_probe(dist, rank, world)
            if rank == 0 and not p2p_probe["ok"]:
                print("bench.py: the peer-to-peer probe failed ({}); not using that transport".format(p2p_probe["why"]), file=sys.stderr)
            if args.exchange == "p2p" and not p2p_probe["ok"]:
                raise SystemExit("--exchange p2p, but the peer-to-peer probe failed: {}".format(p2p_probe["why"]))
        wd.stage("device + nccl group", 300)
    # (from here on this process uses its GPU; the probe's children have come and gone — on a one-GPU rehearsal box that keeps the
    #  number of processes holding the card at the number of ranks)
    args.tunable_op = enable_tunable_op(args.tunable_op)
    torch.cuda.set_device(local_rank)
    if sharded and not rehearsal:
        # the nccl (= RCCL) group of the same ranks: the transport of the fall-backs and of the A/B runs.  An ERROR while forming or
        # checking it is agreed on and survived (the job then has the peer-to-peer transport, or the host-staged all-reduce, only);
        # a rank that hangs in here ends in the watchdog.
        from paddlexde_amd.utils import exchange as X0

        err = None
        try:
            try:
                nccl_pg = dist.new_group(backend="nccl", device_id=device)
            except TypeError:  # (an older signature)
                nccl_pg = dist.new_group(backend="nccl")
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe, group=nccl_pg)  # the RCCL communicator exists and works before anything is timed
            if float(probe.item()) != world:
                err = "the nccl group's first all-reduce returned {} for {} ranks".format(float(probe.item()), world)
        except Exception as e:  # noqa: BLE001
            err = "{}: {}".format(type(e).__name__, e)
        if X0.agree(err is None):
            rccl_ranks = dist.get_world_size(nccl_pg)
        else:
            print("bench.py[rank {}]: no usable nccl group ({}); continuing without one".format(rank, err or "failed on another rank"),
                  file=sys.stderr)
            nccl_pg = None
    return (world, rank, local_rank, rehearsal, device, wd, dist, nccl_pg, rccl_ranks, p2p_probe, force_dist, sharded)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.probe_p2p:
        raise SystemExit(p2p_probe_child())
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: this process becomes the launcher.  Nothing here has touched the
        # GPU (importing torch and counting devices do not initialise HIP), so starting children is safe.
        raise SystemExit(self_launch(args))
    # stdout carries ONE JSON line: whatever libraries print there meanwhile (RCCL's version banner, tuner notes) goes to
    # stderr; the original stdout is restored just before the line is printed
    sys.stdout.flush()
    global _REAL_STDOUT
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    args.tunable_op = not args.no_tunable_op  # (switched on below, after the transport probe: nothing may open the GPU before it)

    c4_label = None
    if args.workload in ("c4-shard", "c4-n1"):
        # config 4's sizes on ONE GPU: "c4-shard" = 65536 x 64, one rank's rows at N=8 (16 MiB operands, the whole
        # working set Infinity-Cache resident); "c4-n1" = the GLOBAL 524288 x 64 on one rank, the N=1 point of the strong-scaling curve
        args.batch, args.dim = (65536 if args.workload == "c4-shard" else 524288), 64
        c4_label = ("BASELINE.json configs[3], one rank's shard at N=8 on one GPU" if args.workload == "c4-shard"
                    else "BASELINE.json configs[3], the whole problem on one GPU (N=1 point of the strong-scaling curve)")
        args.workload = "c2"
    if args.workload != "c2":
        args.tunable_op = enable_tunable_op(args.tunable_op)  # (single-process workloads: no probe to wait for)
    if args.workload != "c2":  # the side workloads live in bench_side.py
        import types

        import bench_side

        ctx = types.SimpleNamespace(emit=emit, make_problem=make_problem, HBM_PEAK_GBS=HBM_PEAK_GBS)
        return {"rk4": bench_side.rk4_workload, "dense": bench_side.dense_workload, "dde": bench_side.dde_workload}.get(
            args.workload, bench_side.side_workload)(args, ctx)

    world, rank, local_rank, rehearsal, device, wd, dist, nccl_pg, rccl_ranks, p2p_probe, force_dist, sharded = setup_ranks(args)

    import paddlexde_amd
    from paddlexde_amd import _hip
    from paddlexde_amd.utils import _rms_norm
    from paddlexde_amd.xde import BaseODE

    Solver = {"dopri5": paddlexde_amd.Dopri5, "dopri8": paddlexde_amd.Dopri8, "bosh3": paddlexde_amd.Bosh3,
              "fehlberg2": paddlexde_amd.Fehlberg2, "adaptive_heun": paddlexde_amd.AdaptiveHeun}[args.solver]
    n_stage = len(Solver.tableau.alpha)

    # N = 1: config 2 (65536 x 128).  N > 1: config 4 (524288 x 64 GLOBAL, split over the ranks).
    GLOBAL_C4, DIM_C4 = 524288, 64
    explicit = args.batch is not None or args.dim is not None
    if world > 1 and not explicit:
        if GLOBAL_C4 % world:
            raise SystemExit("config 4's 524288 rows do not split evenly over {} ranks; pass --batch".format(world))
        B, D, scaling, cfg_name = GLOBAL_C4 // world, DIM_C4, "strong", "BASELINE.json configs[3]"
    else:
        B = 65536 if args.batch is None else args.batch
        D = (128 if world == 1 else DIM_C4) if args.dim is None else args.dim
        scaling, cfg_name = "weak", ("BASELINE.json configs[1]" if (B, D) == (65536, 128) else (c4_label or "custom size"))
    A, y0 = make_problem(B, D, rank, device)
    if args.dtype == "f64":
        A, y0 = A.double(), y0.double()
    AT = A.T.contiguous()
    func = lambda t, y: y @ AT  # noqa: E731  the user's func stays a framework call

    t_span = torch.tensor([0.0, 1.0e9])
    xde = BaseODE(func, y0=y0, t_span=t_span)
    be = _hip.get_backend()

    # -- how the per-attempt norm sums travel (N > 1): negotiated for the whole group, self-tested, with a stated fall-back ------
    from paddlexde_amd.utils import exchange as X

    exchange, exchange_kind, exchange_report = None, None, None
    if sharded:
        wd.stage("norm-exchange negotiation", 300)
        if args.exchange == "auto":
            prefer = (["p2p"] if (p2p_probe and p2p_probe["ok"]) else []) + ([] if (rehearsal or nccl_pg is None) else ["rccl"]) + ["allreduce"]
        else:
            prefer = [args.exchange]
        exchange, exchange_kind, exchange_report = X.negotiate(None, device, prefer=tuple(prefer), log=(
            (lambda m: print("bench.py: " + m, file=sys.stderr)) if rank == 0 else None))

    def group_for(kind):
        """The process group a solver gets: the all-reduce transport needs the nccl group itself (the control group is gloo: its
        all-reduce goes through the host); the other transports only use the group once per solve, for the global element count."""
        return (nccl_pg if (kind == "allreduce" and nccl_pg is not None) else True) if sharded else None

    def exchange_label(kind):
        if kind == "allreduce" and nccl_pg is None:
            return "all-reduce (torch.distributed, gloo through the host: rehearsal)"
        return X.NAMES[kind] + (" over the nccl (= RCCL) backend" if kind == "allreduce" else "")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_run(kind, ex, steps, warmup, events, pipeline=None, blocks=1):
        """Build a solver on transport `kind`, let it settle, warm up, and time EXACTLY `steps` attempted steps between barriers:
        every rank's own clock stops after its stream has drained, the job's time is the MAX over the ranks."""
        if os.environ.get("XDE_BENCH_TEST_HANG") == str(rank):  # test hook: this rank never joins the set-up's first collective
            time.sleep(10 ** 6)
        with torch.no_grad():
            func(None, y0)  # the framework picks (TunableOp: times) its GEMM here, on every rank, before anything is exchanged ...
        barrier()  # ... and the ranks start the solve together: the first norm exchange does not have to absorb seconds of skew
        pipeline = args.pipeline if pipeline is None else pipeline
        solver = Solver(xde=xde, y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm, pipeline=pipeline, process_group=group_for(kind),
                        norm_exchange=ex)
        solver.y0 = y0
        solver._before_integrate(np.asarray([0.0, 1.0e9], dtype=np.float32))
        # pipeline="auto" on a small state starts eagerly and captures its hipGraph after AUTO_GRAPH_AFTER attempts: let it settle
        # (setup, like the GEMM tuning) before the W warm-up steps, so that no capture falls into the timed region
        settle = 0
        while pipeline == "auto" and solver._auto_state in (None, "sync-then-graph") and settle < 64:
            solver.advance(4)
            settle += 4
        if solver._auto_state == "graph" or pipeline == "graph":
            solver.advance(solver.GRAPH_ATTEMPTS + 1)  # both graphs a budgeted advance replays (4 attempts, 1 attempt) now exist
            settle += solver.GRAPH_ATTEMPTS + 1
        # ... and whatever the pipeline, SETTLE_MS (100 ms) of untimed attempts of set-up (their number is reported as `settle_steps`): the
        # chip needs tens of milliseconds of this load before a 20-step block (6 ms) reads what the next one reads — first block of three
        # after 24 / 96 / 256 settle attempts + 5 warm-up steps at the headline size: 0.3169 / 0.3135 / 0.3131 ms per step against
        # 0.3108 / 0.3093 / 0.3122 for the blocks after it (profiles/r05_settle.txt); at the c4 shard 256 attempts (41 ms) still left the
        # first block 2 % behind, hence a time, not a count.  The timed region stays EXACTLY K attempts.
        # (a rehearsal — several ranks time-slicing ONE card — is not a measurement; XDE_BENCH_SETTLE_IN_REHEARSAL=1 settles there too: a soak)
        if SETTLE_MS > 0 and (not rehearsal or os.environ.get("XDE_BENCH_SETTLE_IN_REHEARSAL") == "1"):
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            solver.advance(32)
            torch.cuda.synchronize()
            per_step = (time.perf_counter() - t_s) / 32
            more = min(4096, max(0, int(SETTLE_MS * 1e-3 / max(per_step, 1e-6)) - 32))
            if dist is not None:  # (every rank takes the same number of attempts: the norm exchange is collective)
                box = torch.tensor([more], dtype=torch.int64)
                dist.all_reduce(box, op=dist.ReduceOp.MAX)
                more = int(box.item())
            solver.advance(more)
            settle += 32 + more
        if events:
            be.prof_enable(args.event_period)  # (already during the warm-up steps: a stream's FIRST dispatch-stamped launches are not free)
        solver.advance(warmup)
        barrier()
        if events:
            be.prof_enable(args.event_period)  # counters and samples start from zero at the timed region

        def block():
            """EXACTLY `steps` attempted steps between barriers; this rank's clock stops after its stream has drained; MAX over ranks."""
            t0 = time.perf_counter()
            cb = solver.advance(steps)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            barrier()
            if dist is not None:
                tmax = torch.tensor([el], dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                el = float(tmax.item())
            return cb, el

        c, elapsed = block()  # THE timed region: `value` / `ms_per_step` come from this block alone
        # ... and the same block twice more, back to back on the same solver: the line then says how far a 20-step block moves from
        # one run to the next on this box (`ms_per_step_blocks`), which one shot cannot
        repeats = [elapsed]
        for _ in range(blocks - 1):
            _c, el = block()  # (`c`, the counters the line reports, stay those of the timed region)
            repeats.append(el)
        prof = None
        if events:
            prof = be.prof_collect()  # (kernel averages over all the blocks)
            be.prof_enable(False)
        timed_run.blocks = repeats
        return solver, c, elapsed, prof, settle

    wd.stage("set-up + warm-up + timed region", 600)
    solver, c, elapsed, prof, settle = timed_run(exchange_kind, exchange, args.steps, args.warmup, not args.no_kernel_events, blocks=3)
    block_ms = [1e3 * el / args.steps for el in timed_run.blocks]
    # func's own share of a step, measured apart (the same GEMM, back to back, on torch's stream — where func runs)
    with torch.no_grad():
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            func(None, y0)
        ev0.record()
        for _ in range(30):
            func(None, y0)
        ev1.record()
        ev1.synchronize()
        func_ms_per_step = n_stage * ev0.elapsed_time(ev1) / 30.0

    N_local = B * D
    N_global = N_local * world
    ms_per_step = 1e3 * elapsed / args.steps
    value = N_global * args.steps / elapsed

    out = {
        "metric": "integrated states/sec (batch*dim/step_time) " + args.solver,
        "value": value,
        "unit": "states/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        # the same `steps`-step block three times back to back (the first is the one `value` is computed from): the spread of a short
        # timed region on THIS box, in the line itself
        "ms_per_step_blocks": block_ms,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": "{}: linear ODE dy/dt=Ay, {} adaptive (rtol 1e-5, atol 1e-7), global batch={} x dim={} = {} rows per GPU "
                        "x {} GPU(s), func = torch matmul{}".format(cfg_name if args.solver == "dopri5" else "custom solver", args.solver, B * world, D, B, world,
                                                                  " (framework GEMM picked by PyTorch TunableOp)" if args.tunable_op else ""),
            "global_batch": B * world,
            "rows_per_gpu": B,
            "dim": D,
            "pipeline": args.pipeline if args.pipeline != "auto" else "auto -> " + str(solver._auto_state),
            "parallelism": "batch-sharded x{} (error-norm sums only: {})".format(world, exchange_label(exchange_kind))
                           if world > 1 else "single GPU",
        },
        "norm_exchange": exchange_label(exchange_kind) if sharded else None,
        # ranks of the nccl (= RCCL) group this job formed and checked (it carries the sums when the transport is "rccl" / "allreduce", and is the
        # fall-back otherwise); 0 = no RCCL group (one GPU, or a gloo rehearsal)
        "rccl_ranks": rccl_ranks,
        "solver": {"n_steps": int(c.n_steps), "n_accept": int(c.n_accept), "n_reject": int(c.n_reject), "t": float(c.t1),
                   "dt": float(c.dt), "settle_steps": settle},
    }
    if prof is not None:
        kern = {}
        for name, rec in prof.items():
            if rec["launches"]:
                avg_us = 1e3 * rec["ms"] / rec["launches"]
                kern[name] = {
                    "launches": rec["launches"],
                    "avg_us": avg_us,
                    "algorithmic_GBps": (rec["bytes"] / rec["launches"]) / (avg_us * 1e-6) / 1e9 if rec["bytes"] else None,
                }
        out["kernels"] = kern
        comb = prof["combine"]
        achieved = comb["bytes"] / (comb["ms"] * 1e-3) / 1e9 if comb["ms"] > 0 else 0.0
        traffic, traffic_source = pmc_traffic(B, D, args.dtype) if world == 1 else (None, "PMC passes are taken on one GPU")
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "xde_combine_kernel<{}, RK, vec> (one launch per stage; with an FSAL pair the last one also emits the partial error sum)".format(
                "float" if args.dtype == "f32" else "double"),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "bytes_per_launch": comb["bytes"] / max(comb["launches"], 1),
            "avg_launch_us": 1e3 * comb["ms"] / max(comb["launches"], 1),
        }
        if isinstance(traffic_source, dict) and traffic_source.get("rocprofv3_avg_launch_us"):
            # beside the live figure, the RECORDED one: rocprofv3's mean duration of the same launches (committed kernel summary, same
            # stamp as the counter passes) — the events of a sampled launch read ~1 us longer than the profiler does
            us = traffic_source.pop("rocprofv3_avg_launch_us")
            out["roofline"]["rocprofv3"] = {"avg_launch_us": us, "frac": out["roofline"]["bytes_per_launch"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                           "file": traffic_source.pop("rocprofv3_file", None), "kernel_stamp": traffic_source.get("kernel_stamp")}
        en = prof["errnorm"]
        if en["ms"] > 0:
            a2 = en["bytes"] / (en["ms"] * 1e-3) / 1e9
            out["roofline_errnorm"] = {"bound": "hbm", "achieved": a2, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a2 / HBM_PEAK_GBS,
                                       "avg_launch_us": 1e3 * en["ms"] / en["launches"]}
        combines = comb["launches"] / max(prof["errnorm"]["launches"], 1)  # stage combines (+ the solution combine of a non-FSAL pair) per attempt
        per_step = {"combine": combines if prof["errnorm"]["launches"] else n_stage, "errnorm": 1, "control": 1,
                    "finalize": 1 if (sharded and exchange_kind != "p2p") else 0}
        solver_ms = sum(per_step[k] * prof[k]["ms"] / prof[k]["launches"] for k in per_step if prof[k]["launches"])
        out["solver_kernel_ms_per_step"] = solver_ms
        out["solver_only_states_per_s"] = N_local / (solver_ms * 1e-3) if solver_ms > 0 else None
        # what of a step is inside NO kernel (dependent launch boundaries, the host's enqueue when it is the slower side): the step minus
        # this library's kernels (events, in situ) minus func's GEMMs (timed apart, back to back: in situ they run on colder caches)
        out["func_ms_per_step"] = func_ms_per_step
        out["gap_ms_per_step"] = ms_per_step - (solver_ms + func_ms_per_step)
        out["gap_ms_per_step_blocks"] = [b - (solver_ms + func_ms_per_step) for b in block_ms]

    def emit_main_line(expired_stage=None):
        """The job's one JSON line.  Also what a watchdog does when an EXTRA measurement below outlives its limit: the headline is
        never lost to one — but the line then names the stage that hung (`watchdog_expired`) and the process exits non-zero."""
        if expired_stage is not None:
            out["watchdog_expired"] = expired_stage
        if rank == 0:
            emit(out)
        return 0 if expired_stage is None else 75

    if sharded:
        # who ran where: one row per rank, and whether each rank's device can address each other rank's (hipDeviceCanAccessPeer)
        mine = {"rank": rank, "device": device.index, "name": torch.cuda.get_device_name(device),
                "pci_bus_id": getattr(torch.cuda.get_device_properties(device), "pci_bus_id", None)}
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        out["devices"] = rows
        out["peer_access"] = [[1 if (a["device"] == b_["device"] or torch.cuda.can_device_access_peer(a["device"], b_["device"])) else 0
                               for b_ in rows] for a in rows] if not rehearsal else "rehearsal: every rank on device 0"
        out["norm_exchange_report"] = {"asked": args.exchange, "tried": exchange_report, "p2p_probe": p2p_probe}

    del solver
    if world > 1 and not args.no_ab:
        # The other transports on the same ranks, same shard, straight after the timed region (short runs, outside `value`): one visit to
        # a multi-GPU node answers which transport is fastest there.
        wd.stage("extra: the other norm-exchange transports", 420, on_expire=emit_main_line)
        ab = {exchange_kind: {"ms_per_step": ms_per_step, "steps": args.steps, "headline": True}}
        others = [k for k in ("p2p", "rccl", "allreduce") if k != exchange_kind and not (k == "p2p" and not (p2p_probe and p2p_probe["ok"]))
                  and not (k == "rccl" and (rehearsal or nccl_pg is None))]
        for kind in others:
            try:
                ex2, k2, _ = X.negotiate(None, device, prefer=(kind,))
            except Exception as e:  # (group-consistent: every rank lands here together)
                ab[kind] = {"error": "{}: {}".format(type(e).__name__, e)[:300]}
                continue
            steps2 = min(args.steps, 60)
            _s, _c, el2, _p, _ = timed_run(kind, ex2, steps2, min(args.warmup, 10), False)
            ab[kind] = {"ms_per_step": 1e3 * el2 / steps2, "steps": steps2}
            del _s
            if kind == "p2p" and args.pipeline in ("auto", "lag"):
                # the one transport whose sharded attempt can be CAPTURED: the same ranks replaying hipGraphs of whole attempts
                # (host floor 50 us per attempt against 129 us enqueued eagerly) — says whether the hosts of a full node keep up
                _s, _c, el3, _p, _ = timed_run(kind, ex2, steps2, min(args.warmup, 10), False, pipeline="graph")
                ab["p2p, pipeline=graph"] = {"ms_per_step": 1e3 * el3 / steps2, "steps": steps2}
                del _s
            if ex2 is not None:
                ex2.close()
        if exchange_kind == "p2p" and args.pipeline in ("auto", "lag"):
            steps2 = min(args.steps, 60)
            _s, _c, el3, _p, _ = timed_run("p2p", exchange, steps2, min(args.warmup, 10), False, pipeline="graph")
            ab["p2p, pipeline=graph"] = {"ms_per_step": 1e3 * el3 / steps2, "steps": steps2}
            del _s
        out["exchange_ab"] = ab
    if sharded:
        # BASELINE.json configs[3] names "RCCL error-norm all-reduce": that transport's step time at top level, whatever the headline ran on
        # ("rccl": ncclAllReduce issued on the solver's stream; else torch.distributed's all_reduce over the nccl group)
        ab_ = out.get("exchange_ab") or {exchange_kind: {"ms_per_step": ms_per_step, "steps": args.steps}}
        out["rccl_allreduce_ms_per_step"] = None  # (no RCCL group: one-GPU rehearsal over gloo)
        for kind, label in (("rccl", "ncclAllReduce on the solver's stream (RcclExchange)"),
                            ("allreduce", "torch.distributed.all_reduce over the nccl (= RCCL) group")):
            if nccl_pg is not None and isinstance(ab_.get(kind), dict) and "ms_per_step" in ab_[kind]:
                out["rccl_allreduce_ms_per_step"] = ab_[kind]["ms_per_step"]
                out["rccl_allreduce_transport"] = label + (" — the headline transport" if kind == exchange_kind
                                                           else " — a {}-step run after the timed region".format(ab_[kind]["steps"]))
                break

    if rank == 0 and world == 1 and not sharded and not args.no_cpu_baseline and args.dtype == "f32":
        wd.stage("cpu baseline", 600, on_expire=emit_main_line)
        out["cpu_baseline"], fused = cpu_baseline(B, D)
        if fused is not None:
            out["cpu_baseline_fused"] = fused
    if rank == 0 and world == 1 and not sharded and not args.no_odeint and (B, D) == (65536, 128) and args.solver == "dopri5":
        wd.stage("whole odeint() calls", 300, on_expire=emit_main_line)
        if os.environ.get("XDE_BENCH_TEST_HANG") == "extra":  # test hook: an EXTRA measurement that never returns
            time.sleep(10 ** 6)
        try:
            out.update(odeint_calls(func, y0, args))
        except Exception as e:  # never lose the headline to the extra measurement
            out["odeint_calls_error"] = "{}: {}".format(type(e).__name__, e)

    if exchange is not None:  # every rank leaves the per-step transport together, before rank 0 goes off on its own below
        wd.stage("closing the exchange", 120, on_expire=emit_main_line)
        dist.barrier()
        exchange.close()
        exchange = None
    if world > 1 and scaling == "strong" and not args.no_n1:
        wd.stage("extra: the same global problem on one GPU", 420, on_expire=emit_main_line)
        if rank == 0:
            # the same GLOBAL problem on this rank alone, so that a strong-scaling efficiency can be computed from this one line
            # (the driver's own `--gpus 1` line is config 2, a different amount of work); the other ranks wait at the barrier below
            torch.cuda.empty_cache()
            try:
                out["n1_same_workload"] = time_unsharded(GLOBAL_C4, DIM_C4, args.dtype, args.pipeline, device, min(args.steps, 60), min(args.warmup, 10))
                if rehearsal:
                    out["n1_same_workload"]["note"] = "rehearsal: measured while the other ranks idle on the SAME GPU"
            except Exception as e:  # never lose the N-rank line to the extra measurement
                out["n1_same_workload"] = {"error": "{}: {}".format(type(e).__name__, e)}
        dist.barrier()

    emit_main_line()
    if dist is not None:
        # (the line is out; a teardown that hangs — a rank that never reaches the barrier — still ends this process non-zero, named on stderr)
        wd.stage("teardown", 90, on_expire=lambda stage: 76)
        dist.barrier()
        dist.destroy_process_group()
    wd.done()


if __name__ == "__main__":
    main()
