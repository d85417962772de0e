"""N>1 path on CPU: world_size-2 `gloo` run of the batch-sharded Dopri5 (rows split across ranks, the ONLY
data-path collective is the all-reduce of the error-norm partial sums per attempted step).  Uses the numpy test
double for the kernels; checks that (i) both ranks take the identical step sequence, (ii) the sharded solution
equals the unsharded one on the same rows, (iii) the initial step uses the global norm too."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from . import problems as P


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem(B, D):
    A = P.skew_matrix(D)
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(0))
    # rows of very different magnitude: each shard alone would choose a different step size
    y0[B // 2 :] *= 25.0
    return A, y0


def _solve(y0, A, pg, norm_name="rms", pipeline="sync", exchange=None):
    from paddlexde_amd import Dopri5
    from paddlexde_amd.utils import _linf_norm, _rms_norm
    from paddlexde_amd.xde import BaseODE

    t = torch.linspace(0.0, 1.0, 4)
    xde = BaseODE(lambda t_, y: y @ A.T, y0=y0, t_span=t)
    s = Dopri5(xde=xde, y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm if norm_name == "rms" else _linf_norm, process_group=pg,
               record_trace=True, pipeline=pipeline, norm_exchange=exchange)
    return s.integrate(t), s


def _worker(rank, world, port, out_dir, norm_name, pipeline, device="cpu"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from paddlexde_amd import _hip

        if device == "cpu":
            from ._cpu_double import NumpyDoubleBackend

            _hip._set_backend_for_testing(NumpyDoubleBackend())
        torch.set_num_threads(1)
        B, D = 64, 16
        A, y0 = _problem(B, D)
        rows = slice(rank * B // world, (rank + 1) * B // world)
        sol, s = _solve(y0[rows].contiguous().to(device), A.to(device), True, norm_name, pipeline)
        np.savez(os.path.join(out_dir, "rank{}.npz".format(rank)), sol=sol.cpu().numpy(),
                 trace=np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("norm_name,pipeline,world", [("rms", "sync", 2), ("rms", "lag", 2), ("linf", "sync", 2), ("rms", "lag", 3),
                                                      ("rms", "sync", 8), ("rms", "lag", 8), ("linf", "lag", 8)])
def test_two_rank_sharded_equals_unsharded(tmp_path, cpu_double, norm_name, pipeline, world):
    """world 3 splits the 64 rows 21 / 21 / 22: uneven shards (the global element count is itself all-reduced).  world 8 (64 rows ->
    8 x 8) is north_star's geometry — "the 8 GPUs of one node" — executed before hardware does."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), norm_name, pipeline), nprocs=world, join=True)
    rs = [np.load(tmp_path / "rank{}.npz".format(r)) for r in range(world)]
    r0 = rs[0]
    # (i) lock-step: identical (t0, dt, ratio, accept) on every rank, bit for bit
    for r in rs[1:]:
        assert np.array_equal(r0["trace"], r["trace"])
    # (ii) equals the single-process run over the whole batch
    B, D = 64, 16
    A, y0 = _problem(B, D)
    full, s = _solve(y0, A, None, norm_name, pipeline)
    full = full.numpy()
    tr = np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace])
    assert tr.shape == r0["trace"].shape
    assert np.array_equal(tr[:, 3], r0["trace"][:, 3])
    assert np.allclose(tr[:, :3], r0["trace"][:, :3], rtol=1e-6)
    got = np.concatenate([r["sol"] for r in rs], axis=1)
    assert P.rel_err(got, full) <= 1e-6
    # (iii) the shards really are coupled: the small-magnitude shard alone would have taken different steps
    alone, s_alone = _solve(y0[: B // 2].contiguous(), A, None, norm_name, pipeline)
    assert len(s_alone.trace) != len(s.trace) or not np.allclose([x[1] for x in s_alone.trace], tr[:, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", ["sync", "lag"])
def test_two_ranks_on_one_gpu_hip_kernels(tmp_path, pipeline):
    """The same check with the HIP kernels: two processes share cuda:0, gloo carries the 32-double reduction
    (RCCL refuses two ranks on one device; the 8-GPU run uses the nccl backend through the same code path)."""
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), "rms", pipeline, "cuda:0"), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["trace"], r1["trace"])
    B, D = 64, 16
    A, y0 = _problem(B, D)
    full, s = _solve(y0.to("cuda:0"), A.to("cuda:0"), None, "rms", pipeline)
    tr = np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace])
    assert tr.shape == r0["trace"].shape and np.array_equal(tr[:, 3], r0["trace"][:, 3])
    got = np.concatenate([r0["sol"], r1["sol"]], axis=1)
    assert P.rel_err(got, full.cpu().numpy()) <= 1e-6


# ----------------------------------------------------------------------------------------------
# batch-sharded odeint_adjoint: forward and backward both under the global error norm
# ----------------------------------------------------------------------------------------------
class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.W1 = torch.nn.Parameter(0.4 * torch.randn(4, 8, generator=g, dtype=torch.float64))
        self.W2 = torch.nn.Parameter(0.4 * torch.randn(8, 4, generator=g, dtype=torch.float64))

    def forward(self, t, y):
        return torch.tanh(y @ self.W1) @ self.W2 - 0.1 * y


def _adjoint_run(y0, pg, norm="seminorm", t_grad=False, trace=None, device="cpu"):
    from paddlexde_amd import Dopri5, odeint_adjoint
    from paddlexde_amd.utils import _rms_norm

    m = _Net().to(device)
    y0 = y0.clone().to(device).requires_grad_(True)
    t = torch.linspace(0.0, 1.0, 4, dtype=torch.float64).to(device).requires_grad_(t_grad)
    opts = {"norm": _rms_norm, "dtype": torch.float64}
    if pg:
        opts["process_group"] = True
    # "seminorm": the adjoint's step control looks at (adj_t, y, adj_y) only; the parameter adjoints stay per-rank partial sums
    # until ONE all-reduce at the end.  default: the step control also looks at every parameter adjoint, so those are summed
    # over the group at every evaluation of the augmented dynamics.  Either way the all-reduced norm is the unsharded one.
    adj = {"dtype": torch.float64, **({"process_group": True} if pg else {})}
    if norm == "seminorm":
        adj["norm"] = "seminorm"
    if trace is not None:
        adj["_step_hook"] = lambda i, y0_, y1_, ks, c: trace.append((c.t0, c.dt_last, c.ratio, bool(c.accept)))
    sol = odeint_adjoint(m, y0, t, solver=Dopri5, rtol=1e-7, atol=1e-9, options=opts, adjoint_options=adj)
    w = torch.linspace(-1.0, 1.0, sol.numel(), dtype=torch.float64).reshape(sol.shape).to(device)
    return m, y0, sol, w, t


def _adjoint_problem():
    B = 12
    y_all = torch.randn(B, 4, generator=torch.Generator().manual_seed(11), dtype=torch.float64)
    y_all[B // 2 :] *= 4.0
    return B, y_all


def _adjoint_worker(rank, world, port, out_dir, norm, t_grad):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from paddlexde_amd import _hip

        from ._cpu_double import NumpyDoubleBackend

        _hip._set_backend_for_testing(NumpyDoubleBackend())
        torch.set_num_threads(1)
        B, y_all = _adjoint_problem()
        rows = slice(rank * B // world, (rank + 1) * B // world)
        trace = []
        m, y0, sol, _, t = _adjoint_run(y_all[rows].contiguous(), True, norm, t_grad, trace)
        w_all = torch.linspace(-1.0, 1.0, 4 * B * 4, dtype=torch.float64).reshape(4, B, 4)
        (sol * w_all[:, rows]).sum().backward()
        np.savez(os.path.join(out_dir, "adj{}.npz".format(rank)), sol=sol.detach().numpy(), gy=y0.grad.numpy(),
                 gW1=m.W1.grad.numpy(), gW2=m.W2.grad.numpy(), gt=t.grad.numpy() if t_grad else np.zeros(0),
                 trace=np.asarray([[a, b, c, float(d)] for a, b, c, d in trace]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("norm,t_grad,world", [("seminorm", False, 2), ("default", False, 2), ("default", True, 2), ("seminorm", True, 2),
                                              ("default", False, 3), ("seminorm", False, 8), ("default", False, 8), ("default", True, 8)])
def test_sharded_adjoint_equals_unsharded(tmp_path, cpu_double, norm, t_grad, world):
    """Forward AND adjoint backward batch-sharded over the ranks, under the default adjoint norm (odeint_adjoint.py:284-287: it
    looks at every parameter adjoint, which is a sum over ALL rows) and under "seminorm": every rank takes the unsharded run's
    backward step sequence, solution rows and d/dy0 rows equal the unsharded run's, and EVERY rank ends with the parameter (and
    time) gradients of the global loss."""
    port = _free_port()
    mp.spawn(_adjoint_worker, args=(world, port, str(tmp_path), norm, t_grad), nprocs=world, join=True)
    rs = [np.load(tmp_path / "adj{}.npz".format(r)) for r in range(world)]
    B, y_all = _adjoint_problem()
    trace = []
    m, y0, sol, w, t = _adjoint_run(y_all, False, norm, t_grad, trace)
    (sol * w).sum().backward()
    tr = np.asarray([[a, b, c, float(d)] for a, b, c, d in trace])
    for r in rs:
        # lock-step with each other (bit for bit) and with the unsharded run (same decisions, same steps to rounding)
        assert np.array_equal(r["trace"], rs[0]["trace"])
        assert r["trace"].shape == tr.shape and np.array_equal(r["trace"][:, 3], tr[:, 3])
        assert np.allclose(r["trace"][:, :2], tr[:, :2], rtol=1e-6, atol=1e-12)  # t0, dt (row sums are ordered differently)
        # (the first attempts' error estimates are rounding noise, ratio ~1e-7: absolute floor there)
        assert np.allclose(r["trace"][:, 2], tr[:, 2], rtol=1e-6, atol=1e-9)
        assert P.rel_err(r["gW1"], m.W1.grad.numpy()) <= 1e-8
        assert P.rel_err(r["gW2"], m.W2.grad.numpy()) <= 1e-8
        if t_grad:
            assert P.rel_err(r["gt"], t.grad.numpy()) <= 1e-8
    assert P.rel_err(np.concatenate([r["sol"] for r in rs], axis=1), sol.detach().numpy()) <= 1e-9
    assert P.rel_err(np.concatenate([r["gy"] for r in rs], axis=0), y0.grad.numpy()) <= 1e-8


def test_sharded_solver_refuses_a_user_norm(cpu_double):
    from paddlexde_amd import Dopri5
    from paddlexde_amd.xde import BaseODE

    y0 = torch.ones(4, 2)
    with pytest.raises(NotImplementedError, match="cannot be all-reduced"):
        Dopri5(xde=BaseODE(lambda t_, y: -y, y0=y0, t_span=torch.tensor([0.0, 1.0])), y0=y0, rtol=1e-5, atol=1e-7,
               norm=lambda x: x.abs().max(), process_group=True)


# ----------------------------------------------------------------------------------------------
# the production transport: torch.distributed "nccl" (= RCCL) carrying the norm all-reduce
# ----------------------------------------------------------------------------------------------
def _nccl_graph_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from paddlexde_amd import Dopri5
        from paddlexde_amd.utils import RcclExchange, _rms_norm
        from paddlexde_amd.xde import BaseODE

        A = P.skew_matrix(16).to("cuda:0")
        y0 = torch.randn(64, 16, generator=torch.Generator().manual_seed(0)).to("cuda:0")
        t = torch.linspace(0.0, 30.0, 7)

        def solve(pipeline, pg, ex):
            s = Dopri5(xde=BaseODE(lambda t_, y: y @ A.T, y0=y0, t_span=t), y0=y0, rtol=1e-7, atol=1e-9, norm=_rms_norm, pipeline=pipeline,
                       process_group=pg, norm_exchange=ex, record_trace=True)
            return s.integrate(t), list(s.trace)

        ref, tr0 = solve("sync", None, None)
        ex = RcclExchange()
        refused = False
        try:
            solve("graph", True, ex)
        except NotImplementedError:
            refused = True  # off by default: a captured RCCL collective has not been exercised across GPUs
        ex.capturable = True
        got, tr1 = solve("graph", True, ex)
        torch.cuda.synchronize()
        ex.close()
        np.savez(os.path.join(out_dir, "nccl_graph.npz"), equal=bool(torch.equal(got, ref)), same_trace=tr1 == tr0, attempts=len(tr1), refused=refused)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_direct_rccl_exchange_can_be_captured_with_one_rank(tmp_path):
    """Opt-in (`exchange.capturable = True` / XDE_RCCL_CAPTURE=1): pipeline="graph" records the in-stream ncclAllReduce together with
    the attempt's kernels and replays it — with ONE rank (all a one-GPU box can run) ~200 attempts replay bit-identically to the
    unsharded solve.  By default the combination is refused."""
    mp.spawn(_nccl_graph_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl_graph.npz")
    assert bool(r["refused"]) and bool(r["equal"]) and bool(r["same_trace"]) and int(r["attempts"]) > 100


def _nccl_worker(rank, world, port, out_dir, pipeline, direct=False, norm_name="rms"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        B, D = 4096, 64
        A, y0 = _problem(B, D)
        ex = None
        if direct:  # ncclAllReduce issued on the solver's own stream instead of torch.distributed.all_reduce
            from paddlexde_amd.utils import RcclExchange

            ex = RcclExchange()
            probe = torch.arange(32, dtype=torch.float64, device="cuda:0")
            ex.exchange(probe, 1)  # linf: max of the first 16, sum of the last 16 (one rank: the identity)
            assert torch.equal(probe.cpu(), torch.arange(32, dtype=torch.float64)) and ex.async_error() is None
        sol, s = _solve(y0.to("cuda:0"), A.to("cuda:0"), True, norm_name, pipeline, exchange=ex)
        if ex is not None:
            ex.close()
        np.savez(os.path.join(out_dir, "nccl.npz"), sol=sol.cpu().numpy(), trace=np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", ["sync", "lag"])
def test_rccl_backend_world_size_one(tmp_path, pipeline):
    """`process_group=True` over the nccl backend (RCCL) with one rank: xde_error_norm_partial -> xde_norm_finalize ->
    all_reduce on the device (no host staging) -> xde_rk_control(sums) — the exact per-attempt sequence of the 8-GPU run.  The
    result equals the unsharded run bit for bit (same fixed-order reduction, the all-reduce of one rank is the identity)."""
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path), pipeline), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl.npz")
    B, D = 4096, 64
    A, y0 = _problem(B, D)
    full, s = _solve(y0.to("cuda:0"), A.to("cuda:0"), None, "rms", pipeline)
    tr = np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace])
    assert np.array_equal(tr, r["trace"])
    assert np.array_equal(full.cpu().numpy(), r["sol"])


@pytest.mark.gpu
@pytest.mark.parametrize("norm_name,pipeline", [("rms", "sync"), ("rms", "lag"), ("linf", "lag")])
def test_direct_rccl_exchange_world_size_one(tmp_path, norm_name, pipeline):
    """`norm_exchange=RcclExchange()`: its own communicator (unique id by broadcast_object_list, ncclCommInitRank) and
    ncclAllReduce on the solver's stream, one rank — bit-identical to the unsharded run, like the torch.distributed transport."""
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path), pipeline, True, norm_name), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl.npz")
    B, D = 4096, 64
    A, y0 = _problem(B, D)
    full, s = _solve(y0.to("cuda:0"), A.to("cuda:0"), None, norm_name, pipeline)
    tr = np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace])
    assert np.array_equal(tr, r["trace"])
    assert np.array_equal(full.cpu().numpy(), r["sol"])


def test_direct_rccl_exchange_is_not_offered_to_the_graph_pipeline(cpu_double):
    from paddlexde_amd import Dopri5
    from paddlexde_amd.utils import PeerExchange, RcclExchange
    from paddlexde_amd.xde import BaseODE

    assert PeerExchange.capturable and not RcclExchange.capturable
    y0 = torch.ones(4, 2)
    stand_in = type("X", (), {"capturable": False})()
    with pytest.raises(NotImplementedError, match="peer-to-peer norm exchange"):
        Dopri5(xde=BaseODE(lambda t_, y: -y, y0=y0, t_span=torch.tensor([0.0, 1.0])), y0=y0, rtol=1e-5, atol=1e-7,
               norm=__import__("paddlexde_amd").utils._rms_norm, process_group=True, norm_exchange=stand_in, pipeline="graph")


# ----------------------------------------------------------------------------------------------
# one-shot peer-to-peer norm exchange (utils.PeerExchange / xde_p2p_*): rehearsal with two processes on ONE GPU
# ----------------------------------------------------------------------------------------------
def _p2p_worker(rank, world, port, out_dir, norm_name, pipeline):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # set-up transport only (handles, element counts)
    try:
        from paddlexde_amd.utils import PeerExchange

        B, D = 256, 32
        A, y0 = _problem(B, D)
        rows = slice(rank * B // world, (rank + 1) * B // world)
        ex = PeerExchange()
        try:
            # the exchange alone: 100 rounds of known vectors, sum and max, checked exactly
            sums = torch.zeros(32, dtype=torch.float64, device="cuda:0")
            for i in range(100):
                sums.copy_(torch.arange(32, dtype=torch.float64) * (rank + 1) + i)
                ex.exchange(sums, 0)
                want = sum(torch.arange(32, dtype=torch.float64) * (r + 1) + i for r in range(world))
                assert torch.equal(sums.cpu(), want), (i, sums.cpu(), want)
                sums.copy_(torch.arange(32, dtype=torch.float64) * (rank + 1) - i)
                ex.exchange(sums, 1)
                parts = [torch.arange(32, dtype=torch.float64) * (r + 1) - i for r in range(world)]
                want = torch.cat([torch.stack(parts).max(0).values[:16], torch.stack(parts).sum(0)[16:]])
                assert torch.equal(sums.cpu(), want), (i, sums.cpu(), want)
            assert ex.error() == 0
            assert ex.fused_control  # the default: finalize + exchange + controller of an attempt as ONE launch (xde_p2p_rk_control)
            sol, s = _solve(y0[rows].contiguous().to("cuda:0"), A.to("cuda:0"), True, norm_name, pipeline, exchange=ex)
            # the same solve with the three launches the fused one replaces (xde_norm_finalize, xde_p2p_exchange, xde_rk_control)
            ex.fused_control = False
            sol3, s3 = _solve(y0[rows].contiguous().to("cuda:0"), A.to("cuda:0"), True, norm_name, pipeline, exchange=ex)
            ex.fused_control = True
            # the all-reduce twin (gloo): a captured step cannot hold it, so the graph case is compared with the lag pipeline
            sol2, s2 = _solve(y0[rows].contiguous().to("cuda:0"), A.to("cuda:0"), True, norm_name, "lag" if pipeline == "graph" else pipeline)
        finally:
            ex.close()
        tr = lambda so: np.asarray([[a, b, c, float(d)] for a, b, c, d in so.trace])  # noqa: E731
        np.savez(os.path.join(out_dir, "p2p{}.npz".format(rank)), sol=sol.cpu().numpy(), trace=tr(s), sol_ar=sol2.cpu().numpy(),
                 trace_ar=tr(s2), sol_3=sol3.cpu().numpy(), trace_3=tr(s3))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("norm_name,pipeline,world", [("rms", "sync", 2), ("rms", "lag", 2), ("linf", "lag", 2), ("rms", "graph", 2),
                                                      ("rms", "graph", 3), ("rms", "lag", 3), ("rms", "lag", 4), ("rms", "graph", 4)])
# (every pipeline that goes through xde_p2p_control_kernel's publish wave meets MORE THAN ONE peer — lag and graph at three
#  and at four ranks; a rank costs ~7 s of start-up.  Four ranks + the test process = 5 processes on the card: the pool allows 6)
def test_peer_exchange_two_ranks_on_one_gpu(tmp_path, norm_name, pipeline, world):
    """The IPC-mapped mailboxes carry the per-attempt norm sums instead of an all-reduce: both ranks stay in lock-step and the
    run is BIT-identical to the all-reduce run (two summands: the rank-ordered sum is the all-reduce's sum; with four ranks
    sharing the GPU: identical across ranks, equal to the all-reduce run to rounding).  The exchange
    counter lives in device memory, so the sharded step can also be captured and replayed (pipeline="graph")."""
    mp.spawn(_p2p_worker, args=(world, _free_port(), str(tmp_path), norm_name, pipeline), nprocs=world, join=True)
    rs = [np.load(tmp_path / "p2p{}.npz".format(r)) for r in range(world)]
    for r in rs[1:]:
        assert np.array_equal(rs[0]["trace"], r["trace"])  # lock-step, bit for bit, whatever the world size
    for r in rs:
        # ONE launch (xde_p2p_rk_control) == the three launches it replaces, bit for bit, whatever the world size
        assert np.array_equal(r["trace"], r["trace_3"]) and np.array_equal(r["sol"], r["sol_3"])
        if world == 2:
            assert np.array_equal(r["trace"], r["trace_ar"]) and np.array_equal(r["sol"], r["sol_ar"])
        else:
            # three / four summands: the mailbox sums them in RANK order, the all-reduce in its own order — equal to fp64 rounding
            assert r["trace"].shape == r["trace_ar"].shape and np.array_equal(r["trace"][:, 3], r["trace_ar"][:, 3])
            assert np.allclose(r["trace"][:, :3], r["trace_ar"][:, :3], rtol=1e-6, atol=1e-12)
            assert P.rel_err(r["sol"], r["sol_ar"]) <= 1e-6
    assert len(rs[0]["trace"]) > 5


def _p2p_timeout_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from paddlexde_amd import _hip
        from paddlexde_amd.utils import PeerExchange

        ex = PeerExchange()
        ex.SPIN_LIMIT = 20_000  # ~ a millisecond of polling instead of a second
        outcome = "skipped"
        try:
            if rank == 0:  # rank 1 never posts: rank 0's exchange must give up, not hang the GPU
                from paddlexde_amd import Dopri5
                from paddlexde_amd.utils import _rms_norm
                from paddlexde_amd.xde import BaseODE

                sums = torch.arange(32, dtype=torch.float64, device="cuda:0")
                ex.exchange(sums, 0)
                torch.cuda.synchronize()
                assert ex.error() == 1  # the first exchange
                assert sums[:16].abs().sum().item() == 0.0 and (sums[16:] == 1.0).all()  # poisoned: "non-finite" count > 0
                # ... which stops the solve; the host then reports the exchange instead of a bogus non-finite state
                y0 = torch.ones(4, 2, device="cuda:0")
                s_ = Dopri5(xde=BaseODE(lambda t_, y: -y, y0=y0, t_span=torch.tensor([0.0, 1.0])), y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm,
                            process_group=True, norm_exchange=ex)
                c = _hip.XdeCtrl()
                c.status = _hip.STATUS_NONFINITE
                try:
                    s_._raise_status(c)
                    outcome = "no error"
                except _hip.XdeError as e:
                    outcome = str(e)
        finally:
            ex.close()
        with open(os.path.join(out_dir, "timeout{}.txt".format(rank)), "w") as fh:
            fh.write(outcome)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_peer_exchange_gives_up_when_a_rank_never_arrives(tmp_path):
    """The wait inside xde_p2p_exchange is bounded: a peer that never posts makes the exchange poison its sums (the controller
    stops the solve at once) and raise the mailbox's error flag; the host reports the exchange, not a bogus non-finite state."""
    mp.spawn(_p2p_timeout_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msg = open(tmp_path / "timeout0.txt").read()
    assert "peer-to-peer norm exchange" in msg and "timed out" in msg, msg
    assert open(tmp_path / "timeout1.txt").read() == "skipped"


def _p2p_abort_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time

        from paddlexde_amd.utils import PeerExchange

        ex = PeerExchange()
        rec = {}
        try:
            sums = torch.arange(32, dtype=torch.float64, device="cuda:0") * (rank + 1)
            ex.exchange(sums, 0)  # exchange 1: everybody arrives
            torch.cuda.synchronize()
            rec["first_ok"] = bool(ex.error() == 0 and float(sums[1]) == sum(r + 1 for r in range(world)))
            dist.barrier()
            if rank == 0:
                ex.SPIN_LIMIT = 20_000  # gives up after about a millisecond: ranks 1 and 2 are not there yet
                t0 = time.perf_counter()
                ex.exchange(sums, 0)
                torch.cuda.synchronize()
                rec["seconds"] = time.perf_counter() - t0
                rec["info"] = list(ex.error_info())
                dist.barrier()  # (a) rank 0 has failed and told its peers
                dist.barrier()  # (b)
            elif rank == 1:
                dist.barrier()  # (a)
                # this rank had completed exchange 1 and arrives at exchange 2 AFTER rank 0 gave up: with a spin limit worth many
                # seconds it must still stop at once, because rank 0 marked this mailbox
                ex.SPIN_LIMIT = 2_000_000_000
                t0 = time.perf_counter()
                ex.exchange(sums, 0)
                torch.cuda.synchronize()
                rec["seconds"] = time.perf_counter() - t0
                rec["info"] = list(ex.error_info())
                rec["stop_vector"] = bool(sums[:16].abs().sum().item() == 0.0 and (sums[16:] == 1.0).all())
                # ... and so does every later exchange (the failure is sticky, nothing is posted any more)
                ex.exchange(sums, 0)
                torch.cuda.synchronize()
                rec["info_after"] = list(ex.error_info())
                dist.barrier()  # (b)
            else:
                dist.barrier()  # (a)
                dist.barrier()  # (b) this rank never posts exchange 2 at all
                rec["info"] = list(ex.error_info())  # it was told as well: its next exchange would stop at once
        finally:
            ex.close()
        import json

        with open(os.path.join(out_dir, "abort{}.json".format(rank)), "w") as fh:
            json.dump(rec, fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_peer_exchange_failure_stops_every_rank_of_the_group(tmp_path):
    """'on timeout only the timing-out rank stops'.  Now the rank whose wait runs out marks every peer's mailbox:
    a peer that arrives later — or is still waiting with a much longer limit — stops at once with the stop vector, names the rank
    that told it, and never posts again; no rank of a failed group runs on alone.  Three ranks on one GPU."""
    import json

    mp.spawn(_p2p_abort_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    r0, r1, r2 = [json.load(open(tmp_path / "abort{}.json".format(r))) for r in range(3)]
    assert r0["first_ok"] and r1["first_ok"] and r2["first_ok"]
    assert r0["info"] == [2, None]  # its own wait ran out at exchange 2
    assert r1["info"] == [2, 0] and r1["stop_vector"] and r1["seconds"] < 2.0, r1  # told by rank 0, at once (its own limit: minutes)
    assert r1["info_after"] == [2, 0]
    assert r2["info"] == [0, 0]  # never ran exchange 2, so no failure of its own yet — but it has been told by rank 0 as well


def _nccl_adjoint_worker(rank, world, port, out_dir, norm):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        _, y_all = _adjoint_problem()
        m, y0, sol, w, t = _adjoint_run(y_all, True, norm, True, device="cuda:0")
        (sol * w).sum().backward()
        np.savez(os.path.join(out_dir, "nccl_adj.npz"), sol=sol.detach().cpu().numpy(), gy=y0.grad.cpu().numpy(), gW1=m.W1.grad.cpu().numpy(),
                 gW2=m.W2.grad.cpu().numpy(), gt=t.grad.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("norm", ["default", "seminorm"])
def test_rccl_backend_world_size_one_adjoint(tmp_path, norm):
    """The sharded odeint_adjoint over the nccl backend (RCCL) with one rank, HIP kernels: the per-evaluation sums of the
    parameter / time adjoints (default norm), the final all-reduce ("seminorm") and the norm exchange all run on the device;
    with one rank every sum is the identity, so solution and all gradients equal the unsharded run's."""
    mp.spawn(_nccl_adjoint_worker, args=(1, _free_port(), str(tmp_path), norm), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl_adj.npz")
    _, y_all = _adjoint_problem()
    m, y0, sol, w, t = _adjoint_run(y_all, False, norm, True, device="cuda:0")
    (sol * w).sum().backward()
    assert P.rel_err(r["sol"], sol.detach().cpu().numpy()) <= 1e-9
    for key, ref in (("gy", y0.grad), ("gW1", m.W1.grad), ("gW2", m.W2.grad), ("gt", t.grad)):
        assert P.rel_err(r[key], ref.cpu().numpy()) <= 1e-8, key


# ----------------------------------------------------------------------------------------------
# choosing the transport of the norm sums: group-safe set-up (utils/exchange.py)
# ----------------------------------------------------------------------------------------------
def _negotiate_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json

        from paddlexde_amd import _hip
        from paddlexde_amd.utils import exchange as X

        rec = {}
        # (1) a step that fails on ONE rank only is an exception on every rank, raised at the same point — nobody is left
        #     waiting in the collective that would have followed
        try:
            X.raise_together(RuntimeError("no such library") if rank == 1 else None, "set-up", None)
            rec["one_rank_fails"] = "no error"
        except _hip.XdeError as e:
            rec["one_rank_fails"] = str(e)
        X.raise_together(None, "set-up", None)  # all fine: returns
        rec["agree"] = [X.agree(True), X.agree(rank == 0)]
        # (2) on a box without a GPU neither the mailboxes nor an RCCL communicator can be built: every rank falls back, together,
        #     to the group's own all-reduce, and says why
        ex, kind, report = X.negotiate(None, None, prefer=("p2p", "rccl", "allreduce"))
        rec["kind"], rec["exchange_is_none"], rec["report"] = kind, ex is None, report
        with open(os.path.join(out_dir, "neg{}.json".format(rank)), "w") as fh:
            json.dump(rec, fh)
    finally:
        dist.destroy_process_group()


def test_exchange_negotiation_is_group_safe_without_a_gpu(tmp_path):
    """a rank failing before a constructor's collective must not leave its peers waiting in it."""
    import json

    if torch.cuda.is_available():
        pytest.skip("the CPU statement of the fall-back; the GPU suite negotiates for real")
    mp.spawn(_negotiate_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [json.load(open(tmp_path / "neg{}.json".format(r))) for r in range(2)]
    assert "failed on another rank" in r0["one_rank_fails"] and "no such library" in r1["one_rank_fails"]
    assert r0["agree"] == r1["agree"] == [True, False]
    for r in (r0, r1):
        assert r["kind"] == "allreduce" and r["exchange_is_none"]
        assert [x["transport"] for x in r["report"]] == ["p2p", "rccl", "allreduce"]
        assert [x["adopted"] for x in r["report"]] == [False, False, True] and all(x.get("why") for x in r["report"][:2])


def _negotiate_gpu_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json

        from paddlexde_amd.utils import exchange as X

        rec = {}
        dev = torch.device("cuda", 0)
        # (1) the preferred transport works here (mailboxes are IPC-mapped between processes on one GPU): adopted after its self-test
        ex, kind, report = X.negotiate(None, dev, prefer=("p2p", "rccl", "allreduce"))
        rec["first"] = [kind, report]
        ok, why = X.selftest(ex, None, rounds=5)
        rec["selftest_again"] = [ok, why]
        ex.close()
        # (2) RCCL refuses two ranks on ONE device: the communicator cannot be built, on every rank; the group falls back TOGETHER
        ex2, kind2, report2 = X.negotiate(None, dev, prefer=("rccl", "allreduce"))
        rec["second"] = [kind2, ex2 is None, report2]
        # (3) a transport whose self-test fails on ONE rank only is dropped by all of them
        from paddlexde_amd.utils import PeerExchange

        real = PeerExchange.exchange

        def corrupt(self, sums, norm_kind):
            real(self, sums, norm_kind)
            if rank == 1:
                sums[3] += 1.0  # this rank "receives" a wrong sum

        PeerExchange.exchange = corrupt
        try:
            ex3, kind3, report3 = X.negotiate(None, dev, prefer=("p2p", "allreduce"))
        finally:
            PeerExchange.exchange = real
        rec["third"] = [kind3, ex3 is None, report3]
        with open(os.path.join(out_dir, "negg{}.json".format(rank)), "w") as fh:
            json.dump(rec, fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_exchange_negotiation_on_the_gpu_adopts_falls_back_and_never_splits(tmp_path):
    """utils/exchange.py with the real transports, two ranks on one GPU: the peer-to-peer transport is adopted after its self-test;
    RCCL — which cannot build a communicator for two ranks on one device — is dropped by both ranks with the library's own message
    and the group lands on the all-reduce; a self-test that fails on ONE rank drops the transport on BOTH.  Run under a deadline: a
    hang inside a communicator initialisation must fail this test, not the suite."""
    import json
    import time

    ctx = mp.spawn(_negotiate_gpu_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=False)
    deadline = time.time() + 240
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for p in ctx.processes:  # (exactly the two processes this test started)
                if p.is_alive():
                    p.kill()
            pytest.fail("the negotiation did not finish in 240 s")
    r0, r1 = [json.load(open(tmp_path / "negg{}.json".format(r))) for r in range(2)]
    for r in (r0, r1):
        assert r["first"][0] == "p2p" and r["first"][1] == [{"transport": "p2p", "adopted": True}] and r["selftest_again"][0] is True
        kind2, none2, rep2 = r["second"]
        assert kind2 == "allreduce" and none2 and [x["transport"] for x in rep2] == ["rccl", "allreduce"] and rep2[0]["adopted"] is False
        kind3, none3, rep3 = r["third"]
        assert kind3 == "allreduce" and none3 and rep3[0] == {"transport": "p2p", "adopted": False, "why": rep3[0]["why"]}
    assert "RcclExchange" in r0["second"][2][0]["why"] or "ncclCommInitRank" in r0["second"][2][0]["why"] or "another rank" in r0["second"][2][0]["why"]
    assert "sum round" in r1["third"][2][0]["why"] and "another rank" in r0["third"][2][0]["why"]
