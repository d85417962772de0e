"""Child process of tests/test_gpu_world8.py — north_star's geometry ("the batch dimension shards across the 8 GPUs of one node") on ONE
card, in ONE process: eight ranks as eight host threads, each with its own stream, its own mailbox (xde_p2p_alloc) and the pointers
of all eight (what xde_p2p_import hands a rank on a real node), its 65536 x 64 shard of BASELINE configs[3]'s 524288 x 64 state, the
real kernels and the fused finalize -> exchange -> controller launch (xde_p2p_rk_control) with SEVEN peers per rank.

Why threads and not processes: the GPU pool ends a job that has more than 6 processes on one card, so eight one-card rank processes
(bench.py's rehearsal mode) cannot run there; the device side — 8 mailbox rows, 7 peers, the rank-ordered sum of 8 vectors, lock-step
of 8 controllers — does not care which host process enqueued a rank's launches.  What this does NOT cover is the host-side set-up of a
real group (IPC handles, the gloo / nccl control plane): tests/test_sharded_gloo.py runs that at world 8 on the CPU double, and
with 2-4 real processes on the card.

A rank's controller launch spins (bounded) until its seven peers have posted: every rank's stream therefore needs a hardware queue
of its own — a launch queued BEHIND a spinning one on the same queue would never start.  GPU_MAX_HW_QUEUES (4 by default) is raised
before the runtime starts, and a short co-residency probe runs first on the very streams the solve will use: if two ranks share a queue
all the same, the probe's bounded wait runs out and the run ends as a SKIP (exit code 77: a fact about the box, not about the library);
nothing hangs.

    python tests/_world8_child.py exchange     # known vectors through xde_p2p_exchange, 8 ranks, sum and max, 50 rounds
    python tests/_world8_child.py solve sync   # config 4 at full size, 8 shards, against tests/golden/config4_full.npz
"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from paddlexde_amd import Dopri5, _hip  # noqa: E402
from paddlexde_amd.utils import PeerExchange, _rms_norm  # noqa: E402
from paddlexde_amd.xde import BaseODE  # noqa: E402
from tests import problems as P  # noqa: E402

WORLD = 8
DEV = "cuda:0"


class InProcessExchange(PeerExchange):
    """PeerExchange's launch side over mailboxes that all live in this process: a peer's mailbox is addressed by its own pointer (on a
    node: by the mapping xde_p2p_import returned).  No process group, no IPC handle; the owner of `mailboxes` frees them."""

    SPIN_LIMIT = 100_000_000  # ~5 s: ranks that share a hardware queue must end in a diagnosis, not in 20 s x every exchange

    def __init__(self, rank, world, mailboxes):  # noqa: super().__init__ is the multi-process set-up this stands in for
        self.group, self.world, self.rank = None, world, rank
        self.device = torch.device(DEV)
        self.lib = _hip.load_library()
        self._local = mailboxes[rank]
        self._peers = (C.c_void_p * world)(*mailboxes)
        self._opened = []

    def __del__(self):
        pass


class Rank(Dopri5):
    """Dopri5 on one shard of a state whose other shards are held by the other threads: the global element count — the one thing the
    sharded solver asks its process group for at set-up (solver/_rk_norms.py::_global_counts) — is known here (equal shards)."""

    def _global_counts(self):
        return [c * WORLD for c in self._seg_count_local]


def mailboxes():
    lib = _hip.load_library()
    out = []
    for _ in range(WORLD):
        p = C.c_void_p()
        assert lib.xde_p2p_alloc(C.byref(p)) == 0, lib.xde_last_error().decode()
        out.append(p.value)
    return lib, out


_STREAMS = []


def rank_streams():
    """The eight rank streams, made ONCE per process: the co-residency probe and the solve must run on the same streams (which
    hardware queue a stream lands on is decided when it is created)."""
    if not _STREAMS:
        _STREAMS.extend(torch.cuda.Stream(device=DEV) for _ in range(WORLD))
    return _STREAMS


def run_ranks(body):
    """body(rank) on WORLD threads, each on a stream of its own; the first exception of any rank is raised here."""
    errors, results = [], [None] * WORLD
    streams = rank_streams()
    start = threading.Barrier(WORLD)

    def run(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(streams[r]):
                start.wait(60)
                results[r] = body(r)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, e))
            start.abort()

    ts = [threading.Thread(target=run, args=(r,), name="rank{}".format(r)) for r in range(WORLD)]
    [t.start() for t in ts]
    [t.join(900) for t in ts]
    if any(t.is_alive() for t in ts):
        print("FAILED: a rank thread is still running after 900 s", flush=True)
        os._exit(3)
    if errors:
        r, e = errors[0]
        raise RuntimeError("rank {}: {}: {}".format(r, type(e).__name__, e))
    return results


SKIP_EXIT = 77


def co_residency_probe():
    """Eight exchange launches that wait for one another can only finish if the eight streams run CONCURRENTLY, i.e. sit on eight
    hardware queues.  Three rounds with a short bounded wait: if they do not come through, the box (or the runtime's stream-to-queue
    mapping) does not give this process eight concurrent queues — an environment fact, reported as a skip (exit code 77), not as a
    failure of the library; what follows in this process then has a real meaning."""
    lib, mbs = mailboxes()
    exs = [InProcessExchange(r, WORLD, mbs) for r in range(WORLD)]
    for e in exs:
        e.SPIN_LIMIT = 40_000_000  # ~2 s

    def body(r):
        sums = torch.zeros(32, dtype=torch.float64, device=DEV)
        for i in range(3):
            sums.fill_(float(r + i))
            exs[r].exchange(sums, _hip.NORM_RMS)
        torch.cuda.current_stream().synchronize()
        return exs[r].error()

    errs = run_ranks(body)
    torch.cuda.synchronize()
    for p in mbs:
        lib.xde_p2p_free(p)
    if any(errs):
        print("SKIP: the eight rank streams of this process do not run concurrently (exchange errors {}): GPU_MAX_HW_QUEUES={} was not "
              "honoured or the queues are oversubscribed".format(errs, os.environ.get("GPU_MAX_HW_QUEUES")), flush=True)
        os._exit(SKIP_EXIT)


def exchange():
    co_residency_probe()
    lib, mbs = mailboxes()
    exs = [InProcessExchange(r, WORLD, mbs) for r in range(WORLD)]
    base = torch.arange(32, dtype=torch.float64)

    def body(r):
        sums = torch.zeros(32, dtype=torch.float64, device=DEV)
        for i in range(50):
            sums.copy_((base * (r + 1) + i).to(DEV, non_blocking=False))
            exs[r].exchange(sums, _hip.NORM_RMS)
            want = sum(base * (q + 1) + i for q in range(WORLD))
            got = sums.cpu()
            assert torch.equal(got, want), (r, i, got[:4].tolist(), want[:4].tolist(), exs[r].error_info())
            sums.copy_((base * (r + 1) - i).to(DEV))
            exs[r].exchange(sums, _hip.NORM_LINF)
            parts = torch.stack([base * (q + 1) - i for q in range(WORLD)])
            want = torch.cat([parts.max(0).values[:16], parts.sum(0)[16:]])
            got = sums.cpu()
            assert torch.equal(got, want), (r, i, got[:4].tolist(), want[:4].tolist(), exs[r].error_info())
        assert exs[r].error() == 0
        return True

    run_ranks(body)
    torch.cuda.synchronize()
    for p in mbs:
        lib.xde_p2p_free(p)
    print("OK exchange world={} rounds=100".format(WORLD), flush=True)


def solve(pipeline):
    z = np.load(os.path.join(ROOT, "tests", "golden", "config4_full.npz"))
    B, D = int(z["B"]), int(z["D"])
    A = P.skew_matrix(D).float().to(DEV)
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(0))
    assert np.array_equal(y0[torch.from_numpy(z["rows"])].numpy(), z["y0_rows"])  # the fixture's seeded inputs
    t = torch.from_numpy(z["t"])
    shards = [y0[r * B // WORLD:(r + 1) * B // WORLD].contiguous().to(DEV) for r in range(WORLD)]
    assert all(s.shape == (65536, 64) for s in shards)
    with torch.no_grad():
        _ = shards[0] @ A.T  # the framework picks its GEMM once, before eight threads ask for it at the same time
    torch.cuda.synchronize()
    co_residency_probe()
    lib, mbs = mailboxes()
    exs = [InProcessExchange(r, WORLD, mbs) for r in range(WORLD)]

    def body(r):
        lo = r * B // WORLD
        s = Rank(xde=BaseODE(lambda t_, y: y @ A.T, y0=shards[r], t_span=t), y0=shards[r], rtol=1e-5, atol=1e-7, norm=_rms_norm,
                 pipeline=pipeline, record_trace=True, process_group=True, norm_exchange=exs[r])
        with torch.no_grad():
            sol = s.integrate(t)
        mine = [(i, int(row) - lo) for i, row in enumerate(z["rows"]) if lo <= row < lo + B // WORLD]
        idx = torch.tensor([row for _, row in mine], device=DEV, dtype=torch.long)
        return {"trace": np.asarray([[a, b, c, float(d)] for a, b, c, d in s.trace]), "which": np.asarray([i for i, _ in mine], dtype=np.int64),
                "rows": sol[:, idx].cpu().numpy(), "counts": (s.stats["n_accept"], s.stats["n_reject"], s.stats["nfe"]),
                "error": exs[r].error_info()}

    t0 = time.time()
    rs = run_ranks(body)
    wall = time.time() - t0
    torch.cuda.synchronize()
    for p in mbs:
        lib.xde_p2p_free(p)
    ref = z["trace"]
    for r, out in enumerate(rs):
        assert out["error"][0] == 0, (r, out["error"])
        assert np.array_equal(out["trace"], rs[0]["trace"]), "rank {} left lock-step".format(r)  # bit for bit, eight controllers
        tr = out["trace"]
        assert tr.shape == ref.shape and np.array_equal(tr[:, 3], ref[:, 3])  # the fixture's (global) accept / reject sequence
        assert np.allclose(tr[:, 1], ref[:, 1], rtol=2e-2, atol=0)  # dt: as test_config4_sharded_two_ranks_on_one_gpu_vs_golden
        assert tuple(out["counts"]) == tuple(int(x) for x in z["counts"]), out["counts"]
    got = np.empty_like(z["sol_rows"])
    seen = 0
    for out in rs:
        if len(out["which"]):
            got[:, out["which"]] = out["rows"]
            seen += len(out["which"])
    assert seen == len(z["rows"])
    worst = float(np.abs(got - z["sol_rows"]).max())
    assert worst <= 1e-5 * float(z["sol_abs_max"]), worst
    import json

    print(json.dumps({"ok": True, "world": WORLD, "rows_per_rank": B // WORLD, "dim": D, "pipeline": pipeline, "transport":
                      "xde_p2p_rk_control, 8 in-process mailboxes, 7 peers per rank", "attempts": int(len(ref)), "counts": [int(x) for x in z["counts"]],
                      "worst_abs_err_rows": worst, "bar": 1e-5 * float(z["sol_abs_max"]), "wall_s_functional_not_a_measurement": round(wall, 2),
                      "device": torch.cuda.get_device_name(0)}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "exchange":
        exchange()
    else:
        solve(sys.argv[2] if len(sys.argv) > 2 else "sync")
