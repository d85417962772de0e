"""Choosing the transport of the per-attempt norm sums for a batch-sharded solve, safely for the whole group.

The batch-sharded integrator has one coupling between its ranks: 32 doubles per attempted step (the error norm reduces over the
whole batch, reference utils/ode_utils.py:8-9,80-82).  Three transports carry them:

  "p2p"        utils.PeerExchange — one-shot stores into IPC-mapped mailboxes over xGMI; with ``fused_control`` the whole
               finalize -> exchange -> controller of an attempt is ONE launch (xde_p2p_rk_control);
  "rccl"       utils.RcclExchange — ncclAllReduce issued directly on the solver's stream;
  "allreduce"  torch.distributed.all_reduce on the process group (always available).

Two rules keep a multi-rank job from hanging or splitting at set-up time (round-3 advice: a rank that failed before its peers'
collective left them waiting in it):

  * a constructor does every step that can fail on ONE rank (loading a library, allocating, exporting, importing a handle)
    outside any collective, and the ranks AGREE (`agree`: a MIN all-reduce of a success flag on the group's own backend) before the
    next collective step is entered.  A failure anywhere is therefore an exception on EVERY rank, raised at the same point;
  * `negotiate` tries the preferred transports in order; each candidate is also self-tested (known vectors, exact result, on every
    rank) and the verdict is agreed before it is adopted.  The group always ends up on one and the same transport.

What remains fatal by design: a rank that dies or hangs INSIDE a collective initialisation (ncclCommInitRank) — nothing in-process
can rescue its peers; bench.py's per-stage watchdog turns that into a diagnosed non-zero exit.
"""
import torch

from .. import _hip


def _control_tensor(value, group):
    """A one-element tensor the group's backend can reduce: host memory for gloo, the current device for nccl."""
    import torch.distributed as dist

    on_host = dist.get_backend(group) == "gloo"
    return torch.tensor([value], dtype=torch.float64, device="cpu" if on_host else torch.device("cuda", torch.cuda.current_device()))


def agree(ok, group=None):
    """True iff ``ok`` is true on EVERY rank of ``group`` (collective: every rank must call it at the same point)."""
    import torch.distributed as dist

    t = _control_tensor(1.0 if ok else 0.0, group)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return float(t.item()) == 1.0


def raise_together(local_error, what, group=None):
    """After a rank-local step: if any rank failed, raise on all of them (the failing ranks with their own reason)."""
    if agree(local_error is None, group):
        return
    if local_error is not None:
        raise _hip.XdeError("{}: {}: {}".format(what, type(local_error).__name__, local_error))
    raise _hip.XdeError("{}: failed on another rank of the group".format(what))


def selftest(exchange, group=None, rounds=3):
    """Known vectors through ``exchange`` (sum, then max-and-sum), compared exactly on every rank; the verdict is agreed.  Small
    integers: every order of summation gives the same doubles."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    m = _hip.XDE_MAX_SEG
    dev = exchange.device
    base = torch.arange(2 * m, dtype=torch.float64)
    ok, why = True, None

    def note(msg):
        nonlocal ok, why
        if ok:
            ok, why = False, msg

    try:
        # (every rank runs EVERY round whatever it has seen so far: a rank that left early would leave its peers waiting in the
        #  next exchange until their bounded wait ran out)
        for i in range(rounds):
            sums = (base * (rank + 1) + i).to(dev)
            exchange.exchange(sums, _hip.NORM_RMS)
            want = sum(base * (r + 1) + i for r in range(world))
            if not torch.equal(sums.cpu(), want):
                note("sum round {}: got {} want {}".format(i, sums.cpu()[:4].tolist(), want[:4].tolist()))
            sums = (base * (rank + 1) - i).to(dev)
            exchange.exchange(sums, _hip.NORM_LINF)
            parts = torch.stack([base * (r + 1) - i for r in range(world)])
            want = torch.cat([parts.max(0).values[:m], parts.sum(0)[m:]])
            if not torch.equal(sums.cpu(), want):
                note("max round {}: got {} want {}".format(i, sums.cpu()[:4].tolist(), want[:4].tolist()))
        if ok and exchange.error():
            note("the exchange reports a failed round")
    except Exception as e:  # noqa: BLE001 - whatever went wrong, the group must hear about it
        note("{}: {}".format(type(e).__name__, e))
    return agree(ok, group), why


NAMES = {
    "p2p": "peer-to-peer mailbox exchange over xGMI, fused with the controller launch (xde_p2p_rk_control)",
    "rccl": "in-stream ncclAllReduce (RCCL)",
    "allreduce": "all-reduce (torch.distributed)",
}


def negotiate(group=None, device=None, prefer=("p2p", "rccl", "allreduce"), log=None):
    """-> (exchange or None, kind, report).  ``exchange`` goes into ``options["norm_exchange"]`` (None: the group's own all_reduce).
    Collective: every rank of ``group`` calls it with the same ``prefer``.  ``report``: what was tried and why it was dropped."""
    from .p2p import PeerExchange
    from .rccl import RcclExchange

    report = []
    for kind in prefer:
        if kind == "allreduce":
            report.append({"transport": kind, "adopted": True})
            return None, kind, report
        cls = {"p2p": PeerExchange, "rccl": RcclExchange}[kind]
        ex, why = None, None
        try:
            ex = cls(group, device)  # group-consistent by construction: raises on every rank or on none ...
        except Exception as e:  # noqa: BLE001
            why = "{}: {}".format(type(e).__name__, e)
        built = agree(ex is not None, group)  # ... and checked all the same, before the next collective
        ok = built
        if built:
            ok, why = selftest(ex, group)
        if ok:
            report.append({"transport": kind, "adopted": True})
            return ex, kind, report
        if ex is not None:
            try:
                if built:
                    ex.close()  # every rank holds one: the (collective) close is safe
                else:
                    ex.abandon()  # rank-local release only
            except Exception:  # noqa: BLE001
                pass
        report.append({"transport": kind, "adopted": False, "why": why or "failed on another rank"})
        if log is not None:
            log("norm exchange '{}' not adopted ({}); trying the next transport".format(kind, why or "failed on another rank"))
    raise _hip.XdeError("no norm-exchange transport could be set up: {}".format(report))
