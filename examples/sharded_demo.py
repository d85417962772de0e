#!/usr/bin/env python3
"""Batch-sharded integration over the GPUs of one node — the multi-GPU counterpart of the reference's data-parallel launch
(example/D3STN/README.md:53-59: `python -m paddle.distributed.launch --gpus 0,...,7 train_dde.py`).

Every rank owns its rows of the batch; the ONE thing the ranks share is the step size, i.e. the global error norm
(utils/ode_utils.py:8-9,80-82 reduce over every element of the batch): 32 doubles are all-reduced per attempted step, everything else
is local.  The step sequence is therefore the one a single GPU would take on the whole batch, and identical on every rank.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_demo.py
    XDE_DEMO_REHEARSAL=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/sharded_demo.py   # one GPU, gloo
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from paddlexde_amd import Dopri5, odeint  # noqa: E402
from paddlexde_amd.utils import _rms_norm, negotiate_exchange  # noqa: E402


def main():
    rehearsal = os.environ.get("XDE_DEMO_REHEARSAL", "0") == "1"  # all ranks on cuda:0, gloo carries the collectives
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("gloo" if rehearsal else "nccl", **({} if rehearsal else {"device_id": dev}))

    B, D = 65536, 64  # GLOBAL batch; rank r integrates rows [r B / world, (r + 1) B / world)
    g = torch.Generator().manual_seed(1)
    U = 0.1 * torch.randn(D, D, generator=g)
    A = (U - U.T).to(dev)
    y0_all = torch.randn(B, D, generator=torch.Generator().manual_seed(0))
    y0_all[B // 2:] *= 25.0  # rows of very different magnitude: each shard alone would choose a different step size
    rows = slice(rank * B // world, (rank + 1) * B // world)
    y0 = y0_all[rows].to(dev)
    t = torch.linspace(0.0, 1.0, 5)

    options = {"norm": _rms_norm, "process_group": True}
    # how the 32 doubles travel: the fastest transport the whole group can set up and self-test — one-shot stores into IPC-mapped
    # mailboxes over xGMI (fused with the controller launch), else ncclAllReduce on the solver's own stream, else the group's all_reduce
    exchange, kind, _ = negotiate_exchange(None, dev, prefer=("p2p", "allreduce") if rehearsal else ("p2p", "rccl", "allreduce"))
    if exchange is not None:
        options["norm_exchange"] = exchange
    sol = odeint(lambda t_, y: y @ A.T, y0, t, solver=Dopri5, rtol=1e-5, atol=1e-7, options=options)  # [T, B / world, D]

    # check: the rotation conserves every row's norm, and all ranks took the same steps (their last rows' norms gathered)
    drift = float(((sol[-1].norm(dim=1) - y0.norm(dim=1)).abs() / y0.norm(dim=1)).max())
    worst = torch.tensor([drift], device="cpu" if rehearsal else dev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    if rank == 0:
        print("ranks {}  global batch {} x {}  rows per rank {}  norm exchange '{}'  worst relative norm drift over all ranks {:.2e}".format(
            world, B, D, B // world, kind, float(worst)))
    if exchange is not None:
        exchange.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
