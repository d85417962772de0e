#!/usr/bin/env python3
"""Host-side cost of one attempted Dopri5 step in the speculative ("lag") pipeline: the state is made so small (256 x 64) that the
GPU is never the bottleneck, so time per step = what the host needs to ENQUEUE a step.  At config 4's per-GPU shard (65536 x 64) the
GPU needs ~165 us per step: a host that needs more than that is the bottleneck of the 8-GPU run.  Also with the sharded code path
(XDE_BENCH_FORCE_DIST-style: a one-rank nccl group, finalize -> all-reduce -> controller) and a cProfile breakdown.

    python3 profiles/tools/host_profile.py [--dist] [--profile]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", action="store_true")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--pipeline", default="lag")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if args.dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    from paddlexde_amd import Dopri5
    from paddlexde_amd.utils import _rms_norm
    from paddlexde_amd.xde import BaseODE

    g = torch.Generator().manual_seed(1)
    U = 0.1 * torch.randn(args.dim, args.dim, generator=g)
    AT = (U - U.T).T.contiguous().to(dev)
    y0 = torch.randn(args.batch, args.dim, generator=torch.Generator().manual_seed(0)).to(dev)
    func = lambda t, y: y @ AT  # noqa: E731
    s = Dopri5(xde=BaseODE(func, y0=y0, t_span=torch.tensor([0.0, 1e9])), y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm,
               pipeline=args.pipeline, process_group=(True if args.dist else None))
    s.y0 = y0
    s._before_integrate(np.asarray([0.0, 1e9], dtype=np.float32))
    s.advance(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.advance(args.steps)
    host_done = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("batch {} x {} pipeline={} dist={}: {:.1f} us/step wall, {:.1f} us/step until the host had enqueued everything".format(
        args.batch, args.dim, args.pipeline, args.dist, 1e6 * el / args.steps, 1e6 * host_done / args.steps))
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        s.advance(args.steps)
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(22)
    if args.dist:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
