"""Round-5 lab: would splitting the batch into chunks on separate streams let a chunk's stage combine (HBM-bound) overlap another
chunk's func GEMM (partly compute-bound)?  Not a solver: a chain that has a Dopri5 attempt's launch shape — six times
[stage combine with 1..6 derivatives -> func = torch.mm] on config 2's state (65536 x 128 fp32) — captured as ONE hipGraph of `STEPS`
attempts, (a) whole batch on one stream, (b) 2 / 4 row chunks, each its own chain on its own stream, joined once per attempt (where the
global error norm would be).  Prints microseconds per attempt, alternating the variants `REPS` times.

    python3 profiles/tools/overlap_lab_r05.py [B] [D]
"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from paddlexde_amd import _hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
STEPS, REPS = 20, 6
dev = torch.device("cuda:0")
bench.enable_tunable_op(True)
be = _hip.get_backend()
g = torch.Generator().manual_seed(0)
AT = (torch.randn(D, D, generator=g) / D**0.5).to(dev)
COEF = [0.2, 0.075, 0.225, 0.3, 0.1, 0.05]


class Chain:
    def __init__(self, rows):
        self.y0 = torch.randn(rows, D, generator=g).to(dev)
        self.ks = [torch.randn(rows, D, generator=g).to(dev) for _ in range(7)]
        self.out = torch.empty_like(self.y0)

    def attempt(self):
        for i in range(1, 7):
            be.stage_combine(self.out, self.y0, self.ks[:i], COEF[:i], _hip.COMBINE_RK, dt_host=0.01)
            torch.mm(self.out, AT, out=self.ks[i])


def build(n_chunks):
    chains = [Chain(B // n_chunks) for _ in range(n_chunks)]
    side = [torch.cuda.Stream() for _ in range(n_chunks - 1)]
    for c in chains:  # eager warm-up: the framework picks its GEMM here
        c.attempt()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        main = torch.cuda.current_stream()
        for _ in range(STEPS):
            for s in side:
                s.wait_stream(main)
            chains[0].attempt()
            for s, c in zip(side, chains[1:]):
                with torch.cuda.stream(s):
                    c.attempt()
            for s in side:
                main.wait_stream(s)
    return graph, chains


def timed(graph):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / (5 * STEPS)


variants = {n: build(n) for n in (1, 2, 4)}
for n, (gr, _) in variants.items():
    timed(gr)
res = {n: [] for n in variants}
for _ in range(REPS):
    for n, (gr, _) in variants.items():
        res[n].append(timed(gr))
print("# {} x {} fp32, {} attempts per graph, {} alternating repetitions; us per attempt (6 x [combine, GEMM])".format(B, D, STEPS, REPS))
for n, v in res.items():
    print("chunks {}: median {:7.1f}  min {:7.1f}  max {:7.1f}".format(n, statistics.median(v), min(v), max(v)))
