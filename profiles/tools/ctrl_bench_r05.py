"""Round 5: the controller launch taken apart on the final library — the round-2 table
(profiles/r02_ctrl_decomposition.txt) re-taken, plus the sharded attempt's one-launch variant (xde_p2p_rk_control) on one rank.

    python3 profiles/tools/ctrl_bench_r05.py [c2|c4]        (XDE_CTRL_FLAGS=0|1|3|7 for the publish / fetch variants)

Durations are the kernels' own dispatch begin -> end (hipExtLaunchKernelGGL-stamped events, as rocprofv3 reports them), 400 launches
each, back to back and behind a streaming kernel (cache state as in a real step).  World sizes above one are not timed here: on ONE
GPU the peers' launches are other processes' dispatches, which the card time-slices — a poll then waits for a scheduler, not for
xGMI; what a rank's launch costs apart from the wait for its peers is what world = 1 shows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist

from paddlexde_amd import Dopri5, _hip
from paddlexde_amd.utils import PeerExchange, _rms_norm
from paddlexde_amd.xde import BaseODE

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
B, D = (65536, 128) if which == "c2" else (65536, 64)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
dist.init_process_group("gloo", rank=0, world_size=1)
be = _hip.get_backend()
y0 = torch.randn(B, D, device=dev)
A = torch.randn(D, D, device=dev) * 0.05
t_span = torch.tensor([0.0, 1e9])


def solver(**kw):
    s = Dopri5(xde=BaseODE(lambda t, y: y @ A, y0=y0, t_span=t_span), y0=y0, rtol=1e-5, atol=1e-7, norm=_rms_norm, pipeline="lag", **kw)
    s.y0 = y0
    s._before_integrate(np.asarray([0.0, 1e9], dtype=np.float32))
    s.advance(5)
    torch.cuda.synchronize()
    return s


s = solver()
ex = PeerExchange(None, dev)
sp = solver(process_group=True, norm_exchange=ex)
ctrl_nomirror = torch.zeros(_hip.C.sizeof(_hip.XdeCtrl), dtype=torch.uint8, device=dev)
ctrl_nomirror.copy_(s._ctrl)
sums = torch.zeros(32, dtype=torch.float64, device=dev)
sums[0] = 1e6
junk = torch.empty(1 << 24, device=dev)
print("# state {} x {} fp32; XDE_CTRL_FLAGS={}; norm grid {} records".format(B, D, os.environ.get("XDE_CTRL_FLAGS", "7 (default)"),
                                                                           os.environ.get("XDE_NORM_GRID", "512 (default)")))


def run(name, fn, kid, n=400, behind_stream=False):
    def one():
        if behind_stream:
            junk.mul_(1.0001)
        fn()
    for _ in range(30):
        one()
    torch.cuda.synchronize()
    be.prof_enable(1)
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    r = be.prof_collect()[kid]
    be.prof_enable(False)
    print("%-72s %7.2f us  (%d launches)" % (name, 1e3 * r["ms"] / max(r["launches"], 1), r["launches"]), flush=True)


for behind in (False, True):
    tag = "  [behind a streaming kernel]" if behind else ""
    run("A  control: 512 partial records + host mirror (production)" + tag, lambda: be.rk_control(s._ctrl, s._params, s._ws, None, s._t_span_dev, None, s._t_stage), "control", behind_stream=behind)
    run("B  control: partial records, no mirror" + tag, lambda: be.rk_control(ctrl_nomirror, s._params, s._ws, None, s._t_span_dev, None, s._t_stage), "control", behind_stream=behind)
    run("C  control: finalised sums, no mirror" + tag, lambda: be.rk_control(ctrl_nomirror, s._params, None, sums, s._t_span_dev, None, s._t_stage), "control", behind_stream=behind)
    run("C' control: finalised sums + mirror" + tag, lambda: be.rk_control(s._ctrl, s._params, None, sums, s._t_span_dev, None, s._t_stage), "control", behind_stream=behind)
    run("D  finalize alone (records -> 32 sums)" + tag, lambda: be.norm_finalize(s._ws, 0, sums), "finalize", behind_stream=behind)
    run("E  p2p control, world 1: records + mailbox + poll + mirror (production)" + tag,
        lambda: be.p2p_rk_control(sp._ctrl, sp._params, sp._ws, ex, sp._t_span_dev, None, sp._t_stage), "control", behind_stream=behind)
    run("E' p2p control, world 1, no mirror" + tag,
        lambda: be.p2p_rk_control(ctrl_nomirror, sp._params, sp._ws, ex, sp._t_span_dev, None, sp._t_stage), "control", behind_stream=behind)
    run("F  p2p exchange alone, world 1 (32 sums -> mailbox -> poll -> sums)" + tag, lambda: ex.exchange(sums, _hip.NORM_RMS), "finalize", behind_stream=behind)
ex.close()
dist.destroy_process_group()
