"""The spiral neural-ODE demo (examples/ode_demo.py, counterpart of the reference's example/ode_demo.py) trains: the loss goes
down when back-propagating through odeint(RK4), through odeint_adjoint(RK4) and through odeint_adjoint(Dopri5)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


@pytest.mark.parametrize("adjoint,solver", [(False, "rk4"), (True, "rk4"), (True, "dopri5")])
def test_demo_loss_decreases(adjoint, solver):
    import ode_demo

    steps = 120 if solver == "rk4" else 60
    losses = ode_demo.train(max_steps=steps, adjoint=adjoint, solver=solver, log_every=0)
    head, tail = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    assert tail < 0.9 * head, (head, tail)


def test_dde_demo_trains_weights_and_lags():
    """examples/dde_demo.py (counterpart of the reference's example/dde_demo.py): ddeint(RK4) with learned delays."""
    import dde_demo

    losses, lag_grad = dde_demo.train(max_steps=70, log_every=0)
    head, tail = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    assert tail < 0.9 * head, (head, tail)
    assert lag_grad > 0.0


def test_sharded_demo_runs_with_two_ranks_on_one_gpu():
    """examples/sharded_demo.py under torch.distributed.run, rehearsed with two ranks on this GPU (gloo) and with one rank over nccl
    (where the per-step all-reduce is RcclExchange's in-stream ncclAllReduce)."""
    import socket
    import subprocess

    def free_port():
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    demo = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "sharded_demo.py")
    for n, extra in ((2, {"XDE_DEMO_REHEARSAL": "1"}), (1, {})):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                            "--master-port", str(free_port()), demo], capture_output=True, text=True, timeout=300, env=dict(env, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("ranks")][0]
        assert "ranks {} ".format(n) in line and float(line.split()[-1]) < 1e-4, line


@pytest.mark.parametrize("solver", ["rk4", "dopri5"])
def test_foreign_framework_demo_trains_through_the_adjoint(solver):
    """examples/foreign_adjoint_demo.py: the spiral MLP as a layer of ANOTHER framework (tensors that expose only DLPack, a hand-written
    vjp, RMSprop on the caller's own storage) trained through `AdjointProblem` — no autograd of any framework anywhere; the loss falls as
    it does for the torch demo ("for 'drops in under the existing Paddle models' the training path is the one that matters")."""
    import foreign_adjoint_demo

    steps = 120 if solver == "rk4" else 60
    losses = foreign_adjoint_demo.train(max_steps=steps, solver=solver, log_every=0)
    head, tail = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    assert tail < 0.9 * head, (head, tail)
