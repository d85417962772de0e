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
