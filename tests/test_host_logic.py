"""Host logic on CPU: the cases of tests/_e2e_cases.py with the numpy test double of the kernel backend
(tests/_cpu_double.py).  Exercises the product's Python drivers (solver loops, operand plans, pipelines,
tuple flattening, adjoint) without a GPU; the HIP kernels themselves are covered by the `-m gpu` suite."""
import pytest

from ._dde_cases import *  # noqa: F401,F403
from ._e2e_cases import *  # noqa: F401,F403
from ._kernel_oracle_cases import *  # noqa: F401,F403
from ._replay_cases import *  # noqa: F401,F403


@pytest.fixture
def dev(cpu_double):
    return "cpu"
