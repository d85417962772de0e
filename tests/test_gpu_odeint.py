"""GPU parity, end to end: the cases of tests/_e2e_cases.py with the HIP backend on cuda:0."""
import pytest

from ._dde_cases import *  # noqa: F401,F403
from ._e2e_cases import *  # noqa: F401,F403
from ._kernel_oracle_cases import *  # noqa: F401,F403
from ._replay_cases import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return "cuda:0"
