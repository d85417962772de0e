"""GPU parity, end to end: the cases of tests/_{forward,options,pipeline,adjoint}_cases.py with the HIP backend on cuda:0."""
import pytest

from ._adjoint_cases import *  # noqa: F401,F403
from ._dde_cases import *  # noqa: F401,F403
from ._forward_cases import *  # noqa: F401,F403
from ._kernel_oracle_cases import *  # noqa: F401,F403
from ._options_cases import *  # noqa: F401,F403
from ._pipeline_cases import *  # noqa: F401,F403
from ._replay_cases import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return "cuda:0"
