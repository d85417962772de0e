from .ddeint import ddeint  # noqa: F401
from .ddeint_adjoint import ddeint_adjoint  # noqa: F401
from .odeint import odeint  # noqa: F401
from .odeint_adjoint import AdjointProblem, odeint_adjoint  # noqa: F401
