from .odeint import odeint  # noqa: F401
from .odeint_adjoint import odeint_adjoint  # noqa: F401
