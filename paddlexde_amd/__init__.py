"""paddlexde_amd — MI355X-native integration engine behind the paddlexde ``odeint`` / ``odeint_adjoint`` API.

The compute path is libxde_hip.so (hand-written gfx950 kernels, C ABI in include/xde_hip.h) driven from
Python over ctypes with torch-ROCm tensors as device buffers.  There is no CPU fallback.
"""
__version__ = "0.1.0"

from .functional import AdjointProblem, ddeint, ddeint_adjoint, odeint, odeint_adjoint  # noqa: F401
from .interpolation import BezierSpline, CubicHermiteSpline, LinearInterpolation  # noqa: F401
from .solver import *  # noqa: F401,F403
from .xde import BaseDDE, BaseODE, BaseXDE  # noqa: F401
