from .adams import AdamsBashforthMoulton  # noqa: F401
from .euler import Euler  # noqa: F401
from .midpoint import Midpoint  # noqa: F401
from .rk4 import RK4  # noqa: F401
