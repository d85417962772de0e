from .base_ode import BaseODE  # noqa: F401
from .base_xde import BaseXDE  # noqa: F401
from .base_dde import BaseDDE, HistoryIndex  # noqa: F401,E402
