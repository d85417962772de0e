from .adaptive_heun import AdaptiveHeun  # noqa: F401
from .bosh3 import Bosh3  # noqa: F401
from .dopri5 import Dopri5  # noqa: F401
from .dopri8 import Dopri8  # noqa: F401
from .fehlberg2 import Fehlberg2  # noqa: F401
