from .ode_utils import _linf_norm, _mixed_norm, _rms_norm, _zero_norm  # noqa: F401
from .graphed import GraphedFunc  # noqa: F401,E402
from .p2p import PeerExchange  # noqa: F401,E402
from .rccl import RcclExchange  # noqa: F401,E402
from .exchange import negotiate as negotiate_exchange  # noqa: F401,E402
