from .ode_utils import _linf_norm, _mixed_norm, _rms_norm, _zero_norm  # noqa: F401
