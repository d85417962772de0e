"""`AdaptiveHeun` (reference: paddlexde/solver/adaptive_solver/adaptive_heun.py:23-26); the tableau lives in _tableaus.py."""
from ._tableaus import AdaptiveHeun  # noqa: F401
