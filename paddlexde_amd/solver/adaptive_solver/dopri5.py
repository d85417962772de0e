"""`Dopri5` (reference: paddlexde/solver/adaptive_solver/dopri5.py:58-61); the tableau lives in _tableaus.py."""
from ._tableaus import Dopri5  # noqa: F401
