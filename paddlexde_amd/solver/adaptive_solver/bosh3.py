"""`Bosh3` (reference: paddlexde/solver/adaptive_solver/bosh3.py:21-24); the tableau lives in _tableaus.py."""
from ._tableaus import Bosh3  # noqa: F401
