"""`Fehlberg2` (reference: paddlexde/solver/adaptive_solver/fehlberg2.py:18-21); the tableau lives in _tableaus.py."""
from ._tableaus import Fehlberg2  # noqa: F401
