"""Embedded Runge-Kutta solver classes (the names the reference exports from this package)."""
from . import _tableaus, dopri8

Dopri5, Bosh3, Fehlberg2, AdaptiveHeun = _tableaus.Dopri5, _tableaus.Bosh3, _tableaus.Fehlberg2, _tableaus.AdaptiveHeun
Dopri8 = dopri8.Dopri8

__all__ = ["AdaptiveHeun", "Bosh3", "Dopri5", "Dopri8", "Fehlberg2"]
