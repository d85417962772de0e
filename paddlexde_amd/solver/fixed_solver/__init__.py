"""Fixed-grid solver classes (the names the reference exports from this package)."""
from . import adams, euler, midpoint, rk4

RK4, Euler, Midpoint = rk4.RK4, euler.Euler, midpoint.Midpoint
AdamsBashforthMoulton = adams.AdamsBashforthMoulton

__all__ = ["AdamsBashforthMoulton", "Euler", "Midpoint", "RK4"]
