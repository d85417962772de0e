"""Euler (reference: paddlexde/solver/fixed_solver/euler.py:4-11)."""
from ... import _hip
from ..base_fixed_solver import FixedSolver


class Euler(FixedSolver):
    order = 1

    @staticmethod
    def _time_values(dt):
        return (dt,)

    def step(self, t0, t1, y0):
        dt = self._host_dt(t0, t1)
        (dtt,) = self._times(t0, dt)
        dy = self._f(t0, dtt, y0)
        y1 = self._combine(y0, [dy], [1.0], _hip.COMBINE_FUSE, dt, out=self._y1_out)
        return y1, dy
