"""Midpoint (reference: paddlexde/solver/fixed_solver/midpoint.py:4-18)."""
from ... import _hip
from ..base_fixed_solver import FixedSolver


class Midpoint(FixedSolver):
    order = 2

    @staticmethod
    def _time_values(dt):
        return (dt, 0.5 * dt, 0.5 * dt)  # dt, half_dt, t0 + half_dt

    def _time_values_tagged(self, dt):
        v = self._time_values(dt)
        return [(v[0], False), (v[1], False), (v[2], True)]

    def _times(self, t0, dt):
        if self._row is not None:
            return [self._row[j : j + 1] for j in range(3)]
        v = self._time_values(dt)
        return [self._tdev(v[0], t0), self._tdev(v[1], t0), self._tdev(type(dt)(t0.item()) + v[2], t0)]

    def step(self, t0, t1, y0):
        dt = self._host_dt(t0, t1)
        half_dt = 0.5 * dt
        dtt, hdt, t_half = self._times(t0, dt)
        dy_half = self._f(t0, hdt, y0)
        y_half = self._combine(y0, [dy_half], [1.0], _hip.COMBINE_FUSE, half_dt)
        dy = self._f(t_half, dtt, y_half)
        y1 = self._combine(y0, [dy], [1.0], _hip.COMBINE_FUSE, dt, out=self._y1_out)
        return y1, dy
