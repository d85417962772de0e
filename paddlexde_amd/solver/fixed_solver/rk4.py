"""RK4 (reference: paddlexde/solver/fixed_solver/rk4.py:4-10 — uses the *alt* step)."""
from ..base_fixed_solver import FixedSolver, _one_third, _two_thirds


class RK4(FixedSolver):
    """``options={"variant": "classic"}`` selects the textbook RK4 (``rk4_step_func``, base_fixed_solver.py:146-164,
    which the reference defines but never uses); the default "alt" is what the reference's RK4 runs."""

    order = 4

    def __init__(self, *args, variant="alt", **kwargs):
        super().__init__(*args, **kwargs)
        if variant not in ("alt", "classic"):
            raise ValueError("variant must be 'alt' (reference) or 'classic'")
        self.variant = variant
        # the classical variant builds its stage times as per-step constants (no table row): not replayable from a graph
        self.graphable = variant == "alt"

    @staticmethod
    def _time_values(dt):
        # dt, dt/3, t0 + dt/3, t0 + 2dt/3  (base_fixed_solver.py:168-174); the last two are offsets from t0
        return (dt, dt * _one_third, dt * _one_third, dt * _two_thirds)

    def _time_values_tagged(self, dt):
        v = self._time_values(dt)
        return [(v[0], False), (v[1], False), (v[2], True), (v[3], True)]

    def _times(self, t0, dt):
        if self._row is not None:
            return [self._row[j : j + 1] for j in range(4)]
        v = self._time_values(dt)
        t0h = type(dt)(t0.item())
        return [self._tdev(v[0], t0), self._tdev(v[1], t0), self._tdev(t0h + v[2], t0), self._tdev(t0h + v[3], t0)]

    def step(self, t0, t1, y0):
        dt = self._host_dt(t0, t1)
        f0 = self._f(t0, self._times(t0, dt)[0], y0)
        if self.variant == "classic":
            return self.rk4_step_func(t0, t1, y0, f0=f0), f0
        return self.rk4_alt_step_func(t0, t1, y0, f0=f0), f0
