"""ODE problem wrapper (reference: paddlexde/xde/base_ode.py:9-62)."""
from .base_xde import BaseXDE


class BaseODE(BaseXDE):
    def __init__(self, func, y0, t_span):
        super().__init__(name="ODE", var_nums=1, y0=y0, t_span=t_span)
        # keep func out of nn.Module registration semantics differences: plain attribute is fine for
        # both nn.Module and bare callables
        self.func = func
        self.init_y0(y0)

    def init_y0(self, y0):
        self.__dict__["y0"] = y0  # (a plain attribute: see BaseXDE.__init__)

    def handle(self, h, ts):
        pass

    def move(self, t0, dt, y0):
        """base_ode.py:47-49 — ``dt`` is ignored."""
        return self.call_func(t0, y0)

    def fuse(self, dy, dt, y0):
        """base_ode.py:51-58"""
        return dy * dt + y0

    def call_func(self, t, y0):
        return self.func(t, y0)
