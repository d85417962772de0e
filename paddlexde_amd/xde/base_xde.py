"""Problem-wrapper protocol (reference: paddlexde/xde/base_xde.py:8-107).

Solvers bind ``xde.move`` / ``xde.fuse`` exactly as the reference does
(solver/base_fixed_solver.py:62-64, solver/base_adaptive_solver.py:12-14).  ``move`` is the
framework call into the user's ``func``; ``fuse`` is kept for API parity but the solvers never
call it on the hot path: its ``dy * dt + y0`` is what xde_stage_combine computes in fused form.
"""
from abc import ABC, abstractmethod

import torch.nn as nn


class BaseXDE(ABC, nn.Module):
    def __init__(self, name, var_nums, y0, t_span):
        super().__init__()
        # (plain attributes, set past nn.Module.__setattr__'s parameter / buffer / submodule bookkeeping: a wrapper is built per
        # odeint() call, and that bookkeeping was 15 of a call's ~100 us of host set-up, profiles/r06_odeint_tail.txt)
        d = self.__dict__
        d["name"] = name
        d["var_nums"] = var_nums
        d["t_span"] = t_span
        d["pred_len"] = t_span.shape

    def method(self):
        print(f"current method is {self.name}.")
        return self.name

    @abstractmethod
    def init_y0(self, input):
        raise NotImplementedError

    @abstractmethod
    def handle(self, h, ts):
        raise NotImplementedError

    @abstractmethod
    def move(self, t0, dt, y0):
        raise NotImplementedError

    @abstractmethod
    def fuse(self, dy, dt, y0):
        raise NotImplementedError

    def unflatten(self, input, length):
        raise NotImplementedError

    def flatten(self, input):
        raise NotImplementedError

    def format(self, sol):
        """Identity (the reference calls an undefined ``xde.format``, functional/odeint.py:33 — SURVEY D1)."""
        return sol

    def on_integrate_step_end(self, y0=None, y1=None, t0=None, t1=None):
        pass

    @abstractmethod
    def call_func(self, **kwargs):
        raise NotImplementedError
