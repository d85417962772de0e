"""Re-armable interval solves: one captured graph = the initial-step heuristic + the first attempted step of a 2-point solve.

odeint_adjoint's backward pass is one short solve per output interval (paddlexde/functional/odeint_adjoint.py:134-159): same state
layout, same tolerances, same direction, 1-2 attempted steps each — and, for a small state, launch-bound.  Instead of one solver
object, one control block and ~170 launches per interval, ONE solver is kept for the whole sweep (and the next backward pass): the
state lives in a static buffer (``state``), the two output times are uploaded into a static pair, and a replayed hipGraph re-arms
the control block on the device (xde_initial_step_fused with t_start = NaN and seq0 < 0), runs the heuristic's two evaluations and
the first attempt, writes the output row and hands the state over (xde_dense_commit).  A second graph holds one more attempt for the
intervals that need it.  Same kernels, same operands, same order as the eager solve: bit-identical results
(tests/_adjoint_cases.py::test_adjoint_captured_interval_solves).
"""
import numpy as np
import torch

from .. import _hip
from ._common import np_dtype


class IntervalSolves:
    """The re-arming machinery of ONE stepper (an ``AdaptiveRKSolver``), reached through ``solver.intervals_prepare`` /
    ``solver.interval_solve`` / ``solver.interval_state``."""

    def __init__(self, stepper):
        self.s = stepper
        self.first_graph = self.next_graph = None

    def supported(self):
        """Whether the stepper's options allow the captured interval solve (else: one ordinary solve per interval)."""
        s = self.s
        return bool(s.y0.is_cuda and not s._custom_norm and s._chunks is None and s.process_group is None
                    and not s._replay and s.first_step is None and s.step_t is None and not s._has_callbacks
                    and s._step_hook is None and not s.record_trace and s._reuse_f0 and s._stats_out is None
                    and s.pipeline == "auto" and s.y0.dim() == 1)

    def prepare(self, t_span, capture=True):
        """Static buffers for 2-point solves in the direction of ``t_span`` (two host times), one eager solve of that span from
        the constructor's ``y0`` as warm-up and — ``capture`` — the two graphs.  Main thread, outside autograd nodes (see
        utils/graphed.py); raises when the capture fails (the caller keeps the per-interval solves)."""
        s = self.s
        be = s.backend
        t_span = s._setup(t_span)
        if len(t_span) != 2 or not t_span[0] != t_span[1]:
            raise ValueError("intervals_prepare needs two distinct output times")
        y0, dev = s.y0, s.y0.device
        # One static device block of five doubles: [d0, d1 | t0, t1 | t0 in the time dtype].  The last three — the interval's two
        # output times as the kernels read them and its start time as func is handed it — are written by ONE host-to-device copy
        # per interval from a pinned mirror; the first two are the heuristic's norms (separate launches of a larger state), so that
        # the scalar kernel finds (d0, d1, start time) side by side (xde_initial_step, phase 2).
        self.pinned = torch.zeros(24, dtype=torch.uint8).pin_memory()
        self.block = torch.zeros(40, dtype=torch.uint8, device=dev)
        self.res = self.block[:24].view(torch.float64)
        s._t_span_dev = self.block[16:32].view(torch.float64)
        self.t0 = self.block[32:40].view(s.dtype)[0]
        self._upload(t_span)
        self.base = (y0.clone(), None)
        self.y1 = torch.empty_like(y0)
        self.hs = torch.zeros(5, dtype=torch.float64, device=dev)
        self.tprobe = torch.empty((), dtype=torch.promote_types(s.dtype, y0.dtype), device=dev)
        s._solution = torch.empty((2,) + tuple(y0.shape), dtype=y0.dtype, device=dev)
        self.first_graph = self.next_graph = None
        s._ctrl_ready = True
        # (a control block whose launch count the host mirror agrees with, before the in-graph re-arming keeps counting from it)
        be.ctrl_init(s._ctrl, s._params, float(t_span[0]), 0.0, 2, s._t_span_dev, s._step_t_dev, s._t_stage)
        with torch.no_grad(), torch.autograd.set_multithreading_enabled(False):
            nfe0 = s.nfe
            self.solve(t_span)  # eager: func's lazy initialisations, the allocator's blocks
            if capture:
                torch.cuda.synchronize(dev)
                self.first_graph = be.capture(self._first, s._ctrl, launches=1)
                self.next_graph = be.capture(self._attempt, s._ctrl, launches=1)
            s.nfe = nfe0
        return s

    def _upload(self, t):
        host = self.pinned.numpy()
        host[:16].view(np.float64)[:] = np.asarray(t, dtype=np.float64)
        host[16:24].view(np_dtype(self.s.dtype))[0] = t[0]
        self.block[16:].copy_(self.pinned, non_blocking=True)

    @property
    def state(self):
        """The static state buffer an interval solve starts from (write the state into it, do not replace it)."""
        return self.base[0]

    def _first(self):
        s = self.s
        be = s.backend
        y0 = self.base[0]
        nan = float("nan")
        # (f0 stays where func wrote it — recorded: a block of this graph's private pool, kept allocated by this reference, which the
        # second graph addresses too)
        f0 = s._eval(self.t0, y0)
        self.base = (y0, f0)
        if s._fused_first_step():  # small state: two one-workgroup launches around the Euler probe
            be.initial_step_fused(0, f0, None, y0, s._xsegs, self.hs, s._params, nan, self.tprobe, s._ctrl, t_span_dev=s._t_span_dev,
                                  keep_seq=True)
            be.stage_combine(self.y1, y0, [f0], [1.0], _hip.COMBINE_FUSE, ctrl=s._ctrl)  # fuse(f0, h0, y0)
            f1 = s._eval(self.tprobe, self.y1)
            be.initial_step_fused(1, f1, f0, y0, s._xsegs, self.hs, s._params, nan, None, s._ctrl, 2, s._t_span_dev, s._step_t_dev,
                                  s._t_stage, keep_seq=True)
        elif s._tail_first_step():  # a larger state: the norm passes as launches of their own, one one-workgroup launch after each
            be.scaled_norm2_partial(f0, y0, s.rtol, s.atol, s._xsegs, s._norm_kind, s._ws)
            be.initial_step_tail(0, s._ws, self.hs, s._params, nan, self.tprobe, s._ctrl, t_span_dev=s._t_span_dev, keep_seq=True)
            be.stage_combine(self.y1, y0, [f0], [1.0], _hip.COMBINE_FUSE, ctrl=s._ctrl)  # fuse(f0, h0, y0)
            f1 = s._eval(self.tprobe, self.y1)
            be.scaled_norm_partial(f1, f0, y0, float(s.rtol), float(s.atol), s._xsegs, s._norm_kind, s._ws, 0)
            be.initial_step_tail(1, s._ws, self.hs, s._params, nan, None, s._ctrl, 2, s._t_span_dev, s._step_t_dev, s._t_stage,
                                 keep_seq=True)
        else:  # the separate launches of _select_initial_step_device, the start time read on the device
            res, hs = self.res, self.hs
            s._scaled_norm_into(y0, None, y0, s.rtol, s.atol, res[0:1])
            s._scaled_norm_into(f0, None, y0, s.rtol, s.atol, res[1:2])
            be.initial_step(2, res, hs, s._params, nan, self.tprobe, s._ctrl)  # h0 -> ctrl.dt, t0 + h0 -> t_probe
            be.stage_combine(self.y1, y0, [f0], [1.0], _hip.COMBINE_FUSE, ctrl=s._ctrl)  # fuse(f0, h0, y0)
            f1 = s._eval(self.tprobe, self.y1)
            s._scaled_norm_into(f1, f0, y0, s.rtol, s.atol, res[0:1])
            be.initial_step(1, res, hs, s._params, nan, None, s._ctrl)
            be.ctrl_init(s._ctrl, s._params, nan, 0.0, 2, s._t_span_dev, s._step_t_dev, s._t_stage, first_step_dev=hs[3:4], keep_seq=True)
        self._attempt()

    def _attempt(self):
        s = self.s
        base = self.base
        y1, ks = s._attempt(base)
        idx, coef = s._mid_plan
        s.backend.dense_commit(s._solution, [ks[j] for j in idx], coef, base[0], y1, ks[-1], s._ctrl, s._t_span_dev,
                               _hip.dtype_code(s.dtype))

    def solve(self, t_span):
        """Integrate ``state`` from ``t_span[0]`` to ``t_span[1]`` (host times, the prepared direction); returns the state at
        ``t_span[1]`` — a view of a static row, valid until the next call.  ``state`` is overwritten."""
        s = self.s
        be = s.backend
        tt = np_dtype(s.dtype)
        t = np.asarray([t_span[0], t_span[1]], dtype=tt)
        d = s._direction
        if not d * t[1] > d * t[0]:
            if t[1] == t[0]:  # (an output time repeated: the state itself, as integrate() fills such rows)
                s._solution[1].copy_(self.base[0])
                return s._solution[1]
            raise AssertionError("interval_solve: the interval runs against the prepared direction")
        self._upload(t)
        graphs = self.first_graph is not None
        with torch.no_grad():
            first = True
            while True:
                if graphs:
                    (handle,) = (self.first_graph if first else self.next_graph).replay()
                    c = be.ctrl_wait(handle)
                else:
                    (self._first if first else self._attempt)()
                    c = be.ctrl_read(s._ctrl)
                if graphs:
                    s.nfe += s._n_stage + (2 if first else 0)
                first = False
                s._raise_status(c)
                if c.done:
                    break
        s._last = c
        return s._solution[1]
