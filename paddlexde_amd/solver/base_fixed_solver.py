"""Fixed-grid solvers on the HIP combine kernel.

Reference: paddlexde/solver/base_fixed_solver.py:14-197.  The loop structure, the ``step`` protocol
``(t0, t1, y0) -> (y1, dy0)``, the option names and the output layout (``concat(axis=-2)``,
SURVEY D3) are the reference's; every ``fuse`` (``dy * dt + y0``, xde/base_ode.py:58) and every
stage formula is one xde_stage_combine launch instead of 2-9 eager element-wise ops.

Times: the reference slices ``time_grid[i-1:i]`` (shape ``[1]`` tensors) and derives stage times with
eager ops.  Here the whole table of stage times is computed once on the host, in the time dtype and
with the reference's op order, and uploaded once; ``func`` receives shape-``[1]`` device views of it.
"""
import abc
import ctypes as C
import threading

import numpy as np
import torch

from .. import _hip
from ..xde.base_dde import DDE_DAMPING, BaseDDE
from ..xde.base_ode import BaseODE
from ..xde.base_xde import BaseXDE
from ._autograd import CombineFn
from ._common import as_operand, np_dtype, storage_ptr, t_span_to_host, upload

_one_third = 1 / 3
_two_thirds = 2 / 3
_one_sixth = 1 / 6


class FixedSolver(metaclass=abc.ABCMeta):
    order: int

    graphable = True  # the step's control flow does not depend on data (False: AdamsBashforthMoulton)
    GRAPH_MIN_STEPS = 4
    AUTO_GRAPH_MIN_STEPS = 24  # pipeline="auto": steps needed to amortise a capture (~2 ms against ~100 us saved per step)
    AUTO_GRAPH_MAX_BYTES = 8 << 20  # ... and only for launch-bound (small) states

    def __init__(self, xde, y0, step_size=None, grid_constructor=None, interp="linear", perturb=False, pipeline="auto", **kwargs):
        self.xde = xde
        self.y0 = y0
        self.dtype = y0.dtype
        self.step_size = step_size
        self.interp = interp
        self.perturb = perturb
        if pipeline not in ("auto", "sync", "lag", "graph"):
            raise ValueError("pipeline must be 'auto', 'sync' or 'graph' ('lag' means 'sync' for a fixed grid)")
        self.pipeline = pipeline
        self._g_ctrls = None  # graph pipeline: device dt sources of the step's combines, in call order
        self._g_slot = 0
        self._rec = None  # recording pass: the dt every combine of a step receives, for all steps at once

        # base_fixed_solver.py:45-47 — KeyError when absent, as in the reference
        self.atol = kwargs["atol"]
        self.rtol = kwargs["rtol"]
        self.norm = kwargs["norm"]

        if step_size is not None and grid_constructor is not None:
            raise ValueError("step_size and grid_constructor are mutually exclusive arguments.")
        if step_size is not None or grid_constructor is not None:
            # In the reference the loop walks only len(t_span) grid points whatever the grid is
            # (base_fixed_solver.py:126-127), so sub-stepping never worked.
            raise NotImplementedError("step_size / grid_constructor sub-stepping is broken in the reference (SURVEY D7)")
        self.grid_constructor = lambda y0, t: t

        self.move = self.xde.move
        self.fuse = self.xde.fuse
        self.on_integrate_step_end = self.xde.on_integrate_step_end
        # The reference binds this hook (base_fixed_solver.py:64) and never calls it; here it IS called, once per step of the eager
        # loop, as `xde.on_integrate_step_end(y0, y1, t0, t1)` (state before / after the step, device tensors; t0 / t1 the shape-[1]
        # device views the step itself received).  The base classes' hook does nothing, so only a wrapper that overrides it sees a
        # difference — and such a wrapper keeps the solve on the eager loop: a captured step cannot call back into Python, so
        # pipeline="graph" refuses it and "auto" does not capture.
        self._step_end_hook = getattr(type(xde), "on_integrate_step_end", None) is not BaseXDE.on_integrate_step_end
        if self._step_end_hook and pipeline == "graph":
            raise NotImplementedError("pipeline='graph' replays a captured step and cannot call xde.on_integrate_step_end; "
                                      "use pipeline='sync' (or the default 'auto', which then keeps the eager loop)")
        # the wrapper's fuse is what xde_stage_combine computes: BaseODE's `dy*dt + y0` or BaseDDE's damped form
        fuse_impl = getattr(type(xde), "fuse", None)
        if fuse_impl is BaseODE.fuse:
            self._damping = 0.0
        elif fuse_impl is BaseDDE.fuse:
            self._damping = DDE_DAMPING
        else:
            raise NotImplementedError("only BaseODE.fuse / BaseDDE.fuse are mapped onto the HIP combine kernel")

        self.backend = _hip.get_backend()
        self.nfe = 0
        self._dt = None  # host dt (numpy scalar of the time dtype) while integrate() drives step()
        self._t0_host = None  # host t0 of the current step while integrate() drives step()
        self._row = None  # current row of the uploaded time table
        self._y1_out = None  # where the step's final combine should write (a slice of the output)
        self._tdev_cache = {}

    # -- framework call -----------------------------------------------------------------------
    def _f(self, t, dt, y):
        if self._rec is not None:
            return None
        self.nfe += 1
        f = self.move(t, dt, y)
        f = as_operand(f, like=y)
        if storage_ptr(f) == storage_ptr(y):
            f = f.clone()
        return f

    def _combine(self, y0, ks, coef, mode, dt, scale=1.0, out=None, emit=None):
        """One xde_stage_combine launch.  ``emit`` (a FUSE launch that is not being differentiated): the weights of the step's final
        sum over the operands this launch holds — it then also writes that partial sum and ``(out, partial)`` is returned."""
        if self._rec is not None:
            self._rec.append(dt)
            return None if emit is None else (None, None)
        damp = self._damping if mode != _hip.COMBINE_RK else 0.0
        part = torch.empty_like(y0) if emit is not None else None
        if self._g_ctrls is not None:  # graph pipeline: dt is read from device memory (rewritten before every replay)
            ctrl = self._g_ctrls[self._g_slot]
            self._g_slot += 1
            if out is None:
                out = torch.empty_like(y0)
            self.backend.stage_combine(out, y0, ks, coef, mode, scale=scale, ctrl=ctrl, damping=damp, out2=part, coef2=emit)
            return out if emit is None else (out, part)
        if emit is None and torch.is_grad_enabled() and (y0.requires_grad or any(k.requires_grad for k in ks)):
            # discretise-then-optimise: keep the autograd graph through the combine
            return CombineFn.apply(self.backend, list(coef), mode, scale, float(dt), damp, y0, *ks)
        if out is None:
            out = torch.empty_like(y0)
        self.backend.stage_combine(out, y0, ks, coef, mode, scale=scale, dt_host=float(dt), damping=damp, out2=part, coef2=emit)
        return out if emit is None else (out, part)

    def _combine_pre(self, y0, pre, ks, coef, dt, scale, out=None):
        """The final weighted sum with its leading terms pre-summed (xde_stage_combine_pre_weighted): reads y0, ``pre`` and the newest
        derivative(s)."""
        if self._rec is not None:
            self._rec.append(dt)
            return None
        if out is None:
            out = torch.empty_like(y0)
        ctrl = None
        if self._g_ctrls is not None:
            ctrl = self._g_ctrls[self._g_slot]
            self._g_slot += 1
        self.backend.stage_combine_pre_weighted(out, y0, pre, ks, coef, scale=scale, dt_host=0.0 if ctrl is not None else float(dt), ctrl=ctrl,
                                                damping=self._damping)
        return out

    def _presum_ok(self, y0, ks):
        """Whether the last stage-input launch may emit the final sum's leading terms: the backend offers it and nothing here is being
        differentiated (the autograd node of a combine has one output).  Same bits either way."""
        if not hasattr(self.backend, "stage_combine_pre_weighted"):
            return False
        if self._rec is not None:
            return True  # (recording pass: the same launches, in the same order, either way)
        return not (torch.is_grad_enabled() and (y0.requires_grad or any(k.requires_grad for k in ks)))

    # -- time handling ----------------------------------------------------------------------------
    def _host_dt(self, t0, t1):
        if self._dt is not None:
            return self._dt
        return np_dtype(t0.dtype)((t1 - t0).item())  # direct step() call outside integrate(): one device read

    @staticmethod
    def _time_values(dt):
        """Host scalars (time dtype) the step hands to ``move`` besides t0/t1, in the order ``_times`` returns them."""
        return ()

    def _times(self, t0, dt):
        vals = self._time_values(dt)
        if self._row is not None:
            return [self._row[j : j + 1] for j in range(len(vals))]
        return [self._tdev(v, t0) for v in vals]

    def _tdev(self, value, like):
        key = (float(value), like.dtype)
        t = self._tdev_cache.get(key)
        if t is None:
            t = torch.full((1,), float(value), dtype=like.dtype, device=like.device)
            if len(self._tdev_cache) < 1024:
                self._tdev_cache[key] = t
        return t

    @abc.abstractmethod
    def step(self, t0, t1, y0):
        """Propose a step from t0 to t1. Returns (y1, dy0)."""
        raise NotImplementedError

    # -- base_fixed_solver.py:103-144 ------------------------------------------------------------
    def integrate(self, t_span):
        if not torch.is_tensor(t_span):
            t_span = torch.as_tensor(t_span)
        y0 = self.y0
        self.backend.require_device(y0)
        if y0.dim() < 2:
            raise ValueError("fixed solvers concatenate on axis -2: y0 needs >= 2 dims (reference layout [..., L, D])")
        pred_len = len(t_span)
        t_dtype = t_span.dtype if t_span.dtype in (torch.float32, torch.float64) else torch.float32
        t_host = t_span_to_host(t_span, np_dtype(t_dtype))
        # (values rounded to the time dtype on the host, as t_span.astype would; no blocking pageable copy)
        t_dev = t_span.detach().to(device=y0.device, dtype=t_dtype) if t_span.is_cuda else upload(t_host, y0.device)

        # one upload: per step [t-like values the step passes to move()], computed for all steps at once (element-wise numpy
        # arithmetic in the time dtype gives the values the per-step scalar expressions give)
        table = None
        if pred_len > 1:
            dts = t_host[1:] - t_host[:-1]
            cols = [np.broadcast_to(t_host[:-1] + v if is_time else v, dts.shape) for v, is_time in self._time_values_tagged(dts)]
            if cols:
                table = upload(np.stack(cols, axis=1).astype(np_dtype(t_dtype)), y0.device)

        tracking = torch.is_grad_enabled() and y0.requires_grad
        y0 = as_operand(y0 if tracking else y0.detach())
        L, D = y0.shape[-2], y0.shape[-1]
        lead = y0.shape[:-2]
        out = torch.empty(*lead, pred_len * L, D, dtype=y0.dtype, device=y0.device)
        direct = (int(np.prod(lead)) == 1) if len(lead) else True  # output rows are contiguous slices
        out.narrow(-2, 0, L).copy_(y0)

        # pipeline="graph": one captured step replayed over the grid (no autograd, data-independent step).  "auto" (default)
        # takes it for inference-style calls (grad mode off) on small states with enough steps to pay for the capture, behind
        # the capture guard, and falls back to the eager loop below if the capture is refused or fails.
        can_graph = (self.graphable and not self._step_end_hook and not tracking and not torch.is_grad_enabled() and table is not None and y0.is_cuda
                     and self.interp != "cubic" and pred_len - 1 >= self.GRAPH_MIN_STEPS
                     and threading.current_thread() is threading.main_thread() and not torch.cuda.is_current_stream_capturing())
        if self.pipeline == "graph" and can_graph:
            return self._integrate_graph(t_host, t_dev, table, y0, out, L)
        if (self.pipeline == "auto" and can_graph and pred_len - 1 >= self.AUTO_GRAPH_MIN_STEPS
                and y0.numel() * y0.element_size() <= self.AUTO_GRAPH_MAX_BYTES):
            nfe0 = self.nfe
            try:
                return self._integrate_graph(t_host, t_dev, table, y0, out, L, guard=True)
            except Exception:  # refused by the guard, or func cannot be captured: eager loop, from the start
                self.nfe = nfe0
                self._g_ctrls = None

        try:
            for i in range(1, pred_len):
                t0, t1 = t_dev[i - 1 : i], t_dev[i : i + 1]
                self._dt = t_host[i] - t_host[i - 1]
                self._t0_host = t_host[i - 1]
                self._row = table[i - 1] if table is not None else None
                dst = out.narrow(-2, i * L, L)
                self._y1_out = dst.view(y0.shape) if (direct and dst.data_ptr() % 16 == 0) else None
                y1, dy0 = self.step(t0, t1, y0)
                if self.interp == "cubic":
                    # base_fixed_solver.py:133-137: an extra step(t1, t1, y1) supplies dy1; the Hermite cubic
                    # evaluated at t == t1 is y1 itself (h00 = h10 = h11 = 0, h01 = 1), so only the NFE matter.
                    self._dt = t_host[i] - t_host[i]
                    self._t0_host = t_host[i]
                    self._row = None
                    self._y1_out = None
                    self.step(t1, t1, y1)
                # "linear": linear_interp returns y1 when t == t1 (interp_fn.py:7-8); any other value: raw y1
                if y1.data_ptr() != dst.data_ptr():  # (differentiable copy when y1 carries an autograd graph)
                    dst.copy_(y1)
                if self._step_end_hook:
                    self.on_integrate_step_end(y0, y1, t0, t1)
                y0 = y1
        finally:
            self._dt = None
            self._t0_host = None
            self._row = None
            self._y1_out = None
        return out

    # -- hipGraph pipeline: one captured step, replayed over the grid ------------------------------------------------
    def _record_combine_dts(self, dts):
        """The ``dt`` argument of every ``_combine`` call of one step, in call order, as arrays over all steps: ``step`` is
        run once on the ARRAY of step sizes with ``func`` and the kernels switched off (its host arithmetic is element-wise
        numpy in the time dtype, so each entry is the scalar the eager loop would pass)."""
        self._rec, self._dt, self._t0_host = [], dts, None
        row = torch.zeros(max(len(self._time_values(dts[:1])), 1))
        self._row = row
        try:
            self.step(row[0:1], row[0:1], None)
            return [np.broadcast_to(np.asarray(v), dts.shape).astype(np.float64) for v in self._rec]
        finally:
            self._rec, self._dt, self._row = None, None, None

    def _integrate_graph(self, t_host, t_dev, table, y0, out, L, guard=False):
        """``options={"pipeline": "graph"}`` (no autograd, data-independent step): the first step runs eagerly, then ONE step
        — the combines, the framework ops of ``func``, the state hand-over — is captured into a hipGraph and replayed; per
        step the host rewrites the step's times and step sizes in device memory (two small copies) and stores the row.
        Same kernels, same operands: bit-identical to the eager loop.  For launch-latency-bound (small) states."""
        be = self.backend
        dev = y0.device
        n_steps = len(t_host) - 1
        dts = t_host[1:] - t_host[:-1]
        dt_table = upload(np.stack(self._record_combine_dts(dts), axis=1), dev)  # [n_steps, K] fp64
        K = dt_table.shape[1]
        nb = C.sizeof(_hip.XdeCtrl)
        ctrls = torch.zeros(K * nb, dtype=torch.uint8, device=dev)
        ctrl_dt = ctrls.view(torch.float64).view(K, nb // 8)[:, 2]  # the `dt` field of each control block
        ctrl_list = [ctrls[k * nb : (k + 1) * nb] for k in range(K)]
        times = torch.cat([t_dev[:-1, None], t_dev[1:, None], table], dim=1).contiguous()  # per step: t0, t1, stage times
        row = torch.empty(times.shape[1], dtype=times.dtype, device=dev)
        y_cur, y_next = y0.clone(), torch.empty_like(y0)

        def body():
            self._row, self._y1_out, self._g_ctrls, self._g_slot = row[2:], y_next, ctrl_list, 0
            self._dt, self._t0_host = dts[0], t_host[0]  # placeholders: every dt the kernels use comes from `ctrls`
            try:
                y1, _ = self.step(row[0:1], row[1:2], y_cur)
            finally:
                self._row = self._y1_out = self._g_ctrls = self._dt = self._t0_host = None
            if y1.data_ptr() != y_next.data_ptr():
                y_next.copy_(y1)
            y_cur.copy_(y_next)

        def load(i):
            row.copy_(times[i])
            ctrl_dt.copy_(dt_table[i])

        nfe0 = self.nfe
        load(0)
        if guard:  # "auto": a func that differentiates w.r.t. parameter leaves must never be captured (utils/graphed.py)
            from ..utils.graphed import _AutogradTargetProbe

            with _AutogradTargetProbe() as probe:
                body()
            if probe.hit is not None:
                raise RuntimeError("capture refused: func calls " + probe.hit)
        else:
            body()  # step 1, eagerly (warm-up of func and of the allocator)
        out.narrow(-2, L, L).copy_(y_cur)
        per_step = self.nfe - nfe0
        from ..utils.graphed import CapturedGraph

        g = CapturedGraph()  # (replays of a graph that holds memset nodes are synchronised: see its docstring)
        with g.capture(capture_error_mode="thread_local"):
            body()
        g.finish()
        self.nfe = nfe0 + per_step  # recording executes nothing
        for i in range(1, n_steps):
            load(i)
            g.replay()
            out.narrow(-2, (i + 1) * L, L).copy_(y_cur)
            self.nfe += per_step
        return out

    # -- re-armable one-step solves: odeint_adjoint's backward over a fixed grid -----------------------------------------
    # The backward sweep of a fixed-grid solve is one STEP per output interval, each from a fresh solver: with the captured
    # dynamics that was 2 input copies + 1 clone around every evaluation.  Here ONE solver is kept for the sweep (and the next
    # backward pass): the state lives in a static buffer, the step's times and step sizes go up in one host-to-device copy, and
    # the whole step — func's framework ops, the combines, the hand-over — is one graph replay.  Same kernels, same operands as
    # the eager step: bit-identical.
    _IV_SLOTS = 8

    def intervals_supported(self):
        return bool(self.graphable and not self._step_end_hook and self.interp != "cubic" and self.y0.is_cuda and self.y0.dim() >= 2
                    and self.pipeline in ("auto", "sync", "graph"))

    def _iv_host_rows(self, t_host):
        """(times row in the time dtype: t0, t1, the step's time values; the dt of every combine of the step, fp64) for one step."""
        dts = t_host[1:] - t_host[:-1]
        cols = [np.broadcast_to(t_host[:-1] + v if is_time else v, dts.shape) for v, is_time in self._time_values_tagged(dts)]
        row = np.concatenate([t_host[:1], t_host[1:]] + [np.asarray(c) for c in cols]).astype(t_host.dtype)
        return row, np.asarray([v[0] for v in self._record_combine_dts(dts)], dtype=np.float64)

    def intervals_prepare(self, t_span, t_dtype, capture=True):
        """Static buffers for one-step solves, one eager step over ``t_span`` (two host times) from the constructor's ``y0`` as
        warm-up, and — ``capture`` — the step's graph.  Main thread, outside autograd nodes (utils/graphed.py)."""
        from ..utils.graphed import CapturedGraph

        dev = self.y0.device
        self.backend.require_device(self.y0)
        y0 = as_operand(self.y0.detach())
        self._iv_tt = tt = np_dtype(t_dtype)
        row, dtrow = self._iv_host_rows(np.asarray([t_span[0], t_span[1]], dtype=tt))
        K, nb, item = len(dtrow), C.sizeof(_hip.XdeCtrl), np.dtype(tt).itemsize
        # one static device block: K control blocks (the combines read their `dt` field), then the step's times; its pinned mirror
        size = K * nb + len(row) * 8
        # (the host never waits for a step here, so the mirror is a ring: a slot is rewritten only after the copy that read it ran)
        self._iv_pinned = torch.zeros(self._IV_SLOTS, size, dtype=torch.uint8).pin_memory()
        self._iv_events = [None] * self._IV_SLOTS
        self._iv_slot = 0
        self._iv_block = torch.zeros(size, dtype=torch.uint8, device=dev)
        self._iv_ctrl_list = [self._iv_block[k * nb : (k + 1) * nb] for k in range(K)]
        self._iv_row = self._iv_block[K * nb : K * nb + len(row) * item].view(t_dtype)
        host = self._iv_pinned.numpy()
        self._iv_host_dt = [h[: K * nb].view(np.float64).reshape(K, nb // 8)[:, 2] for h in host]  # (the `dt` field of each control block)
        self._iv_host_row = [h[K * nb : K * nb + len(row) * item].view(tt) for h in host]
        self._iv_y = (y0.clone(), torch.empty_like(y0))
        self._iv_graph = None
        with torch.no_grad(), torch.autograd.set_multithreading_enabled(False):
            nfe0 = self.nfe
            self.interval_solve(t_span)  # eager: func's lazy initialisations, the allocator's blocks
            self._iv_per_step = self.nfe - nfe0
            if capture:
                torch.cuda.synchronize(dev)
                g = CapturedGraph()
                with g.capture(capture_error_mode="thread_local"):
                    self._iv_body()
                g.finish()
                self._iv_graph = g
            self.nfe = nfe0
        return self

    @property
    def interval_state(self):
        """The static state buffer a one-step solve starts from and leaves its result in."""
        return self._iv_y[0]

    def _iv_body(self):
        y_cur, y_next = self._iv_y
        row = self._iv_row
        self._row, self._y1_out, self._g_ctrls, self._g_slot = row[2:], y_next, self._iv_ctrl_list, 0
        self._dt, self._t0_host = self._iv_tt(1.0), self._iv_tt(0.0)  # placeholders: every dt the kernels use comes from the block
        try:
            y1, _ = self.step(row[0:1], row[1:2], y_cur)
        finally:
            self._row = self._y1_out = self._g_ctrls = self._dt = self._t0_host = None
        y_cur.copy_(y1)

    def interval_solve(self, t_span):
        """One step from ``t_span[0]`` to ``t_span[1]`` (host times) on ``interval_state``, in place; returns it."""
        row, dtrow = self._iv_host_rows(np.asarray([t_span[0], t_span[1]], dtype=self._iv_tt))
        i = self._iv_slot
        self._iv_slot = (i + 1) % self._IV_SLOTS
        if self._iv_events[i] is not None:
            self._iv_events[i].synchronize()
        self._iv_host_row[i][:] = row
        self._iv_host_dt[i][:] = dtrow
        self._iv_block.copy_(self._iv_pinned[i], non_blocking=True)
        if self._iv_events[i] is None:
            self._iv_events[i] = torch.cuda.Event()
        self._iv_events[i].record()
        with torch.no_grad():
            if self._iv_graph is not None:
                self._iv_graph.replay()
                self.nfe += self._iv_per_step
            else:
                self._iv_body()
        return self._iv_y[0]

    def _time_values_tagged(self, dt):
        """[(value, is_offset_from_t0)] matching ``_time_values``; default: every value is a plain dt-like."""
        return [(v, False) for v in self._time_values(dt)]

    # -- base_fixed_solver.py:146-164 (classical RK4; unused by the reference's RK4 class) ------------
    def rk4_step_func(self, t0, t1, y0, f0=None):
        dt = self._host_dt(t0, t1)
        half_dt = dt * 0.5
        t0h = self._t0_host if self._t0_host is not None else type(dt)(t0.item())
        dtt, hdt, t_half = self._tdev(dt, t0), self._tdev(half_dt, t0), self._tdev(t0h + half_dt, t0)
        k1 = f0
        if k1 is None:
            k1 = self._f(t0, dtt, y0)
        k2 = self._f(t_half, hdt, self._combine(y0, [k1], [1.0], _hip.COMBINE_FUSE, half_dt))
        k3 = self._f(t_half, hdt, self._combine(y0, [k2], [1.0], _hip.COMBINE_FUSE, half_dt))
        k4 = self._f(t1, hdt, self._combine(y0, [k3], [1.0], _hip.COMBINE_FUSE, dt))
        return self._combine(y0, [k1, k2, k3, k4], [1.0, 2.0, 2.0, 1.0], _hip.COMBINE_WFUSE, dt, scale=_one_sixth,
                             out=self._y1_out)

    # -- base_fixed_solver.py:166-197 ---------------------------------------------------------------
    def rk4_alt_step_func(self, t0, t1, y0, f0=None):
        """The reference's "3/8-rule" variant as written: stage-3 input is ``k1 - k2/3``."""
        dt = self._host_dt(t0, t1)
        dtt, d13, t_one_third, t_two_thirds = self._times(t0, dt)
        k1 = f0
        if k1 is None:
            k1 = self._f(t0, dtt, y0)
        k2 = self._f(t_one_third, d13, self._combine(y0, [k1], [1.0], _hip.COMBINE_FUSE, dt * _one_third))
        k3 = self._f(t_two_thirds, d13, self._combine(y0, [k1, k2], [1.0, -_one_third], _hip.COMBINE_FUSE, dt))
        if self._presum_ok(y0, [k1, k2, k3]):
            # the launch that forms k4's input holds k1..k3 anyway: it also emits `fuse(k1) + 3 fuse(k2) + 3 fuse(k3)` (left to right), and
            # the final launch reads y0, that partial sum and k4 — 18 N -> 17 N elements per step, same association, same bits
            y4, part = self._combine(y0, [k1, k2, k3], [1.0, -1.0, 1.0], _hip.COMBINE_FUSE, dt, emit=[1.0, 3.0, 3.0])
            k4 = self._f(t1, t_one_third, y4)
            if self._rec is not None or not (torch.is_grad_enabled() and k4.requires_grad):
                return self._combine_pre(y0, part, [k4], [1.0], dt, 0.125, out=self._y1_out)
        else:
            k4 = self._f(t1, t_one_third, self._combine(y0, [k1, k2, k3], [1.0, -1.0, 1.0], _hip.COMBINE_FUSE, dt))
        return self._combine(y0, [k1, k2, k3, k4], [1.0, 3.0, 3.0, 1.0], _hip.COMBINE_WFUSE, dt, scale=0.125,
                             out=self._y1_out)
