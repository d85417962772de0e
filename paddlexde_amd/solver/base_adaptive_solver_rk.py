"""Embedded Runge-Kutta stepper on the HIP kernels with a device-resident controller.

Reference: paddlexde/solver/base_adaptive_solver_rk.py:28-292 (+ utils/ode_utils.py:28-97).

What differs from the reference by design (results are the same):
  * the stage derivatives ``k_j`` stay the tensors ``func`` returned (SoA, zero copy) instead of being
    scattered into a stage-innermost ``k[..., S+1]`` buffer (:155-170);
  * each ``y_i = y0 + sum_j k_j (beta_ij dt)`` (:166-168) is ONE xde_stage_combine launch;
  * the error estimate, tolerance scaling, norm and ``isfinite`` pass (:180, :201, ode_utils.py:80-82) are ONE
    xde_error_norm_partial launch + the controller launch;
  * accept/reject, ``optimal_step_size``, clipping, stage times and output bookkeeping run on the device
    (xde_rk_control); ``dt`` lives in device memory, so enqueueing a step needs no host knowledge of it;
  * dense output is evaluated lazily, only for accepted steps that cover an output time (:262 refits on
    every accepted step), and the quartic's coefficients are never materialised;
  * reverse time runs natively with a signed ``dt`` (the reference's loop cannot).

Layout: this module is the STEPPER — construction, the framework call, one attempted step (``_attempt``), dense output, status
errors, the reference's public ``step(next_t)``; how the host drives attempts is one class per pipeline in ``_rk_pipelines.py``
(``options["pipeline"]``: "sync" | "lag" | "graph" | "auto"), the operand plans of a tableau are ``_rk_plans.py``, the re-armable
interval solves odeint_adjoint's backward runs on are ``_rk_intervals.py``.
"""
import bisect
import collections

import numpy as np
import torch

from .. import _hip
from ..utils.ode_utils import native_norm_spec
from ._common import as_operand, direction_of, np_dtype, scalar, scalar_const, storage_ptr, t_span_to_host, upload, upload_const
from ._rk_intervals import IntervalSolves
from ._rk_norms import NormReductions
from ._rk_pipelines import AutoPipeline, GraphPipeline, LagPipeline, SyncPipeline
from ._rk_plans import build_plans
from .base_adaptive_solver import AdaptiveSolver

_ButcherTableau = collections.namedtuple("_ButcherTableau", "alpha, beta, c_sol, c_error")

_RungeKuttaState = collections.namedtuple("_RungeKuttaState", "y1, f1, t0, t1, dt, interp_coeff")

_STATUS_MSG = {
    _hip.STATUS_DT_UNDERFLOW: "underflow in dt {}",
    _hip.STATUS_NONFINITE: "non-finite values in state `y`: {}",
    _hip.STATUS_MAX_STEPS: "max_num_steps exceeded ({}>={})",
}


_PLANS = {}  # solver class -> RKPlans (a function of the tableau only)


class AdaptiveRKSolver(NormReductions, AdaptiveSolver):
    order: int
    tableau: _ButcherTableau
    mid: list
    SINGLE_MAX_ELEMS = 1 << 16  # largest state (elements) served by the one-workgroup norm + controller / initial-step kernels

    def __init__(
        self,
        xde,
        y0,
        rtol,
        atol,
        min_step=0,
        max_step=float("inf"),
        first_step=None,
        step_t=None,
        jump_t=None,
        safety=0.9,
        ifactor=10.0,
        dfactor=0.2,
        max_num_steps=2**31 - 1,
        dtype=torch.float32,
        pipeline="auto",
        controller="I",
        pi_beta=0.04,
        process_group=None,
        norm_exchange=None,
        record_trace=False,
        _replay=None,
        _step_hook=None,
        reuse_f0=False,
        stats_out=None,
        callback_step=None,
        callback_accept_step=None,
        callback_reject_step=None,
        callback_accept=None,
        callback_reject=None,
        _xde_segments=None,
        _xde_segment_shapes=None,
        _short_solves=False,
        _fused_first_step=True,
        **kwargs,
    ):
        super().__init__(xde=xde, dtype=dtype, y0=y0, **kwargs)
        if jump_t is not None:
            raise NotImplementedError("jump_t calls a non-existent self.func in the reference (SURVEY D7)")
        if pipeline not in ("auto", "sync", "lag", "graph"):
            raise ValueError("pipeline must be 'auto', 'sync', 'lag' or 'graph'")
        if controller not in ("I", "PI"):
            raise ValueError("controller must be 'I' (reference) or 'PI' (opt-in)")
        if dtype not in (torch.float32, torch.float64):
            raise TypeError("dtype (time dtype) must be torch.float32 or torch.float64")
        if pipeline == "graph" and process_group is not None and not getattr(norm_exchange, "capturable", False):
            raise NotImplementedError("pipeline='graph' with a process_group needs the peer-to-peer norm exchange "
                                      "(norm_exchange=PeerExchange(...)): a torch.distributed all-reduce cannot be replayed "
                                      "from a captured step; or use 'sync' / 'lag'")
        tt = np_dtype(dtype)
        # base_adaptive_solver_rk.py:56-69: time-like scalars are tensors of `dtype`
        self.rtol = tt(rtol)
        self.atol = tt(atol)
        self.min_step = tt(min_step)
        self.max_step = tt(max_step)
        self.first_step = None if first_step is None else tt(first_step)
        self.safety = tt(safety)
        self.ifactor = tt(ifactor)
        self.dfactor = tt(dfactor)
        self.max_num_steps = int(max_num_steps)
        self.dtype = dtype
        self.step_t = None if step_t is None else np.asarray(torch.as_tensor(step_t).cpu().numpy(), dtype=tt)
        self.jump_t = None
        self.pipeline = pipeline
        self.controller = controller
        self.pi_beta = float(pi_beta)
        self.process_group = process_group
        # how the per-attempt norm sums travel between the ranks of `process_group`: None = torch.distributed.all_reduce
        # (RCCL), or a utils.PeerExchange (one-shot peer-to-peer stores into IPC-mapped mailboxes, rank-ordered sum)
        self.norm_exchange = norm_exchange
        if norm_exchange is not None and process_group is None:
            raise ValueError("norm_exchange needs a process_group (it replaces that group's all-reduce)")
        self.record_trace = bool(record_trace)
        self.trace = []  # (t0, dt, ratio, accept) per attempted step when record_trace is set
        # parity harness: a prescribed (dt, accept) sequence the device controller follows (xde_ctrl_params_t.replay)
        # and a callable(index, y0, y1, ks, ctrl) invoked after every attempt of the "sync" pipeline
        self._replay = None if _replay is None else [(float(h), bool(a)) for h, a in _replay]
        self._step_hook = _step_hook
        # The reference evaluates func(t0, y0) twice before the first attempt: once for the state (`_before_integrate`, :83) and
        # once more inside `select_initial_step` (f0=None, :84-87).  Same arguments, same value.  reuse_f0=True hands the first
        # result to the heuristic instead: func is called once less.  `nfe` / stats["nfe"] stay the number of calls func really
        # received (a func that counts its calls, or regularises on them, sees exactly that number); stats["nfe_reference"] is what
        # the reference would report for the same solve (+1 here).  Off by default; odeint_adjoint switches it on for its
        # backward intervals, where that evaluation is one of nine per interval.
        self._reuse_f0 = bool(reuse_f0)
        self._fused_first = bool(_fused_first_step)
        # odeint_adjoint's backward runs one solve per output interval, most of them a single attempted step long: the speculative
        # pipeline then waits for the verdict of a solve's FIRST attempt (whose step size the host never saw) before it enqueues a
        # second one, instead of discarding a whole attempt per interval
        self._short_solves = bool(_short_solves)
        # options["stats_out"] = {}: a dict of the caller's that receives the solve's counters (attempts, accepted, rejected, func
        # evaluations, final time and step) when it ends — `odeint()` returns the solution only, as the reference's does
        self._stats_out = stats_out
        if self._replay is not None and step_t is not None:
            raise NotImplementedError("a prescribed step sequence and step_t clipping do not combine")
        # Step callbacks.  The reference names them and leaves the calls commented out (`self.func.callback_step(t0, y0, dt)` at the top
        # of every attempt, `callback_accept_step` / `callback_reject_step` on its verdict, base_adaptive_solver_rk.py:186,259,275).
        # Here they are live: options["callback_step" | "callback_accept_step" | "callback_reject_step"] (short forms
        # "callback_accept" / "callback_reject"), or methods of those names on the user's func, are called as `cb(t0, y0, dt)` —
        # t0 and dt 0-dim HOST tensors of the time dtype, y0 the state the attempt starts from (a device tensor).  They need the
        # host to know every attempt's verdict before the next one is enqueued, i.e. the "sync" pipeline: "auto" resolves to it,
        # an explicit "lag" / "graph" refuses.
        func = getattr(xde, "func", None)
        pick = lambda opt, *names: opt if opt is not None else next(  # noqa: E731
            (getattr(func, n) for n in names if callable(getattr(func, n, None))), None)
        self._cb_step = pick(callback_step, "callback_step")
        self._cb_accept = pick(callback_accept_step if callback_accept_step is not None else callback_accept, "callback_accept_step")
        self._cb_reject = pick(callback_reject_step if callback_reject_step is not None else callback_reject, "callback_reject_step")
        self._has_callbacks = any(cb is not None for cb in (self._cb_step, self._cb_accept, self._cb_reject))
        if self._has_callbacks:
            if pipeline not in ("auto", "sync"):
                raise NotImplementedError("step callbacks are called by the host between attempts: pipeline='{}' enqueues attempts "
                                          "ahead of their verdicts (use 'sync', or the default 'auto')".format(pipeline))
            self.pipeline = pipeline = "sync"
        if _step_hook is not None:
            if pipeline not in ("auto", "sync"):
                raise NotImplementedError("_step_hook observes attempts of pipeline='sync'")
            self.pipeline = pipeline = "sync"
        # SMALL states (configs 3 and 5: launch-latency-bound): xde_error_norm_control runs as ONE workgroup that walks the segments
        # and goes straight on to the controller — no partial records, no tickets, one launch and one hipGraph node less per
        # attempt.  SINGLE_MAX_ELEMS = largest state (elements) served that way.  (At large sizes one ticketed launch for
        # norm + controller was measured no faster than two — 36.4 us vs 23.1 + 12 us on config 2 — and is not offered.)
        self._single_max = self.SINGLE_MAX_ELEMS

        self.backend = _hip.get_backend()
        self.nfe = 0  # calls func has received
        self._nfe_skipped = 0  # calls the reference would have made on top (reuse_f0)
        self.stats = {}

        # -- operand plans (a function of the tableau only: built once per solver class) ---------------
        plans = _PLANS.get(type(self))
        if plans is None:
            plans = _PLANS[type(self)] = build_plans(self.tableau, self.mid)
        (self._n_stage, self._stage_plan, self._fsal, self._sol_plan, self._err_plan, self._mid_plan, self._fuse_err,
         self._err2_coef, self._stage_nt, self._presum) = plans
        if not hasattr(self.backend, "stage_combine_pre"):
            self._presum = {}

        # -- how the host drives the attempts (one object per pipeline, all over this stepper), and the re-armable interval solves --
        self._sync = SyncPipeline(self)
        self._lag = LagPipeline(self)
        self._graph_pl = GraphPipeline(self, self._sync)
        self._auto = AutoPipeline(self, self._sync, self._lag, self._graph_pl)
        self._intervals = IntervalSolves(self)
        self._init_peek = None

        # -- segments / norm ---------------------------------------------------------------------
        n = self.y0.numel()
        segs = [(0, n)] if _xde_segments is None else [(int(s), int(l)) for s, l in _xde_segments]
        spec = native_norm_spec(self.norm)
        self._custom_norm = spec is None
        self._segs = segs
        self._seg_shapes = _xde_segment_shapes
        if spec is None:
            # an arbitrary callable: err/tol is materialised by one kernel (xde_error_ratio), the user's norm runs
            # as framework ops on it and its scalar feeds the device controller (as a 1-segment "linf" value)
            if not callable(self.norm):
                raise TypeError("options['norm'] must be callable")
            if process_group is not None:
                raise NotImplementedError("a user norm callable cannot be all-reduced over a process_group; use _rms_norm / "
                                          "_linf_norm (or, for odeint_adjoint, the default adjoint norm or \"seminorm\")")
            self._norm_kind = _hip.NORM_LINF
            self._norm_segs = [(0, n)]
            self._seg_count_local = [1.0]
        elif spec[0] in ("rms", "linf"):
            self._norm_kind = _hip.NORM_RMS if spec[0] == "rms" else _hip.NORM_LINF
            # a plain norm over a (padded) tuple state: pads are zero, so one segment over the whole
            # buffer with the true element count gives the same value
            self._norm_segs = [(0, n)]
            self._seg_count_local = [float(sum(l for _, l in segs))]
        elif spec[0] == "mixed":
            self._norm_kind = _hip.NORM_RMS
            k = len(segs) if spec[1] is None else min(int(spec[1]), len(segs))
            self._norm_segs = segs[:k]
            self._seg_count_local = [float(l) for _, l in self._norm_segs]
        else:
            raise ValueError("unknown native norm spec {!r}".format(spec))
        # One norm launch reduces up to XDE_MAX_SEG segments.  A tuple state with more of them (odeint_adjoint's default norm
        # has one segment per parameter tensor: a module with > 13 of them) is reduced in chunks of XDE_MAX_SEG segments —
        # one partial + finalize + result launch per chunk, each over its own segments only, so nothing is read twice — and
        # the chunk results are max-combined on the device; the controller then takes that scalar the way it takes a custom
        # norm's (a 1-segment "linf" value).  Same value as the single launch: max over segments of the per-segment RMS.
        self._small_state = (not self._custom_norm and len(self._norm_segs) <= _hip.XDE_MAX_SEG
                             and 0 < sum(l for _, l in self._norm_segs) <= self._single_max)
        self._chunks = None
        self._ctrl_norm_kind = self._norm_kind
        if len(self._norm_segs) > _hip.XDE_MAX_SEG:
            m = _hip.XDE_MAX_SEG
            self._chunks = [(i, _hip.make_segments(self._norm_segs[i : i + m])) for i in range(0, len(self._norm_segs), m)]
            self._xsegs = self._chunks[0][1]
            self._ctrl_norm_kind = _hip.NORM_LINF
        else:
            self._xsegs = _hip.make_segments(self._norm_segs)

    # ------------------------------------------------------------------------------------------
    # framework call
    # ------------------------------------------------------------------------------------------
    def _eval(self, t, y, live=()):
        """``xde.move`` -> func(t, y); returns a kernel-ready tensor that aliases nothing we still need (``live``: the
        storage pointers of the tensors still in use)."""
        self.nfe += 1
        if torch.is_grad_enabled():
            with torch.no_grad():  # the adaptive path is forward-only; gradients come from odeint_adjoint
                f = self.move(t, None, y)
        else:
            f = self.move(t, None, y)
        f = as_operand(f, like=y)
        sp = storage_ptr(f)
        if sp == storage_ptr(y) or sp in live:
            f = f.clone()
        return f

    def _scalar_t(self, value, dtype):
        return scalar(value, dtype, self.y0.device)

    # ------------------------------------------------------------------------------------------
    # base_adaptive_solver_rk.py:81-114
    # ------------------------------------------------------------------------------------------
    def _before_integrate(self, t_span):
        # The evaluations at the start time need nothing of the set-up below, so they are enqueued FIRST: the GPU works through them
        # while the host builds the solve's buffers, uploads and parameter block (a whole `odeint()` call at config 2 is host-bound
        # up to its first attempt: profiles/r06_odeint_tail.txt).  Same evaluations, same order as the reference's.
        tt = np_dtype(self.dtype)
        if not isinstance(t_span, np.ndarray) or t_span.dtype != tt:
            t_span = t_span_to_host(t_span, tt)
        self.y0 = y0 = as_operand(self.y0.detach())
        self.backend.require_device(y0)
        # f0 = move(t_span[0], t_span[1] - t_span[0], y0)                                          :83
        self._t0_dev = scalar_const(t_span[0], self.dtype, y0.device, self.backend._stream_of(y0.device) if hasattr(self.backend, "_stream_of") else None)
        f0 = self._eval(self._t0_dev, y0)
        f0_dup = None
        if self.first_step is None and not self._reuse_f0 and not self._custom_norm:
            # f0 is recomputed inside select_initial_step (f0=None), as in the reference         :84-87
            f0_dup = self._eval(self._t0_dev, y0, live=(storage_ptr(f0),))
        t_span = self._setup(t_span)
        be, p, d, y0 = self.backend, self._params, self._direction, self.y0
        first_dev = None
        self._ctrl_ready = False
        if self.first_step is None:
            f0_again = f0 if self._reuse_f0 else f0_dup
            if self._reuse_f0:
                self._nfe_skipped += 1  # (the call the reference makes here and this solve does not)
            if self._custom_norm:  # (a user's norm callable runs as framework ops: the heuristic's scalars go through the host)
                first_step = self.select_initial_step(t_span[0], y0, self.order - 1, self.rtol, self.atol, f0=f0_again)
            else:
                first_step, first_dev = None, self._select_initial_step_device(t_span[0], y0, f0=f0_again)
        else:
            first_step = self.first_step
        self.rk_state = _RungeKuttaState(y0, f0, t_span[0], t_span[0], first_step, None)
        if not self._ctrl_ready:  # (the fused initial step has constructed the control block already)
            be.ctrl_init(self._ctrl, p, float(t_span[0]), 0.0 if first_step is None else float(d * abs(first_step)), len(t_span),
                         self._t_span_dev, self._step_t_dev, self._t_stage, first_step_dev=first_dev)
        # The speculative pipeline wants to know where the FIRST attempt lands if it is accepted (`t_plan` of the block just
        # constructed — on the device when the heuristic chose the step): a copy of the block is enqueued here, behind the
        # heuristic's kernels and ahead of the first attempt's, and read when the second attempt is about to be enqueued
        self._init_peek = None
        if (hasattr(be, "ctrl_peek_async")
                and (self.pipeline == "lag" or (self.pipeline == "auto" and self._auto.pick() == "lag"))):
            # (the large-state heuristic's last launch has published that block to the host mirror already: nothing to enqueue)
            h = be.ctrl_init_handle(self._ctrl) if hasattr(be, "ctrl_init_handle") else None
            self._init_peek = h if h is not None else be.ctrl_peek_async(self._ctrl)

    def _setup(self, t_span):
        """Buffers, output times and controller parameters of a solve over ``t_span`` (everything before the first evaluation)."""
        be = self.backend
        tt = np_dtype(self.dtype)
        if not isinstance(t_span, np.ndarray) or t_span.dtype != tt:  # direct callers (the step() API) pass tensors
            t_span = t_span_to_host(t_span, tt)
        self.y0 = y0 = as_operand(self.y0.detach())
        be.require_device(y0)
        dev = y0.device
        self._base = None
        self._kept = None  # step() API: operands of the last accepted step
        self._auto_state = None  # what pipeline='auto' resolved to
        self._graph_pl.reset()
        self._direction = direction_of(t_span)
        self._t_host = t_span
        self._work = w = be.acquire_work(dev, y0.dtype)  # recycled by integrate() when the solve has ended
        # (read-only for every kernel; keyed by the work set's stream: the copy is ordered before this solve's launches)
        self._t_span_dev = upload_const(t_span.astype(np.float64), dev, getattr(w, "key", None))
        self._t_stage, self._ctrl, self._ws, self._sums = w.t_stage, w.ctrl, w.ws, w.sums
        self._t_views = getattr(w, "t_views", None) or [self._t_stage[i] for i in range(self._n_stage)]  # the 0-dim stage times handed to func
        self._seg_count = self._global_counts()
        self._csums = be.new_sums(dev) if self._chunks is not None else None  # the controller's input in chunked mode
        self._scratch = torch.empty_like(y0)
        # the partial error sum of the last stage goes into the stage scratch buffer: by then the previous stage's
        # input it held has been consumed by func (one buffer less in the step's working set)
        self._ebuf = self._scratch if self._fuse_err else None
        # the partial sum a stage's launch emits for the NEXT stage lives from that launch to the next one — across the func call that
        # reads the scratch buffer — so it has a buffer of its own
        self._sbuf = torch.empty_like(y0) if self._presum else None

        # step_t handling                                                                        :95-111
        d = self._direction
        if self.step_t is None:
            step_t = np.asarray([], dtype=tt)
        else:
            st = self.step_t
            st = st[d * st >= d * t_span[0]]
            step_t = np.sort(d * st) * d
        self._step_t_host = step_t
        self._step_t_dev = upload(step_t.astype(np.float64), dev) if len(step_t) else None
        self.next_step_index = min(bisect.bisect((d * step_t).tolist(), d * t_span[0]), len(step_t) - 1)

        p = _hip.XdeCtrlParams()
        p.rtol, p.atol = float(self.rtol), float(self.atol)
        p.min_step, p.max_step = float(self.min_step), float(self.max_step)
        p.safety, p.ifactor, p.dfactor = float(self.safety), float(self.ifactor), float(self.dfactor)
        p.order = float(self.order)
        p.max_num_steps = self.max_num_steps
        p.time_dtype = _hip.dtype_code(self.dtype)
        p.state_dtype = _hip.dtype_code(y0.dtype)
        p.direction = d
        p.norm_kind = self._ctrl_norm_kind
        p.n_stage = self._n_stage
        p.n_seg = 1 if self._chunks is not None else len(self._norm_segs)
        p.n_step_t = len(step_t)
        p.pi_controller = 1 if self.controller == "PI" else 0
        p.pi_beta = self.pi_beta
        for i, a in enumerate(self.tableau.alpha):
            p.alpha[i] = float(a)
        for i, c in enumerate([1.0] if self._chunks is not None else self._seg_count):
            p.seg_count[i] = float(c)
        self._replay_dev = None
        if self._replay:
            tab = np.asarray([[d * abs(h), 1.0 if a else 0.0] for h, a in self._replay], dtype=np.float64)
            self._replay_dev = upload(tab, dev)  # kept alive by the solver: params hold its raw pointer
            p.replay, p.n_replay = self._replay_dev.data_ptr(), len(self._replay)
        self._params = p
        return t_span

    def _after_integrate(self):
        w, self._work = getattr(self, "_work", None), None
        if w is not None and self.pipeline != "graph" and self._auto_state != "graph":  # a captured graph keeps addressing its buffers
            self.backend.release_work(w)

    def _fused_first_step(self):
        """One-workgroup initial step (xde_initial_step_fused): small state, native norm in one launch's worth of segments, one GPU,
        no prescribed step sequence.  (``_fused_first`` = False keeps the separate launches: same results, the parity tests' A/B.)"""
        return (self._small_state and self._chunks is None and self.process_group is None and not self._replay
                and hasattr(self.backend, "initial_step_fused") and self._fused_first)

    def _tail_first_step(self):
        """The heuristic of a state above the one-workgroup kernels' reach in 4 launches instead of 12 (xde_scaled_norm2_partial,
        xde_initial_step_tail): a native norm in one launch's worth of segments, one GPU (a sharded run exchanges the sums between
        finalize and result), no prescribed step sequence.  (``_fused_first`` = False keeps the separate launches: same results.)"""
        return (not self._custom_norm and self._chunks is None and self.process_group is None and not self._replay
                and hasattr(self.backend, "initial_step_tail") and self._fused_first)

    def _select_initial_step_device(self, t0, y0, f0=None):
        """``select_initial_step`` (base_adaptive_solver.py:33-72) with its scalar arithmetic on the device: the three
        norms feed two one-thread launches (xde_initial_step) instead of two blocking reads; the first step never visits
        the host.  Same op order and dtypes as the host version above it in the class hierarchy."""
        be = self.backend
        dev = y0.device
        sdt = _hip.dtype_code(y0.dtype)
        t0h = np_dtype(self.dtype)(t0)
        if f0 is None:
            # (the reference's second evaluation at the start time gets the start time's tensor of the first one: same value)
            t0_dev = getattr(self, "_t0_dev", None)
            f0 = self._eval(t0_dev if t0_dev is not None else self._scalar_t(t0h, self.dtype), y0)
        t_probe = torch.empty((), dtype=torch.promote_types(self.dtype, y0.dtype), device=dev)
        if self._fused_first_step():
            # small state: the three norms, the scalar arithmetic and the control block's construction in TWO one-workgroup launches
            # (+ the Euler probe's combine); _before_integrate skips its ctrl_init
            hs = torch.zeros(5, dtype=torch.float64, device=dev)
            be.initial_step_fused(0, f0, None, y0, self._xsegs, hs, self._params, float(t0h), t_probe, self._ctrl)
            y1 = torch.empty_like(y0)
            be.stage_combine(y1, y0, [f0], [1.0], _hip.COMBINE_FUSE, ctrl=self._ctrl)  # fuse(f0, h0, y0)
            f1 = self._eval(t_probe, y1)
            be.initial_step_fused(1, f1, f0, y0, self._xsegs, hs, self._params, float(t0h), None, self._ctrl, len(self._t_host),
                                  self._t_span_dev, self._step_t_dev, self._t_stage)
            self._first_step_dbg = (hs[4:5], hs)
            self._ctrl_ready = True
            return hs[3:4]
        if self._tail_first_step():
            # d0 and d1 in ONE pass over (y0, f0); finalize + result + the scalars (+ the control block's construction) folded into
            # one one-workgroup launch per phase: 12 launches -> 4; _before_integrate skips its ctrl_init
            hs = torch.empty(5, dtype=torch.float64, device=dev)  # (every entry is written by the two tail launches)
            be.scaled_norm2_partial(f0, y0, self.rtol, self.atol, self._xsegs, self._norm_kind, self._ws)
            be.initial_step_tail(0, self._ws, hs, self._params, float(t0h), t_probe, self._ctrl)
            y1 = torch.empty_like(y0)
            be.stage_combine(y1, y0, [f0], [1.0], _hip.COMBINE_FUSE, ctrl=self._ctrl)  # fuse(f0, h0, y0)
            f1 = self._eval(t_probe, y1)
            be.scaled_norm_partial(f1, f0, y0, float(self.rtol), float(self.atol), self._xsegs, self._norm_kind, self._ws, 0)
            be.initial_step_tail(1, self._ws, hs, self._params, float(t0h), None, self._ctrl, len(self._t_host), self._t_span_dev,
                                 self._step_t_dev, self._t_stage)
            self._first_step_dbg = (hs[4:5], hs)
            self._ctrl_ready = True
            return hs[3:4]
        res = torch.empty(2, dtype=torch.float64, device=dev)
        hs = torch.zeros(4, dtype=torch.float64, device=dev)

        def norm_into(a, b, out):
            self._scaled_norm_into(a, b, y0, self.rtol, self.atol, out)

        norm_into(y0, None, res[0:1])
        norm_into(f0, None, res[1:2])
        be.initial_step(0, res, hs, self._params, float(t0h), t_probe, self._ctrl)  # h0 -> ctrl.dt, t0 + h0 -> t_probe
        y1 = torch.empty_like(y0)
        be.stage_combine(y1, y0, [f0], [1.0], _hip.COMBINE_FUSE, ctrl=self._ctrl)  # fuse(f0, h0, y0)
        f1 = self._eval(t_probe, y1)
        norm_into(f1, f0, res[0:1])
        be.initial_step(1, res, hs, self._params, float(t0h), None, self._ctrl)
        self._first_step_dbg = (res, hs)  # (third norm, [d0, d1, h0, first step]): read by the kernel-level parity tests only
        return hs[3:4]

    # ------------------------------------------------------------------------------------------
    # one attempted step = _runge_kutta_step (:129-181) + error ratio + controller (:183-284)
    # ------------------------------------------------------------------------------------------
    def _attempt(self, base, alt=None):
        be = self.backend
        ctrl = self._ctrl
        y0, f0 = base
        y0_alt, k0_alt = alt if alt is not None else (None, None)
        ks = [f0]
        S = self._n_stage
        y_stage = None
        live = {storage_ptr(x) for x in ([y0, f0] + ([y0_alt, k0_alt] if alt is not None else []))}
        fuse = self._fuse_err and not self._custom_norm
        for i in range(S):
            idx, coef = self._stage_plan[i]
            out = torch.empty_like(y0) if (i == S - 1 and self._fsal) else self._scratch
            pre = self._presum.get(i)
            emit = self._presum.get(i + 1)
            if i == S - 1 and fuse:
                be.stage_combine(out, y0, [ks[j] for j in idx], coef, _hip.COMBINE_RK, ctrl=ctrl, y0_alt=y0_alt, k0_alt=k0_alt,
                                 out2=self._ebuf, coef2=self._err2_coef, nt_mask=self._stage_nt[i])
            elif pre is not None:  # y0 + ((partial sum of the previous launch) + k_i (beta_ii dt)): 3 arrays in
                be.stage_combine_pre(out, y0, self._sbuf, [ks[j] for j in pre[1]], pre[2], ctrl=ctrl, y0_alt=y0_alt,
                                     nt_mask=self._stage_nt[i] if fuse else 0)
            elif emit is not None:  # this launch also writes the next stage's partial sum from the operands it holds
                be.stage_combine(out, y0, [ks[j] for j in idx], coef, _hip.COMBINE_RK, ctrl=ctrl, y0_alt=y0_alt, k0_alt=k0_alt,
                                 out2=self._sbuf, coef2=emit[0], nt_mask=self._stage_nt[i] if fuse else 0)
            else:
                be.stage_combine(out, y0, [ks[j] for j in idx], coef, _hip.COMBINE_RK, ctrl=ctrl, y0_alt=y0_alt, k0_alt=k0_alt,
                                 nt_mask=self._stage_nt[i] if fuse else 0)
            ks.append(self._eval(self._t_views[i], out, live=live))
            live.add(storage_ptr(ks[-1]))
            y_stage = out
        if self._fsal:
            y1 = y_stage
        else:
            idx, coef = self._sol_plan
            y1 = torch.empty_like(y0)
            be.stage_combine(y1, y0, [ks[j] for j in idx], coef, _hip.COMBINE_RK, ctrl=ctrl, y0_alt=y0_alt, k0_alt=k0_alt)
        idx, coef = self._err_plan
        if self._custom_norm:
            # err/tol materialised once (with the speculative pipeline's operand select and the non-finite count of y0 riding
            # along), the user's callable on it as framework ops, its scalar into the controller: nothing here needs the host,
            # so a norm callable that stays on the device runs under "lag" and "graph" too
            r = torch.empty_like(y0)
            self._sums.zero_()
            m = _hip.XDE_MAX_SEG
            be.error_ratio(r, [ks[j] for j in idx], coef, y0, y1, float(self.rtol), float(self.atol), ctrl=ctrl, y0_alt=y0_alt,
                           k0_alt=k0_alt, nonfinite_out=self._sums[m : m + 1])
            self._sums[0:1].copy_(self._user_norm(r).reshape(1))
            be.rk_control(ctrl, self._params, None, self._sums, self._t_span_dev, self._step_t_dev, self._t_stage)
            return y1, ks
        if self._chunks is not None:
            def partial(xs):
                if fuse:
                    be.error_norm_partial([ks[-1]], [coef[-1]], y0, y1, float(self.rtol), float(self.atol), xs, self._norm_kind,
                                          self._ws, ctrl=ctrl, y0_alt=y0_alt, e_pre=self._ebuf)
                else:
                    be.error_norm_partial([ks[j] for j in idx], coef, y0, y1, float(self.rtol), float(self.atol), xs,
                                          self._norm_kind, self._ws, ctrl=ctrl, y0_alt=y0_alt, k0_alt=k0_alt)

            m = _hip.XDE_MAX_SEG
            self._reduce_chunks(partial, self._csums[0:1], nonfinite_out=self._csums[m : m + 1])
            be.rk_control(ctrl, self._params, None, self._csums, self._t_span_dev, self._step_t_dev, self._t_stage)
            return y1, ks
        if self.process_group is None and self._small_state:
            # single GPU, small state: error norm + controller in ONE one-workgroup launch
            if fuse:
                be.error_norm_control([ks[-1]], [coef[-1]], y0, y1, self._xsegs, self._ws, ctrl, self._params, self._t_span_dev,
                                      self._step_t_dev, self._t_stage, y0_alt=y0_alt, e_pre=self._ebuf)
            else:
                be.error_norm_control([ks[j] for j in idx], coef, y0, y1, self._xsegs, self._ws, ctrl, self._params,
                                      self._t_span_dev, self._step_t_dev, self._t_stage, y0_alt=y0_alt, k0_alt=k0_alt)
            return y1, ks
        if fuse:
            be.error_norm_partial([ks[-1]], [coef[-1]], y0, y1, float(self.rtol), float(self.atol), self._xsegs,
                                  self._norm_kind, self._ws, ctrl=ctrl, y0_alt=y0_alt, e_pre=self._ebuf)
        else:
            be.error_norm_partial([ks[j] for j in idx], coef, y0, y1, float(self.rtol), float(self.atol), self._xsegs,
                                  self._norm_kind, self._ws, ctrl=ctrl, y0_alt=y0_alt, k0_alt=k0_alt)
        if self.process_group is None:
            be.rk_control(ctrl, self._params, self._ws, None, self._t_span_dev, self._step_t_dev, self._t_stage)
        elif getattr(self.norm_exchange, "fused_control", False) and y0.is_cuda:
            # peer-to-peer exchange: partial records -> per-segment sums -> mailboxes -> controller, ONE launch
            be.p2p_rk_control(ctrl, self._params, self._ws, self.norm_exchange, self._t_span_dev, self._step_t_dev, self._t_stage)
        else:
            be.norm_finalize(self._ws, 0, self._sums)
            self._allreduce_sums(self._sums)
            be.rk_control(ctrl, self._params, None, self._sums, self._t_span_dev, self._step_t_dev, self._t_stage)
        return y1, ks

    def _dense(self, solution, base, y1, ks, alt=None, expect_step=-1):
        idx, coef = self._mid_plan
        y0_alt, k0_alt = alt if alt is not None else (None, None)
        self.backend.dense_eval(solution, [ks[j] for j in idx], coef, base[0], y1, ks[-1], self._ctrl, self._t_span_dev,
                                _hip.dtype_code(self.dtype), y0_alt=y0_alt, k0_alt=k0_alt, expect_step=expect_step)

    def _raise_status(self, c):
        if c.status == _hip.STATUS_OK:
            return
        if c.status == _hip.STATUS_DT_UNDERFLOW:
            raise AssertionError(_STATUS_MSG[c.status].format(c.dt))
        if c.status == _hip.STATUS_NONFINITE:
            if self.norm_exchange is not None and self.norm_exchange.error():
                q, by = self.norm_exchange.error_info()
                raise _hip.XdeError("peer-to-peer norm exchange {} timed out{}: a rank of the process group did not arrive "
                                    "(died, or fell out of lock-step); every rank of the group stops".format(
                                        q, "" if by is None else " on rank {}, which told this rank".format(by)))
            comm_error = getattr(self.norm_exchange, "async_error", lambda: None)()
            if comm_error:  # a failed collective leaves garbage in the sums: report the transport, not the state
                raise _hip.XdeError("the norm exchange's RCCL communicator reports: {}".format(comm_error))
            raise AssertionError(_STATUS_MSG[c.status].format("{} non-finite element(s)".format(int(c.nonfinite))))
        if c.status == _hip.STATUS_MAX_STEPS:
            raise AssertionError(_STATUS_MSG[c.status].format(c.steps_in_interval, self.max_num_steps))
        raise AssertionError("solver status {}".format(c.status))

    def _finish(self, c, base):
        self.stats = {
            "n_steps": int(c.n_steps),
            "n_accept": int(c.n_accept),
            "n_reject": int(c.n_reject),
            "nfe": int(self.nfe),
            "nfe_reference": int(self.nfe + self._nfe_skipped),
            "t": float(c.t1),
            "dt_next": float(c.dt),
        }
        if self._stats_out is not None:
            self._stats_out.update(self.stats)
        self.rk_state = _RungeKuttaState(base[0], base[1], c.t0, c.t1, c.dt, None)

    def _run(self, solution):
        self._solution = solution
        self._begin()
        self.advance(None)
        self._solution = None
        self._graph_pl.reset()  # releases the captured graphs and their private memory pools

    def _begin(self):
        """State of a solve that has not made an attempt yet (after ``_before_integrate``)."""
        self._base = (self.rk_state.y1, self.rk_state.f1)
        self._pending = None  # lag pipeline: (y1, [f1], read handle) of the newest, unresolved attempt
        self._n_attempts = 0
        self._last = None
        self._graph_pl.reset()

    # (names other modules know these by: bench.py's settle loop, the tests' policy assertions)
    GRAPH_ATTEMPTS = GraphPipeline.ATTEMPTS
    AUTO_GRAPH_MAX_BYTES = AutoPipeline.GRAPH_MAX_BYTES
    AUTO_GRAPH_AFTER = AutoPipeline.GRAPH_AFTER

    def advance(self, max_attempts=None):
        """Run attempted steps until every output is produced (``max_attempts=None``) or exactly
        ``max_attempts`` attempts were made.  Returns the last control block read from the device.

        Callable after ``_before_integrate`` as a stepping API (bench.py drives it with a fixed count)."""
        if getattr(self, "_base", None) is None:
            self._solution = None
            self._begin()
        pipeline = {"auto": self._auto, "lag": self._lag, "graph": self._graph_pl, "sync": self._sync}[self.pipeline]
        with torch.no_grad():  # once per advance, not once per func evaluation
            c = pipeline.advance(max_attempts)
        self._finish(c, self._base)
        return c

    # -- re-armable interval solves (odeint_adjoint's backward sweep): see _rk_intervals.py ---------------------------------------
    def intervals_supported(self):
        return self._intervals.supported()

    def intervals_prepare(self, t_span, capture=True):
        return self._intervals.prepare(t_span, capture=capture)

    def interval_solve(self, t_span):
        return self._intervals.solve(t_span)

    @property
    def interval_state(self):
        return self._intervals.state

    # base_adaptive_solver_rk.py:116-127
    def step(self, next_t):
        """Advance the solve until an accepted step reaches ``next_t`` and return the dense-output value there — the
        reference's public per-solver method (``while next_t > rk_state.t1: _adaptive_step``; then ``interp_evaluate``).
        Call after ``_before_integrate(t_span)``; ``integrate`` is this loop with the output bookkeeping on the device
        for ALL rows at once, and a manual ``step(t_i)`` loop gives the same rows bit for bit.  ``next_t`` may be any time
        inside or after the last accepted step (several calls can land in one step: no new attempt is made for them).
        Host-driven by nature: attempts are resolved one at a time whatever ``pipeline`` says."""
        be = self.backend
        if getattr(self, "_base", None) is None:
            self._solution = None
            self._begin()
        y0 = self.y0
        d = self._direction
        tt = np_dtype(self.dtype)
        nt = tt(float(next_t))
        # the controller and the dense-output kernel now look at this one-entry output list
        self._t_span_dev = t_dev = upload(np.asarray([nt], dtype=np.float64), y0.device)
        be.ctrl_retarget(self._ctrl, self._params, t_dev, 1)
        c = self._last = be.ctrl_read(self._ctrl)
        row = torch.empty((1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
        if c.out_end > c.out_begin:
            # already covered by the retained step: interpolate there                              :127
            # (ode_utils.py:65-67: the interpolant is valid on [t0, t1] of that step only)
            if not d * tt(c.t0) <= d * nt:
                raise AssertionError("invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}".format(c.t0, nt, c.t1))
            self._dense(row, *self._kept)
        else:
            if not d * nt > d * tt(c.t1):
                if c.n_accept != 0:
                    raise AssertionError("invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}".format(c.t0, nt, c.t1))
                # nothing stepped yet and next_t is the start time: the reference evaluates its initial interpolant
                # ([y0]*5, base_adaptive_solver_rk.py:91) on the empty interval; the value there is y0
                row[0] = y0
                return row[0]
            while True:
                base, y1, ks, c = self._sync.attempt_and_resolve()
                if c.accept:
                    self._kept = (base, y1, ks)
                    self._base = (y1, ks[-1])
                self._raise_status(c)
                if c.accept and c.out_end > c.out_begin:
                    self._dense(row, base, y1, ks)
                    break
        self._last = c
        self._finish(c, self._base)
        return row[0]
