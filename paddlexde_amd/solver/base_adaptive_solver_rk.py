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

Pipelines (``options["pipeline"]``):
  "sync"  one 256-byte device->host read per attempted step (default; NFE identical to the reference);
  "lag"   speculative: attempt n+1 is enqueued before the host knows whether attempt n was accepted; its
          kernels pick (y0, f0) between the two candidates from ctrl->accept on the device.  The GPU never
          waits for the host.  One extra (discarded) attempt runs after the last output.
  "graph" one whole attempted step captured into a hipGraph and replayed (launch-bound small states).
  "auto"  (default) picks among them per solve: "lag" when an operand is larger than
          AUTO_GRAPH_MAX_BYTES (the step is bandwidth-bound; also with a process_group); otherwise it starts in "sync" and,
          if the solve is still running after AUTO_GRAPH_AFTER attempts, captures the step and continues as "graph" —
          provided the capture is safe (main thread, no capture in progress, func does not differentiate with respect to
          parameter leaves) and succeeds; else it stays in "sync".  Results are bit-identical whatever is picked.
"""
import bisect
import collections
import os

import numpy as np
import torch

from .. import _hip
from ..utils.ode_utils import native_norm_spec
from ._common import as_operand, direction_of, np_dtype, scalar, storage_ptr, t_span_to_host, upload
from .base_adaptive_solver import AdaptiveSolver

_ButcherTableau = collections.namedtuple("_ButcherTableau", "alpha, beta, c_sol, c_error")

_RungeKuttaState = collections.namedtuple("_RungeKuttaState", "y1, f1, t0, t1, dt, interp_coeff")

_STATUS_MSG = {
    _hip.STATUS_DT_UNDERFLOW: "underflow in dt {}",
    _hip.STATUS_NONFINITE: "non-finite values in state `y`: {}",
    _hip.STATUS_MAX_STEPS: "max_num_steps exceeded ({}>={})",
}


def _nz_plan(coefs, upto=None):
    """Operand indices with non-zero coefficient; index 0 (f0, the select-able operand) always first."""
    n = len(coefs) if upto is None else upto
    idx = [0] + [j for j in range(1, n) if float(coefs[j]) != 0.0]
    return idx, _hip.dbl_array([float(coefs[j]) for j in idx])  # marshalled once: the C double[] the kernels take


_PLANS = {}


def _build_plans(tab, mid):
    """Operand plans of a tableau: only non-zero entries are read."""
    n_stage = len(tab.alpha)
    stage_plan = [_nz_plan(beta, upto=i + 1) for i, beta in enumerate(tab.beta)]
    c_sol = [float(c) for c in tab.c_sol]
    last_beta = [float(b) for b in tab.beta[-1]]
    # :172-176 "This property (true for Dormand-Prince) lets us save a few FLOPs."
    fsal = c_sol[-1] == 0 and c_sol[:-1] == last_beta
    sol_plan = _nz_plan(c_sol)
    err_plan = _nz_plan(tab.c_error)
    mid_plan = _nz_plan(mid)
    # Error-estimate fusion: if the last stage (whose output is y1, FSAL) loads every operand the error estimate
    # needs except the last derivative, it emits the partial sum as a second output and the error-norm kernel
    # reads {e_partial, k_last, y0, y1} instead of all the k's (Dopri5: 8N -> 4N + 1N written).
    last_idx = stage_plan[-1][0]
    err_idx = err_plan[0]
    S = n_stage
    fuse_err = fsal and err_idx[-1] == S and set(err_idx[:-1]) <= set(last_idx) and float(tab.c_error[S]) != 0.0
    err2_coef = _hip.dbl_array([float(tab.c_error[j]) for j in last_idx]) if fuse_err else None
    # Pre-summed stages (round 4): the launch of stage i-1 holds k_0..k_{i-1} in registers anyway, so it can emit
    # `sum_j k_j (beta_ij dt)` over them as a second output (one array written); stage i then reads y0, that partial sum and its newest
    # derivative k_i — 3 arrays instead of len(idx_i) + 1 — with the same left-to-right association, i.e. the same bits.  Worth it
    # from 4 operands on; a pre-summed stage cannot emit for the next one (it no longer holds the old derivatives), and the last stage of
    # an FSAL pair stays full (it emits the partial error estimate from all its operands).  Dopri5: stage 5 <- stage 4, 32 N -> 30 N
    # elements per step through the stage combines.  XDE_PRESUM=0 switches it off (same results; measured side by side).
    presum = {}
    if os.environ.get("XDE_PRESUM", "1") != "0":
        i = S - 2 if fuse_err else S - 1
        while i >= 1:
            idx_i, idx_p = stage_plan[i][0], stage_plan[i - 1][0]
            # (exactly the previous launch's operands: an operand stage i does not use would enter the emitted sum as `k_j * 0`, which
            # is NaN for a non-finite k_j the full stage never reads — ADVICE r04; every tableau shipped here has equal sets)
            if idx_i[-1] == i and len(idx_i) >= 4 and idx_i[:-1] == idx_p:
                emit = _hip.dbl_array([float(tab.beta[i][j]) for j in idx_p])
                presum[i] = (emit, [i], _hip.dbl_array([float(tab.beta[i][i])]))
                i -= 2
            else:
                i -= 1
    # what each stage's launch really READS (positions = bits of its nt mask)
    read_plan = [([i] if i in presum else list(stage_plan[i][0])) for i in range(n_stage)]
    # cache-policy hint per stage: bit p set = operand p of that stage's list is read there for the last time in an
    # accepted step (later readers: stages, the unfused error estimate; dense output is rare and lazy)
    last_use = {}
    for i, idx_i in enumerate(read_plan):
        for j in idx_i:
            last_use[j] = i
    later = set(sol_plan[0]) if not fsal else set()
    if not fuse_err:
        later |= set(err_plan[0])
    # XDE_STAGE_NT_MODE (measurement knob, results never depend on it): "lastuse" (default) as described; "old" = every operand
    # but the newest derivative (which the framework's GEMM has just written) is streamed, y0 included (bit 31); "all"; "none"
    mode = os.environ.get("XDE_STAGE_NT_MODE", "lastuse")
    stage_nt = []
    for i, idx_i in enumerate(read_plan):
        m = 0
        for pos, j in enumerate(idx_i):
            if mode == "lastuse":
                hit = last_use[j] == i and j not in later
            elif mode == "old":
                hit = pos != len(idx_i) - 1 or (last_use[j] == i and j not in later)
            elif mode == "oldk":
                hit = pos != len(idx_i) - 1 or (last_use[j] == i and j not in later)
            else:
                hit = mode == "all"
            if hit:
                m |= 1 << pos
        if mode in ("old", "all"):
            m |= 1 << 31
        stage_nt.append(m)
    return n_stage, stage_plan, fsal, sol_plan, err_plan, mid_plan, fuse_err, err2_coef, stage_nt, presum


class AdaptiveRKSolver(AdaptiveSolver):
    order: int
    tableau: _ButcherTableau
    mid: list

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
        # odeint_adjoint's backward runs one solve per output interval, most of them a single attempted step long: the speculative
        # pipeline then waits for the verdict of a solve's FIRST attempt (whose step size the host never saw) before it enqueues a
        # second one, instead of discarding a whole attempt per interval
        self._short_solves = bool(_short_solves) and os.environ.get("XDE_SHORT_SOLVES", "1") != "0"  # (0: for measuring it)
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
        # XDE_FUSE_CONTROL=1: error norm + controller as ONE launch (xde_error_norm_control, last-workgroup-done).
        # Bit-identical, but measured no faster than two launches (the controller's latency chain just moves into
        # the tail of the norm kernel: 36.4 us vs 23.1 + 12 us on config 2), so it is off by default.
        self._fuse_control = os.environ.get("XDE_FUSE_CONTROL", "0") == "1"
        # ... except for SMALL states (configs 3 and 5: launch-latency-bound): there xde_error_norm_control runs as ONE workgroup
        # that walks the segments and goes straight on to the controller — no partial records, no tickets, one launch and
        # one hipGraph node less per attempt.  XDE_SINGLE_ELEMS = largest state (elements) served that way; 0 = off.
        self._single_max = int(os.environ.get("XDE_SINGLE_ELEMS", str(1 << 16)))
        # the initial-step heuristic's scalar arithmetic runs on the device (xde_initial_step; no host read before the first
        # attempt).  XDE_HOST_FIRST_STEP=1 takes the host version (select_initial_step), which a custom norm always does.
        self._device_first_step = os.environ.get("XDE_HOST_FIRST_STEP", "0") != "1"

        self.backend = _hip.get_backend()
        self.nfe = 0  # calls func has received
        self._nfe_skipped = 0  # calls the reference would have made on top (reuse_f0)
        self.stats = {}

        # -- operand plans (a function of the tableau only: built once per solver class) ---------------
        plans = _PLANS.get(type(self))
        if plans is None:
            plans = _PLANS[type(self)] = _build_plans(self.tableau, self.mid)
        (self._n_stage, self._stage_plan, self._fsal, self._sol_plan, self._err_plan, self._mid_plan, self._fuse_err,
         self._err2_coef, self._stage_nt, self._presum) = plans
        if not hasattr(self.backend, "stage_combine_pre"):
            self._presum = {}

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
    # reductions (optionally all-reduced across the batch-sharding process group)
    # ------------------------------------------------------------------------------------------
    def _allreduce_sums(self, sums):
        if self.process_group is None:
            return
        if self.norm_exchange is not None and sums.is_cuda:
            self.norm_exchange.exchange(sums, self._norm_kind)
            return
        import torch.distributed as dist

        group = None if self.process_group is True else self.process_group
        buf = sums
        staged = sums.is_cuda and dist.get_backend(group) == "gloo"
        if staged:  # rehearsal transport (several ranks on one GPU): gloo reduces on the host
            buf = sums.cpu()
        if self._norm_kind == _hip.NORM_RMS:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(buf[: _hip.XDE_MAX_SEG], op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(buf[_hip.XDE_MAX_SEG :], op=dist.ReduceOp.SUM, group=group)
        if staged:
            sums.copy_(buf)

    def _global_counts(self):
        counts = list(self._seg_count_local)
        if self.process_group is not None:
            import torch.distributed as dist

            group = None if self.process_group is True else self.process_group
            cdev = "cpu" if dist.get_backend(group) == "gloo" else self.y0.device
            c = torch.tensor(counts, dtype=torch.float64, device=cdev)
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
            counts = c.tolist()
        return counts

    def _user_norm(self, flat):
        """Apply a user-supplied norm callable to a flat state-like tensor (tuple state: to the tuple of views)."""
        if self._seg_shapes is not None:
            arg = tuple(flat[s : s + l].view(shape) for (s, l), shape in zip(self._segs, self._seg_shapes))
        else:
            arg = flat.view(self.y0.shape)
        v = self.norm(arg)
        v = v if torch.is_tensor(v) else torch.as_tensor(float(v))
        return v.detach().abs().to(device=flat.device, dtype=torch.float64).reshape(())

    def _scaled_norms(self, pairs, y0, rtol, atol):
        """norm(a / scale) or norm((a - b) / scale) for each (a, b) pair; one host read for all of them."""
        if self._custom_norm:
            scale = float(atol) + y0.abs() * float(rtol)
            vals = [self._user_norm(((a - b) if b is not None else a) / scale) for a, b in pairs]
            return torch.stack(vals).tolist()
        res = torch.empty(len(pairs), dtype=torch.float64, device=y0.device)
        for i, (a, b) in enumerate(pairs):
            self._scaled_norm_into(a, b, y0, rtol, atol, res[i : i + 1])
        return res.tolist()

    def _reduce_chunks(self, launch_partial, out, nonfinite_out=None):
        """Run one norm over all segments: ``launch_partial(xsegs)`` enqueues the partial kernel for one chunk of segments;
        the scalar norm (max over every segment) lands in ``out`` (device double[1])."""
        be = self.backend
        sdt = _hip.dtype_code(self.y0.dtype)
        m = _hip.XDE_MAX_SEG
        if self._chunks is None:
            launch_partial(self._xsegs)
            be.norm_finalize(self._ws, 0, self._sums)
            self._allreduce_sums(self._sums)
            be.norm_result(self._sums, self._seg_count, self._norm_kind, sdt, out)
            return
        cres = torch.empty(len(self._chunks), dtype=torch.float64, device=self.y0.device)
        nf = None
        for ci, (first, xs) in enumerate(self._chunks):
            launch_partial(xs)
            be.norm_finalize(self._ws, 0, self._sums)
            self._allreduce_sums(self._sums)
            be.norm_result(self._sums, self._seg_count[first : first + xs.n_seg], self._norm_kind, sdt, cres[ci : ci + 1])
            if nonfinite_out is not None:
                part = self._sums[m : m + xs.n_seg].sum()
                nf = part if nf is None else nf + part
        out.copy_(cres.max().reshape(out.shape))  # torch.max propagates NaN, like the kernels' max over segments
        if nonfinite_out is not None:
            nonfinite_out.copy_(nf.reshape(nonfinite_out.shape))

    def _scaled_norm_into(self, a, b, y0, rtol, atol, out):
        be = self.backend
        self._reduce_chunks(
            lambda xs: be.scaled_norm_partial(a, b, y0, float(rtol), float(atol), xs, self._norm_kind, self._ws, 0), out)

    # ------------------------------------------------------------------------------------------
    # base_adaptive_solver_rk.py:81-114
    # ------------------------------------------------------------------------------------------
    def _before_integrate(self, t_span):
        t_span = self._setup(t_span)
        be, p, d, y0 = self.backend, self._params, self._direction, self.y0

        # f0 = move(t_span[0], t_span[1] - t_span[0], y0)                                          :83
        f0 = self._eval(self._scalar_t(t_span[0], self.dtype), y0)
        first_dev = None
        self._ctrl_ready = False
        if self.first_step is None:
            # f0 is recomputed inside select_initial_step (f0=None), as in the reference         :84-87
            f0_again = f0 if self._reuse_f0 else None
            if f0_again is not None:
                self._nfe_skipped += 1  # (the call the reference makes here and this solve does not)
            if self._custom_norm or not self._device_first_step:
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
        if (hasattr(be, "ctrl_peek_async") and os.environ.get("XDE_SHORT_SOLVES", "1") != "0"
                and (self.pipeline == "lag" or (self.pipeline == "auto" and self._auto_pick() == "lag"))):
            self._init_peek = be.ctrl_peek_async(self._ctrl)

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
        self._graph_warmup = None
        self._direction = direction_of(t_span)
        self._t_host = t_span
        self._t_span_dev = upload(t_span.astype(np.float64), dev)
        self._work = w = be.acquire_work(dev, y0.dtype)  # recycled by integrate() when the solve has ended
        self._t_stage, self._ctrl, self._ws, self._sums = w.t_stage, w.ctrl, w.ws, w.sums
        self._t_views = [self._t_stage[i] for i in range(self._n_stage)]  # the 0-dim stage times handed to func
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
        no prescribed step sequence.  XDE_FUSED_FIRST_STEP=0 keeps the separate launches (same results; measured side by side)."""
        return (self._small_state and self._chunks is None and self.process_group is None and not self._replay
                and hasattr(self.backend, "initial_step_fused") and os.environ.get("XDE_FUSED_FIRST_STEP", "1") != "0")

    def _select_initial_step_device(self, t0, y0, f0=None):
        """``select_initial_step`` (base_adaptive_solver.py:33-72) with its scalar arithmetic on the device: the three
        norms feed two one-thread launches (xde_initial_step) instead of two blocking reads; the first step never visits
        the host.  Same op order and dtypes as the host version above it in the class hierarchy."""
        be = self.backend
        dev = y0.device
        sdt = _hip.dtype_code(y0.dtype)
        t0h = np_dtype(self.dtype)(t0)
        if f0 is None:
            f0 = self._eval(self._scalar_t(t0h, self.dtype), y0)
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
        if self.process_group is None and (self._fuse_control or self._small_state):
            # single GPU: error norm + controller in ONE launch (the last workgroup to arrive runs the controller)
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
        self._base = (self.rk_state.y1, self.rk_state.f1)
        self._pending = None  # lag pipeline: (y1, [f1], read handle) of the newest, unresolved attempt
        self._n_attempts = 0
        self._last = None
        self._graph = None
        self.advance(None)
        self._solution = None
        self._graph = None  # releases the captured graphs and their private memory pools
        self._graphs = {}

    AUTO_GRAPH_MAX_BYTES = 8 << 20  # per state operand; above it the step is bandwidth-bound and "lag" wins
    AUTO_GRAPH_AFTER = 16  # attempts made eagerly before a capture is worth its ~2 ms

    def _auto_pick(self):
        """The pipeline `auto` resolves to for this solve (see the module docstring)."""
        y0 = self.y0
        if self.process_group is not None or y0.numel() * y0.element_size() > self.AUTO_GRAPH_MAX_BYTES:
            return "lag"
        return "sync-then-graph" if y0.is_cuda else "sync"

    def _auto_may_capture(self):
        import threading

        return (threading.current_thread() is threading.main_thread() and self.y0.is_cuda
                and not torch.cuda.is_current_stream_capturing())

    def _advance_auto(self, max_attempts):
        to_end = max_attempts is None
        if self._auto_state is None:
            self._auto_state = self._auto_pick()
        if self._auto_state == "lag":
            return self._advance_lag(max_attempts)
        if self._auto_state == "graph":
            return self._advance_graph(max_attempts)
        if self._auto_state == "sync":
            return self._advance_sync(max_attempts)
        # "sync-then-graph": eager attempts first; short solves (the adjoint's 1-3 step intervals) end here
        done = 0
        left = self.AUTO_GRAPH_AFTER - self._n_attempts
        if left > 1:
            n = left - 1 if to_end else min(left - 1, max_attempts)
            c = self._advance_sync(n, stop_on_done=to_end)
            done += n
            if (to_end and c.done) or (not to_end and done >= max_attempts):
                return c
        # one more eager attempt under the capture guard: does func differentiate w.r.t. parameter leaves?
        from ..utils.graphed import _AutogradTargetProbe

        with _AutogradTargetProbe() as probe:
            c = self._advance_sync(1, stop_on_done=to_end)
        done += 1
        if probe.hit is not None or not self._auto_may_capture():
            self._auto_state = "sync"
        else:
            self._auto_state = "graph"
            self._graph_warmup = 0
        if (to_end and c.done) or (not to_end and done >= max_attempts):
            return c
        rest = None if to_end else max_attempts - done
        if self._auto_state == "graph":
            nfe0 = self.nfe
            try:
                return self._advance_graph(rest)
            except AssertionError:
                raise  # the solver's own status errors
            except Exception:  # the capture failed (func syncs with the host, allocates pinned memory, ...): stay eager
                if getattr(self, "_graph", None) is not None:
                    raise  # the failure came after a successful capture: not ours to hide
                self.nfe = nfe0
                self._auto_state = "sync"
        return self._advance_sync(rest)

    def advance(self, max_attempts=None):
        """Run attempted steps until every output is produced (``max_attempts=None``) or exactly
        ``max_attempts`` attempts were made.  Returns the last control block read from the device.

        Callable after ``_before_integrate`` as a stepping API (bench.py drives it with a fixed count)."""
        if getattr(self, "_base", None) is None:
            self._solution = None
            self._base = (self.rk_state.y1, self.rk_state.f1)
            self._pending = None
            self._n_attempts = 0
            self._last = None
            self._graph = None
        with torch.no_grad():  # once per advance, not once per func evaluation
            if self.pipeline == "auto":
                c = self._advance_auto(max_attempts)
            elif self.pipeline == "lag":
                c = self._advance_lag(max_attempts)
            elif self.pipeline == "graph":
                c = self._advance_graph(max_attempts)
            else:
                c = self._advance_sync(max_attempts)
        self._finish(c, self._base)
        return c

    def _callbacks_before(self, c, base):
        """`callback_step(t0, y0, dt)` at the top of an attempt (base_adaptive_solver_rk.py:186); returns the arguments for the verdict's
        callback.  `c`: the newest control block (None before the first attempt: the first step size was chosen on the device)."""
        if c is None:
            c = self.backend.ctrl_read(self._ctrl)
        args = (torch.tensor(c.t1, dtype=self.dtype), base[0], torch.tensor(c.dt, dtype=self.dtype))
        if self._cb_step is not None:
            self._cb_step(*args)
        return args

    def _callbacks_after(self, c, args):
        """`callback_accept_step` / `callback_reject_step` on the attempt's verdict (:259, :275)."""
        cb = self._cb_accept if c.accept else self._cb_reject
        if cb is not None:
            cb(*args)

    def _advance_sync(self, max_attempts, stop_on_done=None):
        be = self.backend
        c = self._last
        done = 0
        if stop_on_done is None:
            stop_on_done = max_attempts is None
        while max_attempts is None or done < max_attempts:
            base = self._base
            cb_args = self._callbacks_before(c, base) if self._has_callbacks else None
            y1, ks = self._attempt(base)
            self._n_attempts += 1
            done += 1
            c = be.ctrl_read(self._ctrl)  # the step's one host sync
            if self.record_trace:
                self.trace.append((c.t0, c.dt_last, c.ratio, bool(c.accept)))
            if self._step_hook is not None:
                self._step_hook(self._n_attempts - 1, base[0], y1, ks, c)
            if cb_args is not None:
                self._callbacks_after(c, cb_args)
            if c.accept:
                if c.out_end > c.out_begin and self._solution is not None:
                    self._dense(self._solution, base, y1, ks)
                self._base = (y1, ks[-1])
            self._raise_status(c)
            del y1, ks  # (dead now: released before the next attempt allocates, so that it reuses these very blocks)
            if c.done and stop_on_done:
                break
        self._last = c
        return c

    GRAPH_WARMUP_ATTEMPTS = 2
    # Attempted steps captured per graph.  A graph launch costs the GPU ~8 us of idle time between the last node of one
    # replay and the first node of the next (rocprofv3 trace of config 5: profiles/r02_c5_graph_gaps.txt), so a solve that
    # runs to its end replays graphs of several attempts; attempts past the last output are device-side no-ops.
    GRAPH_ATTEMPTS = 4
    assert 2 * GRAPH_ATTEMPTS < _hip.XDE_MIRROR_SLOTS  # an unread block must never be overwritten (the slot is a seqlock too)

    def _graph_of(self, k):
        """The captured graph of ``k`` consecutive attempted steps on the static operands ``self._gbase``."""
        be = self.backend
        g = self._graphs.get(k)
        if g is None:
            def body():
                nfe0 = self.nfe  # evaluations are accounted per resolved replay, not while recording
                base = self._gbase
                for _ in range(k):
                    y1, ks = self._attempt(base)
                    if self._solution is not None:  # rows of this step + the state hand-over, one launch
                        idx, coef = self._mid_plan
                        be.dense_commit(self._solution, [ks[j] for j in idx], coef, base[0], y1, ks[-1], self._ctrl,
                                        self._t_span_dev, _hip.dtype_code(self.dtype))
                    else:
                        be.commit(self._ctrl, base[0], y1, base[1], ks[-1])
                self.nfe = nfe0

            g = self._graphs[k] = be.capture(body, self._ctrl, launches=k)
            self._graph = g
        return g

    def _advance_graph(self, max_attempts):
        be = self.backend
        to_end = max_attempts is None
        done = 0
        if getattr(self, "_graph", None) is None:
            # eager warm-up (also lets short integrations finish without paying for a capture)
            warm = self.GRAPH_WARMUP_ATTEMPTS if self._graph_warmup is None else self._graph_warmup
            n_warm = warm if to_end else min(warm, max_attempts)
            if n_warm > 0:
                c = self._advance_sync(n_warm, stop_on_done=to_end)
                done += n_warm
                if (to_end and c.done) or (not to_end and done >= max_attempts):
                    return c
            y0, f0 = self._base
            self._gbase = (y0.clone(), f0.clone())  # static operands of the captured step
            self._graphs = {}
            self._graph_of(self.GRAPH_ATTEMPTS if to_end else min(self.GRAPH_ATTEMPTS, max(max_attempts - done, 1)))
            self._base = self._gbase
        K = self.GRAPH_ATTEMPTS
        pending = collections.deque()
        issued = 0
        finished = False
        while True:
            left = None if to_end else max_attempts - done - issued
            if not finished and (to_end or left > 0) and len(pending) <= K:
                # a budgeted advance (bench.py times EXACTLY its step count) ends on single-attempt replays
                k = K if (to_end or left >= K) else (left if left in self._graphs else 1)
                pending.extend(self._graph_of(k).replay())
                issued += k
                continue
            if not pending:
                break
            c = be.ctrl_wait(pending.popleft())
            if finished:
                continue  # replays issued past the last output are device-side no-ops (controller `done` guard)
            self._n_attempts += 1
            self.nfe += self._n_stage
            if self.record_trace:
                self.trace.append((c.t0, c.dt_last, c.ratio, bool(c.accept)))
            self._raise_status(c)
            self._last = c
            if c.done and to_end:
                finished = True
        return self._last

    def _resolve_pending(self):
        c = self.backend.ctrl_wait(self._pending[2])
        if self.record_trace:
            self.trace.append((c.t0, c.dt_last, c.ratio, bool(c.accept)))
        if c.accept:
            self._base = (self._pending[0], self._pending[1][-1])
        self._pending = None
        self._raise_status(c)
        return c

    def _advance_lag(self, max_attempts):
        """Speculative pipeline: the host resolves attempt n-1 only after attempt n is enqueued.

        The end of a solve is not speculated over (round 4): the block of attempt n-1 says where attempt n will land if it is accepted
        (`t_plan`); when that is at or past the last output time, the host waits for attempt n's verdict before it enqueues anything
        else — a stall of one poll instead of a whole discarded attempt (six func evaluations and ~35 N elements of traffic; 9 % of
        config 2's `odeint` over [0, 1]).  Only a solve that ends with its very FIRST attempts, whose step size the host never
        saw, would still pay for one discarded attempt: `_before_integrate` therefore enqueues a copy of the freshly constructed block
        (ahead of the first attempt's kernels), whose `t_plan` is read here — by then long on the host — before a second attempt is
        enqueued.  (`_short_solves`, odeint_adjoint's hint that its interval solves are short, covers a backend without that copy.)"""
        be = self.backend
        c = self._last
        done = 0
        to_end = max_attempts is None
        d = self._direction
        t_last = float(self._t_host[-1])
        while to_end or done < max_attempts:
            base = self._base
            alt = (self._pending[0], self._pending[1][-1]) if self._pending is not None else None
            y1, ks = self._attempt(base, alt)
            self._n_attempts += 1
            done += 1
            if self._solution is not None:
                self._dense(self._solution, base, y1, ks, alt, expect_step=self._n_attempts)
            handle = be.ctrl_read_async(self._ctrl)
            planned_end = None  # where the attempt just enqueued lands if accepted (known from its predecessor's block)
            if self._pending is not None:
                c = self._resolve_pending()
                if c.done and to_end:
                    # the attempt just enqueued is a device-side no-op (ctrl->done guards the controller and
                    # the dense kernel); its func evaluations are the price of never stalling the GPU
                    self.nfe -= self._n_stage
                    self._last = c
                    return c
                planned_end = c.t_plan
            elif c is not None:
                planned_end = c.t_plan  # (the predecessor was resolved synchronously: see below)
            # Only the proposal (y1, f1 = ks[-1]) of the unresolved attempt is kept: its other stage derivatives are dead once its
            # dense-output launch is enqueued, and released HERE they are the blocks the next attempt's func writes into — the step
            # cycles through ~12 state-sized buffers instead of ~18 (config 4's shard: 192 MiB instead of 288, i.e. inside the
            # 256 MiB Infinity Cache instead of spilling out of it).
            self._pending = (y1, ks[-1:], handle)
            del ks
            if planned_end is None and getattr(self, "_init_peek", None) is not None:
                # the first attempt of the solve: its landing point is in the block the heuristic / ctrl_init constructed
                planned_end = be.ctrl_peek_result(self._init_peek).t_plan
                self._init_peek = None
            if planned_end is None and self._short_solves:
                planned_end = t_last  # (no copy of that block: a solve that is expected to be short takes its first attempt as its last)
            if to_end and planned_end is not None and d * planned_end >= d * t_last:
                c = self._resolve_pending()  # this attempt ends the solve if it is accepted: do not speculate past it
                if c.done:
                    self._last = c
                    return c
        if self._pending is not None:  # drain: the caller gets a fully resolved state
            c = self._resolve_pending()
        self._last = c
        return c

    # ------------------------------------------------------------------------------------------
    # re-armable interval solves: one captured graph = the initial-step heuristic + the first attempted step of a 2-point solve
    # ------------------------------------------------------------------------------------------
    # odeint_adjoint's backward pass is one short solve per output interval (functional/odeint_adjoint.py:134-159): same state
    # layout, same tolerances, same direction, 1-2 attempted steps each — and, for a small state, launch-bound.  Instead of one
    # solver object, one control block and ~170 launches per interval, ONE solver is kept for the whole sweep (and the next
    # backward pass): the state lives in a static buffer (`interval_state`), the two output times are uploaded into a static
    # pair, and a replayed hipGraph re-arms the control block on the device (xde_initial_step_fused with t_start = NaN and
    # seq0 < 0), runs the heuristic's two evaluations and the first attempt, writes the output row and hands the state over
    # (xde_dense_commit).  A second graph holds one more attempt for the intervals that need it.  Same kernels, same operands,
    # same order as the eager solve: bit-identical results (tests/_e2e_cases.py::test_adjoint_captured_interval_solves).
    def intervals_supported(self):
        """Whether this solver's options allow the captured interval solve (else: one ordinary solve per interval)."""
        return bool(self.y0.is_cuda and not self._custom_norm and self._chunks is None and self.process_group is None
                    and not self._replay and self.first_step is None and self.step_t is None and not self._has_callbacks
                    and self._step_hook is None and not self.record_trace and self._reuse_f0 and self._stats_out is None
                    and self.pipeline == "auto" and self._device_first_step and self.y0.dim() == 1)

    def intervals_prepare(self, t_span, capture=True):
        """Static buffers for 2-point solves in the direction of ``t_span`` (two host times), one eager solve of that span from
        the constructor's ``y0`` as warm-up and — ``capture`` — the two graphs.  Main thread, outside autograd nodes (see
        utils/graphed.py); raises when the capture fails (the caller keeps the per-interval solves)."""
        be = self.backend
        t_span = self._setup(t_span)
        if len(t_span) != 2 or not t_span[0] != t_span[1]:
            raise ValueError("intervals_prepare needs two distinct output times")
        y0, dev = self.y0, self.y0.device
        # One static device block of five doubles: [d0, d1 | t0, t1 | t0 in the time dtype].  The last three — the interval's two output
        # times as the kernels read them and its start time as func is handed it — are written by ONE host-to-device copy per interval
        # from a pinned mirror; the first two are the heuristic's norms (separate launches of a larger state), so that the scalar
        # kernel finds (d0, d1, start time) side by side (xde_initial_step, phase 2).
        self._iv_pinned = torch.zeros(24, dtype=torch.uint8).pin_memory()
        self._iv_block = torch.zeros(40, dtype=torch.uint8, device=dev)
        self._iv_res = self._iv_block[:24].view(torch.float64)
        self._t_span_dev = self._iv_block[16:32].view(torch.float64)
        self._iv_t0 = self._iv_block[32:40].view(self.dtype)[0]
        self._iv_upload(t_span)
        self._gbase = (y0.clone(), None)
        self._iv_y1 = torch.empty_like(y0)
        self._iv_hs = torch.zeros(5, dtype=torch.float64, device=dev)
        self._iv_tprobe = torch.empty((), dtype=torch.promote_types(self.dtype, y0.dtype), device=dev)
        self._solution = torch.empty((2,) + tuple(y0.shape), dtype=y0.dtype, device=dev)
        self._iv_first_graph = self._iv_next_graph = None
        self._ctrl_ready = True
        # (a control block whose launch count the host mirror agrees with, before the in-graph re-arming keeps counting from it)
        be.ctrl_init(self._ctrl, self._params, float(t_span[0]), 0.0, 2, self._t_span_dev, self._step_t_dev, self._t_stage)
        with torch.no_grad(), torch.autograd.set_multithreading_enabled(False):
            nfe0 = self.nfe
            self.interval_solve(t_span)  # eager: func's lazy initialisations, the allocator's blocks
            if capture:
                torch.cuda.synchronize(dev)
                self._iv_first_graph = be.capture(self._iv_first, self._ctrl, launches=1)
                self._iv_next_graph = be.capture(self._iv_attempt, self._ctrl, launches=1)
            self.nfe = nfe0
        return self

    def _iv_upload(self, t):
        host = self._iv_pinned.numpy()
        host[:16].view(np.float64)[:] = np.asarray(t, dtype=np.float64)
        host[16:24].view(np_dtype(self.dtype))[0] = t[0]
        self._iv_block[16:].copy_(self._iv_pinned, non_blocking=True)

    @property
    def interval_state(self):
        """The static state buffer an interval solve starts from (write the state into it, do not replace it)."""
        return self._gbase[0]

    def _iv_first(self):
        be = self.backend
        y0 = self._gbase[0]
        nan = float("nan")
        # (f0 stays where func wrote it — recorded: a block of this graph's private pool, kept allocated by this reference, which the
        # second graph addresses too)
        f0 = self._eval(self._iv_t0, y0)
        self._gbase = (y0, f0)
        if self._fused_first_step():  # small state: two one-workgroup launches around the Euler probe
            be.initial_step_fused(0, f0, None, y0, self._xsegs, self._iv_hs, self._params, nan, self._iv_tprobe, self._ctrl,
                                  t_span_dev=self._t_span_dev, keep_seq=True)
            be.stage_combine(self._iv_y1, y0, [f0], [1.0], _hip.COMBINE_FUSE, ctrl=self._ctrl)  # fuse(f0, h0, y0)
            f1 = self._eval(self._iv_tprobe, self._iv_y1)
            be.initial_step_fused(1, f1, f0, y0, self._xsegs, self._iv_hs, self._params, nan, None, self._ctrl, 2, self._t_span_dev,
                                  self._step_t_dev, self._t_stage, keep_seq=True)
        else:  # the separate launches of _select_initial_step_device, the start time read on the device
            res, hs = self._iv_res, self._iv_hs
            self._scaled_norm_into(y0, None, y0, self.rtol, self.atol, res[0:1])
            self._scaled_norm_into(f0, None, y0, self.rtol, self.atol, res[1:2])
            be.initial_step(2, res, hs, self._params, nan, self._iv_tprobe, self._ctrl)  # h0 -> ctrl.dt, t0 + h0 -> t_probe
            be.stage_combine(self._iv_y1, y0, [f0], [1.0], _hip.COMBINE_FUSE, ctrl=self._ctrl)  # fuse(f0, h0, y0)
            f1 = self._eval(self._iv_tprobe, self._iv_y1)
            self._scaled_norm_into(f1, f0, y0, self.rtol, self.atol, res[0:1])
            be.initial_step(1, res, hs, self._params, nan, None, self._ctrl)
            be.ctrl_init(self._ctrl, self._params, nan, 0.0, 2, self._t_span_dev, self._step_t_dev, self._t_stage,
                         first_step_dev=hs[3:4], keep_seq=True)
        self._iv_attempt()

    def _iv_attempt(self):
        base = self._gbase
        y1, ks = self._attempt(base)
        idx, coef = self._mid_plan
        self.backend.dense_commit(self._solution, [ks[j] for j in idx], coef, base[0], y1, ks[-1], self._ctrl, self._t_span_dev,
                                  _hip.dtype_code(self.dtype))

    def interval_solve(self, t_span):
        """Integrate ``interval_state`` from ``t_span[0]`` to ``t_span[1]`` (host times, the prepared direction); returns the
        state at ``t_span[1]`` — a view of a static row, valid until the next call.  ``interval_state`` is overwritten."""
        be = self.backend
        tt = np_dtype(self.dtype)
        t = np.asarray([t_span[0], t_span[1]], dtype=tt)
        d = self._direction
        if not d * t[1] > d * t[0]:
            if t[1] == t[0]:  # (an output time repeated: the state itself, as integrate() fills such rows)
                self._solution[1].copy_(self._gbase[0])
                return self._solution[1]
            raise AssertionError("interval_solve: the interval runs against the prepared direction")
        self._iv_upload(t)
        graphs = self._iv_first_graph is not None
        with torch.no_grad():
            first = True
            while True:
                if graphs:
                    (handle,) = (self._iv_first_graph if first else self._iv_next_graph).replay()
                    c = be.ctrl_wait(handle)
                else:
                    (self._iv_first if first else self._iv_attempt)()
                    c = be.ctrl_read(self._ctrl)
                if graphs:
                    self.nfe += self._n_stage + (2 if first else 0)
                first = False
                self._raise_status(c)
                if c.done:
                    break
        self._last = c
        return self._solution[1]

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
            self._base = (self.rk_state.y1, self.rk_state.f1)
            self._pending = None
            self._n_attempts = 0
            self._last = None
            self._graph = None
        y0 = self.y0
        d = self._direction
        tt = np_dtype(self.dtype)
        nt = tt(float(next_t))
        # the controller and the dense-output kernel now look at this one-entry output list
        self._t_span_dev = t_dev = upload(np.asarray([nt], dtype=np.float64), y0.device)
        be.ctrl_retarget(self._ctrl, self._params, t_dev, 1)
        c = be.ctrl_read(self._ctrl)
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
                base = self._base
                cb_args = self._callbacks_before(c, base) if self._has_callbacks else None
                y1, ks = self._attempt(base)
                self._n_attempts += 1
                c = be.ctrl_read(self._ctrl)
                if self.record_trace:
                    self.trace.append((c.t0, c.dt_last, c.ratio, bool(c.accept)))
                if cb_args is not None:
                    self._callbacks_after(c, cb_args)
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
