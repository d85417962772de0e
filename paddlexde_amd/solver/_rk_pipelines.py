"""How the host drives the attempted steps of an adaptive solve: one class per pipeline (``options["pipeline"]``).

The reference's driver is ``while next_t > rk_state.t1: rk_state = self._adaptive_step(rk_state)`` with ~17 host reads of device
scalars per attempt (paddlexde/solver/base_adaptive_solver_rk.py:116-127, :183-284).  Here the controller lives on the device
(xde_rk_control) and the host only decides WHEN it looks at a verdict:

  SyncPipeline   one poll of the control block's host mirror per attempted step; NFE identical to the reference; the only pipeline
                 that can call back into Python (step callbacks, the parity harness's step hook).
  LagPipeline    speculative: attempt n+1 is enqueued before the host knows whether attempt n was accepted; its kernels pick
                 (y0, f0) between the two candidates from ctrl->accept on the device.  The GPU never waits for the host — except at
                 the end of a solve, which is not speculated over.
  GraphPipeline  whole attempts captured into a hipGraph of GRAPH_ATTEMPTS attempts and replayed (launch-bound small states).
  AutoPipeline   (default) picks among them per solve: "lag" when an operand is larger than AUTO_GRAPH_MAX_BYTES (the step is
                 bandwidth-bound; also with a process_group); otherwise it starts in "sync" and, if the solve is still running after
                 AUTO_GRAPH_AFTER attempts, captures the step and continues as "graph" — provided the capture is safe (main thread,
                 no capture in progress, func does not differentiate with respect to parameter leaves) and succeeds.

Each pipeline is a strategy over ONE stepper (the ``AdaptiveRKSolver``: ``_attempt``, ``_dense``, ``_raise_status`` and the state of
the solve — ``_base``, ``_pending``, ``_last``, ``_n_attempts``, ``trace``, ``nfe``).  Results are bit-identical whichever runs.
"""
import collections
import threading

import torch

from .. import _hip


class SyncPipeline:
    def __init__(self, stepper):
        self.s = stepper

    def _callbacks_before(self, c, base):
        """`callback_step(t0, y0, dt)` at the top of an attempt (base_adaptive_solver_rk.py:186); returns the arguments for the
        verdict's callback.  `c`: the newest control block (None before the first attempt: the first step size was chosen on the
        device)."""
        s = self.s
        if c is None:
            c = s.backend.ctrl_read(s._ctrl)
        args = (torch.tensor(c.t1, dtype=s.dtype), base[0], torch.tensor(c.dt, dtype=s.dtype))
        if s._cb_step is not None:
            s._cb_step(*args)
        return args

    def _callbacks_after(self, c, args):
        """`callback_accept_step` / `callback_reject_step` on the attempt's verdict (:259, :275)."""
        cb = self.s._cb_accept if c.accept else self.s._cb_reject
        if cb is not None:
            cb(*args)

    def attempt_and_resolve(self):
        """One attempted step, its verdict read: ``(base, y1, ks, c)``.  The accepted proposal becomes the new base."""
        s = self.s
        base = s._base
        cb_args = self._callbacks_before(s._last, base) if s._has_callbacks else None
        y1, ks = s._attempt(base)
        s._n_attempts += 1
        c = s.backend.ctrl_read(s._ctrl)  # the step's one host sync
        if s.record_trace:
            s.trace.append((c.t0, c.dt_last, c.ratio, bool(c.accept)))
        if s._step_hook is not None:
            s._step_hook(s._n_attempts - 1, base[0], y1, ks, c)
        if cb_args is not None:
            self._callbacks_after(c, cb_args)
        s._last = c
        return base, y1, ks, c

    def advance(self, max_attempts, stop_on_done=None):
        s = self.s
        c = s._last
        done = 0
        if stop_on_done is None:
            stop_on_done = max_attempts is None
        while max_attempts is None or done < max_attempts:
            base, y1, ks, c = self.attempt_and_resolve()
            done += 1
            if c.accept:
                if c.out_end > c.out_begin and s._solution is not None:
                    s._dense(s._solution, base, y1, ks)
                s._base = (y1, ks[-1])
            s._raise_status(c)
            del base, y1, ks  # (dead now: released before the next attempt allocates, so that it reuses these very blocks)
            if c.done and stop_on_done:
                break
        s._last = c
        return c


class LagPipeline:
    """Speculative pipeline: the host resolves attempt n-1 only after attempt n is enqueued.

    The end of a solve is not speculated over (round 4): the block of attempt n-1 says where attempt n will land if it is accepted
    (`t_plan`); when that is at or past the last output time, the host waits for attempt n's verdict before it enqueues anything else
    — a stall of one poll instead of a whole discarded attempt (six func evaluations and ~35 N elements of traffic; 9 % of config 2's
    `odeint` over [0, 1]).  Only a solve that ends with its very FIRST attempts, whose step size the host never saw, would still pay
    for one discarded attempt: `_before_integrate` therefore enqueues a copy of the freshly constructed block (ahead of the first
    attempt's kernels), whose `t_plan` is read here — by then long on the host — before a second attempt is enqueued.
    (`_short_solves`, odeint_adjoint's hint that its interval solves are short, covers a backend without that copy.)"""

    def __init__(self, stepper):
        self.s = stepper

    def _resolve_pending(self):
        s = self.s
        c = s.backend.ctrl_wait(s._pending[2])
        if s.record_trace:
            s.trace.append((c.t0, c.dt_last, c.ratio, bool(c.accept)))
        if c.accept:
            s._base = (s._pending[0], s._pending[1][-1])
        s._pending = None
        s._raise_status(c)
        return c

    def advance(self, max_attempts):
        s = self.s
        be = s.backend
        c = s._last
        done = 0
        to_end = max_attempts is None
        d = s._direction
        t_last = float(s._t_host[-1])
        while to_end or done < max_attempts:
            base = s._base
            alt = (s._pending[0], s._pending[1][-1]) if s._pending is not None else None
            y1, ks = s._attempt(base, alt)
            s._n_attempts += 1
            done += 1
            if s._solution is not None:
                s._dense(s._solution, base, y1, ks, alt, expect_step=s._n_attempts)
            handle = be.ctrl_read_async(s._ctrl)
            planned_end = None  # where the attempt just enqueued lands if accepted (known from its predecessor's block)
            if s._pending is not None:
                c = self._resolve_pending()
                if c.done and to_end:
                    # the attempt just enqueued is a device-side no-op (ctrl->done guards the controller and
                    # the dense kernel); its func evaluations are the price of never stalling the GPU
                    s.nfe -= s._n_stage
                    s._last = c
                    return c
                planned_end = c.t_plan
            elif c is not None:
                planned_end = c.t_plan  # (the predecessor was resolved synchronously: see below)
            # Only the proposal (y1, f1 = ks[-1]) of the unresolved attempt is kept: its other stage derivatives are dead once its
            # dense-output launch is enqueued, and released HERE they are the blocks the next attempt's func writes into — the step
            # cycles through ~12 state-sized buffers instead of ~18 (config 4's shard: 192 MiB instead of 288, i.e. inside the
            # 256 MiB Infinity Cache instead of spilling out of it).
            s._pending = (y1, ks[-1:], handle)
            del ks
            if planned_end is None and s._init_peek is not None:
                # the first attempt of the solve: its landing point is in the block the heuristic / ctrl_init constructed
                planned_end = be.ctrl_peek_result(s._init_peek).t_plan
                s._init_peek = None
            if planned_end is None and s._short_solves:
                planned_end = t_last  # (no copy of that block: a solve that is expected to be short takes its first attempt as its last)
            if to_end and planned_end is not None and d * planned_end >= d * t_last:
                c = self._resolve_pending()  # this attempt ends the solve if it is accepted: do not speculate past it
                if c.done:
                    s._last = c
                    return c
        if s._pending is not None:  # drain: the caller gets a fully resolved state
            c = self._resolve_pending()
        s._last = c
        return c


class GraphPipeline:
    WARMUP_ATTEMPTS = 2
    # Attempted steps captured per graph.  A graph launch costs the GPU ~8 us of idle time between the last node of one
    # replay and the first node of the next (rocprofv3 trace of config 5: profiles/r02_c5_graph_gaps.txt), so a solve that
    # runs to its end replays graphs of several attempts; attempts past the last output are device-side no-ops.
    ATTEMPTS = 4
    assert 2 * ATTEMPTS < _hip.XDE_MIRROR_SLOTS  # an unread block must never be overwritten (the slot is a seqlock too)

    def __init__(self, stepper, sync):
        self.s, self.sync = stepper, sync
        self.reset()

    def reset(self):
        """Forget the captured graphs (they hold private memory pools and address the solve's static operands)."""
        self.graphs = {}
        self.base = None  # static operands (y0, f0) of the captured step
        self.warmup = None  # eager attempts before the capture (None: WARMUP_ATTEMPTS)

    @property
    def captured(self):
        return bool(self.graphs)

    def graph_of(self, k):
        """The captured graph of ``k`` consecutive attempted steps on the static operands."""
        s = self.s
        be = s.backend
        g = self.graphs.get(k)
        if g is None:
            def body():
                nfe0 = s.nfe  # evaluations are accounted per resolved replay, not while recording
                base = self.base
                for _ in range(k):
                    y1, ks = s._attempt(base)
                    if s._solution is not None:  # rows of this step + the state hand-over, one launch
                        idx, coef = s._mid_plan
                        be.dense_commit(s._solution, [ks[j] for j in idx], coef, base[0], y1, ks[-1], s._ctrl, s._t_span_dev,
                                        _hip.dtype_code(s.dtype))
                    else:
                        be.commit(s._ctrl, base[0], y1, base[1], ks[-1])
                s.nfe = nfe0

            g = self.graphs[k] = be.capture(body, s._ctrl, launches=k)
        return g

    def advance(self, max_attempts):
        s = self.s
        be = s.backend
        to_end = max_attempts is None
        done = 0
        if not self.captured:
            # eager warm-up (also lets short integrations finish without paying for a capture)
            warm = self.WARMUP_ATTEMPTS if self.warmup is None else self.warmup
            n_warm = warm if to_end else min(warm, max_attempts)
            if n_warm > 0:
                c = self.sync.advance(n_warm, stop_on_done=to_end)
                done += n_warm
                if (to_end and c.done) or (not to_end and done >= max_attempts):
                    return c
            y0, f0 = s._base
            self.base = (y0.clone(), f0.clone())
            self.graph_of(self.ATTEMPTS if to_end else min(self.ATTEMPTS, max(max_attempts - done, 1)))
            s._base = self.base
        K = self.ATTEMPTS
        pending = collections.deque()
        issued = 0
        finished = False
        while True:
            left = None if to_end else max_attempts - done - issued
            if not finished and (to_end or left > 0) and len(pending) <= K:
                # a budgeted advance (bench.py times EXACTLY its step count) ends on single-attempt replays
                k = K if (to_end or left >= K) else (left if left in self.graphs else 1)
                pending.extend(self.graph_of(k).replay())
                issued += k
                continue
            if not pending:
                break
            c = be.ctrl_wait(pending.popleft())
            if finished:
                continue  # replays issued past the last output are device-side no-ops (controller `done` guard)
            s._n_attempts += 1
            s.nfe += s._n_stage
            if s.record_trace:
                s.trace.append((c.t0, c.dt_last, c.ratio, bool(c.accept)))
            s._raise_status(c)
            s._last = c
            if c.done and to_end:
                finished = True
        return s._last


class AutoPipeline:
    GRAPH_MAX_BYTES = 8 << 20  # per state operand; above it the step is bandwidth-bound and "lag" wins
    GRAPH_AFTER = 16  # attempts made eagerly before a capture is worth its ~2 ms

    def __init__(self, stepper, sync, lag, graph):
        self.s, self.sync, self.lag, self.graph = stepper, sync, lag, graph

    def pick(self):
        """The pipeline `auto` resolves to for this solve (see the module docstring)."""
        s = self.s
        y0 = s.y0
        if s.process_group is not None or y0.numel() * y0.element_size() > self.GRAPH_MAX_BYTES:
            return "lag"
        return "sync-then-graph" if y0.is_cuda else "sync"

    def may_capture(self):
        return (threading.current_thread() is threading.main_thread() and self.s.y0.is_cuda
                and not torch.cuda.is_current_stream_capturing())

    def advance(self, max_attempts):
        s = self.s
        to_end = max_attempts is None
        if s._auto_state is None:
            s._auto_state = self.pick()
        if s._auto_state == "lag":
            return self.lag.advance(max_attempts)
        if s._auto_state == "graph":
            return self.graph.advance(max_attempts)
        if s._auto_state == "sync":
            return self.sync.advance(max_attempts)
        # "sync-then-graph": eager attempts first; short solves (the adjoint's 1-3 step intervals) end here
        done = 0
        left = self.GRAPH_AFTER - s._n_attempts
        if left > 1:
            n = left - 1 if to_end else min(left - 1, max_attempts)
            c = self.sync.advance(n, stop_on_done=to_end)
            done += n
            if (to_end and c.done) or (not to_end and done >= max_attempts):
                return c
        # one more eager attempt under the capture guard: does func differentiate w.r.t. parameter leaves?
        from ..utils.graphed import _AutogradTargetProbe

        with _AutogradTargetProbe() as probe:
            c = self.sync.advance(1, stop_on_done=to_end)
        done += 1
        if probe.hit is not None or not self.may_capture():
            s._auto_state = "sync"
        else:
            s._auto_state = "graph"
            self.graph.warmup = 0
        if (to_end and c.done) or (not to_end and done >= max_attempts):
            return c
        rest = None if to_end else max_attempts - done
        if s._auto_state == "graph":
            nfe0 = s.nfe
            try:
                return self.graph.advance(rest)
            except AssertionError:
                raise  # the solver's own status errors
            except Exception:  # the capture failed (func syncs with the host, allocates pinned memory, ...): stay eager
                if self.graph.captured:
                    raise  # the failure came after a successful capture: not ours to hide
                s.nfe = nfe0
                s._auto_state = "sync"
        return self.sync.advance(rest)
