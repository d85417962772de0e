"""``odeint_adjoint`` — O(1)-memory gradients: the adjoint ODE is integrated backwards, interval by interval, on the same kernels.

Reference behaviour restated here (paddlexde/functional/odeint_adjoint.py): the forward solve under ``no_grad`` and what is saved
for the backward (:11-45); the backward sweep over the output intervals with the augmented state ``(adj_t, y, adj_y, *adj_theta)``,
the cotangent ``-adj_y``, the reset ``y <- y_ans[i-1]`` and the jump ``adj_y += grad_y[i-1]`` (:47-167); argument defaults and
validation with the reference's three messages (:170-257); which tensors count as adjoint parameters (:260-277); the default /
``"seminorm"`` / user adjoint norms (:280-327).

How it is organised here (not the reference's layout):
  * everything the backward needs is ONE immutable record (``_BackwardPlan``) built by ``odeint_adjoint`` and handed to the autograd
    node, instead of a dozen positional arguments copied onto ``ctx`` one by one;
  * the augmented state lives in the flat, 16-byte-segment layout the kernels integrate for the WHOLE sweep (``_AugmentedState``):
    the reference rebuilds a tuple per interval, here the reset and the jump are two in-place copies into the flat buffer;
  * the adjoint norm is an object (``SegmentMaxNorm``) that states what it is — "max over the first n segments of the per-segment
    state norm" — so that the solver maps it onto ONE segmented reduction kernel without inspecting closures;
  * the augmented dynamics can be replayed from a captured HIP graph (``adjoint_options["graph_func"]``) — and with it the sweep's
    2-point solves themselves: ONE re-armable solver serves every interval (and the next backward pass), the initial-step heuristic
    and the first attempted step of an interval are one graph replay (``_sweep_captured``; ``adjoint_options["interval_graph"]``);
  * a batch-sharded backward sums the row-summed adjoints over the process group where the step control looks at them.
Files: this one holds the public entry points, the adjoint norms, the backward plan and the sweep; ``_adjoint_dynamics.py`` the augmented
dynamics (torch autograd / the caller's vjp hook / the functional form a graph records); ``_adjoint_capture.py`` the captured dynamics,
their per-module cache and the re-armable interval solvers.
Deviations documented in SURVEY.md: D4 (tuple state flattened), D5 (reverse-time intervals run natively with a signed dt), D6 (the
gradient w.r.t. ``y0`` is returned instead of ``None``)."""
import warnings
from typing import Any, NamedTuple, Optional

import torch
import torch.nn as nn

from ..utils.ode_utils import _rms_norm, native_norm_spec
from .odeint import _odeint_packed, _pack, _segment_layout, odeint

from ._adjoint_capture import (  # noqa: F401  (the cache and its markers are reached through this module by tests and tools)
    _GRAPH_CACHE,
    _IntervalSolver,
    _NoGraph,
    _NoIntervals,
    _captured_dynamics,
    _first_sweep_interval,
    _interval_key,
)
from ._adjoint_dynamics import (  # noqa: F401
    _N_LEADING,
    _group_sum,
    _is_fixed,
    _make_augmented_dynamics,
    _make_functional_dynamics,
    _time_first,
    _vjp_through_hook,
)


# ----------------------------------------------------------------------------------------------------------------------
# adjoint norm                                                               (reference: odeint_adjoint.py:280-327)
# ----------------------------------------------------------------------------------------------------------------------
class SegmentMaxNorm:
    """``max`` over the leading ``n_segments`` tensors of the augmented state (``None``: all of them) of ``state_norm`` — with
    ``|adj_t|`` for the scalar first segment, which is what any of the state norms gives for one element.

    ``n_segments=None`` is the reference's default adjoint norm (:284-287: ``max(|t|, norm(y), norm(adj_y), mixed(adj_theta))``),
    ``n_segments=3`` its "seminorm" (:301-309: the parameter adjoints do not steer the step size).  When ``state_norm`` is the native
    RMS norm the object carries ``_xde_native = ("mixed", n_segments)`` and the adaptive solver evaluates it as one segmented
    reduction launch; with any other state norm it is an ordinary callable applied to the tuple of segment views."""

    def __init__(self, state_norm, n_segments: Optional[int]):
        self.state_norm = state_norm
        self.n_segments = n_segments
        if native_norm_spec(state_norm) == ("rms",):
            self._xde_native = ("mixed", n_segments)

    def __call__(self, parts):
        parts = tuple(parts) if self.n_segments is None else tuple(parts)[: self.n_segments]
        worst = parts[0].abs()
        for x in parts[1:_N_LEADING]:
            worst = max(worst, self.state_norm(x))
        for x in parts[_N_LEADING:]:  # the parameter adjoints: RMS each (the reference's _mixed_norm), whatever the state norm
            worst = max(worst, _rms_norm(x))
        return worst

    @property
    def watches_parameters(self):
        return self.n_segments is None or self.n_segments > _N_LEADING


def _resolve_adjoint_norm(adjoint_options, state_norm):
    """The norm the backward solve runs with: absent -> the default, ``"seminorm"`` -> the semi-norm, a callable -> itself
    (it receives ``(adj_t, y, adj_y, *adj_theta)``)."""
    chosen = adjoint_options.get("norm")
    if chosen is None:
        return SegmentMaxNorm(state_norm, None)
    if isinstance(chosen, str):
        if chosen != "seminorm":
            raise ValueError("adjoint_options['norm'] must be a callable or \"seminorm\", got {!r}".format(chosen))
        return SegmentMaxNorm(state_norm, _N_LEADING)
    return chosen


# ----------------------------------------------------------------------------------------------------------------------
# adjoint parameters                                                          (reference: odeint_adjoint.py:216-234,260-277)
# ----------------------------------------------------------------------------------------------------------------------
def _module_tensors(func):
    """The tensors of ``func`` that gradients are wanted for.  A DataParallel replica keeps them as plain attributes of its
    sub-modules rather than registered parameters (:263-275)."""
    if not isinstance(func, nn.Module):
        raise TypeError("expected an nn.Module")
    if not getattr(func, "_is_replica", False):
        return tuple(func.parameters())
    live = lambda m: [(name, v) for name, v in vars(m).items() if torch.is_tensor(v) and v.requires_grad]  # noqa: E731
    return tuple(v for _, v in func._named_members(get_members_fn=live))


def _adjoint_parameters(func, given, norm_is_users):
    params = _module_tensors(func) if given is None else tuple(given)
    wanted = tuple(p for p in params if p.requires_grad)
    if len(wanted) < len(params) and norm_is_users:
        warnings.warn(
            "An adjoint parameter was passed without requiring gradient. For efficiency this will be "
            "excluded from the adjoint pass, and will not appear as a tensor in the adjoint norm."
        )
    return wanted


# ----------------------------------------------------------------------------------------------------------------------
# the backward sweep                                                          (reference: odeint_adjoint.py:47-167)
# ----------------------------------------------------------------------------------------------------------------------
class _BackwardPlan(NamedTuple):
    func: Any
    solver: Any  # the adjoint solver class
    rtol: float
    atol: float
    options: dict  # solver options of the backward solves ("norm" resolved)
    time_grad: bool
    graphed: Any  # captured flat dynamics or None
    replay_intervals: Any  # parity harness: one prescribed (dt, accept) table per interval, in the order the intervals are run
    forward_is_fixed: bool
    y0_shape: tuple
    vjp: Any = None  # the caller's vector-Jacobian product (adjoint_options["vjp"], on torch tensors) or None: torch autograd


class _AugmentedState:
    """``(adj_t, y, adj_y, *adj_theta)`` kept in ONE flat buffer with 16-byte-aligned segments for the whole sweep."""

    def __init__(self, y_last, grad_last, adjoint_params):
        parts = [torch.zeros([], dtype=y_last.dtype, device=y_last.device), y_last, grad_last]
        parts += [torch.zeros_like(p) for p in adjoint_params]
        self.shapes = [tuple(x.shape) for x in parts]
        self.dtype, self.segs, self.total = _segment_layout(parts)
        self.flat = _pack(parts, self.segs, self.total, self.dtype, y_last.device)

    def _slice(self, i):
        start, count = self.segs[i]
        return self.flat[start : start + count]

    def subtract_from_adj_t(self, value):
        self._slice(0).sub_(value.to(self.dtype))

    def restart_interval(self, y_saved, grad_here):
        """After an interval's solve: the forward pass's own state replaces the re-integrated one (:155-156), the loss gradient at
        this output time joins the adjoint (:157-159)."""
        self._slice(1).copy_(y_saved.reshape(-1))
        self._slice(2).add_(grad_here.reshape(-1))

    def views(self):
        return [self._slice(i).view(shape) for i, shape in enumerate(self.shapes)]


def _check_sharded_backward(plan, pg, norm_spec, reduce_params):
    fixed = _is_fixed(plan.solver)
    if norm_spec is None and not fixed:
        raise NotImplementedError(
            "a batch-sharded odeint_adjoint needs the default adjoint norm or \"seminorm\" (a user norm callable "
            "cannot be all-reduced)")
    if plan.graphed is not None and (reduce_params or plan.time_grad):
        raise NotImplementedError(
            "adjoint_options['graph_func'] with a process_group needs the \"seminorm\" adjoint norm and no time "
            "gradients (the captured dynamics cannot hold the per-evaluation all-reduce)")


def _interval_solver_for(plan, solve_options, t_host):
    """The captured interval solve prepared for this sweep's options (its lock taken), or None."""
    from ..solver._common import direction_of

    if plan.graphed is None or plan.time_grad or plan.replay_intervals is not None:
        return None
    cache = getattr(plan.graphed, "_intervals", None)
    if not cache:
        return None
    opts = {k: v for k, v in solve_options.items() if k not in ("_xde_flat_func", "_short_solves")}
    span = _first_sweep_interval(t_host)
    if span is None:
        return None
    key = _interval_key(plan.solver, plan.rtol, plan.atol, opts, direction_of(span), t_host.dtype)
    iv = cache.get(key) if key is not None else None
    if not isinstance(iv, _IntervalSolver) or not iv.lock.acquire(blocking=False):
        return None
    return iv


def _sweep_captured(solver, state, t_host, y_ans, grad_y, plan):
    """`_sweep`'s loop on the re-armable solver: the augmented state lives in the solver's static buffer for the whole sweep; per
    interval two output times go up, one graph (seldom two) is replayed, and the row comes back into the state."""
    n_times = len(t_host)
    fixed = _is_fixed(plan.solver)
    flat = solver.interval_state[0] if fixed else solver.interval_state  # (fixed solvers carry the state as [1, total])
    flat.copy_(state.flat)
    state.flat = flat
    times = t_host.tolist()
    for i in range(n_times - 1, 0, -1):
        row = solver.interval_solve((times[i], times[i - 1]))
        if not fixed:  # (the one-step solve leaves its result in the state buffer; the adaptive one in an output row)
            flat.copy_(row)
        state.restart_interval(y_ans[i - 1], grad_y[i - 1])
    parts = [p.clone() for p in state.views()]  # (the static buffer serves the next sweep)
    return parts[2].reshape(plan.y0_shape), None, list(parts[_N_LEADING:])


def _sweep(plan, t_span, y_ans, grad_y, adjoint_params):
    """Integrate the augmented system from the last output time back to the first; returns ``(adj_y0, grad_t_span | None,
    [adj_theta ...])``."""
    n_times = len(t_span)
    if y_ans.numel() == 0 and plan.options.get("process_group") is None:
        # an empty batch (the forward pass served it: solver/base_adaptive_solver.py): nothing flows back — the reference's RMS norm of
        # an empty segment is NaN and its solve ends in "underflow in dt nan"
        zeros = [torch.zeros_like(p) for p in adjoint_params]
        grad_t = torch.zeros(n_times, dtype=t_span.dtype, device=t_span.device) if plan.time_grad else None
        return torch.zeros(plan.y0_shape, dtype=y_ans.dtype, device=y_ans.device), grad_t, zeros
    y_ans = _time_first(y_ans, plan.y0_shape, n_times, plan.forward_is_fixed)
    grad_y = _time_first(grad_y, plan.y0_shape, n_times, plan.forward_is_fixed)
    state = _AugmentedState(y_ans[-1], grad_y[-1], adjoint_params)

    # batch-sharded backward: which of the row-summed adjoints are kept global during the solve (see _make_augmented_dynamics);
    # the others are per-rank partial sums until the one all-reduce at the end
    pg = plan.options.get("process_group")
    norm = plan.options.get("norm")
    spec = native_norm_spec(norm)
    params_steer = not (isinstance(norm, SegmentMaxNorm) and not norm.watches_parameters)
    reduce_params = pg is not None and not _is_fixed(plan.solver) and params_steer  # (a fixed grid has no step control)
    if pg is not None:
        _check_sharded_backward(plan, pg, spec, reduce_params)
    dynamics = _make_augmented_dynamics(plan.func, adjoint_params, plan.time_grad, pg, reduce_params, vjp=plan.vjp)
    solve_options = dict(plan.options)
    if not _is_fixed(plan.solver):
        # one evaluation of the augmented dynamics (func forward + vjp) less per interval: the heuristic's f0 is the state's f0
        solve_options.setdefault("reuse_f0", True)
        # ... and intervals are mostly ONE attempted step long: where the speculative pipeline runs (large states, process groups) it
        # waits for the first attempt's verdict instead of discarding a second attempt per interval
        solve_options["_short_solves"] = True
    if plan.graphed is not None:
        # the captured FLAT dynamics (same segment layout) replaces the unpack -> dynamics -> pack wrapper: 2 input copies + 1 replay
        # + 1 clone per evaluation
        solve_options["_xde_flat_func"] = plan.graphed

    grad_t = torch.empty(n_times, dtype=t_span.dtype, device=t_span.device) if plan.time_grad else None
    t_host = t_span.detach().to("cpu")  # one device->host read of the output times for all intervals
    iv = _interval_solver_for(plan, solve_options, t_host) if (pg is None and n_times > 1) else None
    if iv is not None:
        try:
            return _sweep_captured(iv.solver, state, t_host, y_ans, grad_y, plan)
        finally:
            iv.lock.release()
    for i in range(n_times - 1, 0, -1):
        if plan.time_grad:
            # moving the output time t_i moves the loss by f(t_i, y_i) . dL/dy_i (:137-141)
            moved = plan.func(t_span[i], y_ans[i]).reshape(-1).dot(grad_y[i].reshape(-1))
            if pg is not None:  # a sum over rows: global, like adj_t itself
                (moved,) = _group_sum([moved], pg)
            state.subtract_from_adj_t(moved)
            grad_t[i] = moved
        if plan.replay_intervals is not None:
            solve_options["_replay"] = plan.replay_intervals[n_times - 1 - i]
        rows = _odeint_packed(dynamics, state.flat, state.segs, state.shapes, t_host[i - 1 : i + 1].flip(0), plan.solver,
                              rtol=plan.rtol, atol=plan.atol, options=solve_options)
        state.flat = rows[1]  # the value at t[i-1] (a fresh row: the solver never aliases its input)
        state.restart_interval(y_ans[i - 1], grad_y[i - 1])

    parts = state.views()
    if plan.time_grad:
        grad_t[0] = parts[0]
    adj_params = parts[_N_LEADING:]
    if pg is not None and not reduce_params and len(adj_params):
        # per-rank partial sums so far ("seminorm" never looks at them): ONE all-reduce makes them the gradient of the global loss,
        # identical on every rank — the same thing the default norm's path returns
        adj_params = _group_sum(list(adj_params), pg)
    return parts[2].reshape(plan.y0_shape), grad_t, list(adj_params)


class OdeintAdjointMethod(torch.autograd.Function):
    """Forward: the plain solve, recording nothing (:37-43).  Backward: ``_sweep``."""

    @staticmethod
    def forward(ctx, plan, forward_solve, y0, t_span, *adjoint_params):
        ctx.plan = plan
        with torch.no_grad():
            answer = forward_solve(y0, t_span)
        ctx.save_for_backward(t_span, answer, *adjoint_params)
        return answer

    @staticmethod
    def backward(ctx, grad_answer):
        t_span, answer, *adjoint_params = ctx.saved_tensors
        with torch.no_grad():
            adj_y0, grad_t, adj_params = _sweep(ctx.plan, t_span, answer, grad_answer, tuple(adjoint_params))
        return (None, None, adj_y0, grad_t, *adj_params)  # D6: adj_y0 is returned (the reference drops it)


def _prepare(func, y0, t_span, *, rtol, atol, solver, options, adjoint_rtol, adjoint_atol, adjoint_solver, adjoint_options,
             adjoint_params, vjp, time_grad):
    """Argument defaults and validation of the reference's ``odeint_adjoint`` (:186-238) -> the backward plan and the adjoint
    parameters it will differentiate with respect to."""
    if adjoint_params is None and not isinstance(func, nn.Module):
        raise ValueError(
            "func must be an instance of nn.Module to specify the adjoint parameters; alternatively they "
            "can be specified explicitly via the `adjoint_params` argument. If there are no parameters "
            "then it is allowable to set `adjoint_params=()`."
        )
    # the backward solve inherits what it was not given from the forward solve ...
    adjoint_rtol = rtol if adjoint_rtol is None else adjoint_rtol
    adjoint_atol = atol if adjoint_atol is None else adjoint_atol
    inherit_solver = adjoint_solver is None
    adjoint_solver = solver if inherit_solver else adjoint_solver
    # ... except options across different solvers
    if adjoint_options is None and options is not None and adjoint_solver != solver:
        raise ValueError(
            "If `adjoint_method != method` then we cannot infer `adjoint_options` from `options`. So as "
            "`options` has been passed then `adjoint_options` must be passed as well."
        )
    if adjoint_options is None:
        adjoint_options = {k: v for k, v in (options or {}).items() if k not in ("norm", "from_dlpack")}
    else:
        adjoint_options = dict(adjoint_options)  # the caller's dict is never modified
    adjoint_options.pop("vjp", None)
    adjoint_options.pop("from_dlpack", None)

    if vjp is not None and adjoint_params is not None:
        # with a hook the adjoint parameters only fix the number, shapes and dtype of the parameter adjoints: every one is kept
        # (whether the CALLER's framework marks it trainable is for the hook to know; `None` from it means "no gradient")
        wanted = tuple(adjoint_params)
    else:
        wanted = _adjoint_parameters(func, adjoint_params, norm_is_users=callable(adjoint_options.get("norm")))
    adjoint_options["norm"] = _resolve_adjoint_norm(adjoint_options, options["norm"])

    graphed = _captured_dynamics(func, y0, t_span, adjoint_solver, adjoint_options, wanted, adjoint_rtol, adjoint_atol, vjp=vjp)
    adjoint_options.pop("interval_graph", None)
    plan = _BackwardPlan(
        func=func, solver=adjoint_solver, rtol=adjoint_rtol, atol=adjoint_atol,
        options={k: v for k, v in adjoint_options.items() if k != "_replay_intervals"},
        time_grad=bool(time_grad), graphed=graphed, replay_intervals=adjoint_options.get("_replay_intervals"),
        forward_is_fixed=_is_fixed(solver), y0_shape=tuple(y0.shape), vjp=vjp)
    return plan, wanted


def odeint_adjoint(
    func: callable,
    y0,
    t_span,
    *,
    rtol=1e-7,
    atol=1e-9,
    solver=None,
    options={"norm": _rms_norm},
    event_fn=None,
    adjoint_rtol=None,
    adjoint_atol=None,
    adjoint_solver=None,
    adjoint_options=None,
    adjoint_params=None,
):
    """Same signature, defaults and error behaviour as the reference's (:170-257).

    ``adjoint_options["vjp"] = fn``: the backward pass takes ``fn(t, y, cotangent) -> (f, vjp_t, vjp_y, *vjp_params)`` instead of
    differentiating ``func`` with torch autograd (``_vjp_through_hook``); ``adjoint_params`` then only fixes the parameter adjoints'
    shapes and order.  Tensors of another framework: ``AdjointProblem`` below (this function's result carries a TORCH autograd node)."""
    from ..utils import interop

    if interop.is_foreign(y0) or interop.is_foreign(t_span) or (isinstance(options, dict) and "from_dlpack" in options):
        raise TypeError("odeint_adjoint returns a tensor that carries a torch autograd node; for tensors of another framework wrap "
                        "paddlexde_amd.functional.AdjointProblem(...).forward / .backward in that framework's own autograd "
                        "function (INTEGRATION.md section B)")
    vjp = adjoint_options.get("vjp") if isinstance(adjoint_options, dict) else None
    if not torch.is_tensor(t_span):
        t_span = torch.as_tensor(t_span)
    plan, wanted = _prepare(func, y0, t_span, rtol=rtol, atol=atol, solver=solver, options=options, adjoint_rtol=adjoint_rtol,
                            adjoint_atol=adjoint_atol, adjoint_solver=adjoint_solver, adjoint_options=adjoint_options,
                            adjoint_params=adjoint_params, vjp=vjp, time_grad=t_span.requires_grad)

    def forward_solve(y_start, times):
        return odeint(func, y_start, times, solver=solver, rtol=rtol, atol=atol, options=options)

    return OdeintAdjointMethod.apply(plan, forward_solve, y0, t_span, *wanted)


class AdjointProblem:
    """The adjoint method as a framework-neutral (forward, backward) PAIR — what the autograd function of ANY framework wraps.

    The reference's ``OdeintAdjointMethod`` is a ``paddle.autograd.PyLayer`` whose backward differentiates ``func`` with Paddle itself
    (functional/odeint_adjoint.py:11-167, the vjp at :108-114).  The kernels below it need device pointers only, so the same
    arrangement works for a caller whose tensors are not torch's: the CALLER's framework supplies the vector-Jacobian product
    (``vjp``), this class supplies the two solves::

        prob = AdjointProblem(layer, vjp=paddle_vjp, adjoint_params=params, solver=Dopri5, rtol=1e-7, atol=1e-9,
                              from_dlpack=paddle.from_dlpack)
        ans = prob.forward(y0, t_span)                                   # the plain solve, nothing recorded (:37-43)
        adj_y0, grad_t, grads = prob.backward(t_span, ans, grad_ans)     # the sweep (:47-167); grads: one per adjoint parameter

    ``vjp(t, y, cotangent) -> (f, vjp_t, vjp_y, *vjp_params)`` receives and returns the caller's own tensors (``from_dlpack`` = that
    framework's importer; ``None`` for a zero gradient); ``func(t, y)`` likewise.  Without ``from_dlpack`` everything is torch.
    ``adjoint_params``: tensors (of either kind) that fix number, shapes and dtype of the parameter adjoints.  Arguments, defaults and
    error behaviour otherwise as ``odeint_adjoint``; results are tensors of the caller's framework.  INTEGRATION.md section B has the
    Paddle ``PyLayer`` (12 lines) and the ``paddle.autograd.grad`` hook (10 lines)."""

    def __init__(self, func, *, vjp, adjoint_params, solver=None, rtol=1e-7, atol=1e-9, options={"norm": _rms_norm},
                 adjoint_rtol=None, adjoint_atol=None, adjoint_solver=None, adjoint_options=None, from_dlpack=None):
        from ..utils import interop

        if not callable(vjp):
            raise TypeError("AdjointProblem needs vjp(t, y, cotangent) -> (f, vjp_t, vjp_y, *vjp_params)")
        options = dict(options or {})
        importer = options.pop("from_dlpack", None) if from_dlpack is None else from_dlpack
        options.pop("from_dlpack", None)
        self._importer = importer
        self._func = func if importer is None else interop.adapt_func(func, importer)
        self._vjp = vjp if importer is None else interop.adapt_vjp(vjp, importer)
        self._params = tuple(interop.to_torch(p).detach() for p in adjoint_params)
        self._solver, self._rtol, self._atol, self._options = solver, rtol, atol, options
        self._adjoint = dict(adjoint_rtol=adjoint_rtol, adjoint_atol=adjoint_atol, adjoint_solver=adjoint_solver,
                             adjoint_options=adjoint_options)
        self._plan = None

    def _out(self, x):
        return x if (x is None or self._importer is None) else self._importer(x)

    def _plan_for(self, y0, t_span, time_grad):
        key = (tuple(y0.shape), y0.dtype, y0.device, bool(time_grad))
        if self._plan is None or self._plan[0] != key:
            plan, _ = _prepare(self._func, y0, t_span, rtol=self._rtol, atol=self._atol, solver=self._solver, options=self._options,
                               adjoint_params=self._params, vjp=self._vjp, time_grad=time_grad, **self._adjoint)
            self._plan = (key, plan)
        return self._plan[1]

    def forward(self, y0, t_span):
        """``odeint(func, y0, t_span)`` with nothing recorded; also readies what the backward pass can ready ahead (captures happen
        here: on the caller's thread, outside any autograd engine)."""
        from ..utils import interop

        y0, t_span = interop.to_torch(y0), torch.as_tensor(interop.to_torch(t_span))
        with torch.no_grad():
            self._plan_for(y0, t_span, False)
            return self._out(odeint(self._func, y0, t_span, solver=self._solver, rtol=self._rtol, atol=self._atol, options=self._options))

    def backward(self, t_span, answer, grad_answer, t_requires_grad=False):
        """``(dL/dy0, dL/dt_span | None, [dL/dtheta ...])`` from ``dL/d answer`` — the reference's backward (:47-167) plus the
        gradient with respect to ``y0`` it drops (D6)."""
        from ..utils import interop

        t_span, answer, grad_answer = (interop.to_torch(x) for x in (t_span, answer, grad_answer))
        t_span = torch.as_tensor(t_span)
        fixed, n_times = _is_fixed(self._solver), len(t_span)
        if fixed:  # (fixed solvers fold time into axis -2: the state's shape is the answer's with that axis divided)
            y0_shape = tuple(answer.shape[:-2]) + (answer.shape[-2] // n_times, answer.shape[-1])
        else:
            y0_shape = tuple(answer.shape[1:])
        with torch.no_grad():
            y_first = _time_first(answer, y0_shape, n_times, fixed)[0]
            plan = self._plan_for(y_first, t_span, t_requires_grad)
            adj_y0, grad_t, adj_params = _sweep(plan, t_span, answer, grad_answer.contiguous(), self._params)
        return self._out(adj_y0), self._out(grad_t), [self._out(g) for g in adj_params]
