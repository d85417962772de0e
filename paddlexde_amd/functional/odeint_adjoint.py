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
Deviations documented in SURVEY.md: D4 (tuple state flattened), D5 (reverse-time intervals run natively with a signed dt), D6 (the
gradient w.r.t. ``y0`` is returned instead of ``None``)."""
import os
import threading
import warnings
import collections
import weakref
from typing import Any, NamedTuple, Optional

import torch
import torch.nn as nn

from ..solver.base_fixed_solver import FixedSolver
from ..utils.ode_utils import _rms_norm, native_norm_spec
from .odeint import ScaledTuple, _odeint_packed, _pack, _segment_layout, odeint

_N_LEADING = 3  # adj_t, y, adj_y come first in the augmented state; the parameter adjoints follow


def _is_fixed(solver):
    return isinstance(solver, type) and issubclass(solver, FixedSolver)


def _time_first(x, y0_shape, n_times, fixed):
    """View of a solution / its gradient with time on axis 0 (the fixed-step layout folds time into axis -2)."""
    if not fixed:
        return x
    lead, rows, width = tuple(y0_shape[:-2]), y0_shape[-2], y0_shape[-1]
    return x.reshape(lead + (n_times, rows, width)).movedim(len(lead), 0)


def _group_sum(tensors, pg):
    """Sum a list of small tensors over the batch-sharding process group with ONE all-reduce; returns new tensors."""
    import torch.distributed as dist

    group = None if pg is True else pg
    tensors = [x.contiguous() for x in tensors]
    flat = torch.cat([x.reshape(-1) for x in tensors]) if len(tensors) != 1 else tensors[0].reshape(-1).clone()
    staged = flat.is_cuda and dist.get_backend(group) == "gloo"  # rehearsal transport: gloo reduces on the host
    buf = flat.cpu() if staged else flat
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    if staged:
        flat.copy_(buf)
    outs, off = [], 0
    for x in tensors:
        outs.append(flat[off : off + x.numel()].view(x.shape))
        off += x.numel()
    return outs


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
# augmented dynamics                                                          (reference: odeint_adjoint.py:89-124)
# ----------------------------------------------------------------------------------------------------------------------
def _vjp_of(evaluate, t, y, wrt_params, cotangent, time_grad, retain):
    """``f = evaluate(t, y)`` and ``cotangent^T df/d(t, y, params)``; missing gradients are zeros (:116-122)."""
    with torch.enable_grad():
        # fresh autograd leaves that ALIAS the inputs (the reference copies them, paddle.assign: two more launches per evaluation);
        # nothing writes to either between here and the grad call below, and no graph outlives this function
        t_const = t.detach()
        t_var = t.detach().requires_grad_(True)
        y_var = y.detach().requires_grad_(True)
        # dL/dt is only resolved when asked for: func then sees a time it can be differentiated by
        f = evaluate(t_var if time_grad else t_const, y_var)
        grads = torch.autograd.grad(f, (t_var, y_var) + tuple(wrt_params), cotangent, allow_unused=True, retain_graph=retain)
    filled = [_zeros_like(x) if g is None else g for x, g in zip((t_var, y_var) + tuple(wrt_params), grads)]
    return f.detach(), filled[0], filled[1], filled[2:]


def _vjp_through_hook(hook, t, y, wrt_params, cotangent):
    """The same four results from the CALLER's vector-Jacobian product, ``adjoint_options["vjp"]``: ``hook(t, y, cotangent) -> (f,
    vjp_t, vjp_y, *vjp_params)`` with ``vjp_* = cotangent^T df/d*`` (linear in the cotangent), one entry per adjoint parameter in
    their order; ``None`` stands for a gradient that is identically zero (the time adjoint of an autonomous func).  This is where a
    func of another framework is differentiated BY that framework (the reference's own arrangement: ``paddle.autograd.grad``,
    :108-114) — or where a hand-written / fused vjp replaces ~30 autograd launches; nothing of torch's autograd runs."""
    out = tuple(hook(t, y, cotangent))
    if len(out) != _N_LEADING + len(wrt_params):
        raise ValueError("adjoint_options['vjp'] must return (f, vjp_t, vjp_y, *vjp_params) with one entry per adjoint parameter: "
                         "expected {} values, got {}".format(_N_LEADING + len(wrt_params), len(out)))
    f, vjp_t, vjp_y, *vjp_params = out
    if f is None or vjp_y is None:
        raise ValueError("adjoint_options['vjp'] returned None for f or vjp_y")
    if tuple(f.shape) != tuple(y.shape) or tuple(vjp_y.shape) != tuple(y.shape):
        raise ValueError("adjoint_options['vjp']: f and vjp_y must have the state's shape {}, got {} and {}".format(
            tuple(y.shape), tuple(f.shape), tuple(vjp_y.shape)))
    for p, g in zip(wrt_params, vjp_params):
        if g is not None and g.numel() != p.numel():
            raise ValueError("adjoint_options['vjp']: a parameter vjp has {} elements, its parameter {}".format(g.numel(), p.numel()))
    vjp_t = _zeros_like(t) if vjp_t is None else vjp_t.reshape(())
    filled = [_zeros_like(p) if g is None else g.detach() for p, g in zip(wrt_params, vjp_params)]
    return f.detach(), vjp_t.detach(), vjp_y.detach(), filled


# (shape, dtype, device) -> a zero tensor that is only ever READ: the stand-in for a gradient autograd did not produce (the time
# adjoint of an autonomous func, on every evaluation).  A bounded, least-recently-used cache (a handful of shapes per model; a
# process that walks through many models or batch shapes does not keep every zero it ever needed).  A captured HIP graph bakes in
# the raw ADDRESS of the zero it was recorded with: while a capture records, every zero handed out is also appended to that
# capture's own keep-alive list (_ZERO_SINKS), so an entry evicted from this cache lives exactly as long as a graph that reads it.
_ZEROS = collections.OrderedDict()
_ZEROS_MAX = 32
_ZERO_SINKS = []  # stack of keep-alive lists of the dynamics currently being evaluated for capture


def _zeros_like(x):
    key = (tuple(x.shape), x.dtype, x.device)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(key[0], dtype=x.dtype, device=x.device)
        while len(_ZEROS) > _ZEROS_MAX:
            _ZEROS.popitem(last=False)
    else:
        _ZEROS.move_to_end(key)
    if _ZERO_SINKS and not any(z is k for k in _ZERO_SINKS[-1]):
        _ZERO_SINKS[-1].append(z)
    return z


def _negated_vjp(vjp_t, f, vjp_y, vjp_params):
    """The augmented dynamics' value ``(-vjp_t, f, -vjp_y, -vjp_theta...)`` where the vjp was taken with the cotangent ``+adj_y``:
    the reference's ``-adj_y`` (:108-114) is a launch of its own, a vjp is linear in its cotangent (exactly: every operation of a
    backward graph is sign-symmetric in IEEE arithmetic), and the pack that follows applies the sign for free."""
    members = (vjp_t, f, vjp_y, *vjp_params)
    return ScaledTuple.of(members, [-1.0, 1.0, -1.0] + [-1.0] * len(vjp_params))


def _make_augmented_dynamics(func, adjoint_params, t_requires_grad, pg=None, reduce_params=False, vjp=None):
    """``d/dt (adj_t, y, adj_y, adj_theta) = (vjp_t, f, vjp_y, vjp_theta)`` with the cotangent ``-adj_y`` (taken as ``+adj_y`` and
    negated while the result is packed, see ``_negated_vjp``); only ``y`` and ``adj_y`` are read from the state.

    Batch-sharded run (``pg``): ``f`` and ``vjp_y`` are per-row quantities of this rank's rows, ``vjp_t`` and ``vjp_theta`` are
    sums over rows.  Where the step control looks at them — ``adj_t`` when time gradients are wanted, the parameter adjoints under
    the default adjoint norm (``reduce_params``) — they are summed over the group here, so that every rank integrates the GLOBAL
    ``adj_t`` / ``adj_theta`` and the all-reduced norm is exactly the unsharded one.

    ``vjp``: the caller's vector-Jacobian product instead of torch autograd (``_vjp_through_hook``)."""

    def augmented_dynamics(t, y_aug):
        if vjp is not None:  # the caller's own vector-Jacobian product (adjoint_options["vjp"])
            f, vjp_t, vjp_y, vjp_params = _vjp_through_hook(vjp, t, y_aug[1], adjoint_params, y_aug[2])
        else:
            f, vjp_t, vjp_y, vjp_params = _vjp_of(func, t, y_aug[1], adjoint_params, y_aug[2], t_requires_grad, retain=True)
        if pg is not None:
            shared = ([vjp_t] if t_requires_grad else []) + (list(vjp_params) if reduce_params else [])
            if shared:
                shared = _group_sum(shared, pg)
                if t_requires_grad:
                    vjp_t, shared = shared[0], shared[1:]
                if reduce_params:
                    vjp_params = shared
        return _negated_vjp(vjp_t, f, vjp_y, vjp_params)

    return augmented_dynamics


def _make_functional_dynamics(func, adjoint_params, t_requires_grad):
    """The augmented dynamics for HIP-graph capture: identical arithmetic, but the vjp is taken w.r.t. fresh detached aliases of
    the parameters (substituted with torch.func.functional_call) instead of the parameter leaves themselves.  After a user's
    loss.backward() the real leaves own AccumulateGrad nodes bound to the default stream, and differentiating w.r.t. them inside a
    later stream capture makes the engine synchronise with the default stream — which crashes the capture.  Needs ``func`` to be an
    nn.Module whose parameters are the adjoint parameters."""
    names = {id(p): n for n, p in func.named_parameters()}
    try:
        order = [names[id(p)] for p in adjoint_params]
    except KeyError:
        raise NotImplementedError(
            "adjoint_options['graph_func'] needs func to be an nn.Module and adjoint_params to be (a subset of) its parameters"
        )
    # the capture cache is keyed weakly by the module (_GRAPH_CACHE): what it stores must not keep the module alive
    func_ref = weakref.ref(func)
    del func

    def augmented_dynamics(t, y_aug):
        module = func_ref()
        if module is None:
            raise RuntimeError("the module this captured dynamics was built for no longer exists")
        aliases = tuple(p.detach().requires_grad_(True) for p in adjoint_params)  # same storage, no copy
        evaluate = lambda t_, y_: torch.func.functional_call(module, dict(zip(order, aliases)), (t_, y_))  # noqa: E731
        f, vjp_t, vjp_y, vjp_params = _vjp_of(evaluate, t, y_aug[1], aliases, y_aug[2], t_requires_grad, retain=False)
        return _negated_vjp(vjp_t, f, vjp_y, vjp_params)

    return augmented_dynamics


# ----------------------------------------------------------------------------------------------------------------------
# captured dynamics (adjoint_options["graph_func"])
# ----------------------------------------------------------------------------------------------------------------------
_GRAPH_CACHE = weakref.WeakKeyDictionary()  # func module -> {signature: GraphedFunc}; captures are reused across calls

MAX_GRAPHS_PER_MODULE = 8  # captured dynamics kept per module (each holds static buffers of the state's size)
AUTO_GRAPH_FUNC_MAX_BYTES = 8 << 20  # "auto": states above this are bandwidth-bound, the launches are not the cost
AUTO_GRAPH_FUNC_MIN_INTERVALS = 4  # "auto": output intervals needed to amortise a first capture


class _NoGraph:
    """Cache marker: capturing the dynamics of this module / signature failed once; it runs eagerly."""

    refused = True


def _auto_graph_func(func, y0, t_span, adjoint_params, adjoint_options):
    """Whether adjoint_options["graph_func"] = "auto" captures the augmented dynamics for this call."""
    if not (isinstance(func, nn.Module) and torch.is_tensor(y0) and y0.is_cuda):
        return False
    if threading.current_thread() is not threading.main_thread() or torch.cuda.is_current_stream_capturing():
        return False
    if adjoint_options.get("process_group") is not None:
        return False
    if y0.numel() * y0.element_size() > AUTO_GRAPH_FUNC_MAX_BYTES or len(adjoint_params) == 0:
        return False
    own = {id(p) for p in func.parameters()}
    if any(id(p) not in own for p in adjoint_params):
        return False
    return len(t_span) - 1 >= AUTO_GRAPH_FUNC_MIN_INTERVALS or func in _GRAPH_CACHE


def _graph_time_examples(adjoint_method, adjoint_options, t_span, y0):
    """The time arguments (shape, dtype) the adjoint solver will hand to func, for pre-capturing its HIP graph."""
    dev = y0.device
    if _is_fixed(adjoint_method):
        tt = t_span.dtype if t_span.dtype in (torch.float32, torch.float64) else torch.float32
        return [torch.zeros(1, dtype=tt, device=dev)]
    time_dtype = adjoint_options.get("dtype", torch.float32)
    dtypes = {time_dtype, y0.dtype, torch.promote_types(time_dtype, y0.dtype)}
    return [torch.zeros((), dtype=d, device=dev) for d in dtypes]


# captured interval solves: the whole 2-point solve of one output interval (initial-step heuristic + first attempted step) as one
# hipGraph on a solver that is kept across intervals and backward passes (solver/base_adaptive_solver_rk.py: intervals_prepare)
_INTERVAL_OPTION_KEYS = ("norm", "dtype", "safety", "ifactor", "dfactor", "min_step", "max_step", "max_num_steps", "controller",
                         "pi_beta", "pipeline", "process_group", "reuse_f0")
_FIXED_INTERVAL_OPTION_KEYS = ("norm", "interp", "pipeline", "variant", "process_group")  # (a fixed grid: one STEP per interval)
MAX_INTERVAL_SOLVERS = 4  # per captured dynamics (tolerances x solver x direction)


class _NoIntervals:
    """Cache entry: the interval solve could not be captured for this key (``reason`` says why); one ordinary solve per interval."""

    def __init__(self, reason=""):
        self.reason = reason


class _IntervalSolver:
    def __init__(self, solver):
        self.solver = solver
        self.lock = threading.Lock()  # one sweep at a time owns the static buffers (a second, concurrent one solves per interval)


def _interval_key(solver, rtol, atol, options, direction, t_dtype=None):
    """Cache key of the captured interval solve for these solver options, or None when they rule it out."""
    if not isinstance(solver, type):
        return None
    if os.environ.get("XDE_INTERVAL_GRAPH", "1") == "0":
        return None
    fixed = _is_fixed(solver)
    items = []
    for k, v in options.items():
        if k not in (_FIXED_INTERVAL_OPTION_KEYS if fixed else _INTERVAL_OPTION_KEYS):
            return None
        if k == "norm":
            if fixed:
                continue  # (a fixed grid has no step control: the norm is never called)
            v = native_norm_spec(v)
            if v is None:
                return None
        elif k == "process_group":
            if v is not None:
                return None
        elif k == "pipeline":
            if v not in ("auto", "sync") and not (fixed and v == "graph"):  # (every attempt is resolved before the next: "sync")
                return None
            continue
        elif k == "interp":
            if v != "linear":
                return None
            continue
        elif k == "reuse_f0":
            if not v:
                return None
            continue
        try:
            hash(v)
        except TypeError:
            return None
        items.append((k, v))
    if fixed and t_dtype not in (torch.float32, torch.float64):
        return None
    if fixed:  # (one step per interval: its direction is in the data; its times are handed to func in the output times' dtype)
        return (solver, "fixed", str(t_dtype), tuple(sorted(items, key=lambda kv: kv[0])))
    return (solver, float(rtol), float(atol), int(direction), tuple(sorted(items, key=lambda kv: kv[0])))


def _first_sweep_interval(t_host):
    """The backward sweep's first interval that is not empty — ``(t[i], t[i-1])`` walking back from the end — or None when every
    output time is the same.  Its direction is the sweep's: an output time repeated at the END (``t = [0, 1, 1]``) must not make a
    backward sweep look like a forward one."""
    times = t_host.tolist()
    for i in range(len(times) - 1, 0, -1):
        if times[i] != times[i - 1]:
            return (times[i], times[i - 1])
    return None


def _prepare_intervals(graphed, flat_ex, segs, shapes, t_span, adjoint_solver, rtol, atol, adjoint_options):
    """Build (once per key) the re-armable solver the backward sweep runs its intervals on.  Called where the dynamics is captured:
    on the main thread, outside the autograd node."""
    from ..solver._common import direction_of
    from ..xde.base_ode import BaseODE

    opts = {k: v for k, v in adjoint_options.items() if k not in ("_replay_intervals", "interval_graph")}
    if adjoint_options.get("_replay_intervals") is not None or adjoint_options.get("interval_graph", True) is False:
        return
    if len(t_span) < 2 or _interval_key(adjoint_solver, rtol, atol, opts, 1, t_span.dtype) is None:
        return
    t_host = t_span.detach().to("cpu")
    span = _first_sweep_interval(t_host)
    if span is None:
        return
    key = _interval_key(adjoint_solver, rtol, atol, opts, direction_of(span), t_host.dtype)
    cache = graphed.__dict__.setdefault("_intervals", {})
    if key in cache:
        return
    try:
        opts.pop("reuse_f0", None)
        opts.pop("pipeline", None)
        t_ex = torch.tensor(span, dtype=t_host.dtype)
        if _is_fixed(adjoint_solver):
            s = adjoint_solver(xde=BaseODE(graphed.func, y0=flat_ex, t_span=t_ex), y0=flat_ex, rtol=rtol, atol=atol, **opts)
        else:
            s = adjoint_solver(xde=BaseODE(graphed.func, y0=flat_ex, t_span=t_ex), y0=flat_ex, rtol=rtol, atol=atol, reuse_f0=True,
                               _xde_segments=segs, _xde_segment_shapes=shapes, **opts)
        if not (hasattr(s, "intervals_supported") and s.intervals_supported()):
            cache[key] = _NoIntervals("the solver's options rule it out (intervals_supported)")
            return
        if _is_fixed(adjoint_solver):
            s.intervals_prepare(span, t_host.dtype if t_host.dtype in (torch.float32, torch.float64) else torch.float32)
        else:
            s.intervals_prepare(span)
        entry = _IntervalSolver(s)
    except Exception as e:  # this solve cannot be captured: per-interval solves, and no second attempt for this key
        entry = _NoIntervals("{}: {}".format(type(e).__name__, e))
    while len(cache) >= MAX_INTERVAL_SOLVERS:
        cache.pop(next(iter(cache)))
    cache[key] = entry


def _weak_hook(vjp):
    """``vjp`` behind a weak reference, for a capture that is cached ON the hook's owner (what the cache stores must not keep its own
    key alive); the backward plan holds the hook itself for as long as a backward pass can still run."""
    inner = getattr(vjp, "__wrapped__", None)  # utils.interop.adapt_vjp(hook, importer): the user's hook is the thing to watch
    importer = getattr(vjp, "_from_dlpack", None)
    target = vjp if inner is None else inner
    try:
        ref = weakref.WeakMethod(target) if hasattr(target, "__self__") else weakref.ref(target)
    except TypeError:
        return vjp  # (not weakly referenceable: such an owner never enters the module-wide cache either)

    def call(t, y, cotangent):
        hook = ref()
        if hook is None:
            raise RuntimeError("the vjp hook this captured dynamics was built for no longer exists")
        if inner is not None:
            from ..utils.interop import adapt_vjp

            hook = adapt_vjp(hook, importer)
        return hook(t, y, cotangent)

    return call


def _graph_cache_owner(func, vjp):
    """The object a module-wide capture cache hangs on (weakly): the module, or — with a vjp hook — the hook (its instance, for a bound
    method: the method object itself is made anew on every attribute access)."""
    if vjp is None:
        return func
    hook = getattr(vjp, "__wrapped__", vjp)
    return getattr(hook, "__self__", hook)


def _captured_dynamics(func, y0, t_span, adjoint_solver, adjoint_options, adjoint_params, rtol=None, atol=None, vjp=None):
    """Resolve ``adjoint_options["graph_func"]`` and return the captured FLAT augmented dynamics, or None for the eager one.

    The augmented dynamics (func forward + autograd vjp, ~30 eager launches) is captured into one HIP graph per time-argument
    signature and replayed (config 3's backward: 105 -> 34 ms).  True (or a dict that caches captures across calls) forces it;
    False switches it off; absent / "auto" (the default) uses it when it pays and is safe: a small state (launch-bound), an nn.Module
    func whose parameters are the adjoint parameters, several output intervals to amortise the capture over (or a capture already
    cached for this module), the main thread, no capture in progress, no per-evaluation all-reduce — and falls back to the eager
    dynamics if the capture fails.  The capture has to happen in the caller of the autograd node — on the calling thread and outside
    the node: capturing from the engine's worker thread (where backward runs), or inside its forward while the parameters are its
    inputs, crashes the runtime.

    With a vjp hook (``vjp``) the captured thing is the hook's own launches; "auto" then means OFF — whether another framework's
    kernels land on the capturing stream, and survive a replay, is the caller's knowledge: ``graph_func=True`` states it."""
    mode = adjoint_options.pop("graph_func", "auto")
    forced = mode is True or isinstance(mode, dict)
    if mode == "auto":
        mode = vjp is None and _auto_graph_func(func, y0, t_span, adjoint_params, adjoint_options)
    if not mode:
        return None
    from ..utils.graphed import GraphedFunc

    if vjp is None and not isinstance(func, nn.Module):
        raise NotImplementedError("adjoint_options['graph_func'] needs func to be an nn.Module (or a vjp hook)")
    if isinstance(mode, dict):
        cache = mode
    else:
        try:
            cache = _GRAPH_CACHE.setdefault(_graph_cache_owner(func, vjp), {})
        except TypeError:  # an owner that cannot be referenced weakly: captures live for this call only
            cache = {}
    time_grad = bool(t_span.requires_grad)
    fixed = _is_fixed(adjoint_solver)
    # (the captured kernels address the parameters' storage: a parameter whose storage was swapped — `p.data = ...` — needs a new
    # capture, an in-place update such as an optimiser step does not)
    key = ("aug-flat" if vjp is None else "aug-flat-hook", tuple(y0.shape), y0.dtype, str(y0.device), time_grad, fixed,
           tuple((id(p), p.data_ptr()) for p in adjoint_params))
    graphed = cache.get(key)
    if isinstance(graphed, _NoGraph):
        return None
    # the augmented state in the flat, 16-byte-segment layout the backward will use
    example = [torch.zeros([], dtype=y0.dtype, device=y0.device), y0.detach(), torch.zeros_like(y0)] + [torch.zeros_like(p) for p in adjoint_params]
    adt, segs, total = _segment_layout(example)
    if graphed is None:
        if vjp is None:
            dyn = _make_functional_dynamics(func, adjoint_params, time_grad)
        else:
            dyn = _make_augmented_dynamics(None, adjoint_params, time_grad, vjp=_weak_hook(vjp))
        (s1, n1), (s2, n2) = segs[1], segs[2]
        yshape, dev = tuple(y0.shape), y0.device

        keep = []  # the shared zeros this dynamics reads (their addresses end up inside the captured graph): alive as long as it is

        def flat_dynamics(t, yf):
            # unpack views -> func + vjp -> pack, all inside ONE captured graph
            v = yf[0] if fixed else yf
            _ZERO_SINKS.append(keep)
            try:
                k = _pack(dyn(t, (None, v[s1 : s1 + n1].view(yshape), v[s2 : s2 + n2].view(yshape))), segs, total, adt, dev)
            finally:
                _ZERO_SINKS.pop()
            return k[None, :] if fixed else k

        graphed = GraphedFunc(flat_dynamics, clone_outputs=True)
        graphed._keepalive = keep
        while len(cache) >= MAX_GRAPHS_PER_MODULE:  # a loop over many batch shapes must not pile up captures (oldest first)
            cache.pop(next(iter(cache)))
        cache[key] = graphed
    flat_ex = _pack(example, segs, total, adt, y0.device)
    flat_ex = flat_ex[None, :] if fixed else flat_ex
    try:
        for t_ex in _graph_time_examples(adjoint_solver, adjoint_options, t_span, y0):
            graphed.prepare(t_ex, flat_ex)
    except Exception:
        if forced:
            raise
        # "auto": this func cannot be captured (host synchronisation, unsupported op, ...): eager dynamics, and no second attempt
        # for this module and signature
        cache[key] = _NoGraph()
        return None
    if graphed.refused:
        return None
    if rtol is not None and not time_grad:
        _prepare_intervals(graphed, flat_ex, segs, [tuple(x.shape) for x in example], t_span, adjoint_solver, rtol, atol,
                           adjoint_options)
    return graphed


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
