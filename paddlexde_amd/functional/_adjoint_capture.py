"""Captured forms of the adjoint sweep's work (``adjoint_options["graph_func"]`` / ``["interval_graph"]``): the augmented dynamics as
a replayed hipGraph per time-argument signature, cached on the module (or on the vjp hook's owner), and the re-armable interval
solvers built on it.  Everything here runs in the CALLER of the autograd node — main thread, outside the node (see
``_captured_dynamics``)."""
import threading
import weakref

import torch
import torch.nn as nn

from ..utils.ode_utils import native_norm_spec
from ._adjoint_dynamics import _ZERO_SINKS, _is_fixed, _make_augmented_dynamics, _make_functional_dynamics
from .odeint import _pack, _segment_layout

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


