"""The augmented dynamics of the adjoint sweep (reference: paddlexde/functional/odeint_adjoint.py:89-124) — ``func`` evaluated and
differentiated at ``(t, y)`` with the cotangent ``-adj_y``: by torch autograd, by the caller's own vector-Jacobian product
(``adjoint_options["vjp"]``), or as a functional closure a hipGraph can record.  Used by ``odeint_adjoint.py`` (the sweep) and
``_adjoint_capture.py`` (the captured form)."""
import collections
import weakref

import torch

from ..solver.base_fixed_solver import FixedSolver
from .odeint import ScaledTuple

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


