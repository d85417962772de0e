"""``odeint`` — forward entry point (reference: paddlexde/functional/odeint.py:9-35).

Same signature, defaults and output layouts as the reference (fixed solvers: time concatenated on
axis -2; adaptive solvers: ``[T, *y0.shape]`` — SURVEY D3).  ``xde.format`` is the identity (D1).
A tuple/list ``y0`` is flattened into one padded buffer, integrated, and unflattened time-first
(the torchdiffeq behaviour the reference's commented-out code intended).
"""
from typing import Union

import torch

from ..solver.base_fixed_solver import FixedSolver
from ..utils.ode_utils import _rms_norm
from ..xde import BaseODE


def odeint(
    func: callable,
    y0: Union[tuple, torch.Tensor],
    t_span,
    solver,
    *,
    rtol=1e-7,
    atol=1e-9,
    options: object = {"norm": _rms_norm},
):
    """Integrate ``dy/dt = func(t, y), y(t[0]) = y0`` and return y at every ``t_span`` point.

    ``y0`` / ``t_span`` may be tensors of another framework (anything with ``__dlpack__``): see utils/interop.py;
    ``options["from_dlpack"]`` = that framework's importer makes ``func`` receive, and the call return, its own tensors."""
    from ..utils import interop

    importer = None
    if isinstance(options, dict) and "from_dlpack" in options:
        options = dict(options)
        importer = options.pop("from_dlpack")
    if importer is not None or interop.is_foreign(y0) or interop.is_foreign(t_span):
        y0 = tuple(interop.to_torch(v) for v in y0) if isinstance(y0, (tuple, list)) else interop.to_torch(y0)
        t_span = interop.to_torch(t_span) if interop.is_foreign(t_span) else t_span
        if importer is not None:
            inner = interop.adapt_func(func, importer)
            with torch.no_grad():  # a foreign framework's func records no torch graph (training: functional.AdjointProblem + the caller's vjp)
                sol = odeint(inner, y0, t_span, solver, rtol=rtol, atol=atol, options=options)
            return tuple(importer(v) for v in sol) if isinstance(sol, tuple) else importer(sol)
    if not torch.is_tensor(t_span):
        t_span = torch.as_tensor(t_span)
    if _wants_autograd(func, y0, t_span, solver):
        # The reference's adaptive solvers are eager framework ops, so `odeint(..., solver=Dopri5)` + `loss.backward()`
        # trains there.  Here the adaptive step kernels read dt from device memory and record no autograd graph; rather than
        # hand back a silently detached result, the call is served by the adjoint method (gradients w.r.t. y0, func's
        # parameters and t_span).  Deliberate deviation: continuous adjoint instead of back-propagation through the steps.
        from .odeint_adjoint import odeint_adjoint

        _warn_once_autograd_route()
        params = None if isinstance(func, torch.nn.Module) else ()
        return odeint_adjoint(func, y0, t_span, rtol=rtol, atol=atol, solver=solver, options=options, adjoint_params=params)
    if isinstance(y0, (tuple, list)):
        return _odeint_tuple(func, tuple(y0), t_span, solver, rtol=rtol, atol=atol, options=options)

    xde = BaseODE(func, y0=y0, t_span=t_span)

    s = solver(xde=xde, y0=xde.y0, rtol=rtol, atol=atol, **options)
    solution = s.integrate(t_span)

    solution = xde.format(solution)

    return solution


_ROUTE_WARNED = False


def _warn_once_autograd_route():
    global _ROUTE_WARNED
    if not _ROUTE_WARNED:
        _ROUTE_WARNED = True
        import warnings

        warnings.warn(
            "paddlexde_amd: odeint() with an adaptive solver was called with gradients enabled on y0 / func's parameters / "
            "t_span; the adaptive kernels record no autograd graph, so the call is served by odeint_adjoint (continuous "
            "adjoint). Call odeint_adjoint directly to choose its options, or wrap inference in torch.no_grad().",
            stacklevel=4)


def _wants_autograd(func, y0, t_span, solver):
    """An adaptive solve whose result the caller may differentiate: grad mode is on and y0, t_span or a parameter of func
    requires grad.  (Fixed solvers record their own graph, solver/_autograd.py; tuple states go through untouched.)"""
    if not torch.is_grad_enabled() or not torch.is_tensor(y0):
        return False
    if isinstance(solver, type) and issubclass(solver, FixedSolver):
        return False
    if y0.requires_grad or t_span.requires_grad:
        return True
    return isinstance(func, torch.nn.Module) and any(p.requires_grad for p in func.parameters())


def _segment_layout(tensors):
    """Element offsets of each tensor in the flat buffer; every segment starts 16-byte aligned."""
    dtype = tensors[0].dtype
    for x in tensors:
        if x.dtype != dtype:
            dtype = torch.promote_types(dtype, x.dtype)
    width = 16 // torch.empty((), dtype=dtype).element_size()
    segs, off = [], 0
    for x in tensors:
        n = x.numel()
        segs.append((off, n))
        off += -(-max(n, 1) // width) * width
    return dtype, segs, off


def _pack(tensors, segs, total, dtype, device):
    """The members of a tuple state in one flat buffer (16-byte-aligned segments, pads zero).  On the device this is ONE launch
    (xde_pack_segments) instead of a fill + one copy per member — odeint_adjoint's augmented dynamics packs its 7-member result on
    every evaluation; the framework-op path stays for what the kernel does not take (mixed dtypes, strided members, members that
    carry an autograd graph, the CPU test double)."""
    scales = getattr(tensors, "scales", None)  # ScaledTuple: member s counts as tensors[s] * scales[s]
    tensors = list(tensors)
    if (torch.device(device).type == "cuda" and len(tensors) == len(segs)
            and not (torch.is_grad_enabled() and any(x.requires_grad for x in tensors))):
        from .. import _hip

        be = _hip.get_backend()
        if hasattr(be, "pack_segments"):
            flat = torch.empty(total, dtype=dtype, device=device)
            if be.pack_segments(flat, tensors, segs, scales):
                return flat
    flat = torch.zeros(total, dtype=dtype, device=device)
    for i, (x, (s, n)) in enumerate(zip(tensors, segs)):
        if n:
            flat[s : s + n].copy_(x.reshape(-1))
            if scales is not None and scales[i] != 1.0:
                flat[s : s + n].mul_(scales[i])
    return flat


class ScaledTuple(tuple):
    """A tuple of tensors whose member ``s`` stands for ``self[s] * self.scales[s]``: `_pack` applies the factors while it writes the
    flat buffer (in the pack kernel: for free).  Only ever produced and consumed inside this package."""

    scales = None

    @classmethod
    def of(cls, members, scales):
        out = cls(members)
        out.scales = tuple(float(x) for x in scales)
        assert len(out.scales) == len(out)
        return out


def _odeint_tuple(func, y0, t_span, solver, *, rtol, atol, options):
    shapes = [tuple(x.shape) for x in y0]
    dtype, segs, total = _segment_layout(y0)
    device = y0[0].device
    flat0 = _pack(y0, segs, total, dtype, device)
    sol = _odeint_packed(func, flat0, segs, shapes, t_span, solver, rtol=rtol, atol=atol, options=options)
    T = sol.shape[0]
    return tuple(sol[:, st : st + n].reshape((T,) + shape) for (st, n), shape in zip(segs, shapes))


def _odeint_packed(func, flat0, segs, shapes, t_span, solver, *, rtol, atol, options):
    """Integrate a tuple state that is ALREADY in the flat padded layout of ``_segment_layout`` (pads zero); returns the
    flat solution ``[T, total]``.  odeint_adjoint's backward keeps its augmented state in this form between intervals."""
    dtype, device, total = flat0.dtype, flat0.device, flat0.numel()
    fixed = isinstance(solver, type) and issubclass(solver, FixedSolver)
    options = dict(options)
    # private hook (odeint_adjoint): a ready-made dynamics on the FLAT state with this exact layout, e.g. a
    # HIP-graph-captured one; it replaces the unpack -> func -> pack wrapper below
    ready_flat = options.pop("_xde_flat_func", None)

    def unpack(flat):
        return tuple(flat[s : s + n].view(shape) for (s, n), shape in zip(segs, shapes))

    if fixed:
        # fixed solvers stack time on axis -2: a [1, total] state gives a time-first [T, total] result
        def flat_func(t, y):
            return _pack(func(t, unpack(y[0])), segs, total, dtype, device)[None, :]

        y_in = flat0[None, :]
        opts = options
    else:
        def flat_func(t, y):
            return _pack(func(t, unpack(y)), segs, total, dtype, device)

        y_in = flat0
        opts = options
        opts["_xde_segments"] = segs
        opts["_xde_segment_shapes"] = shapes
    if ready_flat is not None:
        flat_func = ready_flat

    xde = BaseODE(flat_func, y0=y_in, t_span=t_span)
    s = solver(xde=xde, y0=xde.y0, rtol=rtol, atol=atol, **opts)
    return s.integrate(t_span)  # [T, total]
