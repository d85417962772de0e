"""Framework-neutral tensors at the `odeint` boundary (DLPack).

Every caller of the reference passes `paddle.Tensor`s and a `paddle.nn.Layer` (example/ode_demo.py:51,67;
paddlexde/functional/odeint.py:9-18).  The kernels below the boundary only need device pointers, shapes and dtypes, and PyTorch is
this package's allocator and stream owner — so a tensor of ANOTHER framework enters through the DLPack protocol, zero copy:

    from paddlexde_amd import odeint, Dopri5
    sol = odeint(layer, y0_paddle, t_paddle, solver=Dopri5, options={"norm": _rms_norm, "from_dlpack": paddle.from_dlpack})

  * `y0` / `t_span` may be any object with `__dlpack__` (device memory; `__cuda_array_interface__` objects work too): they are viewed as
    torch tensors without a copy (`torch.from_dlpack`);
  * `options["from_dlpack"]` is the CALLER's framework's importer (`paddle.from_dlpack`, `cupy.from_dlpack`, ...): the user's `func`
    then receives `(t, y)` as tensors of its own framework — views of the solver's buffers — and what it returns is viewed back; the
    solution is handed back through the same importer.  Without it, `func` receives torch tensors.

Training goes the same way.  The reference takes the vector-Jacobian product of `func` with the CALLER's framework
(`paddle.autograd.grad(outputs=func_eval, inputs=(t, y) + adjoint_params, grad_outputs=-adj_y)`, functional/odeint_adjoint.py:108-114);
here that product is a hook, `adjoint_options["vjp"] = fn` with `fn(t, y, cotangent) -> (f, vjp_t, vjp_y, *vjp_params)` on the caller's
own tensors (`adapt_vjp` below), and `functional.AdjointProblem` is the framework-neutral (forward, backward) pair a `PyLayer` of the
caller's framework wraps (INTEGRATION.md section B shows the 10-line `paddle.autograd.grad` hook and the PyLayer around it).
Paddle itself is not installed in the build image; the adapters are exercised with a protocol-level stand-in (a class that exposes
nothing but `__dlpack__` / `__dlpack_device__` / `shape` / `dtype`): tests/test_gpu_kernels.py::test_foreign_tensors_through_dlpack,
tests/_adjoint_cases.py::test_adjoint_vjp_hook_on_foreign_tensors_reproduces_config3_gradients.

Stream contract (both adapters): the caller's framework must enqueue its kernels on the stream this package runs on (torch's current
stream: share it, or make the framework's current stream that one) — DLPack's `stream` argument orders the hand-over of each tensor,
not the framework's later kernels.
"""
import torch


def is_foreign(x):
    return not torch.is_tensor(x) and (hasattr(x, "__dlpack__") or hasattr(x, "__cuda_array_interface__"))


def to_torch(x):
    """A torch view (no copy) of a tensor of any framework that speaks DLPack / the CUDA array interface; torch tensors pass."""
    if torch.is_tensor(x):
        return x
    if hasattr(x, "__dlpack__"):
        return torch.from_dlpack(x)
    if hasattr(x, "__cuda_array_interface__"):
        return torch.as_tensor(x, device="cuda")
    raise TypeError("expected a tensor (torch, or any object with __dlpack__ / __cuda_array_interface__), got {}".format(type(x).__name__))


def adapt_func(func, from_dlpack):
    """`func` of the caller's framework -> a callable on torch tensors: inputs are exported with `from_dlpack` (the caller's
    importer consumes torch's `__dlpack__`), the result is viewed back as a torch tensor.  Tuples pass member-wise."""

    def export(x):
        if isinstance(x, (tuple, list)):
            return tuple(export(v) for v in x)
        if not torch.is_tensor(x):
            return x
        return from_dlpack(x.detach())

    def back(x):
        if isinstance(x, (tuple, list)):
            return tuple(back(v) for v in x)
        return to_torch(x)

    def torch_func(t, y):
        return back(func(export(t), export(y)))

    torch_func.__wrapped__ = func
    return torch_func


def adapt_vjp(vjp, from_dlpack):
    """The caller's `vjp(t, y, cotangent) -> (f, vjp_t, vjp_y, *vjp_params)` -> the same contract on torch tensors: the three inputs
    are exported with the caller's importer, every returned tensor is viewed back; `None` members (a gradient the caller's framework
    did not produce) pass as `None`."""

    def export(x):
        return from_dlpack(x.detach()) if torch.is_tensor(x) else x

    def torch_vjp(t, y, cotangent):
        out = vjp(export(t), export(y), export(cotangent))
        return tuple(None if v is None else to_torch(v) for v in out)

    torch_vjp.__wrapped__ = vjp
    torch_vjp._from_dlpack = from_dlpack
    return torch_vjp
