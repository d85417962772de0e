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

Only the forward `odeint` takes foreign tensors: `odeint_adjoint` differentiates `func` with torch autograd, which a foreign
framework's layer cannot serve (INTEGRATION.md section B shows the Paddle-side binding of the C ABI for that case).
Paddle itself is not installed in the build image; the adapter is exercised with a protocol-level stand-in (a class that exposes
nothing but `__dlpack__` / `__dlpack_device__` / `shape` / `dtype`): tests/test_gpu_kernels.py::test_foreign_tensors_through_dlpack.
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
