"""hipGraph capture of the user's ``func`` — for launch-bound problems whose func is many tiny framework ops.

``GraphedFunc(func)`` records ``func(t, y)`` (tensor or tuple state, static shapes, no host syncs) into a HIP graph
per input signature and replays it afterwards: one graph launch instead of one launch per framework op.  Inputs are
copied into static buffers, outputs live in static buffers; by default they are cloned before being returned
because the Runge-Kutta stages keep several results of ``func`` alive at once (the reference copies every result
into its ``k`` buffer, solver/base_adaptive_solver_rk.py:170).

Capture happens on the thread that calls ``prepare()`` / the first ``__call__`` — it must be the main thread:
capturing inside the autograd engine's worker thread (e.g. from a custom Function's backward) crashes the
runtime, so off the main thread an unseen signature is simply evaluated eagerly.

The adjoint's augmented dynamics (functional/odeint_adjoint.py:89-124: func + vjp through autograd) is the main
customer: ``odeint_adjoint(..., adjoint_options={"graph_func": True})`` pre-captures it in ``forward``.
"""
import threading

import torch


def _map(x, fn):
    if isinstance(x, (tuple, list)):
        return tuple(fn(a) for a in x)
    return fn(x)


class _Capture:
    __slots__ = ("graph", "t", "y", "out")


class GraphedFunc:
    def __init__(self, func, warmup=3, clone_outputs=True):
        self.func = func
        self.warmup = int(warmup)
        self.clone_outputs = bool(clone_outputs)
        self._captures = {}
        self.replays = 0
        self.captures = 0
        self.eager_calls = 0

    @staticmethod
    def _signature(t, y):
        ys = y if isinstance(y, (tuple, list)) else (y,)
        return (tuple(t.shape), t.dtype) + tuple((tuple(a.shape), a.dtype, str(a.device)) for a in ys)

    def prepare(self, t, y):
        """Capture the graph for this input signature now (call from the main thread)."""
        first = y[0] if isinstance(y, (tuple, list)) else y
        if not first.is_cuda:
            return None  # host tensors (test double): __call__ evaluates func directly
        key = self._signature(t, y)
        if key in self._captures:
            return self._captures[key]
        c = _Capture()
        c.t = t.detach().clone()
        c.y = _map(y, lambda a: a.detach().clone())
        # func may differentiate (the adjoint's vjp): keep the autograd engine on THIS thread while warming up and
        # capturing — a backward pass executed by the engine's device thread would issue HIP calls from a second
        # thread into a stream that is being captured (observed to crash hipStreamEndCapture intermittently)
        with torch.autograd.set_multithreading_enabled(False):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self.func(c.t, c.y)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            c.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(c.graph, capture_error_mode="thread_local"):
                c.out = self.func(c.t, c.y)
        self._captures[key] = c
        self.captures += 1
        return c

    def __call__(self, t, y):
        first = y[0] if isinstance(y, (tuple, list)) else y
        if not first.is_cuda:
            return self.func(t, y)  # nothing to capture on host tensors (test double)
        c = self._captures.get(self._signature(t, y))
        if c is None:
            if threading.current_thread() is not threading.main_thread():
                self.eager_calls += 1
                return self.func(t, y)
            c = self.prepare(t, y)
        c.t.copy_(t)
        if isinstance(y, (tuple, list)):
            for dst, src in zip(c.y, y):
                dst.copy_(src)
        else:
            c.y.copy_(y)
        c.graph.replay()
        self.replays += 1
        if self.clone_outputs:
            return _map(c.out, lambda a: a.clone())
        return c.out
