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
import ctypes as C
import threading

import torch

# hipGraphNodeType
_NODE_KERNEL, _NODE_MEMCPY, _NODE_MEMSET = 0, 1, 2
_hip_rt = None


def _node_types(graph):
    """Node types of a captured (kept) graph, through the HIP runtime the process already has loaded."""
    global _hip_rt
    if _hip_rt is None:
        _hip_rt = C.CDLL("libamdhip64.so")
    raw = C.c_void_p(graph.raw_cuda_graph())
    n = C.c_size_t(0)
    if _hip_rt.hipGraphGetNodes(raw, None, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (C.c_void_p * max(n.value, 1))()
    if _hip_rt.hipGraphGetNodes(raw, nodes, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    out = []
    for i in range(n.value):
        t = C.c_int(-1)
        if _hip_rt.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        out.append(t.value)
    return out


class CapturedGraph:
    """A captured HIP graph plus the one thing that has to be known about it before replaying it on ROCm 7.2:

    hipGraph MEMSET nodes do not hold their place in the graph.  PyTorch's multi-block reductions zero a semaphore with
    ``hipMemsetAsync`` (``x.sum(0)`` of an [8192, 50] tensor is [memset, kernel]); captured and replayed with ordinary stream
    work around it, such a graph returns the PREVIOUS replay's result in a large fraction of launches (59-163 of 300 in a
    two-node graph; an event recorded after the launch does not cover it, ``hipStreamSynchronize`` does).  In
    ``odeint_adjoint(..., graph_func=True)`` that was a stale bias gradient, 3.6e-3 off, from the second call on.  Graphs of
    kernel (and copy) nodes only — every graph this package's own kernels make — replay correctly by the thousand.

    So: the node types are read back after capture (``hipGraphGetNodes``) and every memset node is replaced, before the graph
    is instantiated, by a fill-KERNEL node with the same destination, pattern, dependencies and dependents
    (``xde_graph_replace_memsets`` in libxde_hip.so).  Should that fail, or the nodes be out of reach (an older PyTorch), the
    graph is replayed in safe mode: a stream synchronise after every launch."""

    def __init__(self):
        try:
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
            self._kept = True
        except TypeError:  # an older PyTorch: no handle on the captured graph
            self.graph = torch.cuda.CUDAGraph()
            self._kept = False
        self.safe_mode = True
        self.node_types = None
        self.memsets_replaced = 0

    def capture(self, **kwargs):
        """Context manager: ``with cg.capture(): ...`` records into the graph; call ``finish()`` afterwards.

        The recording is opened through ``recording()`` (below): while ANY thread of the process records, a captured graph whose last
        reference dies — by a cyclic collection, by an explicit ``gc.collect()`` inside a user's func, or by a plain reference-count
        drop — is parked on a process-wide list instead of being destroyed, and destroyed when the outermost recording has ended."""
        return recording(torch.cuda.graph(self.graph, **kwargs))

    def __del__(self):
        # hipGraphExecDestroy / hipGraphDestroy / the release of the graph's private memory pool must not run in the middle of a stream
        # capture (round 5, gpurun_out/r05f: `Fatal Python error: Aborted` under `weakref.remove` — a collection inside a captured func
        # reaped a dropped module's entry of the per-module capture cache; the destructor's failed HIP call ends in std::terminate)
        try:
            g = self.__dict__.pop("graph", None)
            if g is not None:
                release_when_idle(g)
        except Exception:  # interpreter shutdown: module globals may be gone
            pass

    def finish(self):
        if self._kept:
            try:
                self.node_types = _node_types(self.graph)
                if _NODE_MEMSET in self.node_types:
                    # surgery: every memset node becomes a fill-kernel node with the same edges (xde_graph_replace_memsets)
                    from .. import _hip

                    n = C.c_int(0)
                    lib = _hip.load_library()
                    if lib.xde_graph_replace_memsets(C.c_void_p(self.graph.raw_cuda_graph()), C.byref(n)) != 0:
                        raise RuntimeError(lib.xde_last_error().decode())
                    self.memsets_replaced = n.value
                    self.node_types = _node_types(self.graph)
                self.safe_mode = _NODE_MEMSET in self.node_types
            except Exception:
                self.safe_mode = True
            self.graph.instantiate()
        return self

    def replay(self):
        self.graph.replay()
        if self.safe_mode:
            torch.cuda.current_stream().synchronize()


# ----------------------------------------------------------------------------------------------------------------------
# Recordings open in the process, and what must not be released while one is
# ----------------------------------------------------------------------------------------------------------------------
# Re-entrant: a finalizer (release_when_idle) can run at any allocation, also on a thread that holds the lock
_REC_LOCK = threading.RLock()
_REC = {"depth": 0, "gc_was_enabled": False, "deferred_total": 0}
_DEFERRED = []  # owners of device resources whose last reference died while a recording was open


def recordings_open():
    """Stream captures this package has open in the process (any thread)."""
    return _REC["depth"]


def release_when_idle(obj):
    """For finalizers of objects that own device resources whose release issues HIP calls (a captured graph and its private pool;
    pinned buffers, whose release records an event on the streams that used them).  While a recording is open in the process the
    object is kept alive on a process-wide list, emptied when the outermost recording has ended, and True is returned; with no
    recording open nothing is kept and the caller's reference is the last one."""
    with _REC_LOCK:
        if _REC["depth"] > 0:
            _DEFERRED.append(obj)
            _REC["deferred_total"] += 1
            return True
    return False


class recording:
    """``with recording(torch.cuda.graph(...))``: a stream capture during which nothing that owns a captured graph is released.

    * The resources: ``CapturedGraph.__del__`` (every graph this package captures — ``GraphedFunc``'s captures in the per-module cache
      of ``functional/_adjoint_capture.py``, the pipelines' and interval solvers' graphs of ``_hip.HipBackend.capture``, the fixed
      solvers' captured step) and ``_hip._Peek`` (pinned buffers) go through ``release_when_idle``.  The list is emptied under the
      lock when the depth returns to zero: a thread that wants to open the next recording waits for that.
    * The belt: Python's cyclic collector is held off from the first recording's entry until the last one has closed — counted
      across threads, because captures are opened ``capture_error_mode="thread_local"`` so that other threads keep working: the
      first recording in remembers whether the collector was on, the last one out restores it.  The belt does not cover an explicit
      ``gc.collect()`` in a user's func or a reference-count drop; the deferred release does.  (No collection is forced on entry:
      torch 2.10's ``torch.cuda.graph`` does not collect there either — ``torch.compiler.config.force_cudagraph_gc`` is False — because
      a full collection costs tens of milliseconds, and a solve that switches to its captured pipeline records a graph per call.
      Dead cycles that own captures may therefore be alive when a recording starts: that is what the deferred release is for.)"""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        import gc

        with _REC_LOCK:
            if _REC["depth"] == 0:
                _REC["gc_was_enabled"] = gc.isenabled()
                gc.disable()
            _REC["depth"] += 1
        try:
            return self.ctx.__enter__()
        except BaseException:
            self._close()
            raise

    def __exit__(self, *exc):
        try:
            return self.ctx.__exit__(*exc)
        finally:
            self._close()

    @staticmethod
    def _close():
        import gc

        with _REC_LOCK:
            _REC["depth"] -= 1
            if _REC["depth"] == 0:
                while _DEFERRED:  # (a destructor run here cannot add to the list: the depth is zero)
                    _DEFERRED.pop()
                if _REC["gc_was_enabled"]:
                    gc.enable()


def _map(x, fn):
    if isinstance(x, (tuple, list)):
        return tuple(fn(a) for a in x)
    return fn(x)


class _Capture:
    __slots__ = ("graph", "t", "y", "out")


class _AutogradTargetProbe:
    """While active, notes whether ``torch.autograd.grad`` / ``backward`` (and so ``Tensor.backward``) is asked to
    differentiate with respect to — or accumulate into — real ``nn.Parameter`` leaves.

    That is the one thing a func must not do inside a stream capture on ROCm 7.2 / torch 2.10: once a ``loss.backward()``
    has run, parameter leaves own ``AccumulateGrad`` nodes bound to the default stream; the engine then synchronises the
    capturing stream with the default stream at the end of the pass ("AccumulateGrad node's stream does not match" is the
    warning it prints) and ``hipStreamEndCapture`` takes the process down with a segmentation fault — no Python error to
    catch.  Differentiating w.r.t. fresh detached aliases (``p.detach().requires_grad_()`` + ``torch.func.functional_call``,
    what ``odeint_adjoint(graph_func=True)`` does) is safe: they have no accumulator."""

    _LOCK = threading.RLock()  # one probe at a time: it swaps module attributes of torch.autograd for its duration

    def __init__(self):
        self.hit = None

    def _note(self, what, tensors):
        if tensors is None:
            self.hit = self.hit or (what + " without explicit inputs (accumulates into every parameter leaf)")
            return
        ts = tensors if isinstance(tensors, (tuple, list)) else (tensors,)
        for x in ts:
            if isinstance(x, torch.nn.Parameter):
                self.hit = self.hit or (what + " with respect to an nn.Parameter leaf")
                return

    def __enter__(self):
        self._LOCK.acquire()
        ag = torch.autograd
        self._grad, self._backward = ag.grad, ag.backward
        probe = self

        def grad(outputs, inputs, *a, **k):
            probe._note("torch.autograd.grad", inputs)
            return probe._grad(outputs, inputs, *a, **k)

        def backward(tensors, grad_tensors=None, retain_graph=None, create_graph=False, grad_variables=None, inputs=None):
            probe._note("backward()", inputs)
            return probe._backward(tensors, grad_tensors, retain_graph, create_graph, grad_variables, inputs)

        ag.grad, ag.backward = grad, backward
        return self

    def __exit__(self, *exc):
        torch.autograd.grad, torch.autograd.backward = self._grad, self._backward
        self._LOCK.release()
        return False


class GraphedFunc:
    def __init__(self, func, warmup=3, clone_outputs=True):
        self.func = func
        self.warmup = int(warmup)
        self.clone_outputs = bool(clone_outputs)
        self._captures = {}
        self.replays = 0
        self.captures = 0
        self.eager_calls = 0
        self.safe_mode = False  # True once a capture holds a memset node (see CapturedGraph): replays are synchronised
        self.refused = {}  # signature -> why this func is evaluated eagerly instead of being captured

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
        if key in self.refused:
            return None
        c = _Capture()
        c.t = t.detach().clone()
        c.y = _map(y, lambda a: a.detach().clone())
        # Guard (see _AutogradTargetProbe): one eager evaluation tells whether func differentiates w.r.t. real parameter
        # leaves; if it does, it is never captured — the failure mode would be a segmentation fault inside
        # hipStreamEndCapture, not an exception — and this signature is evaluated eagerly from now on (same results).
        with _AutogradTargetProbe() as probe:
            self.func(c.t, c.y)
        if probe.hit is not None:
            import warnings

            self.refused[key] = probe.hit
            warnings.warn(
                "paddlexde_amd.GraphedFunc: func is not captured into a HIP graph because it calls " + probe.hit + "; "
                "it is evaluated eagerly instead. Differentiate with respect to detached aliases of the parameters "
                "(torch.func.functional_call) to make it capturable.", stacklevel=3)
            return None
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
            c.graph = CapturedGraph()
            with c.graph.capture(capture_error_mode="thread_local"):
                c.out = self.func(c.t, c.y)
            c.graph.finish()
            self.safe_mode = self.safe_mode or c.graph.safe_mode
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
            if c is None:  # refused by the capture guard
                self.eager_calls += 1
                return self.func(t, y)
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
