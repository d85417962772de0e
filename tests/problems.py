"""Test problems shared by the CPU and GPU suites.

The three analytic problems restate the reference's fixtures (tests/testing_utils.py:8-70: ConstantXDE,
SineXDE, LinearXDE on t = linspace(1, 8, 10)); the spiral is the demo system (example/demo_utils.py:136-164).
Each problem offers a numpy ``f_np`` for the oracle and a torch ``f_torch(device)`` for the product; both use
only +, -, * and matmul/sin so that the two frameworks agree to rounding.
"""
import math

import numpy as np
import scipy.linalg
import torch


class Constant:
    a, b = 0.2, 3.0

    def f_np(self, t, y):
        d = y - (np.float32(self.a) * t + np.float32(self.b)).astype(y.dtype)
        return (np.float32(self.a) + d * d * d * d * d).astype(y.dtype)

    def f_torch(self, device):
        a, b = self.a, self.b

        def f(t, y):
            d = y - (a * t + b).to(y.dtype)
            return a + d * d * d * d * d

        return f

    def exact(self, t):
        return (self.a * t + self.b)[:, None]  # [T, 1]


class Sine:
    def f_np(self, t, y):
        t = np.asarray(t, dtype=y.dtype)
        return (2 * y / t + t**4 * np.sin(2 * t) - t**2 + 4 * t**3).astype(y.dtype)

    def f_torch(self, device):
        def f(t, y):
            t = t.to(y.dtype)
            return 2 * y / t + t**4 * torch.sin(2 * t) - t**2 + 4 * t**3

        return f

    def exact(self, t):
        return (
            -0.5 * t**4 * np.cos(2 * t) + 0.5 * t**3 * np.sin(2 * t) + 0.25 * t**2 * np.cos(2 * t) - t**3 + 2 * t**4
            + (math.pi - 0.25) * t**2
        )[:, None]


class Linear:
    def __init__(self, dim=10, seed=0):
        rng = np.random.RandomState(seed)
        U = (rng.randn(dim, dim) * 0.1).astype(np.float32)
        self.A = (2 * U - (U + U.T)).astype(np.float32)
        self.dim = dim

    def f_np(self, t, y):
        return (y @ self.A.T.astype(y.dtype)).astype(y.dtype)

    def f_torch(self, device):
        A = torch.from_numpy(self.A).to(device)

        def f(t, y):
            return y @ A.T.to(y.dtype)

        return f

    def exact(self, t):
        y0 = np.ones((self.dim, 1))
        return np.stack([(scipy.linalg.expm(self.A.astype(np.float64) * ti) @ y0)[:, 0] for ti in t])  # [T, dim]


PROBLEMS = {"constant": Constant, "sine": Sine, "linear": Linear}


def construct_problem(name, npts=10, dtype=np.float32):
    """tests/testing_utils.py:83-98 — returns (problem, y0 [1, D], t [T], sol [T, D])."""
    p = PROBLEMS[name]()
    t = np.linspace(1, 8, npts).astype(np.float32)
    sol = p.exact(t.astype(np.float64)).astype(dtype)
    return p, sol[0][None, :].copy(), t, sol


SPIRAL_A = np.array([[-0.1, 2.0], [-2.0, -0.1]], dtype=np.float32)


def spiral_np(t, y):
    c = y * y * y
    A = SPIRAL_A.astype(y.dtype)
    return np.stack([c[..., 0] * A[0, 0] + c[..., 1] * A[1, 0], c[..., 0] * A[0, 1] + c[..., 1] * A[1, 1]], axis=-1)


def spiral_torch(t, y):
    c = y * y * y
    A = SPIRAL_A
    return torch.stack(
        [c[..., 0] * float(A[0, 0]) + c[..., 1] * float(A[1, 0]), c[..., 0] * float(A[0, 1]) + c[..., 1] * float(A[1, 1])], dim=-1
    )


def skew_matrix(dim, seed=1):
    g = torch.Generator().manual_seed(seed)
    U = 0.1 * torch.randn(dim, dim, generator=g)
    return (U - U.T).contiguous()


def vdp_np(mu):
    def f(t, y):
        x, v = y[..., 0], y[..., 1]
        return np.stack([v, mu * (1 - x * x) * v - x], axis=-1).astype(y.dtype)

    return f


def vdp_torch(mu):
    def f(t, y):
        x, v = y[..., 0], y[..., 1]
        return torch.stack([v, mu * (1 - x * x) * v - x], dim=-1)

    return f


def parity_ok(got, ref, rtol=1e-5, atol=1e-7, slack=1.0):
    """north_star bar: |got - ref| <= atol + rtol * |ref| (optionally with a stated slack factor)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return bool((np.abs(got - ref) <= slack * (atol + rtol * np.abs(ref))).all())


def worst(got, ref, rtol=1e-5, atol=1e-7):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float((np.abs(got - ref) / (atol + rtol * np.abs(ref))).max())


def rel_err(got, ref):
    """max |got - ref| / max |ref|: relative error against the solution's scale (fp32 free-running bar)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300))


def ulp_atol(ref, k, floor=1e-7):
    """k fp32 ulps at the scale of `ref` (its largest magnitude).  north_star's absolute tolerance, 1e-7, is BELOW one fp32 ulp
    of any quantity larger than 1 (ulp(1) = 1.19e-7, ulp(4) = 4.8e-7): where two fp32 implementations differ by one rounding of
    an O(1) intermediate (a GEMM summed in another order), elements of the state that pass through zero cannot agree to 1e-7.
    Tests whose func is not bit-identical on both sides state the bar's absolute part in ulps of the state's scale instead;
    the relative part stays 1e-5."""
    return max(floor, k * float(np.spacing(np.float32(np.abs(np.asarray(ref)).max()))))


def worst_ulps(got, ref, near_zero=1e-2):
    """Largest |got - ref| over the elements whose |ref| is below `near_zero` x the state's scale — the elements for which the
    bar's ABSOLUTE part decides — in units of one fp32 ulp at that scale (its largest magnitude).  This is the number to hold
    against `ulp_atol(ref, k)`'s k."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = np.abs(ref).max()
    small = np.abs(ref) <= near_zero * scale
    if not small.any():
        return 0.0
    return float(np.abs(got - ref)[small].max() / float(np.spacing(np.float32(scale))))


def report(name, rec):
    """Worst observed deviations of a parity test: printed, and appended to $XDE_PARITY_REPORT when set
    (profiles/r03_parity_report.jsonl is one such run of the GPU suite)."""
    import json
    import os

    print("parity report:", name, rec)
    path = os.environ.get("XDE_PARITY_REPORT")
    if path:
        with open(path, "a") as fh:
            fh.write(json.dumps({"test": name, **rec}) + "\n")


class Foreign:
    """A tensor of 'another framework' at the PROTOCOL level (Paddle is not in the image): owns device memory (held through a private
    torch tensor), exposes `__dlpack__` / `__dlpack_device__` / shape / dtype and nothing else the product could use.  `.raw` is that
    framework's own handle on its storage — what ITS kernels (here: torch ops, standing in for them) compute on."""

    imported = []  # what the importer below was handed (type names): evidence that the product exported its buffers through it

    def __init__(self, t):
        self.raw = t
        self.shape, self.dtype = tuple(t.shape), str(t.dtype)

    def __dlpack__(self, stream=None, **kw):
        return self.raw.__dlpack__(stream=stream) if stream is not None else self.raw.__dlpack__()

    def __dlpack_device__(self):
        return self.raw.__dlpack_device__()

    @staticmethod
    def from_dlpack(x):
        """The foreign framework's importer: consumes any `__dlpack__` producer."""
        import torch as _torch

        Foreign.imported.append(type(x).__name__)
        return Foreign(_torch.from_dlpack(x))


# ----------------------------------------------------------------------------------------------------------------------
# the reference's interpolation fixtures (tests/interpolation/test_interpolation.py:13-85), restated as data: a ramp series sampled at
# integer times and evaluated at 21.12 (:16-32), a sine series sampled every 0.01 and evaluated at 16.5 (:52-72); for each of the
# three spline classes the (value, derivative) relative tolerances its test asserts (:34-47, :74-85; paddle.allclose's atol 1e-8)
# ----------------------------------------------------------------------------------------------------------------------
INTERPOLATION_TOLERANCES = {
    "ramp": {"linear": (1e-4, 1e-4), "cubic": (1e-4, 1e-4), "bez": (1e-4, 1e-4)},
    "sine": {"linear": (5e-2, 1e-2), "cubic": (1e-5, 1e-2), "bez": (5e-2, 1e-2)},
}


def interpolation_fixture(kind):
    """-> (series [1, 2000, 2] f32, t [2000] f32, t_eval [1] f32, value target [1, 1, 2], derivative target [1, 1, 2])."""
    zeros = np.zeros(2000, dtype=np.float32)
    if kind == "ramp":
        series = np.stack([np.arange(0, 1000, 0.5).astype(np.float32), zeros], axis=-1)[None]
        t = np.arange(0, 2000, 1).astype(np.float32)  # (int64 in the fixture; the spline classes cast it to float32, interpolate_base.py:27)
        t_eval = np.array([21.12], dtype=np.float32)
        val = np.array([21.12 * 0.5, 0.0], dtype=np.float32)[None, None]
        der = np.array([0.5, 0.0], dtype=np.float32)[None, None]
    elif kind == "sine":
        x = np.arange(0, 20, 0.01).astype(np.float32)
        assert x.shape == (2000,)
        series = np.sin(np.stack([x, zeros], axis=-1))[None].astype(np.float32)
        t = x
        t_eval = np.array([16.5], dtype=np.float32)
        val = np.sin(np.array([16.5, 0.0], dtype=np.float32))[None, None]
        der = np.cos(np.array([16.5, 0.0], dtype=np.float32))[None, None]
        der[:, :, 1] = 0
    else:
        raise KeyError(kind)
    return series, t, t_eval, val, der


def paddle_allclose(x, y, rtol, atol=1e-8):
    """paddle.allclose(x, y): |x - y| <= atol + rtol |y| everywhere (the comparison the reference's tests make)."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    return bool((np.abs(x - y) <= atol + rtol * np.abs(y)).all())


def ulps_apart(got, ref):
    """Largest |got - ref| in units of one ulp of `ref`'s dtype at the LARGEST magnitude in `ref` (0 = the same bits up to signed zeros)."""
    ref = np.asarray(ref)
    scale = np.abs(ref).max()
    if scale == 0:
        return float(np.abs(np.asarray(got, dtype=np.float64)).max() > 0)
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref.astype(np.float64)).max() / float(np.spacing(ref.dtype.type(scale))))
