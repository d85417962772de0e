"""Generates tests/golden/*.npz: small input/output vectors and per-step traces produced by the CPU oracle
(oracle/xde_oracle.py) IN THIS CONTAINER.  The reference cannot be executed here (Paddle is not installed), so
these are oracle-generated regression vectors — they pin the oracle against silent edits and give the GPU suite
fixed expectations; the oracle itself is pinned by tests/test_oracle_pinning.py.

    python -m tests.golden.make_golden
"""
import os

import numpy as np

from oracle import xde_oracle as O

from .. import problems as P

HERE = os.path.dirname(os.path.abspath(__file__))


def _trace(s):
    return np.asarray([[r.t0, r.dt, r.ratio, float(r.accept)] for r in s.trace], dtype=np.float64)


def spiral_rk4():
    y0 = np.array([[2.0, 0.0]], dtype=np.float32)
    t = np.linspace(0.0, 25.0, 1000).astype(np.float32)
    return {"y0": y0, "t": t, "sol": O.odeint(P.spiral_np, y0, t, "rk4")}


def spiral_fixed_small():
    rng = np.random.RandomState(3)
    y0 = rng.uniform(-1.5, 1.5, size=(4, 3, 2)).astype(np.float32)
    t = np.linspace(0.0, 0.3, 7).astype(np.float32)
    out = {"y0": y0, "t": t}
    for m in ("euler", "midpoint", "rk4", "rk4_classic"):
        out["sol_" + m] = O.odeint(P.spiral_np, y0, t, m)
    return out


def linear_dopri5_f64():
    A = P.skew_matrix(16).double().numpy()
    y0 = np.random.RandomState(0).randn(8, 16)
    t = np.linspace(0.0, 2.0, 6)
    sol, s = O.odeint(lambda t_, y: y @ A.T, y0, t, "dopri5", rtol=1e-7, atol=1e-9, options={"norm": O._rms_norm, "dtype": np.float64},
                      return_solver=True)
    return {"A": A, "y0": y0, "t": t, "sol": sol, "trace": _trace(s), "nfe": np.asarray(s.nfe)}


def vdp_dopri5_f64():
    mu = 30.0
    y0 = np.array([2.0, 0.0]) + 0.01 * np.random.RandomState(0).randn(16, 2)
    t = np.array([0.0, 0.5, 1.0])
    sol, s = O.odeint(P.vdp_np(mu), y0, t, "dopri5", rtol=1e-6, atol=1e-8, options={"norm": O._rms_norm, "dtype": np.float64},
                      return_solver=True)
    return {"mu": np.asarray(mu), "y0": y0, "t": t, "sol": sol, "trace": _trace(s), "counts": np.asarray([s.n_accept, s.n_reject, s.nfe])}


def tableaus_f64():
    out = {}
    for name, (order, tab, mid) in O.ADAPTIVE.items():
        out[name + "_alpha"] = tab.alpha
        out[name + "_c_sol"] = tab.c_sol
        out[name + "_c_error"] = tab.c_error
        out[name + "_mid"] = mid
        for i, b in enumerate(tab.beta):
            out["{}_beta{}".format(name, i)] = b
    return out


def _linear_full(B, D, n_rows):
    """BASELINE.json configs 2 / 4 at FULL size on the numpy oracle: dy/dt = A y, A = U - U^T (U = 0.1 randn, seed 1), y0 =
    randn(B, D) seed 0 (both from torch's CPU generator: reproducible on the GPU box), Dopri5 rtol 1e-5 / atol 1e-7, fp32.
    Stored: every attempt's (t0, dt, ratio, accept), the counts, and `n_rows` sampled rows of the solution at t = 0.5, 1."""
    import torch

    A = P.skew_matrix(D).float().numpy()
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(0)).numpy()
    t = np.array([0.0, 0.5, 1.0], dtype=np.float32)
    sol, s = O.odeint(lambda t_, y: y @ A.T, y0, t, "dopri5", rtol=1e-5, atol=1e-7, return_solver=True)
    rows = np.sort(np.random.RandomState(11).choice(B, size=n_rows, replace=False))
    return {"B": np.asarray(B), "D": np.asarray(D), "t": t, "rows": rows, "y0_rows": y0[rows], "sol_rows": sol[:, rows, :],
            "trace": _trace(s), "counts": np.asarray([s.n_accept, s.n_reject, s.nfe]),
            "sol_abs_max": np.asarray(float(np.abs(sol).max()))}


def config2_full():
    return _linear_full(65536, 128, 48)


def config4_full():
    """Config 4's GLOBAL problem (524288 x 64; rank r of 8 owns rows [r*65536, (r+1)*65536)): the step sequence every rank
    must follow, and sampled rows from every shard."""
    return _linear_full(524288, 64, 64)


FULL_CASES = {"config2_full": config2_full, "config4_full": config4_full}  # minutes of CPU time: `--full`

CASES = {
    "spiral_rk4": spiral_rk4,
    "spiral_fixed_small": spiral_fixed_small,
    "linear_dopri5_f64": linear_dopri5_f64,
    "vdp_dopri5_f64": vdp_dopri5_f64,
    "tableaus_f64": tableaus_f64,
}


if __name__ == "__main__":
    import sys

    for name, fn in (FULL_CASES if "--full" in sys.argv else CASES).items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fn())
        print("wrote", name)
