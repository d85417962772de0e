"""Operand plans of an embedded Runge-Kutta tableau: which derivatives each launch of an attempted step reads, and what it emits.

A function of the tableau only (built once per solver class).  Reference: the stage loop and the FSAL shortcut of
paddlexde/solver/base_adaptive_solver_rk.py:155-180 — there every stage multiplies and sums the whole ``k[..., :i+1]`` slice, zeros
included; here only non-zero entries are read, and two fusions move work between launches without changing a bit:

  * error-estimate fusion: if the last stage (whose output is y1, FSAL) loads every operand the error estimate needs except the
    last derivative, it emits the partial sum as a second output and the error-norm kernel reads {e_partial, k_last, y0, y1}
    instead of all the k's (Dopri5: 8N -> 4N + 1N written);
  * pre-summed stages: the launch of stage i-1 holds k_0..k_{i-1} in registers anyway, so it can emit ``sum_j k_j (beta_ij dt)``
    over them as a second output (one array written); stage i then reads y0, that partial sum and its newest derivative k_i — 3
    arrays instead of len(idx_i) + 1 — with the same left-to-right association, i.e. the same bits.  Worth it from 4 operands on; a
    pre-summed stage cannot emit for the next one (it no longer holds the old derivatives), and the last stage of an FSAL pair stays
    full (it emits the partial error estimate from all its operands).  Dopri5: stage 5 <- stage 4, 32 N -> 30 N elements per step
    through the stage combines (3 % of config 2's step, profiles/r04_presum_ab.txt).
"""
import collections

from .. import _hip

RKPlans = collections.namedtuple(
    "RKPlans", "n_stage, stage_plan, fsal, sol_plan, err_plan, mid_plan, fuse_err, err2_coef, stage_nt, presum")


def nz_plan(coefs, upto=None):
    """Operand indices with non-zero coefficient; index 0 (f0, the select-able operand) always first."""
    n = len(coefs) if upto is None else upto
    idx = [0] + [j for j in range(1, n) if float(coefs[j]) != 0.0]
    return idx, _hip.dbl_array([float(coefs[j]) for j in idx])  # marshalled once: the C double[] the kernels take


def build_plans(tab, mid):
    n_stage = len(tab.alpha)
    stage_plan = [nz_plan(beta, upto=i + 1) for i, beta in enumerate(tab.beta)]
    c_sol = [float(c) for c in tab.c_sol]
    last_beta = [float(b) for b in tab.beta[-1]]
    # :172-176 "This property (true for Dormand-Prince) lets us save a few FLOPs."
    fsal = c_sol[-1] == 0 and c_sol[:-1] == last_beta
    sol_plan = nz_plan(c_sol)
    err_plan = nz_plan(tab.c_error)
    mid_plan = nz_plan(mid)
    last_idx = stage_plan[-1][0]
    err_idx = err_plan[0]
    S = n_stage
    fuse_err = fsal and err_idx[-1] == S and set(err_idx[:-1]) <= set(last_idx) and float(tab.c_error[S]) != 0.0
    err2_coef = _hip.dbl_array([float(tab.c_error[j]) for j in last_idx]) if fuse_err else None
    presum = {}
    i = S - 2 if fuse_err else S - 1
    while i >= 1:
        idx_i, idx_p = stage_plan[i][0], stage_plan[i - 1][0]
        # (exactly the previous launch's operands: an operand stage i does not use would enter the emitted sum as `k_j * 0`, which
        # is NaN for a non-finite k_j the full stage never reads — ADVICE r04; every tableau shipped here has equal sets)
        if idx_i[-1] == i and len(idx_i) >= 4 and idx_i[:-1] == idx_p:
            emit = _hip.dbl_array([float(tab.beta[i][j]) for j in idx_p])
            presum[i] = (emit, [i], _hip.dbl_array([float(tab.beta[i][i])]))
            i -= 2
        else:
            i -= 1
    # what each stage's launch really READS
    read_plan = [([i] if i in presum else list(stage_plan[i][0])) for i in range(n_stage)]
    # Cache-policy hint per stage launch (results never depend on it): the kernels compile three whole-launch policies (csrc
    # combine_body, NTP), so the hint is all-or-nothing — set (one bit per operand) when EVERY derivative the launch reads is read
    # there for the last time in an accepted step (later readers: stages, the solution combine of a non-FSAL pair, the unfused error
    # estimate; dense output is rare and lazy).  Dopri5: the last stage, whose five operands die with it.
    last_use = {}
    for i, idx_i in enumerate(read_plan):
        for j in idx_i:
            last_use[j] = i
    later = set(sol_plan[0]) if not fsal else set()
    if not fuse_err:
        later |= set(err_plan[0])
    stage_nt = [((1 << len(idx_i)) - 1) if all(last_use[j] == i and j not in later for j in idx_i) else 0
                for i, idx_i in enumerate(read_plan)]
    return RKPlans(n_stage, stage_plan, fsal, sol_plan, err_plan, mid_plan, fuse_err, err2_coef, stage_nt, presum)
