/*
 * oracle/xde_cpu_kernels.c — TEST / BASELINE INFRASTRUCTURE, not product code.
 *
 * Fused CPU statement (plain C + OpenMP) of the three element-wise passes of one embedded Runge-Kutta attempt, with the
 * operand lists and the op order of include/xde_hip.h's kernels: SURVEY.md section 8(d) baseline "B2" — what a careful CPU
 * implementation of the reference's step costs (one pass per stage instead of the reference's ~10 eager ops per stage).
 *   xde_cpu_stage_combine   yi = y0 + sum_j k_j * (beta_ij * dt)          solver/base_adaptive_solver_rk.py:166-168
 *   (out2)                  sum_j k_j * (dt * c_err_j), the partial error sum of the last stage          :180
 *   xde_cpu_error_norm      err = e_pre + k_S*(dt*c_S); tol = atol + rtol*max(|y0|,|y1|);  sum (err/tol)^2, count of
 *                           non-finite y0                                     utils/ode_utils.py:80-82,8-9; :201
 * Used only by bench.py's cpu_baseline leg and by tests/ (checked against the numpy oracle's formulas, bit for bit: build
 * with -ffp-contract=off).  The product never loads this file's library.
 */
#include <math.h>
#include <stdint.h>

#define XDE_CPU_MAX_K 14

void xde_cpu_stage_combine(float* out, const float* y0, const float* const* k, const double* coef, int nk, double dt_d, int64_t n,
                           float* out2, const double* coef2) {
  const float dt = (float)dt_d;
  float c[XDE_CPU_MAX_K], c2[XDE_CPU_MAX_K];
  for (int j = 0; j < nk; ++j) {
    c[j] = (float)coef[j] * dt;                 /* `beta_i * dt` */
    c2[j] = coef2 ? dt * (float)coef2[j] : 0.f; /* `dt * c_error` */
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float acc = k[0][i] * c[0];
    for (int j = 1; j < nk; ++j) acc = acc + k[j][i] * c[j];
    if (out2) {
      float e = k[0][i] * c2[0];
      for (int j = 1; j < nk; ++j) e = e + k[j][i] * c2[j];
      out2[i] = e;
    }
    out[i] = y0[i] + acc;
  }
}

/* returns sum over i of (err_i / tol_i)^2 in double (per-thread float chunks of 256 summed in double); *nonfinite_out = count */
double xde_cpu_error_norm(const float* e_pre, const float* k_last, double c_last, const float* y0, const float* y1, double rtol_d,
                          double atol_d, double dt_d, int64_t n, int64_t* nonfinite_out) {
  const float dt = (float)dt_d, rtol = (float)rtol_d, atol = (float)atol_d;
  const float cl = dt * (float)c_last;
  double total = 0.0;
  int64_t nf = 0;
#pragma omp parallel for schedule(static) reduction(+ : total, nf)
  for (int64_t b = 0; b < (n + 255) / 256; ++b) {
    const int64_t lo = b * 256, hi = lo + 256 < n ? lo + 256 : n;
    float acc = 0.f;
    for (int64_t i = lo; i < hi; ++i) {
      const float e = e_pre[i] + k_last[i] * cl;
      const float tol = atol + rtol * fmaxf(fabsf(y0[i]), fabsf(y1[i]));
      const float r = fabsf(e / tol);
      acc = acc + r * r;
      nf += isfinite(y0[i]) ? 0 : 1;
    }
    total += (double)acc;
  }
  *nonfinite_out = nf;
  return total;
}
