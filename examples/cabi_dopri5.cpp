// A whole adaptive Dormand-Prince solve through libxde_hip.so from plain C++/HIP — no Python, no torch, no framework: the C ABI of
// include/xde_hip.h takes raw device pointers, and this is what a foreign host (the Paddle-side binding of INTEGRATION.md section B, a
// C++ serving process) does with it.  The "user's func" is a HIP kernel here: 2-D rotations with per-pair angular velocity, so the
// exact solution is known in closed form.
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/cabi_dopri5.cpp -L paddlexde_amd/lib -lxde_hip -Wl,-rpath,$PWD/paddlexde_amd/lib -o /tmp/cabi_dopri5
//   /tmp/cabi_dopri5            # prints the step counts and the worst error against cos/sin; exit code 0 = within tolerance
//
// Reference path this replaces: AdaptiveRKSolver.integrate / _adaptive_step / _runge_kutta_step with the Dopri5 tableau
// (solver/base_adaptive_solver_rk.py:116-292, solver/adaptive_solver/dopri5.py:5-55).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "xde_hip.h"

#define HIP_OK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                             \
      std::exit(2);                                                                            \
    }                                                                                          \
  } while (0)
#define XDE_OK_(x)                                                                             \
  do {                                                                                         \
    int rc_ = (x);                                                                             \
    if (rc_ != XDE_OK) {                                                                       \
      std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, xde_last_error());                       \
      std::exit(3);                                                                            \
    }                                                                                          \
  } while (0)

// the user's func(t, y): rows of `pairs` oscillators (a, b) with a' = -w b, b' = w a
__global__ void rotate(const float* __restrict__ y, float* __restrict__ f, const float* __restrict__ w, int pairs, long n_pairs) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  float om = w[i % pairs];
  float a = y[2 * i], b = y[2 * i + 1];
  f[2 * i] = -(om * b);
  f[2 * i + 1] = om * a;
}

int main() {
  const int rows = 4096, pairs = 32;
  const long n = long(rows) * pairs * 2;
  const double t_out[3] = {0.0, 0.4, 1.0};

  // Dormand-Prince tableau (solver/adaptive_solver/dopri5.py:5-55): stage rows, error weights, midpoint weights; zero entries skipped
  const double alpha[6] = {1 / 5., 3 / 10., 4 / 5., 8 / 9., 1.0, 1.0};
  const double beta[6][6] = {{1 / 5.},
                             {3 / 40., 9 / 40.},
                             {44 / 45., -56 / 15., 32 / 9.},
                             {19372 / 6561., -25360 / 2187., 64448 / 6561., -212 / 729.},
                             {9017 / 3168., -355 / 33., 46732 / 5247., 49 / 176., -5103 / 18656.},
                             {35 / 384., 0, 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84.}};
  const double c_err[7] = {35 / 384. - 1951 / 21600., 0, 500 / 1113. - 22642 / 50085., 125 / 192. - 451 / 720.,
                           -2187 / 6784. - -12231 / 42400., 11 / 84. - 649 / 6300., -1.0 / 60.0};
  const double mid[7] = {6025192743. / 30085553152. / 2, 0, 51252292925. / 65400821598. / 2, -2691868925. / 45128329728. / 2,
                         187940372067. / 1594534317056. / 2, -1776094331. / 19743644256. / 2, 11237099. / 235043384. / 2};

  if (xde_abi_version() != XDE_ABI_VERSION || xde_sizeof_ctrl_params() != (int64_t)sizeof(xde_ctrl_params_t) ||
      xde_sizeof_ctrl() != (int64_t)sizeof(xde_ctrl_t)) {
    std::fprintf(stderr, "header / library mismatch\n");
    return 4;
  }

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  std::vector<float> h_y0(n), h_w(pairs);
  for (int p = 0; p < pairs; ++p) h_w[p] = 1.0f + 2.0f * float(p) / pairs;
  for (long i = 0; i < n / 2; ++i) {
    float ph = 0.001f * float(i % 977);
    h_y0[2 * i] = std::cos(ph);
    h_y0[2 * i + 1] = std::sin(ph);
  }
  float *w_d, *sol;  // sol: [3, n]
  HIP_OK(hipMalloc(&w_d, pairs * sizeof(float)));
  HIP_OK(hipMalloc(&sol, 3 * n * sizeof(float)));
  HIP_OK(hipMemcpy(w_d, h_w.data(), pairs * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(sol, h_y0.data(), n * sizeof(float), hipMemcpyHostToDevice));
  // state buffers: y (current), a pool of derivative buffers k0..k6, the stage scratch, y1
  float* buf[10];
  for (auto& b : buf) HIP_OK(hipMalloc(&b, n * sizeof(float)));
  float *y = buf[0], *y1 = buf[1], *scratch = buf[2];
  float* k[7] = {buf[3], buf[4], buf[5], buf[6], buf[7], buf[8], buf[9]};
  HIP_OK(hipMemcpy(y, h_y0.data(), n * sizeof(float), hipMemcpyHostToDevice));

  void *ctrl, *ws;
  double* t_dev;
  float* t_stage;
  HIP_OK(hipMalloc(&ctrl, xde_sizeof_ctrl()));
  HIP_OK(hipMemset(ctrl, 0, xde_sizeof_ctrl()));
  HIP_OK(hipMalloc(&ws, xde_workspace_bytes()));
  HIP_OK(hipMemset(ws, 0, xde_workspace_bytes()));
  HIP_OK(hipMalloc(&t_dev, sizeof(t_out)));
  HIP_OK(hipMemcpy(t_dev, t_out, sizeof(t_out), hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&t_stage, XDE_MAX_STAGE * sizeof(float)));

  xde_ctrl_params_t p;
  std::memset(&p, 0, sizeof(p));
  p.struct_size = sizeof(p);  // the binding's statement of the layout it was compiled against: checked by every entry point
  p.abi_version = XDE_ABI_VERSION;
  p.rtol = double(1e-5f);
  p.atol = double(1e-7f);
  p.min_step = 0.0;
  p.max_step = INFINITY;
  p.safety = double(0.9f);
  p.ifactor = 10.0;
  p.dfactor = double(0.2f);
  p.order = 5.0;
  p.max_num_steps = 2147483647;
  p.time_dtype = p.state_dtype = XDE_F32;
  p.direction = 1;
  p.norm_kind = XDE_NORM_RMS;
  p.n_stage = 6;
  p.n_seg = 1;
  for (int i = 0; i < 6; ++i) p.alpha[i] = alpha[i];
  p.seg_count[0] = double(n);
  xde_segments_t segs;
  std::memset(&segs, 0, sizeof(segs));
  segs.struct_size = sizeof(segs);
  segs.n_seg = 1;
  segs.seg_start[0] = 0;
  segs.seg_len[0] = n;

  auto func = [&](const float* yin, float* fout) {
    long np = n / 2;
    hipLaunchKernelGGL(rotate, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, yin, fout, w_d, pairs, np);
  };

  XDE_OK_(xde_ctrl_init((xde_ctrl_t*)ctrl, &p, 0.0, 0.01, 3, t_dev, nullptr, t_stage, 0, nullptr, st));
  func(y, k[0]);
  xde_ctrl_t host;
  int attempts = 0;
  for (; attempts < 10000; ++attempts) {
    for (int i = 0; i < 6; ++i) {  // y_i = y + sum_j k_j (beta_ij dt), dt read from the device control block
      const void* ops[6];
      double cf[6];
      int nk = 0;
      for (int j = 0; j <= i; ++j)
        if (j == 0 || beta[i][j] != 0) {
          ops[nk] = k[j];
          cf[nk++] = beta[i][j];
        }
      float* out = (i == 5) ? y1 : scratch;
      XDE_OK_(xde_stage_combine(out, y, nullptr, ops, nullptr, cf, nk, XDE_COMBINE_RK, 1.0, 0.0, (const xde_ctrl_t*)ctrl, n, XDE_F32, nullptr,
                                nullptr, 0.0, 0, st));
      func(out, k[i + 1]);
    }
    const void* eops[6] = {k[0], k[2], k[3], k[4], k[5], k[6]};
    const double ecf[6] = {c_err[0], c_err[2], c_err[3], c_err[4], c_err[5], c_err[6]};
    XDE_OK_(xde_error_norm_partial(eops, nullptr, ecf, 6, y, nullptr, y1, p.rtol, p.atol, 0.0, (const xde_ctrl_t*)ctrl, &segs, XDE_NORM_RMS,
                                   XDE_F32, ws, nullptr, st));
    XDE_OK_(xde_rk_control((xde_ctrl_t*)ctrl, &p, ws, nullptr, t_dev, nullptr, t_stage, nullptr, st));
    XDE_OK_(xde_ctrl_read((const xde_ctrl_t*)ctrl, &host, st));  // (blocking: the simplest pipeline; see DESIGN section 5 for the others)
    if (host.status != XDE_STATUS_OK) {
      std::fprintf(stderr, "solver status %d\n", host.status);
      return 5;
    }
    if (host.accept) {
      if (host.out_end > host.out_begin) {
        const void* mops[6] = {k[0], k[2], k[3], k[4], k[5], k[6]};
        const double mcf[6] = {mid[0], mid[2], mid[3], mid[4], mid[5], mid[6]};
        XDE_OK_(xde_dense_eval(sol, mops, nullptr, mcf, 6, y, nullptr, y1, k[6], (const xde_ctrl_t*)ctrl, t_dev, XDE_F32, n, XDE_F32, -1, st));
      }
      std::swap(y, y1);        // (t, y, f) <- (t1, y1, f1): pointer swaps on the host
      std::swap(k[0], k[6]);   // FSAL: the last stage derivative is the next step's first
    }
    if (host.done) break;
  }
  HIP_OK(hipStreamSynchronize(st));

  std::vector<float> h_sol(3 * n);
  HIP_OK(hipMemcpy(h_sol.data(), sol, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int r = 0; r < 3; ++r)
    for (long i = 0; i < n / 2; ++i) {
      double ph = 0.001 * double(i % 977) + double(h_w[i % pairs]) * t_out[r];
      worst = std::fmax(worst, std::fabs(h_sol[r * n + 2 * i] - std::cos(ph)));
      worst = std::fmax(worst, std::fabs(h_sol[r * n + 2 * i + 1] - std::sin(ph)));
    }
  std::printf("cabi_dopri5: %ld states, %d attempts (%lld accepted, %lld rejected), worst |error| vs cos/sin = %.3e\n", n, attempts + 1,
              (long long)host.n_accept, (long long)host.n_reject, worst);
  return (host.done && worst < 5e-5) ? 0 : 1;
}
