// libxde_hip.so — hand-written gfx950 (MI355X / CDNA4) kernels for the Runge–Kutta hot path of
// paddlexde_amd.  C ABI: include/xde_hip.h (each entry point cites the reference lines it replaces).
//
// All kernels are HBM-bandwidth bound (≈0.5 flop/byte): 16-byte-per-lane coalesced loads, grid-stride
// loops sized to keep every CU's memory queue full (2048 workgroups × 256 threads = 8 waves/SIMD),
// fp32 per-thread accumulation → wave64 __shfl_down → LDS cross-wave → one fp64 partial per workgroup,
// reduced in a fixed order by a single workgroup (bit-reproducible, and identical on every rank after
// the all-reduce).  No MFMA: there is no contraction on this path.
//
// Built with -ffp-contract=off so that element-wise results follow the reference's (unfused) op
// order exactly; the kernels are memory bound, so FMA contraction would buy nothing.
//
// K1: stage combine (all fixed/adaptive stage formulas), its autograd fan-out, the predicated commit.

#include "xde_common.hpp"

using namespace xde;

namespace {

// `y = dy*dt + y0; (dy - lambda*y)*dt + y0` (xde/base_dde.py:55-58), same op order
template <typename T>
__device__ __forceinline__ T fuse_(T dy, T dt, T y0, T lam) {
  if (lam == T(0)) return dy * dt + y0;
  T y = dy * dt + y0;
  return (dy - lam * y) * dt + y0;
}

// ------------------------------------------------------------------------------------------
// K1: stage combine
// ------------------------------------------------------------------------------------------
// NTP — the launch's cache policy, a COMPILE-TIME choice among three (round 5): 0 = default loads; 1 = every derivative k_j streamed
// (non-temporal: this launch reads them for the last time in an accepted step — an FSAL pair's last stage), y0 default (the error
// norm re-reads it); 2 = everything streamed (operands of >= 64 MiB: nothing survives in the Infinity Cache between uses).  Until
// round 4 the policy was a run-time bit per operand: a branch around every load, and — where the two arms got different registers —
// an `s_waitcnt vmcnt(0)` + register copies BETWEEN the loads of one iteration (NK = 6: five loads, wait, two loads), i.e. two
// memory round trips per iteration instead of one.  A mask that is neither "none" nor "all" now takes policy 0.
//
// (Tried in round 5 and NOT kept, profiles/r05_combine_ab.txt — 8 alternating repetitions per variant: requesting the control block's
// words and the select-independent operands BEFORE waiting for the block, the way the error-norm pass does: +0.46 us on the 1-3 operand
// stages and +1.06 us on the pre-summed stage at 16 MiB operands (the compiler sinks the scalar loads behind the vector loads anyway,
// and the peeled first iteration costs registers and code); a software-pipelined main loop on half the grid for the few-stream
// launches: it only wins back what the first change lost.)
template <typename T, int MODE, int NK, bool VEC, bool OUT2, bool PRE = false, int NTP = 0>
__device__ __forceinline__ void combine_body(const CombineArgs& a, const T* __restrict__ y0,
                                             const T* __restrict__ k0, T dt) {
  const T* __restrict__ pre = static_cast<const T*>(a.pre);
  using P = Pack<T, VEC>;
  auto ldk = [](const T* p, int64_t i) { return NTP >= 1 ? P::load_nt(p, i) : P::load(p, i); };
  auto ldy = [](const T* p, int64_t i) { return NTP >= 2 ? P::load_nt(p, i) : P::load(p, i); };
  constexpr int W = P::W;
  T* __restrict__ out = static_cast<T*>(a.out);
  T* __restrict__ out2 = static_cast<T*>(a.out2);
  const T* kp[NK];
  T c[NK];
  T c2[NK];
  kp[0] = k0;
#pragma unroll
  for (int j = 1; j < NK; ++j) kp[j] = static_cast<const T*>(a.k[j]);
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    // reference: tableau cast to the state dtype, then `beta_i * dt` (RK) — or used as is (FUSE/WFUSE)
    c[j] = (MODE == XDE_COMBINE_RK) ? T(a.coef[j]) * dt : T(a.coef[j]);
    // second output: RK -> `dt * tableau.c_error` (or the next stage's `beta * dt`); FUSE -> the weights of a later WFUSE launch
    c2[j] = OUT2 ? (MODE == XDE_COMBINE_RK ? dt * T(a.coef2[j]) : T(a.coef2[j])) : T(0);
  }
  const T scale = T(a.scale);
  const T lam = T(a.damp);
  const int64_t nvec = a.n / W;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  // one vector's arithmetic and stores
  auto finish = [&](int64_t i, const P (&kk)[NK], const P& pr, const P& y) {
    P o;
    P o2;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      if (MODE == XDE_COMBINE_RK) {
        T acc = PRE ? pr.v[w] + kk[0].v[w] * c[0] : kk[0].v[w] * c[0];
#pragma unroll
        for (int j = 1; j < NK; ++j) acc = acc + kk[j].v[w] * c[j];
        o.v[w] = y.v[w] + acc;
        if (OUT2) {
          T e = kk[0].v[w] * c2[0];
#pragma unroll
          for (int j = 1; j < NK; ++j) e = e + kk[j].v[w] * c2[j];
          o2.v[w] = e;
        }
      } else if (MODE == XDE_COMBINE_FUSE) {
        T acc = kk[0].v[w] * c[0];
#pragma unroll
        for (int j = 1; j < NK; ++j) acc = acc + kk[j].v[w] * c[j];
        o.v[w] = fuse_(acc, dt, y.v[w], lam);
        if (OUT2) {  // the leading terms of the step's final weighted sum, from the operands this stage-input launch holds anyway
          T e = fuse_(kk[0].v[w], dt, y.v[w], lam) * c2[0];
#pragma unroll
          for (int j = 1; j < NK; ++j) e = e + fuse_(kk[j].v[w], dt, y.v[w], lam) * c2[j];
          o2.v[w] = e;
        }
      } else {
        T acc = PRE ? pr.v[w] + fuse_(kk[0].v[w], dt, y.v[w], lam) * c[0] : fuse_(kk[0].v[w], dt, y.v[w], lam) * c[0];
#pragma unroll
        for (int j = 1; j < NK; ++j) acc = acc + fuse_(kk[j].v[w], dt, y.v[w], lam) * c[j];
        o.v[w] = acc * scale;
      }
    }
    o.store(out, i);
    if (OUT2) o2.store(out2, i);
  };
  int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  for (; i < nvec; i += stride) {
    // NT: with operands of >= 64 MiB nothing survives in the 256 MiB Infinity Cache between uses anyway; streaming
    // loads then run 12-17 % faster (5.1 -> 5.8 TB/s at 128 MiB x 7 streams).  At the 32 MiB headline size the
    // default policy wins by 18 % (the working set half-fits the cache), so the flag is size-dependent (host).
    // Program order of the loads: the operands whose ADDRESS does not depend on the device-side select come first.  In the
    // speculative pipeline y0 / k0 are picked by ctrl->accept, a scalar load that is still in flight when the wave starts;
    // k_1.. can be requested meanwhile, so the select's memory round trip hides behind them (it cost ~0.9 us per launch).
    P kk[NK];
    P pr;
    if (PRE) pr = P::load_nt(pre, i);  // (written by the previous stage's launch for this one alone: read once, streamed)
#pragma unroll
    for (int j = NK - 1; j >= 1; --j) kk[j] = ldk(kp[j], i);
    P y = ldy(y0, i);
    kk[0] = ldk(kp[0], i);
    finish(i, kk, pr, y);
  }
  if (VEC) {
    // scalar tail (n % W elements), done by the first threads of block 0
    const int64_t tail0 = nvec * W;
    const int64_t i = tail0 + threadIdx.x;
    if (blockIdx.x == 0 && i < a.n) {
      T yv = y0[i];
      T acc;
      if (OUT2) {
        T e = MODE == XDE_COMBINE_RK ? kp[0][i] * c2[0] : fuse_(kp[0][i], dt, yv, lam) * c2[0];
        for (int j = 1; j < NK; ++j) e = e + (MODE == XDE_COMBINE_RK ? kp[j][i] : fuse_(kp[j][i], dt, yv, lam)) * c2[j];
        out2[i] = e;
      }
      if (MODE == XDE_COMBINE_WFUSE) {
        acc = PRE ? pre[i] + fuse_(kp[0][i], dt, yv, lam) * c[0] : fuse_(kp[0][i], dt, yv, lam) * c[0];
        for (int j = 1; j < NK; ++j) acc = acc + fuse_(kp[j][i], dt, yv, lam) * c[j];
        out[i] = acc * scale;
      } else {
        acc = PRE ? pre[i] + kp[0][i] * c[0] : kp[0][i] * c[0];
        for (int j = 1; j < NK; ++j) acc = acc + kp[j][i] * c[j];
        out[i] = (MODE == XDE_COMBINE_RK) ? (yv + acc) : fuse_(acc, dt, yv, lam);
      }
    }
  }
}

// generic operand count (> 8: Dopri8) — runtime loop, same arithmetic order
template <typename T, int MODE, bool VEC, bool OUT2>
__device__ void combine_generic(const CombineArgs& a, const T* __restrict__ y0, const T* __restrict__ k0, T dt) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  T* __restrict__ out = static_cast<T*>(a.out);
  const int nk = a.nk;
  const T scale = T(a.scale);
  const T lam = T(a.damp);
  const int64_t nvec = a.n / W;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nvec; i += stride) {
    P y = P::load(y0, i);
    P acc;
    P e2;
    constexpr bool has2 = OUT2;
    for (int j = 0; j < nk; ++j) {
      const T* kj = j == 0 ? k0 : static_cast<const T*>(a.k[j]);
      P kk = P::load(kj, i);
      T cj = (MODE == XDE_COMBINE_RK) ? T(a.coef[j]) * dt : T(a.coef[j]);
      T c2j = has2 ? (MODE == XDE_COMBINE_RK ? dt * T(a.coef2[j]) : T(a.coef2[j])) : T(0);
#pragma unroll
      for (int w = 0; w < W; ++w) {
        T term = (MODE == XDE_COMBINE_WFUSE) ? fuse_(kk.v[w], dt, y.v[w], lam) * cj : kk.v[w] * cj;
        acc.v[w] = (j == 0) ? term : acc.v[w] + term;
        if (has2) {
          const T t2 = (MODE == XDE_COMBINE_RK) ? kk.v[w] * c2j : fuse_(kk.v[w], dt, y.v[w], lam) * c2j;
          e2.v[w] = (j == 0) ? t2 : e2.v[w] + t2;
        }
      }
    }
    if (has2) e2.store(static_cast<T*>(a.out2), i);
    P o;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      o.v[w] = (MODE == XDE_COMBINE_RK) ? y.v[w] + acc.v[w]
               : (MODE == XDE_COMBINE_FUSE) ? fuse_(acc.v[w], dt, y.v[w], lam)
                                            : acc.v[w] * scale;
    }
    o.store(out, i);
  }
  if (VEC) {
    const int64_t i = nvec * W + threadIdx.x;
    if (blockIdx.x == 0 && i < a.n) {
      T yv = y0[i];
      T acc = T(0);
      T e2 = T(0);
      constexpr bool has2 = OUT2;
      for (int j = 0; j < nk; ++j) {
        const T* kj = j == 0 ? k0 : static_cast<const T*>(a.k[j]);
        T cj = (MODE == XDE_COMBINE_RK) ? T(a.coef[j]) * dt : T(a.coef[j]);
        T term = (MODE == XDE_COMBINE_WFUSE) ? fuse_(kj[i], dt, yv, lam) * cj : kj[i] * cj;
        acc = (j == 0) ? term : acc + term;
        if (has2) {
          const T t2 = (MODE == XDE_COMBINE_RK) ? kj[i] * (dt * T(a.coef2[j])) : fuse_(kj[i], dt, yv, lam) * T(a.coef2[j]);
          e2 = (j == 0) ? t2 : e2 + t2;
        }
      }
      if (has2) static_cast<T*>(a.out2)[i] = e2;
      out[i] = (MODE == XDE_COMBINE_RK) ? yv + acc : (MODE == XDE_COMBINE_FUSE) ? fuse_(acc, dt, yv, lam) : acc * scale;
    }
  }
}

// OUT2 (second output, RK mode only) is a separate instantiation: its two accumulators per element would
// otherwise raise the register budget of EVERY stage launch (62 -> 112 VGPRs, occupancy 8 -> 4 waves/SIMD)
// (WIDE: the two-output launches are register-hungry — two accumulators per element on top of NK + 1 loads in flight — and a kernel's
// allocation is that of its LARGEST case: with every operand count in one kernel the hot cases, Dopri5's 4- and 5-operand launches,
// ran at the 7-operand case's 112 VGPRs = 4 waves per SIMD.  The two-output kernel therefore exists twice: up to 5 operands, and beyond.)
template <typename T, int MODE, bool VEC, bool OUT2, int NTP = 0, bool WIDE = false>
__global__ __launch_bounds__(kBlock) void xde_combine_kernel(CombineArgs a) {
  int sel = 0;
  T dt;
  if (a.ctrl) {
    read_dt_sel<T>(a.ctrl, a.use_sel, dt, sel);
  } else {
    dt = T(a.dt_host);
  }
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  constexpr bool kLow = !(OUT2 && WIDE);   // this kernel holds the cases of 1..5 operands
  constexpr bool kHigh = !OUT2 || WIDE;    // ... and those of 6, 7 and more
  switch (a.nk) {
    case 1: if (kLow) combine_body<T, MODE, 1, VEC, OUT2, false, NTP>(a, y0, k0, dt); break;
    case 2: if (kLow) combine_body<T, MODE, 2, VEC, OUT2, false, NTP>(a, y0, k0, dt); break;
    case 3: if (kLow) combine_body<T, MODE, 3, VEC, OUT2, false, NTP>(a, y0, k0, dt); break;
    case 4: if (kLow) combine_body<T, MODE, 4, VEC, OUT2, false, NTP>(a, y0, k0, dt); break;
    case 5: if (kLow) combine_body<T, MODE, 5, VEC, OUT2, false, NTP>(a, y0, k0, dt); break;
    case 6: if (kHigh) combine_body<T, MODE, 6, VEC, OUT2, false, NTP>(a, y0, k0, dt); break;
    case 7: if (kHigh) combine_body<T, MODE, 7, VEC, OUT2, false, NTP>(a, y0, k0, dt); break;
    default: if (kHigh) combine_generic<T, MODE, VEC, OUT2>(a, y0, k0, dt); break;
  }
}

// A stage whose earlier operands were already summed by the PREVIOUS stage's launch (which held them in registers anyway and emitted
// `sum_j k_j (beta_ij dt)` over them as its second output): this launch reads y0, that partial sum and the newest derivative(s) —
// Dopri5's stage 5: 3 arrays in, 1 out, instead of 6 in, 1 out; the emitting launch writes one array more.  Same left-to-right
// association as the full sum (`((..) + k_3 c_3) + k_4 c_4`, then `y0 +`): bit-identical.  Its own instantiation (registers).
template <typename T, bool VEC, int MODE = XDE_COMBINE_RK, int NTP = 0>
__global__ __launch_bounds__(kBlock) void xde_combine_pre_kernel(CombineArgs a) {
  int sel = 0;
  T dt;
  if (a.ctrl) {
    read_dt_sel<T>(a.ctrl, a.use_sel, dt, sel);
  } else {
    dt = T(a.dt_host);
  }
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(a.k[0]);  // (not the select-able f0: the operands here are the stage's NEWEST derivatives)
  switch (a.nk) {
    case 1: combine_body<T, MODE, 1, VEC, false, true, NTP>(a, y0, k0, dt); break;
    case 2: combine_body<T, MODE, 2, VEC, false, true, NTP>(a, y0, k0, dt); break;
    case 3: combine_body<T, MODE, 3, VEC, false, true, NTP>(a, y0, k0, dt); break;
    default: combine_body<T, MODE, 4, VEC, false, true, NTP>(a, y0, k0, dt); break;
  }
}

// (8..14 operands — Dopri8's later stages, the Adams predictor's long histories — take combine_generic's runtime loop.  A separate
// kernel with compile-time counts was measured in round 3 and is NOT faster here: 63.9 vs 62.6 us per launch on Dopri8's four
// long stages at config 2's size (0.80 of the HBM peak either way; the loop body is short enough for the compiler to keep several
// loads in flight at 8 waves/SIMD).  The error-norm pass is different: see xde_errnorm_wide_kernel.)

// ------------------------------------------------------------------------------------------
// fan-out: outs[j] = g * factor_j  (backward of the combine)
// ------------------------------------------------------------------------------------------
struct FanoutArgs {
  void* outs[XDE_MAX_K + 1];
  double factor[XDE_MAX_K + 1];
  const void* g;
  const double* dt_dev;
  int64_t n;
  int nout;
};

template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_fanout_kernel(FanoutArgs a) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  const T* __restrict__ g = static_cast<const T*>(a.g);
  const T dt = a.dt_dev ? T(*a.dt_dev) : T(1);
  const int nout = a.nout;
  const int64_t nvec = a.n / W;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nvec; i += stride) {
    P gv = P::load(g, i);
    for (int j = 0; j < nout; ++j) {
      const T f = T(a.factor[j]) * dt;
      P o;
#pragma unroll
      for (int w = 0; w < W; ++w) o.v[w] = gv.v[w] * f;
      o.store(static_cast<T*>(a.outs[j]), i);
    }
  }
  if (VEC) {
    const int64_t i = nvec * W + threadIdx.x;
    if (blockIdx.x == 0 && i < a.n)
      for (int j = 0; j < nout; ++j) static_cast<T*>(a.outs[j])[i] = g[i] * (T(a.factor[j]) * dt);
  }
}


// ------------------------------------------------------------------------------------------
// predicated commit (hipGraph pipeline): (y0, f0) <- (y1, f1) when the step was accepted
// ------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_commit_kernel(const xde_ctrl_t* c, T* __restrict__ y0, const T* __restrict__ y1,
                                                            T* __restrict__ f0, const T* __restrict__ f1, int64_t n) {
  if (!c->accept) return;
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  const int64_t nvec = n / W;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nvec; i += stride) {
    P a = P::load(y1, i);
    P b = P::load(f1, i);
    a.store(y0, i);
    b.store(f0, i);
  }
  if (VEC) {
    const int64_t i = nvec * W + threadIdx.x;
    if (blockIdx.x == 0 && i < n) {
      y0[i] = y1[i];
      f0[i] = f1[i];
    }
  }
}


}  // namespace

extern "C" {

static int stage_combine_impl(void* out, const void* y0, const void* y0_alt, const void* const* k, const void* k0_alt,
                              const double* coef, int nk, int mode, double scale, double dt_host, const xde_ctrl_t* ctrl,
                              int64_t n, int dtype, void* out2, const double* coef2, double damping, uint32_t nt_mask,
                              const void* pre, void* stream);

int xde_stage_combine(void* out, const void* y0, const void* y0_alt, const void* const* k, const void* k0_alt,
                      const double* coef, int nk, int mode, double scale, double dt_host, const xde_ctrl_t* ctrl,
                      int64_t n, int dtype, void* out2, const double* coef2, double damping, uint32_t nt_mask, void* stream) {
  return stage_combine_impl(out, y0, y0_alt, k, k0_alt, coef, nk, mode, scale, dt_host, ctrl, n, dtype, out2, coef2, damping, nt_mask,
                            nullptr, stream);
}

int xde_stage_combine_pre(void* out, const void* y0, const void* y0_alt, const void* pre, const void* const* k, const double* coef,
                          int nk, double dt_host, const xde_ctrl_t* ctrl, int64_t n, int dtype, uint32_t nt_mask, void* stream) {
  if (!pre) return fail(XDE_EBADARG, "xde_stage_combine_pre: null pointer");
  if (nk < 1 || nk > 4) return fail(XDE_EBADARG, "xde_stage_combine_pre: 1..4 new operands");
  return stage_combine_impl(out, y0, y0_alt, k, nullptr, coef, nk, XDE_COMBINE_RK, 1.0, dt_host, ctrl, n, dtype, nullptr, nullptr, 0.0,
                            nt_mask, pre, stream);
}

int xde_stage_combine_pre_weighted(void* out, const void* y0, const void* pre, const void* const* k, const double* coef, int nk,
                                   double scale, double dt_host, const xde_ctrl_t* ctrl, int64_t n, int dtype, double damping,
                                   void* stream) {
  if (!pre) return fail(XDE_EBADARG, "xde_stage_combine_pre_weighted: null pointer");
  if (nk < 1 || nk > 4) return fail(XDE_EBADARG, "xde_stage_combine_pre_weighted: 1..4 new operands");
  return stage_combine_impl(out, y0, nullptr, k, nullptr, coef, nk, XDE_COMBINE_WFUSE, scale, dt_host, ctrl, n, dtype, nullptr, nullptr,
                            damping, 0, pre, stream);
}

static int stage_combine_impl(void* out, const void* y0, const void* y0_alt, const void* const* k, const void* k0_alt,
                              const double* coef, int nk, int mode, double scale, double dt_host, const xde_ctrl_t* ctrl,
                              int64_t n, int dtype, void* out2, const double* coef2, double damping, uint32_t nt_mask,
                              const void* pre, void* stream) {
  if (!out || !y0 || !k || !coef) return fail(XDE_EBADARG, "xde_stage_combine: null pointer");
  if (nk < 1 || nk > XDE_MAX_K) return fail(XDE_EBADARG, "xde_stage_combine: nk out of range");
  if (n < 0) return fail(XDE_EBADARG, "xde_stage_combine: negative n");
  if (mode < 0 || mode > 2) return fail(XDE_EBADARG, "xde_stage_combine: bad mode");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_stage_combine: bad dtype");
  if (!pre && (y0_alt == nullptr) != (k0_alt == nullptr)) return fail(XDE_EBADARG, "xde_stage_combine: y0_alt/k0_alt must come together");
  if (y0_alt && !ctrl) return fail(XDE_EBADARG, "xde_stage_combine: operand select needs ctrl");
  if ((out2 == nullptr) != (coef2 == nullptr)) return fail(XDE_EBADARG, "xde_stage_combine: out2/coef2 must come together");
  if (out2 && mode == XDE_COMBINE_WFUSE) return fail(XDE_EBADARG, "xde_stage_combine: second output needs mode RK or FUSE");
  if (out2 && mode == XDE_COMBINE_FUSE && nk > 5) return fail(XDE_EBADARG, "xde_stage_combine: a FUSE launch emits for at most 5 operands");
  if (n == 0) return XDE_OK;
  CombineArgs a;
  memset(&a, 0, sizeof(a));
  a.out = out;
  a.out2 = out2;
  a.pre = pre;
  a.y0[0] = y0;
  a.y0[1] = y0_alt ? y0_alt : y0;
  a.k0_alt = k0_alt ? k0_alt : k[0];
  a.use_sel = y0_alt ? 1 : 0;
  bool vec = aligned16(out) && aligned16(y0) && aligned16(a.y0[1]) && aligned16(a.k0_alt);
  for (int j = 0; j < nk; ++j) {
    if (!k[j]) return fail(XDE_EBADARG, "xde_stage_combine: null k[j]");
    a.k[j] = k[j];
    a.coef[j] = coef[j];
    if (coef2) a.coef2[j] = coef2[j];
    vec = vec && aligned16(k[j]);
  }
  if (out2) vec = vec && aligned16(out2);
  if (pre) vec = vec && aligned16(pre);
  if (damping != 0.0 && mode == XDE_COMBINE_RK) return fail(XDE_EBADARG, "xde_stage_combine: damping applies to FUSE/WFUSE");
  a.damp = damping;
  // cache policy of the launch (combine_body, NTP): operands far larger than the Infinity Cache -> stream everything; the caller marks
  // EVERY derivative as read for the last time (bits 0..nk-1 of nt_mask; bit 31: y0 too) -> stream those; anything else -> default
  int ntp = 0;
  if (big_operand(n, dtype)) ntp = 2;
  else if ((nt_policy() & 4) && nk < 32 && (nt_mask & ((1u << nk) - 1u)) == ((1u << nk) - 1u)) ntp = (nt_mask >> 31) ? 2 : 1;
  a.nt = ntp;
  a.scale = scale;
  a.dt_host = dt_host;
  a.ctrl = ctrl;
  a.n = n;
  a.nk = nk;
  const int width = dtype == XDE_F32 ? 4 : 2;
  const int64_t work = vec ? (n + width - 1) / width : n;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > grid_cap()) blocks = grid_cap();
  if (blocks < 1) blocks = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double elt = dtype == XDE_F32 ? 4.0 : 8.0;
  const int kid = mode == XDE_COMBINE_RK ? XDE_KID_COMBINE : (mode == XDE_COMBINE_FUSE ? XDE_KID_COMBINE_FUSE : XDE_KID_COMBINE_WFUSE);
  ProfScope prof(kid, double(nk + 2 + (out2 ? 1 : 0) + (pre ? 1 : 0)) * double(n) * elt);
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
  if (!vec) ntp = 0;  // (element-wise path: unaligned views, never the hot path)
  if (mode != XDE_COMBINE_RK && ntp == 1) ntp = 0;  // (policy 1 is the adaptive solver's hint)
#define L_(...) XDE_LAUNCH((__VA_ARGS__), g, b, st, prof, a)
// the cache policies of a kernel whose other template arguments are fixed
#define L_NTP3(PFX, ...)                          \
  do {                                            \
    if (ntp == 2) L_(PFX<__VA_ARGS__, 2>);        \
    else if (ntp == 1) L_(PFX<__VA_ARGS__, 1>);   \
    else L_(PFX<__VA_ARGS__, 0>);                 \
  } while (0)
#define L_NTP3W(PFX, ...)                              \
  do {                                                 \
    if (ntp == 2) L_(PFX<__VA_ARGS__, 2, true>);       \
    else if (ntp == 1) L_(PFX<__VA_ARGS__, 1, true>);  \
    else L_(PFX<__VA_ARGS__, 0, true>);                \
  } while (0)
#define L_NTP2(PFX, ...)                          \
  do {                                            \
    if (ntp == 2) L_(PFX<__VA_ARGS__, 2>);        \
    else L_(PFX<__VA_ARGS__, 0>);                 \
  } while (0)
#define LAUNCH_COMBINE(T, MODE)                                 \
  do {                                                          \
    if (vec) L_NTP2(xde_combine_kernel, T, MODE, true, false);  \
    else L_(xde_combine_kernel<T, MODE, false, false>);         \
  } while (0)
#define LAUNCH_COMBINE2(T)                                                                \
  do {                                                                                    \
    if (vec && nk <= 5) L_NTP3(xde_combine_kernel, T, XDE_COMBINE_RK, true, true);        \
    else if (vec) L_NTP3W(xde_combine_kernel, T, XDE_COMBINE_RK, true, true);             \
    else if (nk <= 5) L_(xde_combine_kernel<T, XDE_COMBINE_RK, false, true>);             \
    else L_(xde_combine_kernel<T, XDE_COMBINE_RK, false, true, 0, true>);                 \
  } while (0)
#define LAUNCH_PRE(T)                                                      \
  do {                                                                     \
    if (vec) L_NTP3(xde_combine_pre_kernel, T, true, XDE_COMBINE_RK);      \
    else L_(xde_combine_pre_kernel<T, false>);                             \
  } while (0)
  if (pre && mode == XDE_COMBINE_WFUSE) {  // the fixed-step solvers' final weighted sum, its leading terms pre-summed (RK4: 6 -> 4 arrays)
    if (dtype == XDE_F32) {
      if (vec) L_NTP2(xde_combine_pre_kernel, float, true, XDE_COMBINE_WFUSE); else L_(xde_combine_pre_kernel<float, false, XDE_COMBINE_WFUSE>);
    } else {
      if (vec) L_NTP2(xde_combine_pre_kernel, double, true, XDE_COMBINE_WFUSE); else L_(xde_combine_pre_kernel<double, false, XDE_COMBINE_WFUSE>);
    }
  } else if (pre) {
    if (dtype == XDE_F32) LAUNCH_PRE(float);
    else LAUNCH_PRE(double);
  } else if (out2 && mode == XDE_COMBINE_FUSE) {  // a fixed-step stage input that also emits the final sum's leading terms
    if (dtype == XDE_F32) {
      if (vec) L_NTP2(xde_combine_kernel, float, XDE_COMBINE_FUSE, true, true); else L_(xde_combine_kernel<float, XDE_COMBINE_FUSE, false, true>);
    } else {
      if (vec) L_NTP2(xde_combine_kernel, double, XDE_COMBINE_FUSE, true, true); else L_(xde_combine_kernel<double, XDE_COMBINE_FUSE, false, true>);
    }
  } else if (out2) {
    if (dtype == XDE_F32) LAUNCH_COMBINE2(float);
    else LAUNCH_COMBINE2(double);
  } else if (mode == XDE_COMBINE_RK && vec) {
    if (dtype == XDE_F32) L_NTP3(xde_combine_kernel, float, XDE_COMBINE_RK, true, false);
    else L_NTP3(xde_combine_kernel, double, XDE_COMBINE_RK, true, false);
  } else if (dtype == XDE_F32) {
    if (mode == XDE_COMBINE_RK) LAUNCH_COMBINE(float, XDE_COMBINE_RK);
    else if (mode == XDE_COMBINE_FUSE) LAUNCH_COMBINE(float, XDE_COMBINE_FUSE);
    else LAUNCH_COMBINE(float, XDE_COMBINE_WFUSE);
  } else {
    if (mode == XDE_COMBINE_RK) LAUNCH_COMBINE(double, XDE_COMBINE_RK);
    else if (mode == XDE_COMBINE_FUSE) LAUNCH_COMBINE(double, XDE_COMBINE_FUSE);
    else LAUNCH_COMBINE(double, XDE_COMBINE_WFUSE);
  }
#undef LAUNCH_PRE
#undef L_NTP2
#undef L_NTP3W
#undef L_NTP3
#undef L_
#undef LAUNCH_COMBINE
#undef LAUNCH_COMBINE2
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_scale_fanout(void* const* outs, const void* g, const double* factors, int nout, const double* dt_dev, int64_t n,
                     int dtype, void* stream) {
  if (!outs || !g || !factors) return fail(XDE_EBADARG, "xde_scale_fanout: null pointer");
  if (nout < 1 || nout > XDE_MAX_K + 1) return fail(XDE_EBADARG, "xde_scale_fanout: nout out of range");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_scale_fanout: bad dtype");
  if (n < 0) return fail(XDE_EBADARG, "xde_scale_fanout: negative n");
  if (n == 0) return XDE_OK;
  FanoutArgs a;
  memset(&a, 0, sizeof(a));
  bool vec = aligned16(g);
  for (int j = 0; j < nout; ++j) {
    if (!outs[j]) return fail(XDE_EBADARG, "xde_scale_fanout: null outs[j]");
    a.outs[j] = outs[j];
    a.factor[j] = factors[j];
    vec = vec && aligned16(outs[j]);
  }
  a.g = g;
  a.dt_dev = dt_dev;
  a.n = n;
  a.nout = nout;
  const int width = dtype == XDE_F32 ? 4 : 2;
  const int64_t work = vec ? (n + width - 1) / width : n;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > grid_cap()) blocks = grid_cap();
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_COMBINE, double(nout + 1) * double(n) * (dtype == XDE_F32 ? 4.0 : 8.0));
  dim3 gr(static_cast<unsigned>(blocks)), b(kBlock);
  if (dtype == XDE_F32) {
    if (vec) XDE_LAUNCH((xde_fanout_kernel<float, true>), gr, b, st, prof, a);
    else XDE_LAUNCH((xde_fanout_kernel<float, false>), gr, b, st, prof, a);
  } else {
    if (vec) XDE_LAUNCH((xde_fanout_kernel<double, true>), gr, b, st, prof, a);
    else XDE_LAUNCH((xde_fanout_kernel<double, false>), gr, b, st, prof, a);
  }
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_commit(const xde_ctrl_t* ctrl, void* y0_dst, const void* y1_src, void* f0_dst, const void* f1_src, int64_t n,
               int dtype, void* stream) {
  if (!ctrl || !y0_dst || !y1_src || !f0_dst || !f1_src) return fail(XDE_EBADARG, "xde_commit: null pointer");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_commit: bad dtype");
  if (n < 0) return fail(XDE_EBADARG, "xde_commit: negative n");
  if (n == 0) return XDE_OK;
  const int width = dtype == XDE_F32 ? 4 : 2;
  const bool vec = aligned16(y0_dst) && aligned16(y1_src) && aligned16(f0_dst) && aligned16(f1_src);
  const int64_t work = vec ? (n + width - 1) / width : n;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > grid_cap()) blocks = grid_cap();
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_COMMIT, 4.0 * double(n) * (dtype == XDE_F32 ? 4.0 : 8.0));
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
  if (dtype == XDE_F32) {
    if (vec)
      XDE_LAUNCH((xde_commit_kernel<float, true>), g, b, st, prof, ctrl, static_cast<float*>(y0_dst),
                 static_cast<const float*>(y1_src), static_cast<float*>(f0_dst), static_cast<const float*>(f1_src), n);
    else
      XDE_LAUNCH((xde_commit_kernel<float, false>), g, b, st, prof, ctrl, static_cast<float*>(y0_dst),
                 static_cast<const float*>(y1_src), static_cast<float*>(f0_dst), static_cast<const float*>(f1_src), n);
  } else {
    if (vec)
      XDE_LAUNCH((xde_commit_kernel<double, true>), g, b, st, prof, ctrl, static_cast<double*>(y0_dst),
                 static_cast<const double*>(y1_src), static_cast<double*>(f0_dst), static_cast<const double*>(f1_src), n);
    else
      XDE_LAUNCH((xde_commit_kernel<double, false>), g, b, st, prof, ctrl, static_cast<double*>(y0_dst),
                 static_cast<const double*>(y1_src), static_cast<double*>(f0_dst), static_cast<const double*>(f1_src), n);
  }
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}


}  // extern "C"
