// Device body of the fused error-estimate + tolerance + norm pass (shared by xde_errnorm_kernel and the opt-in
// one-launch error-norm + controller kernel), and the host-side argument setup both entry points use.
#pragma once
#include "xde_common.hpp"
#include "xde_reduce.hpp"

namespace xde {

// ------------------------------------------------------------------------------------------
// K2: fused error combine + tolerance scaling + norm partials (+ non-finite count of y0)
// ------------------------------------------------------------------------------------------
template <typename T, int NK, int NORM, bool VEC, bool NT, bool PRE = false>
__device__ __forceinline__ void errnorm_body(const ErrArgs& a, const T* __restrict__ y0, const T* __restrict__ k0,
                                             T dt, int seg, int lb, int nb, double& acc_out, int& nf_out) {
  const T* __restrict__ epre = static_cast<const T*>(a.e_pre);
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  const T* kp[NK];
  T c[NK];
  kp[0] = k0;
#pragma unroll
  for (int j = 1; j < NK; ++j) kp[j] = static_cast<const T*>(a.k[j]);
#pragma unroll
  for (int j = 0; j < NK; ++j) c[j] = dt * T(a.coef[j]);  // `dt * tableau.c_error`
  const T* __restrict__ y1 = static_cast<const T*>(a.y1);
  const T rtol = T(a.rtol), atol = T(a.atol);
  const int64_t start = a.map.seg_start[seg];
  const int64_t len = a.map.seg_len[seg];
  const int64_t nvec = len / W;
  const int64_t vbase = start / W;  // host guarantees start % W == 0 on the vector path
  const int64_t stride = int64_t(nb) * kBlock;
  T acc = T(0);
  double acc64 = 0.0;  // the per-lane sum moves here every 64 iterations: huge N / small grids stay accurate
  int it = 0;
  int nf = 0;
  auto one = [&](T e, T y0v, T y1v) {
    T tol = atol + rtol * fmax_(abs_(y0v), abs_(y1v));
    T r = e / tol;
    if (NORM == XDE_NORM_RMS) {
      T ar = abs_(r);
      acc = acc + ar * ar;
    } else {
      T ar = abs_(r);
      acc = (ar != ar || acc != acc) ? (ar != ar ? ar : acc) : (ar > acc ? ar : acc);
    }
    nf += finite_(y0v) ? 0 : 1;
  };
  for (int64_t i = int64_t(lb) * kBlock + threadIdx.x; i < nvec; i += stride) {
    // NT: y0 and k_0..k_{NK-2} are read here for the last time when the step is accepted (the common case) and
    // are streamed; y1 and the last operand (f1) become the next step's (y0, f0), which all its stage combines
    // re-read, so they keep the default policy and stay in the Infinity Cache
    // (program order: operands whose address does not depend on the device-side select first, y0 / k0 last — see combine_body)
    P y1v = P::load(y1, vbase + i);
    P ep;
    if (PRE) ep = NT ? P::load_nt(epre, vbase + i) : P::load(epre, vbase + i);
    P kk[NK];
#pragma unroll
    for (int j = NK - 1; j >= 1; --j) kk[j] = (NT && j < NK - 1) ? P::load_nt(kp[j], vbase + i) : P::load(kp[j], vbase + i);
    kk[0] = (NT && 0 < NK - 1) ? P::load_nt(kp[0], vbase + i) : P::load(kp[0], vbase + i);
    P y0v = NT ? P::load_nt(y0, vbase + i) : P::load(y0, vbase + i);
#pragma unroll
    for (int w = 0; w < W; ++w) {
      T e = PRE ? ep.v[w] + kk[0].v[w] * c[0] : kk[0].v[w] * c[0];
#pragma unroll
      for (int j = 1; j < NK; ++j) e = e + kk[j].v[w] * c[j];
      one(e, y0v.v[w], y1v.v[w]);
    }
    if (NORM == XDE_NORM_RMS && (++it & 63) == 0) {
      acc64 += double(acc);
      acc = T(0);
    }
  }
  if (VEC && lb == 0) {
    const int64_t i = start + nvec * W + threadIdx.x;
    if (i < start + len) {
      T e = PRE ? epre[i] + kp[0][i] * c[0] : kp[0][i] * c[0];
      for (int j = 1; j < NK; ++j) e = e + kp[j][i] * c[j];
      one(e, y0[i], y1[i]);
    }
  }
  acc_out = NORM == XDE_NORM_RMS ? acc64 + double(acc) : double(acc);
  nf_out = nf;
}

// non-finite test as ONE VALU instruction (v_cmp_class: sNaN | qNaN | -inf | +inf); the count is taken per WAVE on the scalar unit
// (s_bcnt1 of the compare mask), so it costs the vector pipe nothing more — (x - x) == 0 was sub + compare + select + add per element
__device__ __forceinline__ bool nonfinite_class(float x) { return __builtin_amdgcn_classf(x, 0x207); }
__device__ __forceinline__ bool nonfinite_class(double x) { return __builtin_amdgcn_class(x, 0x207); }

// The FSAL pair's error-norm pass (Dopri5 and friends: e = e_pre + k_S * (dt * c_S), vector path) as a kernel body of its own.
//   * The operands whose address does not depend on the control block (y1, e_pre, k_S) are requested BEFORE the block's
//     dt / accept words are waited for; only y0 (the speculative pipeline's select) follows them.
//   * The loop is software-pipelined by hand: the loads of vector i + stride are issued before vector i is reduced, so a lane
//     keeps 8 x 16 bytes in flight while it divides (at config 4's shard size a lane runs 8 iterations and the launch holds
//     only 2 waves per SIMD: there is nobody else to hide the ~100 VALU instructions of an iteration behind).
// Same per-lane summation order as errnorm_body (same indices, same fp32 accumulate, same 64-iteration fp64 flush): same bits.
template <typename T, int NORM, bool NT>
__device__ __forceinline__ void errnorm_pre_body(const ErrArgs& a, int seg, int lb, int nb, double& acc_out, double& nf_out) {
  using P = Pack<T, true>;
  constexpr int W = P::W;
  const T* __restrict__ epre = static_cast<const T*>(a.e_pre);
  const T* __restrict__ kl = static_cast<const T*>(a.k[0]);
  const T* __restrict__ y1 = static_cast<const T*>(a.y1);
  const int64_t start = a.map.seg_start[seg];
  const int64_t len = a.map.seg_len[seg];
  const int64_t nvec = len / W;
  const int64_t vbase = start / W;
  const int64_t stride = int64_t(nb) * kBlock;
  const int64_t i0 = int64_t(lb) * kBlock + threadIdx.x;
  // the control block's two words are REQUESTED here and waited for only where y0's address is formed, below the first loads
  // (the host launches this kernel only with a control block: a branch on `a.ctrl` here would put the wait back in front)
  const double dtd = __builtin_nontemporal_load(&a.ctrl->dt);
  const int32_t accw = __builtin_nontemporal_load(&a.ctrl->accept);
  const T rtol = T(a.rtol), atol = T(a.atol);
  T acc = T(0);
  double acc64 = 0.0;
  int it = 0;
  int nfw = 0;  // per-wave count, complete in the wave's first lane (it has the smallest index: active whenever any lane is)
  T c = T(0);
  const T* __restrict__ y0 = static_cast<const T*>(a.y0[0]);
  auto one = [&](T e, T y0e, T y1e) {
    T tol = atol + rtol * fmax_(abs_(y0e), abs_(y1e));
    T ar = abs_(e / tol);
    if (NORM == XDE_NORM_RMS) {
      acc = acc + ar * ar;
    } else {
      acc = (ar != ar || acc != acc) ? (ar != ar ? ar : acc) : (ar > acc ? ar : acc);
    }
    nfw += __popcll(__ballot(nonfinite_class(y0e)));
  };
  auto reduce_vec = [&](const P& epv, const P& kv, const P& y0v, const P& y1v) {
#pragma unroll
    for (int w = 0; w < W; ++w) one(epv.v[w] + kv.v[w] * c, y0v.v[w], y1v.v[w]);
    if (NORM == XDE_NORM_RMS && (++it & 63) == 0) {
      acc64 += double(acc);
      acc = T(0);
    }
  };
  if (i0 < nvec) {
    P y1v = P::load(y1, vbase + i0);
    P ep = NT ? P::load_nt(epre, vbase + i0) : P::load(epre, vbase + i0);
    P kk = P::load(kl, vbase + i0);
    // (pinned: without this the scheduler hoists the select — and with it the wait for the control block's words — above the three
    //  loads; 0.2 us per launch at 16 MiB operands, profiles/r04_errnorm_lab.txt k_pipe2)
    __builtin_amdgcn_sched_barrier(0);
    const int sel = (a.use_sel && accw) ? 1 : 0;
    y0 = static_cast<const T*>(a.y0[sel]);
    P y0v = NT ? P::load_nt(y0, vbase + i0) : P::load(y0, vbase + i0);
    c = T(dtd) * T(a.coef[0]);  // `dt * tableau.c_error`
    for (int64_t in = i0 + stride; in < nvec; in += stride) {
      // (no condition inside the body: the next vector's four loads are in flight while the current one is reduced)
      P y1n = P::load(y1, vbase + in);
      P epn = NT ? P::load_nt(epre, vbase + in) : P::load(epre, vbase + in);
      P kn = P::load(kl, vbase + in);
      P y0n = NT ? P::load_nt(y0, vbase + in) : P::load(y0, vbase + in);
      reduce_vec(ep, kk, y0v, y1v);
      y1v = y1n;
      ep = epn;
      kk = kn;
      y0v = y0n;
    }
    reduce_vec(ep, kk, y0v, y1v);
  }
  if (lb == 0) {  // the scalar tail of the segment (len % W elements)
    y0 = static_cast<const T*>(a.y0[(a.use_sel && accw) ? 1 : 0]);
    c = T(dtd) * T(a.coef[0]);
    const int64_t j = start + nvec * W + threadIdx.x;
    const bool t = j < start + len;
    T e = T(0), y0e = T(0), y1e = T(1);
    if (t) {
      e = epre[j] + kl[j] * c;
      y0e = y0[j];
      y1e = y1[j];
    }
    if (t) {
      T tol = atol + rtol * fmax_(abs_(y0e), abs_(y1e));
      T ar = abs_(e / tol);
      if (NORM == XDE_NORM_RMS) acc = acc + ar * ar;
      else acc = (ar != ar || acc != acc) ? (ar != ar ? ar : acc) : (ar > acc ? ar : acc);
    }
    nfw += __popcll(__ballot(t && nonfinite_class(y0e)));
  }
  acc_out = NORM == XDE_NORM_RMS ? acc64 + double(acc) : double(acc);
  nf_out = (threadIdx.x & 63) == 0 ? double(nfw) : 0.0;  // one lane per wave carries the wave's count into the reduction
}

// operand count > 8 (Dopri8): runtime loop over operands, same arithmetic order
template <typename T, int NORM, bool VEC>
__device__ void errnorm_generic(const ErrArgs& a, const T* __restrict__ y0, const T* __restrict__ k0, T dt, int seg,
                                int lb, int nb, double& acc_out, int& nf_out) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  const int nk = a.nk;
  const T* __restrict__ y1 = static_cast<const T*>(a.y1);
  const T rtol = T(a.rtol), atol = T(a.atol);
  const int64_t start = a.map.seg_start[seg];
  const int64_t len = a.map.seg_len[seg];
  const int64_t nvec = len / W;
  const int64_t vbase = start / W;
  const int64_t stride = int64_t(nb) * kBlock;
  T acc = T(0);
  double acc64 = 0.0;  // the per-lane sum moves here every 64 iterations: huge N / small grids stay accurate
  int it = 0;
  int nf = 0;
  auto one = [&](T e, T y0v, T y1v) {
    T tol = atol + rtol * fmax_(abs_(y0v), abs_(y1v));
    T ar = abs_(e / tol);
    if (NORM == XDE_NORM_RMS) {
      acc = acc + ar * ar;
    } else {
      acc = (ar != ar || acc != acc) ? (ar != ar ? ar : acc) : (ar > acc ? ar : acc);
    }
    nf += finite_(y0v) ? 0 : 1;
  };
  for (int64_t i = int64_t(lb) * kBlock + threadIdx.x; i < nvec; i += stride) {
    P y0v = P::load(y0, vbase + i);
    P y1v = P::load(y1, vbase + i);
    P e;
    for (int j = 0; j < nk; ++j) {
      const T* kj = j == 0 ? k0 : static_cast<const T*>(a.k[j]);
      P kk = P::load(kj, vbase + i);
      T cj = dt * T(a.coef[j]);
#pragma unroll
      for (int w = 0; w < W; ++w) e.v[w] = (j == 0) ? kk.v[w] * cj : e.v[w] + kk.v[w] * cj;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) one(e.v[w], y0v.v[w], y1v.v[w]);
    if (NORM == XDE_NORM_RMS && (++it & 63) == 0) {
      acc64 += double(acc);
      acc = T(0);
    }
  }
  if (VEC && lb == 0) {
    const int64_t i = start + nvec * W + threadIdx.x;
    if (i < start + len) {
      T e = T(0);
      for (int j = 0; j < nk; ++j) {
        const T* kj = j == 0 ? k0 : static_cast<const T*>(a.k[j]);
        T term = kj[i] * (dt * T(a.coef[j]));
        e = (j == 0) ? term : e + term;
      }
      one(e, y0[i], y1[i]);
    }
  }
  acc_out = NORM == XDE_NORM_RMS ? acc64 + double(acc) : double(acc);
  nf_out = nf;
}

// 9..14 operands (Dopri8's error estimate reads 11 derivatives): compile-time counts as well, but in a kernel of their own
// (xde_errnorm_wide_kernel) — inside the common kernel their 14 x 16-byte registers per lane cost it 236 VGPRs and scratch.
template <typename T, int NORM, bool VEC>
__device__ __forceinline__ void errnorm_dispatch_wide(const ErrArgs& a, const T* __restrict__ y0, const T* __restrict__ k0, T dt,
                                                      int seg, int lb, int nb, double& acc, int& nf) {
  switch (a.nk) {
    case 9: if (a.nt) errnorm_body<T, 9, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 9, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 10: if (a.nt) errnorm_body<T, 10, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 10, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 11: if (a.nt) errnorm_body<T, 11, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 11, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 12: if (a.nt) errnorm_body<T, 12, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 12, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 13: if (a.nt) errnorm_body<T, 13, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 13, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 14: if (a.nt) errnorm_body<T, 14, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 14, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    default: errnorm_generic<T, NORM, VEC>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
  }
}

template <typename T, int NORM, bool VEC>
__device__ __forceinline__ void errnorm_dispatch(const ErrArgs& a, const T* __restrict__ y0, const T* __restrict__ k0, T dt,
                                                 int seg, int lb, int nb, double& acc, int& nf) {
  if (a.e_pre) {  // host guarantees nk == 1 in this mode: e = e_pre + k_last * (dt * c_last)
    if (a.nt) errnorm_body<T, 1, NORM, VEC, true, true>(a, y0, k0, dt, seg, lb, nb, acc, nf);
    else errnorm_body<T, 1, NORM, VEC, false, true>(a, y0, k0, dt, seg, lb, nb, acc, nf);
  } else
  switch (a.nk) {
    case 1: if (a.nt) errnorm_body<T, 1, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 1, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 2: if (a.nt) errnorm_body<T, 2, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 2, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 3: if (a.nt) errnorm_body<T, 3, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 3, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 4: if (a.nt) errnorm_body<T, 4, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 4, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 5: if (a.nt) errnorm_body<T, 5, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 5, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 6: if (a.nt) errnorm_body<T, 6, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 6, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 7: if (a.nt) errnorm_body<T, 7, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 7, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    case 8: if (a.nt) errnorm_body<T, 8, NORM, VEC, true>(a, y0, k0, dt, seg, lb, nb, acc, nf); else errnorm_body<T, 8, NORM, VEC, false>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
    default: errnorm_generic<T, NORM, VEC>(a, y0, k0, dt, seg, lb, nb, acc, nf); break;
  }
}


inline int setup_err_args(const char* who, const void* const* k, const void* k0_alt, const double* c_err, int nk, const void* y0,
                          const void* y0_alt, const void* y1, double rtol, double atol, double dt_host, const xde_ctrl_t* ctrl,
                          const xde_segments_t* segs, int norm_kind, int dtype, void* ws, const void* e_pre, ErrArgs* out,
                          bool* vec_out, int* nblocks_out, double* bytes_out, int cap_override = 0) {
  const std::string w(who);
  if (!k || !c_err || !y0 || !y1 || !ws || !segs) return fail(XDE_EBADARG, w + ": null pointer");
  if (int rc0 = check_segments(segs)) return rc0;
  if (e_pre && nk != 1) return fail(XDE_EBADARG, w + ": e_pre takes exactly one remaining operand");
  if (nk < 1 || nk > XDE_MAX_K) return fail(XDE_EBADARG, w + ": nk out of range");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, w + ": bad dtype");
  if (norm_kind != XDE_NORM_RMS && norm_kind != XDE_NORM_LINF) return fail(XDE_EBADARG, w + ": bad norm");
  if (!e_pre && (y0_alt == nullptr) != (k0_alt == nullptr)) return fail(XDE_EBADARG, w + ": y0_alt/k0_alt must come together");
  if (y0_alt && !ctrl) return fail(XDE_EBADARG, w + ": operand select needs ctrl");
  ErrArgs& a = *out;
  memset(&a, 0, sizeof(a));
  a.y0[0] = y0;
  a.y0[1] = y0_alt ? y0_alt : y0;
  a.k0_alt = k0_alt ? k0_alt : k[0];
  a.use_sel = y0_alt ? 1 : 0;
  a.y1 = y1;
  const int width = dtype == XDE_F32 ? 4 : 2;
  bool vec = aligned16(y0) && aligned16(a.y0[1]) && aligned16(y1) && aligned16(a.k0_alt) && segs_vec_ok(segs, width);
  for (int j = 0; j < nk; ++j) {
    if (!k[j]) return fail(XDE_EBADARG, w + ": null k[j]");
    a.k[j] = k[j];
    a.coef[j] = c_err[j];
    vec = vec && aligned16(k[j]);
  }
  a.rtol = rtol;
  a.atol = atol;
  a.dt_host = dt_host;
  a.ctrl = ctrl;
  a.slot = slot_ptr(ws, 0);
  a.nk = nk;
  a.e_pre = e_pre;
  if (e_pre) vec = vec && aligned16(e_pre);
  a.nt = nt_policy() & 1;
  {
    int64_t tot = 0;
    for (int s2 = 0; s2 < segs->n_seg; ++s2) tot += segs->seg_len[s2];
    if (big_operand(tot, dtype)) a.nt = 1;
  }
  int rc = build_segmap(segs, vec ? width : 1, vec, &a.map, nblocks_out, cap_override);
  if (rc != XDE_OK) return rc;
  double total = 0;
  for (int s = 0; s < segs->n_seg; ++s) total += double(segs->seg_len[s]);
  *bytes_out = double(nk + 2 + (e_pre ? 1 : 0)) * total * (dtype == XDE_F32 ? 4.0 : 8.0);
  *vec_out = vec;
  return XDE_OK;
}


}  // namespace xde
