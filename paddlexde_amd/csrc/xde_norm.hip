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
// K2: error norm partials, scaled norms of the initial-step heuristic, finalize, the err/tol tensor for custom norms.

#include "xde_common.hpp"
#include "xde_reduce.hpp"
#include "xde_errnorm_device.hpp"

using namespace xde;

namespace {

template <typename T, int NORM, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_errnorm_kernel(ErrArgs a) {
  int sel = 0;
  T dt;
  if (a.ctrl) {
    read_dt_sel<T>(a.ctrl, a.use_sel, dt, sel);
  } else {
    dt = T(a.dt_host);
  }
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  const int seg = find_segment(a.map, blockIdx.x);
  const int lb = blockIdx.x - a.map.seg_blk[seg];
  const int nb = a.map.seg_blk[seg + 1] - a.map.seg_blk[seg];
  double acc = 0.0;
  int nf = 0;
  errnorm_dispatch<T, NORM, VEC>(a, y0, k0, dt, seg, lb, nb, acc, nf);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.slot->nblocks = gridDim.x;
    a.slot->n_seg = a.map.n_seg;
    a.slot->norm_kind = NORM;
  }
  block_reduce_store<NORM>(acc, double(nf), a.slot, seg);
}

// The FSAL pair's pass (e_pre + one derivative, vector path): errnorm_pre_body — loads ahead of the control-block read, hand-pipelined loop
template <typename T, int NORM, bool NT>
__global__ __launch_bounds__(kBlock) void xde_errnorm_pre_kernel(ErrArgs a) {
  const int seg = find_segment(a.map, blockIdx.x);
  const int lb = blockIdx.x - a.map.seg_blk[seg];
  const int nb = a.map.seg_blk[seg + 1] - a.map.seg_blk[seg];
  double acc = 0.0, nf = 0.0;
  errnorm_pre_body<T, NORM, NT>(a, seg, lb, nb, acc, nf);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.slot->nblocks = gridDim.x;
    a.slot->n_seg = a.map.n_seg;
    a.slot->norm_kind = NORM;
  }
  block_reduce_store<NORM>(acc, nf, a.slot, seg);
}

template <typename T, int NORM, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_errnorm_wide_kernel(ErrArgs a) {
  int sel = 0;
  T dt;
  if (a.ctrl) {
    read_dt_sel<T>(a.ctrl, a.use_sel, dt, sel);
  } else {
    dt = T(a.dt_host);
  }
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  const int seg = find_segment(a.map, blockIdx.x);
  const int lb = blockIdx.x - a.map.seg_blk[seg];
  const int nb = a.map.seg_blk[seg + 1] - a.map.seg_blk[seg];
  double acc = 0.0;
  int nf = 0;
  errnorm_dispatch_wide<T, NORM, VEC>(a, y0, k0, dt, seg, lb, nb, acc, nf);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.slot->nblocks = gridDim.x;
    a.slot->n_seg = a.map.n_seg;
    a.slot->norm_kind = NORM;
  }
  block_reduce_store<NORM>(acc, double(nf), a.slot, seg);
}

// ------------------------------------------------------------------------------------------
// scaled norms for select_initial_step:  norm(a / scale)  or  norm((a - b) / scale)
// ------------------------------------------------------------------------------------------
template <typename T, int NORM, bool VEC, bool DIFF>
__global__ __launch_bounds__(kBlock) void xde_scalednorm_kernel(ScaledArgs s) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  const T* __restrict__ a = static_cast<const T*>(s.a);
  const T* __restrict__ b = static_cast<const T*>(s.b);
  const T* __restrict__ y0 = static_cast<const T*>(s.y0);
  const int seg = find_segment(s.map, blockIdx.x);
  const int lb = blockIdx.x - s.map.seg_blk[seg];
  const int nb = s.map.seg_blk[seg + 1] - s.map.seg_blk[seg];
  const T rtol = T(s.rtol), atol = T(s.atol);
  const int64_t start = s.map.seg_start[seg];
  const int64_t len = s.map.seg_len[seg];
  const int64_t nvec = len / W;
  const int64_t vbase = start / W;
  const int64_t stride = int64_t(nb) * kBlock;
  T acc = T(0);
  double acc64 = 0.0;  // (see errnorm_body)
  int it = 0;
  int nf = 0;
  auto one = [&](T av, T bv, T yv) {
    T scale = atol + abs_(yv) * rtol;
    T num = DIFF ? (av - bv) : av;
    T r = abs_(num / scale);
    if (NORM == XDE_NORM_RMS) {
      acc = acc + r * r;
    } else {
      acc = (r != r || acc != acc) ? (r != r ? r : acc) : (r > acc ? r : acc);
    }
    nf += finite_(yv) ? 0 : 1;
  };
  for (int64_t i = int64_t(lb) * kBlock + threadIdx.x; i < nvec; i += stride) {
    P av = P::load(a, vbase + i);
    P yv = P::load(y0, vbase + i);
    P bv = av;
    if (DIFF) bv = P::load(b, vbase + i);
#pragma unroll
    for (int w = 0; w < W; ++w) one(av.v[w], bv.v[w], yv.v[w]);
    if (NORM == XDE_NORM_RMS && (++it & 63) == 0) {
      acc64 += double(acc);
      acc = T(0);
    }
  }
  if (VEC && lb == 0) {
    const int64_t i = start + nvec * W + threadIdx.x;
    if (i < start + len) one(a[i], DIFF ? b[i] : a[i], y0[i]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    s.slot->nblocks = gridDim.x;
    s.slot->n_seg = s.map.n_seg;
    s.slot->norm_kind = NORM;
  }
  block_reduce_store<NORM>(NORM == XDE_NORM_RMS ? acc64 + double(acc) : double(acc), double(nf), s.slot, seg);
}

// norm(y0 / scale) and norm(f0 / scale) of the heuristic's first phase (base_adaptive_solver.py:50-53) in ONE pass over (y0, f0): they share
// `scale`, so the second pass over y0 (and its launch) is saved — SURVEY A6's "scaled-norm of two vectors in one pass".  Each norm's
// partial records go to a slot of their own (d0 -> slot, d1 -> slot_b), element for element what two xde_scalednorm_kernel launches
// on the same grid write: same per-lane order, same fp64 flush period, same workgroup reduction — same bits.
template <typename T, int NORM, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_scalednorm2_kernel(ScaledArgs s, NormSlot* slot_b) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  const T* __restrict__ f0 = static_cast<const T*>(s.a);
  const T* __restrict__ y0 = static_cast<const T*>(s.y0);
  const int seg = find_segment(s.map, blockIdx.x);
  const int lb = blockIdx.x - s.map.seg_blk[seg];
  const int nb = s.map.seg_blk[seg + 1] - s.map.seg_blk[seg];
  const T rtol = T(s.rtol), atol = T(s.atol);
  const int64_t start = s.map.seg_start[seg];
  const int64_t len = s.map.seg_len[seg];
  const int64_t nvec = len / W;
  const int64_t vbase = start / W;
  const int64_t stride = int64_t(nb) * kBlock;
  T acc0 = T(0), acc1 = T(0);
  double acc64_0 = 0.0, acc64_1 = 0.0;  // (see errnorm_body)
  int it = 0;
  int nf = 0;
  auto one = [&](T fv, T yv) {
    const T scale = atol + abs_(yv) * rtol;
    const T r0 = abs_(yv / scale);
    const T r1 = abs_(fv / scale);
    if (NORM == XDE_NORM_RMS) {
      acc0 = acc0 + r0 * r0;
      acc1 = acc1 + r1 * r1;
    } else {
      acc0 = (r0 != r0 || acc0 != acc0) ? (r0 != r0 ? r0 : acc0) : (r0 > acc0 ? r0 : acc0);
      acc1 = (r1 != r1 || acc1 != acc1) ? (r1 != r1 ? r1 : acc1) : (r1 > acc1 ? r1 : acc1);
    }
    nf += finite_(yv) ? 0 : 1;
  };
  for (int64_t i = int64_t(lb) * kBlock + threadIdx.x; i < nvec; i += stride) {
    P fv = P::load(f0, vbase + i);
    P yv = P::load(y0, vbase + i);
#pragma unroll
    for (int w = 0; w < W; ++w) one(fv.v[w], yv.v[w]);
    if (NORM == XDE_NORM_RMS && (++it & 63) == 0) {
      acc64_0 += double(acc0);
      acc64_1 += double(acc1);
      acc0 = T(0);
      acc1 = T(0);
    }
  }
  if (VEC && lb == 0) {
    const int64_t i = start + nvec * W + threadIdx.x;
    if (i < start + len) one(f0[i], y0[i]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    s.slot->nblocks = slot_b->nblocks = gridDim.x;
    s.slot->n_seg = slot_b->n_seg = s.map.n_seg;
    s.slot->norm_kind = slot_b->norm_kind = NORM;
  }
  block_reduce_store<NORM>(NORM == XDE_NORM_RMS ? acc64_0 + double(acc0) : double(acc0), double(nf), s.slot, seg);
  __syncthreads();  // (the reduction's LDS staging is shared by the two calls)
  block_reduce_store<NORM>(NORM == XDE_NORM_RMS ? acc64_1 + double(acc1) : double(acc1), double(nf), slot_b, seg);
}

__global__ __launch_bounds__(kBlock) void xde_finalize_kernel(const NormSlot* slot, double* sums_out) {
  __shared__ double seg_val[XDE_MAX_SEG];
  __shared__ double seg_nf[XDE_MAX_SEG];
  if (threadIdx.x < XDE_MAX_SEG) {
    seg_val[threadIdx.x] = 0.0;
    seg_nf[threadIdx.x] = 0.0;
  }
  __syncthreads();
  reduce_partials(slot, seg_val, seg_nf);
  if (threadIdx.x < XDE_MAX_SEG) {
    sums_out[threadIdx.x] = seg_val[threadIdx.x];
    sums_out[XDE_MAX_SEG + threadIdx.x] = seg_nf[threadIdx.x];
  }
}


struct SegCounts {
  double c[XDE_MAX_SEG];
};

__global__ void xde_norm_result_kernel(const double* sums, SegCounts counts, int n_seg, int norm_kind, int state_dtype,
                                       double* result_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    result_out[0] = norm_from_sums(sums, counts.c, n_seg, norm_kind, state_dtype, nullptr);
  }
}


// ------------------------------------------------------------------------------------------
// element-wise error ratio (materialised only for user-supplied norm callables)
// ------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_ratio_kernel(ErrArgs a, T* __restrict__ out, int64_t n, double* nonfinite_out) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  int sel = 0;
  T dt;
  if (a.ctrl) {
    read_dt_sel<T>(a.ctrl, a.use_sel, dt, sel);
  } else {
    dt = T(a.dt_host);
  }
  const T* __restrict__ y0 = static_cast<const T*>(a.y0[sel]);
  const T* __restrict__ k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  const T* __restrict__ y1 = static_cast<const T*>(a.y1);
  int nf = 0;
  const T rtol = T(a.rtol), atol = T(a.atol);
  const int nk = a.nk;
  const int64_t nvec = n / W;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nvec; i += stride) {
    P y0v = P::load(y0, i);
    P y1v = P::load(y1, i);
    P e;
    for (int j = 0; j < nk; ++j) {
      P kk = P::load(j == 0 ? k0 : static_cast<const T*>(a.k[j]), i);
      T cj = dt * T(a.coef[j]);
#pragma unroll
      for (int w = 0; w < W; ++w) e.v[w] = (j == 0) ? kk.v[w] * cj : e.v[w] + kk.v[w] * cj;
    }
    P o;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      o.v[w] = e.v[w] / (atol + rtol * fmax_(abs_(y0v.v[w]), abs_(y1v.v[w])));
      nf += finite_(y0v.v[w]) ? 0 : 1;
    }
    o.store(out, i);
  }
  if (VEC) {
    const int64_t i = nvec * W + threadIdx.x;
    if (blockIdx.x == 0 && i < n) {
      T e = T(0);
      for (int j = 0; j < nk; ++j) {
        T term = (j == 0 ? k0 : static_cast<const T*>(a.k[j]))[i] * (dt * T(a.coef[j]));
        e = (j == 0) ? term : e + term;
      }
      out[i] = e / (atol + rtol * fmax_(abs_(y0[i]), abs_(y1[i])));
      nf += finite_(y0[i]) ? 0 : 1;
    }
  }
  // the `isfinite(y0).all()` pass (base_adaptive_solver_rk.py:201) rides along: a count, added atomically (integers: exact in
  // any order) to a device double the caller zeroed; non-finite states are the exception, so is the atomic
  if (nonfinite_out && __any(nf != 0)) {
    double c = double(nf);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c != 0.0) atomicAdd(nonfinite_out, c);
  }
}


}  // namespace

extern "C" {

int xde_error_norm_partial(const void* const* k, const void* k0_alt, const double* c_err, int nk, const void* y0,
                           const void* y0_alt, const void* y1, double rtol, double atol, double dt_host,
                           const xde_ctrl_t* ctrl, const xde_segments_t* segs, int norm_kind, int dtype, void* ws,
                           const void* e_pre, void* stream) {
  ErrArgs a;
  bool vec = false;
  int nblocks = 0;
  double bytes = 0;
  int rc = setup_err_args("xde_error_norm_partial", k, k0_alt, c_err, nk, y0, y0_alt, y1, rtol, atol, dt_host, ctrl, segs, norm_kind,
                          dtype, ws, e_pre, &a, &vec, &nblocks, &bytes);
  if (rc != XDE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_ERRNORM, bytes);
  dim3 g(nblocks), b(kBlock);
  // (operands of >= 64 MiB are HBM-served, not Infinity-Cache-served: there the pipelined body's 8 streams per wave cost DRAM page
  //  locality — 82.4 vs 79.0 us at 128 MiB operands — so the general kernel keeps them; below, latency counts and the new body wins)
  int64_t tot_elems = 0;
  for (int s2 = 0; s2 < segs->n_seg; ++s2) tot_elems += segs->seg_len[s2];
  const bool pre = vec && e_pre && nk == 1 && ctrl && a.k0_alt == a.k[0] && !big_operand(tot_elems, dtype);
#define LAUNCH_ERR(T, NORM)                                                       \
  do {                                                                            \
    if (pre && a.nt)                                                              \
      XDE_LAUNCH((xde_errnorm_pre_kernel<T, NORM, true>), g, b, st, prof, a);     \
    else if (pre)                                                                 \
      XDE_LAUNCH((xde_errnorm_pre_kernel<T, NORM, false>), g, b, st, prof, a);    \
    else if (vec && nk > 8 && !e_pre)                                             \
      XDE_LAUNCH((xde_errnorm_wide_kernel<T, NORM, true>), g, b, st, prof, a);    \
    else if (vec)                                                                 \
      XDE_LAUNCH((xde_errnorm_kernel<T, NORM, true>), g, b, st, prof, a);    \
    else                                                                          \
      XDE_LAUNCH((xde_errnorm_kernel<T, NORM, false>), g, b, st, prof, a);   \
  } while (0)
  if (dtype == XDE_F32) {
    if (norm_kind == XDE_NORM_RMS) LAUNCH_ERR(float, XDE_NORM_RMS);
    else LAUNCH_ERR(float, XDE_NORM_LINF);
  } else {
    if (norm_kind == XDE_NORM_RMS) LAUNCH_ERR(double, XDE_NORM_RMS);
    else LAUNCH_ERR(double, XDE_NORM_LINF);
  }
#undef LAUNCH_ERR
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_error_ratio(void* out, const void* const* k, const void* k0_alt, const double* c_err, int nk, const void* y0,
                    const void* y0_alt, const void* y1, double rtol, double atol, double dt_host, const xde_ctrl_t* ctrl,
                    int64_t n, int dtype, double* nonfinite_out, void* stream) {
  if (!out || !k || !c_err || !y0 || !y1) return fail(XDE_EBADARG, "xde_error_ratio: null pointer");
  if ((y0_alt == nullptr) != (k0_alt == nullptr)) return fail(XDE_EBADARG, "xde_error_ratio: y0_alt/k0_alt must come together");
  if (y0_alt && !ctrl) return fail(XDE_EBADARG, "xde_error_ratio: operand select needs ctrl");
  if (nk < 1 || nk > XDE_MAX_K) return fail(XDE_EBADARG, "xde_error_ratio: nk out of range");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_error_ratio: bad dtype");
  if (n < 0) return fail(XDE_EBADARG, "xde_error_ratio: negative n");
  if (n == 0) return XDE_OK;
  ErrArgs a;
  memset(&a, 0, sizeof(a));
  a.y0[0] = y0;
  a.y0[1] = y0_alt ? y0_alt : y0;
  a.use_sel = y0_alt ? 1 : 0;
  a.y1 = y1;
  bool vec = aligned16(out) && aligned16(y0) && aligned16(y1) && aligned16(a.y0[1]) && (!k0_alt || aligned16(k0_alt));
  for (int j = 0; j < nk; ++j) {
    if (!k[j]) return fail(XDE_EBADARG, "xde_error_ratio: null k[j]");
    a.k[j] = k[j];
    a.coef[j] = c_err[j];
    vec = vec && aligned16(k[j]);
  }
  a.k0_alt = k0_alt ? k0_alt : k[0];
  a.rtol = rtol;
  a.atol = atol;
  a.dt_host = dt_host;
  a.ctrl = ctrl;
  a.nk = nk;
  const int width = dtype == XDE_F32 ? 4 : 2;
  const int64_t work = vec ? (n + width - 1) / width : n;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > grid_cap()) blocks = grid_cap();
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_ERRNORM, double(nk + 3) * double(n) * (dtype == XDE_F32 ? 4.0 : 8.0));
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
  if (dtype == XDE_F32) {
    if (vec) XDE_LAUNCH((xde_ratio_kernel<float, true>), g, b, st, prof, a, static_cast<float*>(out), n, nonfinite_out);
    else XDE_LAUNCH((xde_ratio_kernel<float, false>), g, b, st, prof, a, static_cast<float*>(out), n, nonfinite_out);
  } else {
    if (vec) XDE_LAUNCH((xde_ratio_kernel<double, true>), g, b, st, prof, a, static_cast<double*>(out), n, nonfinite_out);
    else XDE_LAUNCH((xde_ratio_kernel<double, false>), g, b, st, prof, a, static_cast<double*>(out), n, nonfinite_out);
  }
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_scaled_norm_partial(const void* av, const void* bv, const void* y0, double rtol, double atol,
                            const xde_segments_t* segs, int norm_kind, int dtype, void* ws, int slot, void* stream) {
  if (!av || !y0 || !ws || !segs) return fail(XDE_EBADARG, "xde_scaled_norm_partial: null pointer");
  if (int rc0 = check_segments(segs)) return rc0;
  if (slot < 0 || slot >= kSlots) return fail(XDE_EBADARG, "xde_scaled_norm_partial: bad slot");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_scaled_norm_partial: bad dtype");
  if (norm_kind != XDE_NORM_RMS && norm_kind != XDE_NORM_LINF) return fail(XDE_EBADARG, "xde_scaled_norm_partial: bad norm");
  ScaledArgs s;
  memset(&s, 0, sizeof(s));
  s.a = av;
  s.b = bv;
  s.y0 = y0;
  s.rtol = rtol;
  s.atol = atol;
  s.slot = slot_ptr(ws, slot);
  const int width = dtype == XDE_F32 ? 4 : 2;
  bool vec = aligned16(av) && aligned16(y0) && (!bv || aligned16(bv)) && segs_vec_ok(segs, width);
  int nblocks = 0;
  int rc = build_segmap(segs, vec ? width : 1, vec, &s.map, &nblocks);
  if (rc != XDE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double total = 0;
  for (int i = 0; i < segs->n_seg; ++i) total += double(segs->seg_len[i]);
  ProfScope prof(XDE_KID_SCALEDNORM, (bv ? 3.0 : 2.0) * total * (dtype == XDE_F32 ? 4.0 : 8.0));
  dim3 g(nblocks), b(kBlock);
#define LAUNCH_SC(T, NORM, DIFF)                                                          \
  do {                                                                                    \
    if (vec)                                                                              \
      XDE_LAUNCH((xde_scalednorm_kernel<T, NORM, true, DIFF>), g, b, st, prof, s);   \
    else                                                                                  \
      XDE_LAUNCH((xde_scalednorm_kernel<T, NORM, false, DIFF>), g, b, st, prof, s);  \
  } while (0)
#define LAUNCH_SC2(T, NORM)              \
  do {                                   \
    if (bv) LAUNCH_SC(T, NORM, true);    \
    else LAUNCH_SC(T, NORM, false);      \
  } while (0)
  if (dtype == XDE_F32) {
    if (norm_kind == XDE_NORM_RMS) LAUNCH_SC2(float, XDE_NORM_RMS);
    else LAUNCH_SC2(float, XDE_NORM_LINF);
  } else {
    if (norm_kind == XDE_NORM_RMS) LAUNCH_SC2(double, XDE_NORM_RMS);
    else LAUNCH_SC2(double, XDE_NORM_LINF);
  }
#undef LAUNCH_SC2
#undef LAUNCH_SC
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_scaled_norm2_partial(const void* f0, const void* y0, double rtol, double atol, const xde_segments_t* segs, int norm_kind,
                             int dtype, void* ws, void* stream) {
  if (!f0 || !y0 || !ws || !segs) return fail(XDE_EBADARG, "xde_scaled_norm2_partial: null pointer");
  if (int rc0 = check_segments(segs)) return rc0;
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_scaled_norm2_partial: bad dtype");
  if (norm_kind != XDE_NORM_RMS && norm_kind != XDE_NORM_LINF) return fail(XDE_EBADARG, "xde_scaled_norm2_partial: bad norm");
  ScaledArgs s;
  memset(&s, 0, sizeof(s));
  s.a = f0;
  s.y0 = y0;
  s.rtol = rtol;
  s.atol = atol;
  s.slot = slot_ptr(ws, 0);
  NormSlot* slot_b = slot_ptr(ws, 1);
  const int width = dtype == XDE_F32 ? 4 : 2;
  const bool vec = aligned16(f0) && aligned16(y0) && segs_vec_ok(segs, width);
  int nblocks = 0;
  int rc = build_segmap(segs, vec ? width : 1, vec, &s.map, &nblocks);  // (the grid xde_scaled_norm_partial takes for these segments)
  if (rc != XDE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double total = 0;
  for (int i = 0; i < segs->n_seg; ++i) total += double(segs->seg_len[i]);
  ProfScope prof(XDE_KID_SCALEDNORM, 2.0 * total * (dtype == XDE_F32 ? 4.0 : 8.0));
  dim3 g(nblocks), b(kBlock);
#define LAUNCH_SC2(T, NORM)                                                                \
  do {                                                                                     \
    if (vec)                                                                               \
      XDE_LAUNCH((xde_scalednorm2_kernel<T, NORM, true>), g, b, st, prof, s, slot_b);      \
    else                                                                                   \
      XDE_LAUNCH((xde_scalednorm2_kernel<T, NORM, false>), g, b, st, prof, s, slot_b);     \
  } while (0)
  if (dtype == XDE_F32) {
    if (norm_kind == XDE_NORM_RMS) LAUNCH_SC2(float, XDE_NORM_RMS);
    else LAUNCH_SC2(float, XDE_NORM_LINF);
  } else {
    if (norm_kind == XDE_NORM_RMS) LAUNCH_SC2(double, XDE_NORM_RMS);
    else LAUNCH_SC2(double, XDE_NORM_LINF);
  }
#undef LAUNCH_SC2
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_norm_finalize(const void* ws, int slot, double* sums_out, void* stream) {
  if (!ws || !sums_out) return fail(XDE_EBADARG, "xde_norm_finalize: null pointer");
  if (slot < 0 || slot >= kSlots) return fail(XDE_EBADARG, "xde_norm_finalize: bad slot");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_FINALIZE, 0.0);
  XDE_LAUNCH(xde_finalize_kernel, dim3(1), dim3(kBlock), st, prof, slot_ptr(ws, slot), sums_out);
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_norm_result(const double* sums, const double* seg_count, int n_seg, int norm_kind, int state_dtype,
                    double* result_out, void* stream) {
  if (!sums || !seg_count || !result_out) return fail(XDE_EBADARG, "xde_norm_result: null pointer");
  if (n_seg < 1 || n_seg > XDE_MAX_SEG) return fail(XDE_EBADARG, "xde_norm_result: n_seg out of range");
  SegCounts p;
  memset(&p, 0, sizeof(p));
  for (int s = 0; s < n_seg; ++s) p.c[s] = seg_count[s];
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(xde_norm_result_kernel, dim3(1), dim3(64), 0, st, sums, p, n_seg, norm_kind, state_dtype, result_out);
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}


}  // extern "C"
