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
// K4: lazy dense output; cubic-Hermite history gather of the DDE caller.

#include "xde_common.hpp"

using namespace xde;

namespace {

// ------------------------------------------------------------------------------------------
// K4: dense output (quartic through y0, y_mid, y1, f0, f1), coefficients never materialised
// ------------------------------------------------------------------------------------------
// interp_fit (utils/ode_utils.py:44-49) + interp_evaluate (:69-77), same op order
template <typename T>
__device__ __forceinline__ T quartic_(T y0v, T y1v, T f0v, T f1v, T ymid, T x, T dt) {
  T ca = T(2) * dt * (f1v - f0v) - T(8) * (y1v + y0v) + T(16) * ymid;
  T cb = dt * (T(5) * f0v - T(3) * f1v) + T(18) * y0v + T(14) * y1v - T(32) * ymid;
  T cc = dt * (f1v - T(4) * f0v) - T(11) * y0v - T(5) * y1v + T(16) * ymid;
  T cd = dt * f0v;
  T total = y0v + x * cd;
  T xp = x;
  xp = xp * x;
  total = total + xp * cc;
  xp = xp * x;
  total = total + xp * cb;
  xp = xp * x;
  total = total + xp * ca;
  return total;
}

// One element range of the launch.  NK > 0: the operand count is a compile-time constant — every stream of a vector
// (k_0..k_{NK-1}, y0, y1, f1) is requested BEFORE the first one is used (round 2's runtime loop had one load in flight
// per lane and reached 0.53 of the HBM peak: 71 us for 9 N 4 B at config 2).  NK == 0: the generic loop (Dopri8's 13 operands).
template <typename T, typename TT, int NK, bool VEC, bool COMMIT>
__device__ __forceinline__ void dense_body(const DenseArgs& a, const T* y0, const T* k0, int ob, int oe, T dt, TT t0, TT t1) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  constexpr int MAXK = NK > 0 ? NK : 1;
  const T* __restrict__ y1 = static_cast<const T*>(a.y1);
  const T* __restrict__ f1p = static_cast<const T*>(a.f1);
  T* out = static_cast<T*>(a.out_base);
  T* cy = static_cast<T*>(a.commit_y0);
  T* cf = static_cast<T*>(a.commit_f0);
  const int nk = NK > 0 ? NK : a.nk;
  const bool dead_nt = !COMMIT && (a.nt & 1);  // y0 and the k_j die with this step (the COMMIT variant overwrites y0 / f0 in place)
  const bool rows_nt = (a.nt & 2) != 0;
  const bool f1_last = a.f1_is_last_k != 0;
  const int64_t nvec = a.n / W;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  const T* kp[MAXK];
  T cm[MAXK];
  if (NK > 0) {
    kp[0] = k0;
#pragma unroll
    for (int j = 1; j < MAXK; ++j) kp[j] = static_cast<const T*>(a.k[j]);
#pragma unroll
    for (int j = 0; j < MAXK; ++j) cm[j] = dt * T(a.mid[j]);  // `dt * self.mid`
  }
  // the output times of this step, as fractions of it (at most a handful; recomputed per vector when there are many)
  constexpr int kXs = 4;
  T xs[kXs];
#pragma unroll
  for (int r = 0; r < kXs; ++r) xs[r] = (ob + r < oe) ? T((TT(a.t_span[ob + r]) - t0) / (t1 - t0)) : T(0);

  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nvec; i += stride) {
    if (COMMIT && oe <= ob) {  // the common replay: no output time inside this step, only the state hand-over
      P y1v = P::load(y1, i);
      P f1v = P::load(f1p, i);
      y1v.store(cy, i);
      f1v.store(cf, i);
      continue;
    }
    P y1v, f1v, y0v, f0v, acc;
    if (NK > 0) {
      P kk[MAXK];
#pragma unroll
      for (int j = MAXK - 1; j >= 1; --j) kk[j] = load_sel<P>(kp[j], i, dead_nt && !(f1_last && j == MAXK - 1));
      y1v = P::load(y1, i);
      if (!(f1_last && MAXK > 1)) f1v = P::load(f1p, i);
      y0v = load_sel<P>(y0, i, dead_nt);
      kk[0] = load_sel<P>(kp[0], i, dead_nt);
      if (f1_last && MAXK > 1) f1v = kk[MAXK - 1];
      f0v = kk[0];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        T s_ = kk[0].v[w] * cm[0];
#pragma unroll
        for (int j = 1; j < MAXK; ++j) s_ = s_ + kk[j].v[w] * cm[j];
        acc.v[w] = s_;
      }
    } else {
      y1v = P::load(y1, i);
      f1v = P::load(f1p, i);
      y0v = P::load(y0, i);
      f0v = P::load(k0, i);
      for (int j = 0; j < nk; ++j) {
        const T* kj = j == 0 ? k0 : static_cast<const T*>(a.k[j]);
        P kk = (j == 0) ? f0v : P::load(kj, i);
        T cj = dt * T(a.mid[j]);
#pragma unroll
        for (int w = 0; w < W; ++w) acc.v[w] = (j == 0) ? kk.v[w] * cj : acc.v[w] + kk.v[w] * cj;
      }
    }
    for (int r = ob; r < oe; ++r) {
      const T x = (r - ob < kXs) ? xs[r - ob] : T((TT(a.t_span[r]) - t0) / (t1 - t0));
      P o;
#pragma unroll
      for (int w = 0; w < W; ++w) o.v[w] = quartic_<T>(y0v.v[w], y1v.v[w], f0v.v[w], f1v.v[w], y0v.v[w] + acc.v[w], x, dt);
      if (rows_nt)
        o.store_nt(out + int64_t(r) * a.n, i);
      else
        o.store(out + int64_t(r) * a.n, i);
    }
    if (COMMIT) {
      y1v.store(cy, i);
      f1v.store(cf, i);
    }
  }
  if (VEC) {
    const int64_t i = nvec * W + threadIdx.x;
    if (blockIdx.x == 0 && i < a.n) {
      T acc = T(0);
      for (int j = 0; j < nk; ++j) {
        const T* kj = j == 0 ? k0 : static_cast<const T*>(a.k[j]);
        T term = kj[i] * (dt * T(a.mid[j]));
        acc = (j == 0) ? term : acc + term;
      }
      const T y0s = y0[i], y1s = y1[i], f0s = k0[i], f1s = f1p[i];
      for (int r = ob; r < oe; ++r) {
        TT xt = (TT(a.t_span[r]) - t0) / (t1 - t0);
        out[int64_t(r) * a.n + i] = quartic_<T>(y0s, y1s, f0s, f1s, y0s + acc, T(xt), dt);
      }
      if (COMMIT) {
        cy[i] = y1s;
        cf[i] = f1s;
      }
    }
  }
}

template <typename T, typename TT, bool VEC, bool COMMIT = false>
__global__ __launch_bounds__(kBlock) void xde_dense_kernel(DenseArgs a) {
  const xde_ctrl_t* c = a.ctrl;
  if (!c->accept) return;  // nothing to emit, nothing to commit
  const bool expected = !(a.expect_step >= 0 && c->n_steps != a.expect_step);
  const int ob = c->out_begin, oe = expected ? c->out_end : c->out_begin;
  if (!COMMIT && oe <= ob) return;
  const int sel = a.use_sel ? (c->sel_used ? 1 : 0) : 0;
  // (no __restrict__ on the operands the COMMIT variant also writes)
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  const T dt = T(TT(c->dt_last));  // `dt.astype(y0.dtype)`
  const TT t0 = TT(c->t0), t1 = TT(c->t1);
  switch (a.nk) {
    case 1: dense_body<T, TT, 1, VEC, COMMIT>(a, y0, k0, ob, oe, dt, t0, t1); break;
    case 2: dense_body<T, TT, 2, VEC, COMMIT>(a, y0, k0, ob, oe, dt, t0, t1); break;
    case 3: dense_body<T, TT, 3, VEC, COMMIT>(a, y0, k0, ob, oe, dt, t0, t1); break;
    case 4: dense_body<T, TT, 4, VEC, COMMIT>(a, y0, k0, ob, oe, dt, t0, t1); break;
    case 5: dense_body<T, TT, 5, VEC, COMMIT>(a, y0, k0, ob, oe, dt, t0, t1); break;
    case 6: dense_body<T, TT, 6, VEC, COMMIT>(a, y0, k0, ob, oe, dt, t0, t1); break;
    case 7: dense_body<T, TT, 7, VEC, COMMIT>(a, y0, k0, ob, oe, dt, t0, t1); break;
    default: dense_body<T, TT, 0, VEC, COMMIT>(a, y0, k0, ob, oe, dt, t0, t1); break;
  }
}


// ------------------------------------------------------------------------------------------
// DDE history gather: cubic-Hermite value and derivative of a [outer, T, D] series at L query times
// (interpolation/interpolate_base.py:50-107 + interpolation/interpolate.py:100-204), one pass
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void xde_hermite_kernel(T* __restrict__ val, T* __restrict__ der,
                                                             const T* __restrict__ his, const T* __restrict__ ts,
                                                             const T* __restrict__ lags, int64_t outer, int Tn, int D, int L) {
  const int64_t total = outer * int64_t(L) * D;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x; e < total; e += stride) {
    const int d = int(e % D);
    const int l = int((e / D) % L);
    const int64_t o = e / (int64_t(D) * L);
    const T tau = lags[l];
    // index = clip(bucketize(tau, t) - 1, 0, T-1); bucketize (right=False) = #{ t_i < tau }
    int lo = 0, hi = Tn;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (ts[mid] < tau) lo = mid + 1; else hi = mid;
    }
    int i = lo - 1;
    i = i < 0 ? 0 : (i > Tn - 1 ? Tn - 1 : i);
    auto h_at = [&](int j) -> T {  // scale1[j] = t[j+1]-t[j], last one repeated
      int jj = j < Tn - 1 ? j : Tn - 2;
      return ts[jj + 1] - ts[jj];
    };
    const T h1 = h_at(i);                 // scale1[i]
    const T h2 = i == 0 ? h_at(0) : h_at(i - 1);  // scale2[i] = concat(scale[:1], scale1[:-1])[i]
    const T s = (tau - ts[i]) / h1;
    const T* row = his + o * int64_t(Tn) * D + d;
    auto ser = [&](int j) -> T { return row[int64_t(j < Tn ? j : Tn - 1) * D]; };
    auto drv = [&](int j) -> T {  // derivs has T+1 entries: finite differences, the last two repeat the last one
      int jj = j < Tn - 1 ? j : Tn - 2;
      return (ser(jj + 1) - ser(jj)) / h_at(jj);
    };
    const T p0 = ser(i) / h1, p1 = ser(i + 1) / h2, d0 = drv(i), d1 = drv(i + 1);
    // [s^3, s^2, s, 1] @ H  and  [3s^2, 2s, 1, 0] @ H  with H = [[2,-2,1,1],[-3,3,-2,-1],[0,0,1,0],[1,0,0,0]]
    const T s2 = s * s, s3 = s2 * s;
    const T c0 = T(2) * s3 - T(3) * s2 + T(1), c1 = T(-2) * s3 + T(3) * s2, c2 = s3 - T(2) * s2 + s, c3 = s3 - s2;
    const T g0 = T(6) * s2 - T(6) * s, g1 = T(-6) * s2 + T(6) * s, g2 = T(3) * s2 - T(4) * s + T(1), g3 = T(3) * s2 - T(2) * s;
    val[e] = (((c0 * p0 + c1 * p1) + c2 * d0) + c3 * d1) * h1;  // evaluate(): result *= scale
    der[e] = ((g0 * p0 + g1 * p1) + g2 * d0) + g3 * d1;         // derivative(): no scale factor
  }
}

// The same arithmetic with everything that depends only on the lag hoisted out of the element loop: each workgroup
// builds the per-lag table (interval index, Hermite basis values, the interval widths, which rows feed the two node
// derivatives) in LDS once, then streams 16-byte vectors along D: three row loads, two stores per output vector.
constexpr int kHermiteMaxL = 128;
template <typename T>
struct HermiteLag {
  int i, mode, zrow, pad;  // mode 0: interior; 1: i == T-2 (d1 repeats d0's rows); 2: i == T-1 (both use rows T-2, T-1)
  T h1, h2, ha, hb, c0, c1, c2, c3, g0, g1, g2, g3;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void xde_hermite_vec_kernel(T* __restrict__ val, T* __restrict__ der,
                                                                 const T* __restrict__ his, const T* __restrict__ ts,
                                                                 const T* __restrict__ lags, int64_t outer, int Tn, int D, int L) {
  using P = Pack<T, true>;
  constexpr int W = P::W;
  __shared__ HermiteLag<T> tab[kHermiteMaxL];
  for (int l = threadIdx.x; l < L; l += kBlock) {
    const T tau = lags[l];
    int lo = 0, hi = Tn;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (ts[mid] < tau) lo = mid + 1; else hi = mid;
    }
    int i = lo - 1;
    i = i < 0 ? 0 : (i > Tn - 1 ? Tn - 1 : i);
    auto h_at = [&](int j) -> T {
      int jj = j < Tn - 1 ? j : Tn - 2;
      return ts[jj + 1] - ts[jj];
    };
    HermiteLag<T> r;
    r.i = i;
    r.mode = i <= Tn - 3 ? 0 : (i == Tn - 2 ? 1 : 2);
    r.zrow = r.mode == 0 ? i + 2 : Tn - 2;
    r.pad = 0;
    r.h1 = h_at(i);
    r.h2 = i == 0 ? h_at(0) : h_at(i - 1);
    r.ha = h_at(i);
    r.hb = h_at(i + 1);
    const T sx = (tau - ts[i]) / r.h1;
    const T s2 = sx * sx, s3 = s2 * sx;
    r.c0 = T(2) * s3 - T(3) * s2 + T(1);
    r.c1 = T(-2) * s3 + T(3) * s2;
    r.c2 = s3 - T(2) * s2 + sx;
    r.c3 = s3 - s2;
    r.g0 = T(6) * s2 - T(6) * sx;
    r.g1 = T(-6) * s2 + T(6) * sx;
    r.g2 = T(3) * s2 - T(4) * sx + T(1);
    r.g3 = T(3) * s2 - T(2) * sx;
    tab[l] = r;
  }
  __syncthreads();
  const int DV = D / W;
  const int64_t total = outer * int64_t(L) * DV;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  const int64_t rowv = int64_t(Tn) * DV;  // vectors per outer slice of his
  for (int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x; e < total; e += stride) {
    int dv, l;
    int64_t o;
    if (total < (int64_t(1) << 31)) {  // 32-bit index arithmetic whenever it fits
      const unsigned eu = unsigned(e), q = eu / unsigned(DV);
      dv = int(eu - q * unsigned(DV));
      const unsigned q2 = q / unsigned(L);
      l = int(q - q2 * unsigned(L));
      o = q2;
    } else {
      dv = int(e % DV);
      l = int((e / DV) % L);
      o = e / (int64_t(DV) * L);
    }
    const HermiteLag<T>& r = tab[l];
    const int64_t base = o * rowv + dv;
    const int i1 = r.i + 1 < Tn ? r.i + 1 : Tn - 1;
    const P X = P::load(his, base + int64_t(r.i) * DV);
    const P Y = P::load(his, base + int64_t(i1) * DV);
    const P Z = P::load(his, base + int64_t(r.zrow) * DV);
    P v, g;
#pragma unroll
    for (int x = 0; x < W; ++x) {
      const T p0 = X.v[x] / r.h1, p1 = Y.v[x] / r.h2;
      const T n0 = r.mode == 2 ? X.v[x] - Z.v[x] : Y.v[x] - X.v[x];
      const T n1 = r.mode == 0 ? Z.v[x] - Y.v[x] : n0;
      const T d0 = n0 / r.ha, d1 = n1 / r.hb;
      v.v[x] = (((r.c0 * p0 + r.c1 * p1) + r.c2 * d0) + r.c3 * d1) * r.h1;
      g.v[x] = ((r.g0 * p0 + r.g1 * p1) + r.g2 * d0) + r.g3 * d1;
    }
    v.store(val, e);
    g.store(der, e);
  }
}


}  // namespace

extern "C" {

}  // extern "C"

static int dense_launch(const char* who_c, void* out_base, const void* const* k, const void* k0_alt, const double* mid, int nk,
                        const void* y0, const void* y0_alt, const void* y1, const void* f1, const xde_ctrl_t* ctrl,
                        const double* t_span_dev, int time_dtype, int64_t n, int dtype, int64_t expect_step, void* commit_y0,
                        void* commit_f0, void* stream) {
  const std::string who(who_c);
  if (!out_base || !k || !mid || !y0 || !y1 || !f1 || !ctrl || !t_span_dev) return fail(XDE_EBADARG, who + ": null pointer");
  if (nk < 1 || nk > XDE_MAX_K) return fail(XDE_EBADARG, who + ": nk out of range");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, who + ": bad dtype");
  if (time_dtype != XDE_F32 && time_dtype != XDE_F64) return fail(XDE_EBADARG, who + ": bad time_dtype");
  if ((y0_alt == nullptr) != (k0_alt == nullptr)) return fail(XDE_EBADARG, who + ": y0_alt/k0_alt must come together");
  if ((commit_y0 == nullptr) != (commit_f0 == nullptr)) return fail(XDE_EBADARG, who + ": commit destinations must come together");
  if (commit_y0 && y0_alt) return fail(XDE_EBADARG, who + ": the fused commit serves the graph pipeline (no operand select)");
  if (n < 0) return fail(XDE_EBADARG, who + ": negative n");
  if (n == 0) return XDE_OK;
  DenseArgs a;
  memset(&a, 0, sizeof(a));
  a.commit_y0 = commit_y0;
  a.commit_f0 = commit_f0;
  a.out_base = out_base;
  a.y0[0] = y0;
  a.y0[1] = y0_alt ? y0_alt : y0;
  a.k0_alt = k0_alt ? k0_alt : k[0];
  a.use_sel = y0_alt ? 1 : 0;
  a.y1 = y1;
  a.f1 = f1;
  a.ctrl = ctrl;
  a.t_span = t_span_dev;
  a.n = n;
  a.nk = nk;
  a.time_dtype = time_dtype;
  a.expect_step = expect_step;
  a.f1_is_last_k = (f1 == k[nk - 1] && nk > 1 && !commit_y0) ? 1 : 0;
  // cache policy (results never depend on it): the step's y0 and k_j are read here for the last time; y1 / f1 become the next
  // step's (y0, f0) and stay cacheable; solution rows are not re-read by the solve
  a.nt = (nt_policy() & 1) ? 3 : 0;
  const int width = dtype == XDE_F32 ? 4 : 2;
  // every output row starts at out_base + r*n elements: rows stay 16-byte aligned only if n % width == 0
  bool vec = aligned16(out_base) && (n % width == 0) && aligned16(y0) && aligned16(a.y0[1]) && aligned16(y1) &&
             aligned16(f1) && aligned16(a.k0_alt);
  if (commit_y0) vec = vec && aligned16(commit_y0) && aligned16(commit_f0);
  for (int j = 0; j < nk; ++j) {
    if (!k[j]) return fail(XDE_EBADARG, who + ": null k[j]");
    a.k[j] = k[j];
    a.mid[j] = mid[j];
    vec = vec && aligned16(k[j]);
  }
  const int64_t work = vec ? (n + width - 1) / width : n;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  // predicated launch: most launches of the speculative pipeline exit at once, so keep the grid small (a no-op
  // launch costs ~5 us with 2048 workgroups); the streaming rate does not depend on the grid between 512 and 4096
  constexpr int dense_grid = 512;  // (512 / 1024 / 2048 workgroups stream at the same rate, profiles/r03_dense.json; a predicated-off launch must stay cheap)
  const int64_t dense_cap = grid_cap() < dense_grid ? grid_cap() : dense_grid;
  if (blocks > dense_cap) blocks = dense_cap;
  if (blocks < 1) blocks = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_DENSE, double(nk + 4) * double(n) * (dtype == XDE_F32 ? 4.0 : 8.0));
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
#define LAUNCH_DENSE(T, TT)                                                              \
  do {                                                                                   \
    if (commit_y0) {                                                                     \
      if (vec)                                                                           \
        XDE_LAUNCH((xde_dense_kernel<T, TT, true, true>), g, b, st, prof, a);            \
      else                                                                               \
        XDE_LAUNCH((xde_dense_kernel<T, TT, false, true>), g, b, st, prof, a);           \
    } else if (vec)                                                                      \
      XDE_LAUNCH((xde_dense_kernel<T, TT, true>), g, b, st, prof, a);                    \
    else                                                                                 \
      XDE_LAUNCH((xde_dense_kernel<T, TT, false>), g, b, st, prof, a);                   \
  } while (0)
  if (dtype == XDE_F32) {
    if (time_dtype == XDE_F32) LAUNCH_DENSE(float, float);
    else LAUNCH_DENSE(float, double);
  } else {
    if (time_dtype == XDE_F32) LAUNCH_DENSE(double, float);
    else LAUNCH_DENSE(double, double);
  }
#undef LAUNCH_DENSE
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

extern "C" {

int xde_dense_eval(void* out_base, const void* const* k, const void* k0_alt, const double* mid, int nk, const void* y0,
                   const void* y0_alt, const void* y1, const void* f1, const xde_ctrl_t* ctrl, const double* t_span_dev,
                   int time_dtype, int64_t n, int dtype, int64_t expect_step, void* stream) {
  return dense_launch("xde_dense_eval", out_base, k, k0_alt, mid, nk, y0, y0_alt, y1, f1, ctrl, t_span_dev, time_dtype, n, dtype,
                      expect_step, nullptr, nullptr, stream);
}

int xde_dense_commit(void* out_base, const void* const* k, const double* mid, int nk, void* y0, const void* y1, void* f0,
                     const void* f1, const xde_ctrl_t* ctrl, const double* t_span_dev, int time_dtype, int64_t n, int dtype,
                     void* stream) {
  if (!k || !f0 || k[0] != f0) return fail(XDE_EBADARG, "xde_dense_commit: k[0] must be f0 (the derivative at the step's start)");
  return dense_launch("xde_dense_commit", out_base, k, nullptr, mid, nk, y0, nullptr, y1, f1, ctrl, t_span_dev, time_dtype, n, dtype,
                      -1, y0, f0, stream);
}

int xde_hermite_gather(void* val_out, void* der_out, const void* his, const void* his_t, const void* lags, int64_t outer,
                       int T, int D, int L, int dtype, void* stream) {
  if (!val_out || !der_out || !his || !his_t || !lags) return fail(XDE_EBADARG, "xde_hermite_gather: null pointer");
  if (outer < 0 || T < 2 || D < 1 || L < 0) return fail(XDE_EBADARG, "xde_hermite_gather: bad sizes (need T >= 2)");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_hermite_gather: bad dtype");
  const int64_t total = outer * int64_t(L) * D;
  if (total == 0) return XDE_OK;
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > grid_cap()) blocks = grid_cap();
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double esz = dtype == XDE_F32 ? 4.0 : 8.0;
  ProfScope prof(XDE_KID_DENSE, 5.0 * double(total) * esz);  // rows i, i+1, i+2 in; value and derivative out
  const int width = dtype == XDE_F32 ? 4 : 2;
  if (D % width == 0 && L <= kHermiteMaxL && aligned16(val_out) && aligned16(der_out) && aligned16(his)) {
    int64_t vblocks = (total / width + kBlock - 1) / kBlock;
    if (vblocks > grid_cap()) vblocks = grid_cap();
    dim3 gv(static_cast<unsigned>(vblocks)), bv(kBlock);
    if (dtype == XDE_F32)
      XDE_LAUNCH(xde_hermite_vec_kernel<float>, gv, bv, st, prof, static_cast<float*>(val_out), static_cast<float*>(der_out),
                 static_cast<const float*>(his), static_cast<const float*>(his_t), static_cast<const float*>(lags), outer, T, D, L);
    else
      XDE_LAUNCH(xde_hermite_vec_kernel<double>, gv, bv, st, prof, static_cast<double*>(val_out), static_cast<double*>(der_out),
                 static_cast<const double*>(his), static_cast<const double*>(his_t), static_cast<const double*>(lags), outer, T, D, L);
    HIP_TRY(hipGetLastError());
    return XDE_OK;
  }
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
  if (dtype == XDE_F32)
    XDE_LAUNCH(xde_hermite_kernel<float>, g, b, st, prof, static_cast<float*>(val_out), static_cast<float*>(der_out),
               static_cast<const float*>(his), static_cast<const float*>(his_t), static_cast<const float*>(lags), outer, T, D, L);
  else
    XDE_LAUNCH(xde_hermite_kernel<double>, g, b, st, prof, static_cast<double*>(val_out), static_cast<double*>(der_out),
               static_cast<const double*>(his), static_cast<const double*>(his_t), static_cast<const double*>(lags), outer, T, D, L);
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}


}  // extern "C"
