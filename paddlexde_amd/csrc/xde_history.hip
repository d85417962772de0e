// libxde_hip.so — the delay-equation caller's history spline beyond the cubic-Hermite default, and HistoryIndex.backward.
//
//   xde_history_gather   HistoryIndex.forward for interp_method "linear" | "cubic" | "bez" (xde/base_dde.py:104-121 over
//                        interpolation/interpolate.py:6-99 / :100-204 / :207-298 on interpolate_base.py:50-107): value AND time
//                        derivative of the spline of `his[outer, T, D]` at the L lags in one pass.  "cubic" is xde_hermite_gather.
//   xde_lag_grad         HistoryIndex.backward (:123-127): grad_lags[l] = sum over every axis but the lag axis of
//                        grad_y * derivative — ONE launch (it was a framework multiply + a framework sum over the D3STN-sized
//                        [9824, 12, 64] tensor).
//
// Linear and Bezier splines are "weighted rows": with i = clip(bucketize(t) - 1, 0, T-1) and s = (t - t_i) / scale1_i,
//     value = (sum_k w_k(s) * his[min(i+k, T-1)] / scale{k+1}_i) * scale1_i,      derivative = sum_k w'_k(s) * his[...] / scale{k+1}_i
// with M = 2 rows, w = [1-s, s] (H = [[-1,1],[1,0]]) for "linear" and M = 4 rows, the Bernstein weights (H of :237-243) for "bez";
// scale1 = the knot spacing t[j+1]-t[j] ("linear") or t[j+3]-t[j] ("bez"), its last value repeated, and scale{k+1} = scale1 shifted
// right by k with its first value repeated (the reference's `_make_series`, conventions kept as written — on a uniform grid every
// scale{k} is the same number).  Bandwidth-bound: M row loads and two stores per output vector; the per-lag quantities are built
// once per workgroup in LDS.

#include "xde_common.hpp"

using namespace xde;

namespace {

constexpr int kHistMaxL = 128;

template <typename T>
struct RowsLag {
  int row[4];      // his rows min(i + k, T - 1)
  T wv[4], wd[4];  // weight of row k (of `his[row] / scale{k+1}`: the reference divides the SERIES, then weights it) in value / derivative
  T h1;            // scale1_i
};

// the per-lag table entry (one thread per lag)
template <typename T, int M>
__device__ __forceinline__ RowsLag<T> make_lag(const T* __restrict__ ts, T tau, int Tn) {
  constexpr int SPAN = M == 2 ? 1 : 3;  // scale = t[j + SPAN] - t[j], j < T - SPAN
  int lo = 0, hi = Tn;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (ts[mid] < tau) lo = mid + 1; else hi = mid;
  }
  int i = lo - 1;  // bucketize(tau, t) - 1 with right=False: #{t_j < tau} - 1
  i = i < 0 ? 0 : (i > Tn - 1 ? Tn - 1 : i);
  auto scale1 = [&](int j) -> T {  // concat(scale, scale[-1:] * SPAN)[j]
    int jj = j < Tn - SPAN ? j : Tn - SPAN - 1;
    return ts[jj + SPAN] - ts[jj];
  };
  RowsLag<T> r;
  r.h1 = scale1(i);
  const T s = (tau - ts[i]) / r.h1;
  T w[4] = {T(0), T(0), T(0), T(0)}, g[4] = {T(0), T(0), T(0), T(0)};
  if (M == 2) {
    // [s, 1] @ [[-1, 1], [1, 0]] = [-s + 1, s];  [1, 0] @ H = [-1, 1]
    w[0] = -s + T(1);
    w[1] = s;
    g[0] = T(-1);
    g[1] = T(1);
  } else {
    // [s^3, s^2, s, 1] @ H,  [3 s^2, 2 s, 1, 0] @ H  with H = [[-1,3,-3,1],[3,-6,3,0],[-3,3,0,0],[1,0,0,0]]
    const T s2 = s * s, s3 = s2 * s;
    w[0] = ((-s3 + T(3) * s2) - T(3) * s) + T(1);
    w[1] = (T(3) * s3 - T(6) * s2) + T(3) * s;
    w[2] = T(-3) * s3 + T(3) * s2;
    w[3] = s3;
    const T a = T(3) * s2, b = T(2) * s;
    g[0] = (-a + T(3) * b) - T(3);
    g[1] = (T(3) * a - T(6) * b) + T(3);
    g[2] = T(-3) * a + T(3) * b;
    g[3] = a;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r.row[k] = i + k < Tn ? i + k : Tn - 1;
    r.wv[k] = w[k];
    r.wd[k] = g[k];
  }
  return r;
}

template <typename T>
struct RowsScale {
  T sc[4];
};

template <typename T, int M>
__device__ __forceinline__ RowsScale<T> make_scales(const T* __restrict__ ts, int i, int Tn) {
  constexpr int SPAN = M == 2 ? 1 : 3;
  auto scale1 = [&](int j) -> T {
    int jj = j < Tn - SPAN ? j : Tn - SPAN - 1;
    return ts[jj + SPAN] - ts[jj];
  };
  RowsScale<T> q;
#pragma unroll
  for (int k = 0; k < 4; ++k) q.sc[k] = scale1(i - k < 0 ? 0 : i - k);
  return q;
}

template <typename T, int M, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_rows_gather_kernel(T* __restrict__ val, T* __restrict__ der, const T* __restrict__ his,
                                                                 const T* __restrict__ ts, const T* __restrict__ lags, int64_t outer,
                                                                 int Tn, int D, int L) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  __shared__ RowsLag<T> tab[kHistMaxL];
  __shared__ RowsScale<T> tsc[kHistMaxL];
  for (int l = threadIdx.x; l < L; l += kBlock) {
    RowsLag<T> r = make_lag<T, M>(ts, lags[l], Tn);
    tab[l] = r;
    tsc[l] = make_scales<T, M>(ts, r.row[0], Tn);  // (row[0] == i)
  }
  __syncthreads();
  const int DV = D / W;
  const int64_t total = outer * int64_t(L) * DV;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  const int64_t rowv = int64_t(Tn) * DV;
  for (int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x; e < total; e += stride) {
    const int dv = int(e % DV);
    const int l = int((e / DV) % L);
    const int64_t o = e / (int64_t(DV) * L);
    const RowsLag<T>& r = tab[l];
    const RowsScale<T>& q = tsc[l];
    const int64_t base = o * rowv + dv;
    P X[M];
#pragma unroll
    for (int k = 0; k < M; ++k) X[k] = P::load(his, base + int64_t(r.row[k]) * DV);
    P v, g;
#pragma unroll
    for (int x = 0; x < W; ++x) {
      T av = r.wv[0] * (X[0].v[x] / q.sc[0]);
      T ad = r.wd[0] * (X[0].v[x] / q.sc[0]);
#pragma unroll
      for (int k = 1; k < M; ++k) {
        const T p = X[k].v[x] / q.sc[k];
        av = av + r.wv[k] * p;
        ad = ad + r.wd[k] * p;
      }
      v.v[x] = av * r.h1;  // evaluate(): result *= scale
      g.v[x] = ad;         // derivative(): no scale factor
    }
    v.store_nt(val, e);
    g.store_nt(der, e);
  }
}

// ------------------------------------------------------------------------------------------
// HistoryIndex.backward: grad_lags[l] = sum_{o, d} grad_y[o, l, d] * der[o, l, d]
// ------------------------------------------------------------------------------------------
// Products are formed in T (the reference's `grad_y * derivative_lags`) and accumulated in fp64; every workgroup leaves one partial
// per lag it worked on, the last workgroup to arrive (agent-scope tickets, two levels) adds the partials of every lag in a fixed
// order (lane-strided, then a shuffle tree): bit-reproducible, one launch.  Two mappings of the [outer, L, D] tensors onto lanes:
//
//   PLANE  (one [L, D] plane fits a workgroup: L * D/W <= 256 lanes — D3STN's 12 x 64): the tensors are walked as ONE contiguous
//          stream of 16-byte vectors; a workgroup's pass covers R whole planes, so a lane keeps the same (lag, column) in every pass
//          and needs one accumulator.  Contiguous 3 KB bursts instead of 256-byte rows 3 KB apart, kPlaneUnroll passes' loads in
//          flight per lane (guarded: the tail is not a serial loop), nb arrivals instead of L * nb.
//   ROWS   (larger planes): workgroup (l, b) walks the rows o = b, b + nb, ... of lag l; rows longer than a workgroup's reach
//          (D/W > 256, or an unaligned D > 256) are walked in strides of the row lanes — any D is served.
constexpr int kLagShards = 16;  // first-level arrival counters (one atomic word serialises its arrivals at ~12 ns each)
constexpr int kPlaneUnroll = 4;
struct LagGradWs {
  unsigned ticket;
  unsigned pad[31];
  struct alignas(128) Shard {
    unsigned count;
    unsigned pad[31];
  } shard[kLagShards];
  double partial[1];  // [L][nb]
};

__device__ __forceinline__ void lag_store_partial(LagGradWs* ws, int64_t at, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(&ws->partial[at]), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial has left this CU before the ticket is taken
}

// Arrival of this workgroup (all its partials stored and waited for, barrier passed); the LAST one of the grid finishes every lag.
template <typename T>
__device__ __forceinline__ void lag_finish(T* __restrict__ out, int L, int nb, LagGradWs* ws, int* s_last) {
  if (threadIdx.x == 0) {
    // two-level arrival: workgroups b, b + 16, ... share a counter; the last arriver of a shard arrives at the top-level one
    const unsigned nsh = gridDim.x < unsigned(kLagShards) ? gridDim.x : unsigned(kLagShards);
    const unsigned sh = blockIdx.x % nsh;
    const unsigned members = (gridDim.x - sh + nsh - 1) / nsh;
    int last = 0;
    if (__hip_atomic_fetch_add(&ws->shard[sh].count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1)
      last = __hip_atomic_fetch_add(&ws->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsh - 1 ? 1 : 0;
    *s_last = last;
  }
  __syncthreads();
  if (!*s_last) return;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // wave w finishes lags w, w + 4, ...: its lanes pull the lag's nb partials lane-strided, EIGHT loads requested before the first
  // is added (a dependent chain of agent-scope loads costs a memory round trip per link: 16 links at nb = 1024 were 24 us of a
  // 24 us kernel), then a shuffle tree — a fixed order.
  for (int j = threadIdx.x >> 6; j < L; j += kWaves) {
    double v = 0.0;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&ws->partial[int64_t(j) * nb]);
    for (int q0 = threadIdx.x & 63; q0 < nb; q0 += 64 * 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + 64 * u;
        t[u] = q < nb ? __longlong_as_double((long long)__hip_atomic_load(src + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) out[j] = T(v);
  }
  // re-armed for the next launch
  if (threadIdx.x < kLagShards) __hip_atomic_store(&ws->shard[threadIdx.x].count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) __hip_atomic_store(&ws->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_lag_grad_plane_kernel(T* __restrict__ out, const T* __restrict__ gy, const T* __restrict__ der,
                                                                    int64_t outer, int D, int L, int R, int nb, LagGradWs* ws) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  __shared__ double s_acc[kBlock];
  __shared__ int s_last;
  const int DV = D / W;
  const int plane = L * DV;          // vectors of one [L, D] plane
  const int active = R * plane;      // lanes at work: R whole planes per pass (<= kBlock)
  const int64_t total = outer * int64_t(plane);
  const int64_t step = int64_t(nb) * active;
  double acc = 0.0;
  if (int(threadIdx.x) < active) {
    // every pass's loads are requested before any product is formed; a pass beyond the end loads nothing (wave-uniform up to the
    // very last vectors) — the tail costs no extra round trip
    for (int64_t v0 = int64_t(blockIdx.x) * active + threadIdx.x; v0 < total; v0 += kPlaneUnroll * step) {
      P a[kPlaneUnroll], c[kPlaneUnroll];
      bool on[kPlaneUnroll];
#pragma unroll
      for (int u = 0; u < kPlaneUnroll; ++u) {
        const int64_t v = v0 + u * step;
        on[u] = v < total;
        if (on[u]) {
          a[u] = P::load_nt(gy, v);
          c[u] = P::load_nt(der, v);
        }
      }
#pragma unroll
      for (int u = 0; u < kPlaneUnroll; ++u)
        if (on[u]) {
#pragma unroll
          for (int x = 0; x < W; ++x) acc += double(a[u].v[x] * c[u].v[x]);
        }
    }
  }
  s_acc[threadIdx.x] = acc;
  __syncthreads();
  if (int(threadIdx.x) < L) {  // lag l: its DV columns of each of the R planes, in a fixed order
    double v = 0.0;
    for (int r = 0; r < R; ++r)
      for (int d = 0; d < DV; ++d) v += s_acc[r * plane + int(threadIdx.x) * DV + d];
    lag_store_partial(ws, int64_t(threadIdx.x) * nb + blockIdx.x, v);
  }
  __syncthreads();
  lag_finish<T>(out, L, nb, ws, &s_last);
}

template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_lag_grad_rows_kernel(T* __restrict__ out, const T* __restrict__ gy, const T* __restrict__ der,
                                                                   int64_t outer, int D, int L, int nb, LagGradWs* ws) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  __shared__ double s_w[kWaves];
  __shared__ int s_last;
  const int l = blockIdx.x / nb, b = blockIdx.x % nb;
  const int DV = D / W;
  const int row_lanes = DV < kBlock ? DV : kBlock;  // lanes that share one row; a longer row is walked in strides of them
  const int rows_per_pass = kBlock / row_lanes;
  const int my_row = threadIdx.x / row_lanes, dv0 = threadIdx.x % row_lanes;
  double acc = 0.0;
  if (my_row < rows_per_pass) {
    const int64_t step = int64_t(nb) * rows_per_pass;
    for (int64_t o = int64_t(b) * rows_per_pass + my_row; o < outer; o += step) {
      const int64_t row = (o * L + l) * DV;
      for (int dv = dv0; dv < DV; dv += row_lanes) {
        const P a = P::load_nt(gy, row + dv);
        const P c = P::load_nt(der, row + dv);
#pragma unroll
        for (int x = 0; x < W; ++x) acc += double(a.v[x] * c.v[x]);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = s_w[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) v += s_w[w];
    lag_store_partial(ws, int64_t(l) * nb + b, v);
  }
  __syncthreads();
  lag_finish<T>(out, L, nb, ws, &s_last);
}

constexpr int kLagMaxBlocksPerLag = 2048;  // (the workspace holds L x 2048 partials)

// PLANE mapping: planes per pass and workgroups.  R whole planes share a workgroup's pass, but no more than 64 / DV of them — the
// per-lag sum over a workgroup's lanes is a serial walk of R * DV LDS words by one thread per lag.
inline void lag_plane_shape(int64_t outer, int plane, int DV, int* R_out, int* nb_out) {
  int R = kBlock / plane;
  const int cap_r = 64 / DV > 0 ? 64 / DV : 1;
  if (R > cap_r) R = cap_r;
  if (R < 1) R = 1;
  if (int64_t(R) > outer) R = int(outer > 0 ? outer : 1);
  const int64_t passes = (outer + R - 1) / R;  // one pass per workgroup at most ...
  constexpr int cap = 512;  // ... and 512 workgroups (256 / 384 / 512 / 768: 14.7 / 13.5 / 12.2 / 13.5 us at the D3STN size, profiles/r05_combine_ab.txt)
  int64_t nb = passes < cap ? passes : cap;
  if (nb > kLagMaxBlocksPerLag) nb = kLagMaxBlocksPerLag;
  if (nb < 1) nb = 1;
  *R_out = R;
  *nb_out = int(nb);
}

inline int lag_rows_blocks(int64_t outer, int D, int L, int width) {
  const int DV = D / width > 0 ? D / width : 1;
  const int row_lanes = DV < kBlock ? DV : kBlock;
  const int rows_per_pass = kBlock / row_lanes;
  int64_t want = (outer + rows_per_pass - 1) / rows_per_pass;  // one pass per workgroup at most
  int64_t cap = 1024 / (L > 0 ? L : 1);  // ~1024 workgroups in all
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return int(want);
}

}  // namespace

extern "C" {

int xde_history_gather(void* val_out, void* der_out, const void* his, const void* his_t, const void* lags, int64_t outer, int T, int D,
                       int L, int dtype, int method, void* stream) {
  if (method == XDE_HISTORY_CUBIC) return xde_hermite_gather(val_out, der_out, his, his_t, lags, outer, T, D, L, dtype, stream);
  if (method != XDE_HISTORY_LINEAR && method != XDE_HISTORY_BEZIER) return fail(XDE_EBADARG, "xde_history_gather: unknown method");
  if (!val_out || !der_out || !his || !his_t || !lags) return fail(XDE_EBADARG, "xde_history_gather: null pointer");
  const int need = method == XDE_HISTORY_LINEAR ? 2 : 4;
  if (outer < 0 || T < need || D < 1 || L < 0)
    return fail(XDE_EBADARG, "xde_history_gather: bad sizes (linear needs T >= 2, bez T >= 4)");
  if (L > kHistMaxL) return fail(XDE_EBADARG, "xde_history_gather: at most 128 lags per launch");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_history_gather: bad dtype");
  if (outer == 0 || L == 0) return XDE_OK;
  const int width = dtype == XDE_F32 ? 4 : 2;
  const bool vec = (D % width) == 0 && aligned16(val_out) && aligned16(der_out) && aligned16(his);
  const int64_t work = outer * int64_t(L) * (vec ? D / width : D);
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > grid_cap()) blocks = grid_cap();
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double es = dtype == XDE_F32 ? 4.0 : 8.0;
  ProfScope prof(XDE_KID_DENSE, double(need + 2) * double(outer) * L * D * es);
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
#define LAUNCH_RG(TY, M)                                                                                                    \
  do {                                                                                                                      \
    if (vec)                                                                                                                \
      XDE_LAUNCH((xde_rows_gather_kernel<TY, M, true>), g, b, st, prof, static_cast<TY*>(val_out), static_cast<TY*>(der_out), \
                 static_cast<const TY*>(his), static_cast<const TY*>(his_t), static_cast<const TY*>(lags), outer, T, D, L);   \
    else                                                                                                                    \
      XDE_LAUNCH((xde_rows_gather_kernel<TY, M, false>), g, b, st, prof, static_cast<TY*>(val_out), static_cast<TY*>(der_out), \
                 static_cast<const TY*>(his), static_cast<const TY*>(his_t), static_cast<const TY*>(lags), outer, T, D, L);   \
  } while (0)
  if (dtype == XDE_F32) {
    if (method == XDE_HISTORY_LINEAR) LAUNCH_RG(float, 2);
    else LAUNCH_RG(float, 4);
  } else {
    if (method == XDE_HISTORY_LINEAR) LAUNCH_RG(double, 2);
    else LAUNCH_RG(double, 4);
  }
#undef LAUNCH_RG
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int64_t xde_lag_grad_workspace_bytes(int L) {
  if (L < 1) L = 1;
  return int64_t(sizeof(LagGradWs)) + int64_t(L) * 2048 * int64_t(sizeof(double));
}

int xde_lag_grad(void* grad_lags_out, const void* grad_y, const void* der, int64_t outer, int D, int L, int dtype, void* ws, void* stream) {
  if (!grad_lags_out || !grad_y || !der || !ws) return fail(XDE_EBADARG, "xde_lag_grad: null pointer");
  if (outer < 0 || D < 1 || L < 1) return fail(XDE_EBADARG, "xde_lag_grad: bad sizes");
  if (L > 2048) return fail(XDE_EBADARG, "xde_lag_grad: at most 2048 lags per launch");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_lag_grad: bad dtype");
  const int width = dtype == XDE_F32 ? 4 : 2;
  // 16-byte vectors when rows are a whole number of them and both tensors are aligned; any other D / alignment is served element-wise
  const bool vec = (D % width) == 0 && aligned16(grad_y) && aligned16(der);
  const int DV = vec ? D / width : D;
  const bool plane = int64_t(L) * DV <= kBlock;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_DENSE, 2.0 * double(outer) * L * D * (dtype == XDE_F32 ? 4.0 : 8.0));
  dim3 b(kBlock);
  if (plane) {
    int R = 1, nb = 1;
    lag_plane_shape(outer, L * DV, DV, &R, &nb);
    dim3 g(static_cast<unsigned>(nb));
#define LAUNCH_LP(TY, V)                                                                                                              \
  XDE_LAUNCH((xde_lag_grad_plane_kernel<TY, V>), g, b, st, prof, static_cast<TY*>(grad_lags_out), static_cast<const TY*>(grad_y), \
             static_cast<const TY*>(der), outer, D, L, R, nb, static_cast<LagGradWs*>(ws))
    if (dtype == XDE_F32) { if (vec) LAUNCH_LP(float, true); else LAUNCH_LP(float, false); }
    else { if (vec) LAUNCH_LP(double, true); else LAUNCH_LP(double, false); }
#undef LAUNCH_LP
  } else {
    const int nb = lag_rows_blocks(outer, D, L, vec ? width : 1);
    dim3 g(static_cast<unsigned>(L * nb));
#define LAUNCH_LR(TY, V)                                                                                                             \
  XDE_LAUNCH((xde_lag_grad_rows_kernel<TY, V>), g, b, st, prof, static_cast<TY*>(grad_lags_out), static_cast<const TY*>(grad_y), \
             static_cast<const TY*>(der), outer, D, L, nb, static_cast<LagGradWs*>(ws))
    if (dtype == XDE_F32) { if (vec) LAUNCH_LR(float, true); else LAUNCH_LR(float, false); }
    else { if (vec) LAUNCH_LR(double, true); else LAUNCH_LR(double, false); }
#undef LAUNCH_LR
  }
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

}  // extern "C"
