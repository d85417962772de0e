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

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

#include "xde_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kSlots = 2;

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                               \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess)                                                           \
      return fail(XDE_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));     \
  } while (0)

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  int x = atoi(v);
  return x > 0 ? x : dflt;
}

// XDE_NT: bit 0 = stream the dead operands of the error-norm kernel with non-temporal loads; bit 1 = stream every
// operand load of the combine / error-norm kernels when one operand is >= XDE_NT_BYTES (default 64 MiB); bit 2 = honour
// the caller's per-operand last-use mask in the stage combines (-1.6 % step time on config 2).  Default 7.
int nt_policy() {
  static int v = [] {
    const char* e = getenv("XDE_NT");
    return (e && *e) ? atoi(e) : 7;
  }();
  return v;
}

bool big_operand(int64_t n, int dtype) {
  static int64_t thr = [] {
    const char* e = getenv("XDE_NT_BYTES");
    return (e && *e) ? atoll(e) : (int64_t(64) << 20);
  }();
  return (nt_policy() & 2) && n * (dtype == XDE_F32 ? 4 : 8) >= thr;
}

// grid of the fused error-norm + controller launch: fewer, longer-running workgroups = fewer ticket arrivals
int fused_grid_cap() {
  static int cap = env_int("XDE_FUSED_GRID", 1024);
  return cap > XDE_MAX_PARTIALS ? XDE_MAX_PARTIALS : cap;
}

int grid_cap() {
  static int cap = env_int("XDE_GRID_BLOCKS", 2048);
  return cap > XDE_MAX_PARTIALS ? XDE_MAX_PARTIALS : cap;
}

// ------------------------------------------------------------------------------------------
// vector types: 16 bytes per lane
// ------------------------------------------------------------------------------------------
template <typename T> struct VecOf;
template <> struct VecOf<float> { using type = float4; static constexpr int W = 4; };
template <> struct VecOf<double> { using type = double2; static constexpr int W = 2; };

template <typename T, bool VEC> struct Pack;
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef double v2d_t __attribute__((ext_vector_type(2)));

template <> struct Pack<float, true> {
  static constexpr int W = 4;
  float v[4];
  __device__ static Pack load(const float* p, int64_t i) {
    float4 x = reinterpret_cast<const float4*>(p)[i];
    return Pack{{x.x, x.y, x.z, x.w}};
  }
  // streaming load (global_load_dwordx4 ... nt): for operands this kernel reads for the last time
  __device__ static Pack load_nt(const float* p, int64_t i) {
    v4f_t x = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p) + i);
    return Pack{{x.x, x.y, x.z, x.w}};
  }
  __device__ void store(float* p, int64_t i) const {
    reinterpret_cast<float4*>(p)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Pack<double, true> {
  static constexpr int W = 2;
  double v[2];
  __device__ static Pack load(const double* p, int64_t i) {
    double2 x = reinterpret_cast<const double2*>(p)[i];
    return Pack{{x.x, x.y}};
  }
  __device__ static Pack load_nt(const double* p, int64_t i) {
    v2d_t x = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(p) + i);
    return Pack{{x.x, x.y}};
  }
  __device__ void store(double* p, int64_t i) const {
    reinterpret_cast<double2*>(p)[i] = make_double2(v[0], v[1]);
  }
};
template <typename T> struct Pack<T, false> {
  static constexpr int W = 1;
  T v[1];
  __device__ static Pack load(const T* p, int64_t i) { return Pack{{p[i]}}; }
  __device__ static Pack load_nt(const T* p, int64_t i) { return Pack{{__builtin_nontemporal_load(p + i)}}; }
  __device__ void store(T* p, int64_t i) const { p[i] = v[0]; }
};

// load with a run-time (wave-uniform) choice of cache policy
template <typename P, typename T>
__device__ __forceinline__ P load_sel(const T* p, int64_t i, bool nt) {
  return nt ? P::load_nt(p, i) : P::load(p, i);
}

template <typename T> __device__ inline bool finite_(T x) { return (x - x) == T(0); }
template <typename T> __device__ inline T abs_(T x) { return x < T(0) ? -x : x; }
__device__ inline float abs_(float x) { return fabsf(x); }
__device__ inline double abs_(double x) { return fabs(x); }
__device__ inline float fmax_(float a, float b) { return fmaxf(a, b); }
__device__ inline double fmax_(double a, double b) { return fmax(a, b); }
// NaN-propagating max (numpy / paddle .max() semantics)
__device__ inline double nanmax_(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }

// ------------------------------------------------------------------------------------------
// kernel argument blocks (passed by value)
// ------------------------------------------------------------------------------------------
struct CombineArgs {
  void* out;
  void* out2;  // optional second output (RK mode): sum_j k_j * (dt * coef2_j), no y0 — the partial error estimate
  double coef2[XDE_MAX_K];
  const void* y0[2];
  const void* k[XDE_MAX_K];
  const void* k0_alt;
  double coef[XDE_MAX_K];
  double scale;
  double dt_host;
  double damp;  // BaseDDE.fuse: (dy - damp * (dy*dt + y0)) * dt + y0   (0 = BaseODE.fuse)
  const xde_ctrl_t* ctrl;
  int64_t n;
  int nk;
  int use_sel;
  int nt;  // non-temporal mask: bit j = stream operand k_j (its last use), bit 31 = stream y0
};

struct SegMap {
  int32_t n_seg;
  int32_t seg_blk[XDE_MAX_SEG + 1];  // block ranges per segment
  int64_t seg_start[XDE_MAX_SEG];
  int64_t seg_len[XDE_MAX_SEG];
};

struct alignas(16) Partial {
  double val;   // sum of squares (RMS) or max |.| (LINF) of the block's elements
  double nf;    // non-finite elements of y0 seen by the block
  int32_t seg;  // segment the block worked on
  int32_t pad[3];
};
static_assert(sizeof(Partial) == 32, "Partial is two 16-byte stores / loads");

constexpr int kTicketShards = 16;  // first-level arrival counters (one atomic word serialises at ~12 ns per add)

struct NormSlot {
  int32_t nblocks;
  int32_t n_seg;
  int32_t norm_kind;
  uint32_t ticket;  // top-level arrivals (one per completed shard) of the fused error-norm + controller launch
  // first-level arrival counters, one 128-byte line each; all ticket words are zero between launches
  struct alignas(128) Shard {
    uint32_t count;
    uint32_t pad[31];
  } shard[kTicketShards];
  Partial p[XDE_MAX_PARTIALS];
};

struct ErrArgs {
  const void* k[XDE_MAX_K];
  const void* k0_alt;
  double coef[XDE_MAX_K];
  const void* y0[2];
  const void* y1;
  double rtol, atol, dt_host;
  const xde_ctrl_t* ctrl;
  NormSlot* slot;
  SegMap map;
  const void* e_pre;  // optional pre-accumulated partial error (second output of the last stage's combine)
  int nk;
  int use_sel;
  int nt;
};

struct ScaledArgs {
  const void* a;
  const void* b;
  const void* y0;
  double rtol, atol;
  NormSlot* slot;
  SegMap map;
};

struct DenseArgs {
  void* out_base;
  const void* k[XDE_MAX_K];
  const void* k0_alt;
  double mid[XDE_MAX_K];
  const void* y0[2];
  const void* y1;
  const void* f1;
  const xde_ctrl_t* ctrl;
  const double* t_span;
  int64_t n;
  int nk;
  int use_sel;
  int time_dtype;
  int64_t expect_step;
};

// fuse(dy, dt, y0): BaseODE `dy*dt + y0` (xde/base_ode.py:58) or, with damping, BaseDDE
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
template <typename T, int MODE, int NK, bool VEC, bool OUT2>
__device__ __forceinline__ void combine_body(const CombineArgs& a, const T* __restrict__ y0,
                                             const T* __restrict__ k0, T dt) {
  const unsigned ntm = unsigned(a.nt);  // bit j: stream operand k_j; bit 31: stream y0
  using P = Pack<T, VEC>;
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
    c2[j] = OUT2 ? dt * T(a.coef2[j]) : T(0);  // `dt * tableau.c_error`
  }
  const T scale = T(a.scale);
  const T lam = T(a.damp);
  const int64_t nvec = a.n / W;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nvec; i += stride) {
    // NT: with operands of >= 64 MiB nothing survives in the 256 MiB Infinity Cache between uses anyway; streaming
    // loads then run 12-17 % faster (5.1 -> 5.8 TB/s at 128 MiB x 7 streams).  At the 32 MiB headline size the
    // default policy wins by 18 % (the working set half-fits the cache), so the flag is size-dependent (host).
    P y = load_sel<P>(y0, i, (ntm >> 31) & 1u);
    P kk[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) kk[j] = load_sel<P>(kp[j], i, (ntm >> j) & 1u);
    P o;
    P o2;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      if (MODE == XDE_COMBINE_RK) {
        T acc = kk[0].v[w] * c[0];
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
      } else {
        T acc = fuse_(kk[0].v[w], dt, y.v[w], lam) * c[0];
#pragma unroll
        for (int j = 1; j < NK; ++j) acc = acc + fuse_(kk[j].v[w], dt, y.v[w], lam) * c[j];
        o.v[w] = acc * scale;
      }
    }
    o.store(out, i);
    if (OUT2) o2.store(out2, i);
  }
  if (VEC) {
    // scalar tail (n % W elements), done by the first threads of block 0
    const int64_t tail0 = nvec * W;
    const int64_t i = tail0 + threadIdx.x;
    if (blockIdx.x == 0 && i < a.n) {
      T yv = y0[i];
      T acc;
      if (OUT2) {
        T e = kp[0][i] * c2[0];
        for (int j = 1; j < NK; ++j) e = e + kp[j][i] * c2[j];
        out2[i] = e;
      }
      if (MODE == XDE_COMBINE_WFUSE) {
        acc = fuse_(kp[0][i], dt, yv, lam) * c[0];
        for (int j = 1; j < NK; ++j) acc = acc + fuse_(kp[j][i], dt, yv, lam) * c[j];
        out[i] = acc * scale;
      } else {
        acc = kp[0][i] * c[0];
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
      T c2j = has2 ? dt * T(a.coef2[j]) : T(0);
#pragma unroll
      for (int w = 0; w < W; ++w) {
        T term = (MODE == XDE_COMBINE_WFUSE) ? fuse_(kk.v[w], dt, y.v[w], lam) * cj : kk.v[w] * cj;
        acc.v[w] = (j == 0) ? term : acc.v[w] + term;
        if (has2) e2.v[w] = (j == 0) ? kk.v[w] * c2j : e2.v[w] + kk.v[w] * c2j;
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
        if (has2) e2 = (j == 0) ? kj[i] * (dt * T(a.coef2[j])) : e2 + kj[i] * (dt * T(a.coef2[j]));
      }
      if (has2) static_cast<T*>(a.out2)[i] = e2;
      out[i] = (MODE == XDE_COMBINE_RK) ? yv + acc : (MODE == XDE_COMBINE_FUSE) ? fuse_(acc, dt, yv, lam) : acc * scale;
    }
  }
}

// OUT2 (second output, RK mode only) is a separate instantiation: its two accumulators per element would
// otherwise raise the register budget of EVERY stage launch (62 -> 112 VGPRs, occupancy 8 -> 4 waves/SIMD)
template <typename T, int MODE, bool VEC, bool OUT2>
__global__ __launch_bounds__(kBlock) void xde_combine_kernel(CombineArgs a) {
  int sel = 0;
  T dt;
  if (a.ctrl) {
    dt = T(a.ctrl->dt);
    if (a.use_sel) sel = a.ctrl->accept ? 1 : 0;
  } else {
    dt = T(a.dt_host);
  }
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  switch (a.nk) {
    case 1: combine_body<T, MODE, 1, VEC, OUT2>(a, y0, k0, dt); break;
    case 2: combine_body<T, MODE, 2, VEC, OUT2>(a, y0, k0, dt); break;
    case 3: combine_body<T, MODE, 3, VEC, OUT2>(a, y0, k0, dt); break;
    case 4: combine_body<T, MODE, 4, VEC, OUT2>(a, y0, k0, dt); break;
    case 5: combine_body<T, MODE, 5, VEC, OUT2>(a, y0, k0, dt); break;
    case 6: combine_body<T, MODE, 6, VEC, OUT2>(a, y0, k0, dt); break;
    case 7: combine_body<T, MODE, 7, VEC, OUT2>(a, y0, k0, dt); break;
    default: combine_generic<T, MODE, VEC, OUT2>(a, y0, k0, dt); break;
  }
}

// ------------------------------------------------------------------------------------------
// block reduction: fp64 value (sum or nan-max) + fp64 non-finite count → one partial per block
// ------------------------------------------------------------------------------------------
template <int NORM>
__device__ __forceinline__ double merge_(double a, double b) {
  return NORM == XDE_NORM_RMS ? a + b : nanmax_(a, b);
}

// TICKET = false: plain store of the partial (a later launch reduces them).
// TICKET = true (fused error-norm + controller): the partial is published to the workgroup that arrives last, inside
// the launch, by the write-through form of cdna_hip_programming.md Guideline 16: 8-byte agent-scope (sc1) stores by ONE
// lane, that lane's `s_waitcnt vmcnt(0)`, then an agent-scope atomic add on the ticket; the add's return value tells
// the last arriver.  Returns true in every thread of the last workgroup.
template <int NORM, bool TICKET = false>
__device__ bool block_reduce_store(double val, double nf, NormSlot* slot, int seg) {
  __shared__ double s_val[kWaves];
  __shared__ double s_nf[kWaves];
  __shared__ int s_last;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    val = merge_<NORM>(val, __shfl_down(val, off, 64));
    nf += __shfl_down(nf, off, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_val[wave] = val;
    s_nf[wave] = nf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = s_val[0], f = s_nf[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
      v = merge_<NORM>(v, s_val[w]);
      f += s_nf[w];
    }
    if (!TICKET) {
      Partial rec;
      rec.val = v;
      rec.nf = f;
      rec.seg = seg;
      rec.pad[0] = rec.pad[1] = rec.pad[2] = 0;
      slot->p[blockIdx.x] = rec;
    } else {
      unsigned long long* w = reinterpret_cast<unsigned long long*>(&slot->p[blockIdx.x]);
      __hip_atomic_store(w + 0, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(w + 1, (unsigned long long)__double_as_longlong(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(w + 2, (unsigned long long)(unsigned)seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have left this CU before the ticket is taken
      // two-level arrival: shard = blockIdx % 16 (blocks b, b+16, ... share a counter); the last arriver of a shard
      // arrives at the top-level counter; the last arriver there is the last workgroup of the launch
      const unsigned nsh = gridDim.x < unsigned(kTicketShards) ? gridDim.x : unsigned(kTicketShards);
      const unsigned sh = blockIdx.x % nsh;
      const unsigned members = (gridDim.x - sh + nsh - 1) / nsh;
      int last = 0;
      unsigned old = __hip_atomic_fetch_add(&slot->shard[sh].count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == members - 1) {
        unsigned old2 = __hip_atomic_fetch_add(&slot->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (old2 == nsh - 1) ? 1 : 0;
      }
      s_last = last;
    }
  }
  if (!TICKET) return false;
  __syncthreads();
  return s_last != 0;
}

__device__ __forceinline__ int find_segment(const SegMap& m, int b) {
  int s = 0;
  while (s + 1 < m.n_seg && b >= m.seg_blk[s + 1]) ++s;
  return s;
}

// ------------------------------------------------------------------------------------------
// K2: fused error combine + tolerance scaling + norm partials (+ non-finite count of y0)
// ------------------------------------------------------------------------------------------
template <typename T, int NK, int NORM, bool VEC, bool NT, bool PRE = false>
__device__ __forceinline__ void errnorm_body(const ErrArgs& a, const T* __restrict__ y0, const T* __restrict__ k0,
                                             T dt, int seg, int lb, int nb, T& acc_out, int& nf_out) {
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
    P y0v = NT ? P::load_nt(y0, vbase + i) : P::load(y0, vbase + i);
    P y1v = P::load(y1, vbase + i);
    P kk[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) kk[j] = (NT && j < NK - 1) ? P::load_nt(kp[j], vbase + i) : P::load(kp[j], vbase + i);
    P ep;
    if (PRE) ep = NT ? P::load_nt(epre, vbase + i) : P::load(epre, vbase + i);
#pragma unroll
    for (int w = 0; w < W; ++w) {
      T e = PRE ? ep.v[w] + kk[0].v[w] * c[0] : kk[0].v[w] * c[0];
#pragma unroll
      for (int j = 1; j < NK; ++j) e = e + kk[j].v[w] * c[j];
      one(e, y0v.v[w], y1v.v[w]);
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
  acc_out = acc;
  nf_out = nf;
}

// operand count > 8 (Dopri8): runtime loop over operands, same arithmetic order
template <typename T, int NORM, bool VEC>
__device__ void errnorm_generic(const ErrArgs& a, const T* __restrict__ y0, const T* __restrict__ k0, T dt, int seg,
                                int lb, int nb, T& acc_out, int& nf_out) {
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
  acc_out = acc;
  nf_out = nf;
}

template <typename T, int NORM, bool VEC>
__device__ __forceinline__ void errnorm_dispatch(const ErrArgs& a, const T* __restrict__ y0, const T* __restrict__ k0, T dt,
                                                 int seg, int lb, int nb, T& acc, int& nf) {
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

template <typename T, int NORM, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_errnorm_kernel(ErrArgs a) {
  int sel = 0;
  T dt;
  if (a.ctrl) {
    dt = T(a.ctrl->dt);
    if (a.use_sel) sel = a.ctrl->accept ? 1 : 0;
  } else {
    dt = T(a.dt_host);
  }
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  const int seg = find_segment(a.map, blockIdx.x);
  const int lb = blockIdx.x - a.map.seg_blk[seg];
  const int nb = a.map.seg_blk[seg + 1] - a.map.seg_blk[seg];
  T acc = T(0);
  int nf = 0;
  errnorm_dispatch<T, NORM, VEC>(a, y0, k0, dt, seg, lb, nb, acc, nf);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.slot->nblocks = gridDim.x;
    a.slot->n_seg = a.map.n_seg;
    a.slot->norm_kind = NORM;
  }
  block_reduce_store<NORM>(double(acc), double(nf), a.slot, seg);
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
  block_reduce_store<NORM>(double(acc), double(nf), s.slot, seg);
}

// ------------------------------------------------------------------------------------------
// fixed-order reduction of block partials → per-segment sums (one workgroup)
// ------------------------------------------------------------------------------------------
template <bool SC1 = false>
__device__ void reduce_partials(const NormSlot* slot, double* seg_val, double* seg_nf, int nblocks_known = -1,
                                int n_seg_known = 0, int norm_kind_known = 0) {
  // Called by all kBlock threads of ONE block; results land in seg_val/seg_nf (shared, XDE_MAX_SEG each).
  // Every thread first pulls its (up to) XDE_MAX_PARTIALS/kBlock records into registers with independent
  // 32-byte loads (one memory round trip), then the per-segment reductions run on registers:
  // strided per-thread order -> wave64 shuffle tree -> 4 wave results summed in order.  Fixed order, so
  // the value is bit-reproducible from launch to launch and identical on every rank.
  constexpr int kPer = XDE_MAX_PARTIALS / kBlock;
  __shared__ double w_val[kWaves];
  __shared__ double w_nf[kWaves];
  // (the fused launch knows its own grid; it must not read the header other workgroups are not ordered with)
  const int nblocks = nblocks_known >= 0 ? nblocks_known : slot->nblocks;
  const int n_seg = nblocks_known >= 0 ? n_seg_known : slot->n_seg;
  const bool rms = (nblocks_known >= 0 ? norm_kind_known : slot->norm_kind) == XDE_NORM_RMS;
  double rv[kPer], rf[kPer];
  int rs[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int b = threadIdx.x + i * kBlock;
    if (b < nblocks) {
      if (SC1) {  // write-through partials of this launch: agent-scope (sc1) loads bypass this CU's L1
        const unsigned long long* w = reinterpret_cast<const unsigned long long*>(&slot->p[b]);
        rv[i] = __longlong_as_double((long long)__hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        rf[i] = __longlong_as_double((long long)__hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        rs[i] = (int)(unsigned)__hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        Partial rec = slot->p[b];
        rv[i] = rec.val;
        rf[i] = rec.nf;
        rs[i] = rec.seg;
      }
    } else {
      rv[i] = 0.0;
      rf[i] = 0.0;
      rs[i] = -1;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = 0; s < n_seg; ++s) {
    double v = 0.0, f = 0.0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      if (rs[i] == s) {
        v = rms ? v + rv[i] : nanmax_(v, rv[i]);
        f += rf[i];
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      double ov = __shfl_down(v, off, 64);
      v = rms ? v + ov : nanmax_(v, ov);
      f += __shfl_down(f, off, 64);
    }
    if (lane == 0) {
      w_val[wave] = v;
      w_nf[wave] = f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double tv = w_val[0], tf = w_nf[0];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) {
        tv = rms ? tv + w_val[w] : nanmax_(tv, w_val[w]);
        tf += w_nf[w];
      }
      seg_val[s] = tv;
      seg_nf[s] = tf;
    }
    __syncthreads();
  }
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

__device__ inline double round_to(double x, int dtype) { return dtype == XDE_F32 ? double(float(x)) : x; }

// norm value from finalised sums: max over segments of sqrt(sum/count) (RMS) or of max (LINF),
// in the state dtype like `_rms_norm` (utils/ode_utils.py:8-9).  NaN propagates.
__device__ double norm_from_sums(const double* seg_val, const double* seg_count, int n_seg, int norm_kind,
                                 int state_dtype, double* per_seg_out) {
  double ratio = 0.0;
  for (int s = 0; s < n_seg; ++s) {
    double r;
    if (norm_kind == XDE_NORM_RMS) {
      double mean = round_to(seg_val[s] / seg_count[s], state_dtype);
      r = round_to(sqrt(mean), state_dtype);
    } else {
      r = round_to(seg_val[s], state_dtype);
    }
    r = fabs(r);
    if (per_seg_out) per_seg_out[s] = r;
    ratio = s == 0 ? r : nanmax_(ratio, r);
  }
  return ratio;
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
// K3: controller
// ------------------------------------------------------------------------------------------
template <typename TT> __device__ inline TT pow_(TT a, TT b);
template <> __device__ inline float pow_<float>(float a, float b) { return powf(a, b); }
template <> __device__ inline double pow_<double>(double a, double b) { return pow(a, b); }
template <typename TT> __device__ inline TT fmin__(TT a, TT b);
template <> __device__ inline float fmin__<float>(float a, float b) { return fminf(a, b); }
template <> __device__ inline double fmin__<double>(double a, double b) { return fmin(a, b); }
template <typename TT> __device__ inline TT fmax__(TT a, TT b);
template <> __device__ inline float fmax__<float>(float a, float b) { return fmaxf(a, b); }
template <> __device__ inline double fmax__<double>(double a, double b) { return fmax(a, b); }

// Plan the pending attempt: step_t clipping (base_adaptive_solver_rk.py:209-215), the underflow and
// max_num_steps assertions (:200, :120-122) and the stage times of _runge_kutta_step (:159-164).
template <typename TT>
__device__ void plan_next(xde_ctrl_t* c, const xde_ctrl_params_t& p, const double* step_t, void* t_stage_out) {
  const TT dir = TT(p.direction);
  TT t0 = TT(c->t1);
  TT dt = TT(c->dt);
  TT t1 = t0 + dt;
  int on = 0;
  if (p.n_step_t > 0 && step_t) {
    TT nxt = TT(step_t[c->next_step_index]);
    if (dir * t0 < dir * nxt && dir * nxt < dir * (t0 + dt)) {
      on = 1;
      t1 = nxt;
      dt = t1 - t0;
    }
  }
  c->on_step_t = on;
  c->dt = double(dt);
  c->t_plan = double(t1);
  if (!c->done) {
    if (!(dir * (t0 + dt) > dir * t0) && c->status == XDE_STATUS_OK) c->status = XDE_STATUS_DT_UNDERFLOW;
    if (c->steps_in_interval >= p.max_num_steps && c->status == XDE_STATUS_OK) c->status = XDE_STATUS_MAX_STEPS;
  }
  // stage times in the state dtype: `t0.astype(t_dtype)`, `ti = t1 if alpha_i == 1 else t0 + alpha_i * dt`
  for (int i = 0; i < p.n_stage; ++i) {
    if (p.state_dtype == XDE_F32) {
      float a = float(p.alpha[i]);
      float ti = (a == 1.0f) ? float(t1) : float(t0) + a * float(dt);
      static_cast<float*>(t_stage_out)[i] = ti;
    } else {
      double a = p.alpha[i];
      double ti = (a == 1.0) ? double(t1) : double(t0) + a * double(dt);
      static_cast<double*>(t_stage_out)[i] = ti;
    }
  }
}

template <typename TT>
__device__ void control_step(xde_ctrl_t* c, const xde_ctrl_params_t& p, double ratio_d, double nonfinite,
                             const double* t_span, const double* step_t, void* t_stage_out) {
  const TT dir = TT(p.direction);
  const TT t0 = TT(c->t1);
  const TT dt = TT(c->dt);
  const TT t1 = TT(c->t_plan);
  const TT min_step = TT(p.min_step), max_step = TT(p.max_step);

  if (nonfinite > 0.0 && c->status == XDE_STATUS_OK) c->status = XDE_STATUS_NONFINITE;

  // accept / reject                                         base_adaptive_solver_rk.py:244-250
  int accept = (ratio_d <= 1.0) ? 1 : 0;  // NaN -> reject
  if (dir * dt > max_step) accept = 0;
  if (dir * dt <= min_step) accept = 1;

  // optimal_step_size                                                  utils/ode_utils.py:85-97
  TT dt_next;
  if (ratio_d == 0.0) {
    dt_next = dt * TT(p.ifactor);
  } else {
    TT dfactor = TT(p.dfactor);
    if (ratio_d < 1.0) dfactor = TT(1);
    TT ratio = TT(ratio_d);
    TT factor;
    if (p.pi_controller) {
      // opt-in PI controller (Hairer's dopri5 form); never used for parity
      TT beta = TT(p.pi_beta);
      TT alpha = TT(1) / TT(p.order) - TT(0.75) * beta;
      TT prev = TT(c->ratio_prev > 1e-4 ? c->ratio_prev : 1e-4);
      factor = fmin__<TT>(TT(p.ifactor), fmax__<TT>(TT(p.safety) * pow_<TT>(prev, beta) / pow_<TT>(ratio, alpha), dfactor));
    } else {
      TT exponent = TT(1) / TT(p.order);
      factor = fmin__<TT>(TT(p.ifactor), fmax__<TT>(TT(p.safety) / pow_<TT>(ratio, exponent), dfactor));
    }
    dt_next = dt * factor;
  }
  // dt_next.clip(min_step, max_step) on the magnitude (direction-aware)
  {
    TT mag = dir * dt_next;
    if (mag < min_step) mag = min_step;
    if (mag > max_step) mag = max_step;
    dt_next = (mag != mag) ? dt_next : dir * mag;
  }

  c->n_steps += 1;
  c->steps_in_interval += 1;
  if (accept) {
    c->n_accept += 1;
    if (ratio_d == ratio_d) c->ratio_prev = ratio_d;
  } else {
    c->n_reject += 1;
  }
  c->sel_used = c->accept;
  c->accept = accept;
  c->ratio = ratio_d;
  c->nonfinite = nonfinite;
  c->t0 = double(t0);
  c->t1 = accept ? double(t1) : double(t0);
  c->dt_last = double(dt);
  c->dt = double(dt_next);

  // outputs covered by this step: step() loops `while next_t > rk_state.t1`  (:116-127)
  int b = c->next_out;
  int e = b;
  if (accept) {
    while (e < c->n_out && dir * TT(t_span[e]) <= dir * t1) ++e;
  }
  c->out_begin = b;
  c->out_end = e;
  c->next_out = e;
  if (e > b) c->steps_in_interval = 0;
  c->done = (e >= c->n_out) ? 1 : 0;

  if (accept && c->on_step_t && c->next_step_index != p.n_step_t - 1) c->next_step_index += 1;  // :263-265

  plan_next<TT>(c, p, step_t, t_stage_out);
}

// The controller workgroup: reduce the partials (or take finalised sums), run the controller on a register copy of
// the control block, write it back to the device block and the pinned host mirror.  FUSED = called by the last
// workgroup of the fused error-norm launch (partials were published write-through inside this launch).
template <bool FUSED>
__device__ void control_block(xde_ctrl_t* c, const xde_ctrl_params_t& p, const NormSlot* slot, const double* sums,
                              const double* t_span, const double* step_t, void* t_stage_out, xde_ctrl_t* mirror,
                              int nblocks, int norm_kind) {
  __shared__ double seg_val[XDE_MAX_SEG];
  __shared__ double seg_nf[XDE_MAX_SEG];
  __shared__ xde_ctrl_t zs;
  constexpr int kWords = sizeof(xde_ctrl_t) / 8;
  constexpr int kSeqWord = offsetof(xde_ctrl_t, seq) / 8;
  // the control block is fetched by the first lanes while the partials are being reduced (it is written only by
  // controller launches, i.e. before this launch started)
  if (threadIdx.x < kWords)
    reinterpret_cast<uint64_t*>(&zs)[threadIdx.x] = reinterpret_cast<const uint64_t*>(c)[threadIdx.x];
  if (sums) {
    if (threadIdx.x < XDE_MAX_SEG) {
      seg_val[threadIdx.x] = sums[threadIdx.x];
      seg_nf[threadIdx.x] = sums[XDE_MAX_SEG + threadIdx.x];
    }
    __syncthreads();
  } else {
    if (threadIdx.x < XDE_MAX_SEG) {
      seg_val[threadIdx.x] = 0.0;
      seg_nf[threadIdx.x] = 0.0;
    }
    __syncthreads();
    if (FUSED)
      reduce_partials<true>(slot, seg_val, seg_nf, nblocks, p.n_seg, norm_kind);
    else
      reduce_partials<false>(slot, seg_val, seg_nf);
  }
  if (threadIdx.x == 0) {
    xde_ctrl_t z = zs;  // all controller arithmetic runs on registers
    z.seq += 1;
    if (z.done) {
      // an attempt enqueued past the last output (speculative / graph replay) is a no-op: nothing to commit,
      // no rows to emit
      z.accept = 0;
      z.out_begin = z.out_end = z.next_out;
    } else {
      double ratio = norm_from_sums(seg_val, p.seg_count, p.n_seg, p.norm_kind, p.state_dtype, z.ratio_seg);
      double nf = 0.0;
      for (int s = 0; s < p.n_seg; ++s) nf += seg_nf[s];
      if (p.time_dtype == XDE_F32)
        control_step<float>(&z, p, ratio, nf, t_span, step_t, t_stage_out);
      else
        control_step<double>(&z, p, ratio, nf, t_span, step_t, t_stage_out);
    }
    zs = z;
  }
  __syncthreads();
  // write-back: one wave instruction to the device block, one to the pinned host mirror slot
  // (slot[seq % SLOTS]); the slot's seq word is stored last, after a system-scope release
  if (threadIdx.x < kWords) {
    const uint64_t word = reinterpret_cast<const uint64_t*>(&zs)[threadIdx.x];
    reinterpret_cast<uint64_t*>(c)[threadIdx.x] = word;
    if (mirror && threadIdx.x != kSeqWord) {
      xde_ctrl_t* ms = mirror + (zs.seq % XDE_MIRROR_SLOTS);
      __hip_atomic_store(reinterpret_cast<uint64_t*>(ms) + threadIdx.x, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (mirror && threadIdx.x < 64) {  // the wave that wrote the words releases them, then publishes seq
    __threadfence_system();
    if (threadIdx.x == 0) {
      xde_ctrl_t* ms = mirror + (zs.seq % XDE_MIRROR_SLOTS);
      __hip_atomic_store(&ms->seq, zs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(kBlock) void xde_control_kernel(xde_ctrl_t* c, xde_ctrl_params_t p, const NormSlot* slot,
                                                             const double* sums, const double* t_span,
                                                             const double* step_t, void* t_stage_out,
                                                             xde_ctrl_t* mirror) {
  control_block<false>(c, p, slot, sums, t_span, step_t, t_stage_out, mirror, 0, 0);
}

// Controller arguments of the fused launch
struct CtrlTail {
  xde_ctrl_t* ctrl;
  xde_ctrl_params_t p;
  const double* t_span;
  const double* step_t;
  void* t_stage_out;
  xde_ctrl_t* mirror;
};

// K2+K3 fused: every workgroup does the error-norm pass and publishes its partial; the workgroup whose ticket is the
// last one acquires (agent scope), reduces all partials in the same fixed order as the standalone controller, runs
// the controller and resets the ticket.  Nobody waits for anybody: no spin, no residency requirement.
template <typename T, int NORM, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_errnorm_control_kernel(ErrArgs a, CtrlTail tl) {
  int sel = 0;
  const T dt = T(a.ctrl->dt);
  if (a.use_sel) sel = a.ctrl->accept ? 1 : 0;
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  const int seg = find_segment(a.map, blockIdx.x);
  const int lb = blockIdx.x - a.map.seg_blk[seg];
  const int nb = a.map.seg_blk[seg + 1] - a.map.seg_blk[seg];
  T acc = T(0);
  int nf = 0;
  errnorm_dispatch<T, NORM, VEC>(a, y0, k0, dt, seg, lb, nb, acc, nf);
  const bool last = block_reduce_store<NORM, true>(double(acc), double(nf), a.slot, seg);
  if (!last) return;
  // ---- last arriver: every other workgroup's partial is in L2/memory (sc1 stores, drained before its ticket) ----
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  control_block<true>(tl.ctrl, tl.p, a.slot, nullptr, tl.t_span, tl.step_t, tl.t_stage_out, tl.mirror, int(gridDim.x), NORM);
  // every workgroup has arrived: re-arm the ticket words for the next launch
  if (threadIdx.x < kTicketShards)
    __hip_atomic_store(&a.slot->shard[threadIdx.x].count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) __hip_atomic_store(&a.slot->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void xde_ctrl_init_kernel(xde_ctrl_t* c, xde_ctrl_params_t p, double t_start, double first_step,
                                     int32_t n_out, const double* t_span, const double* step_t, void* t_stage_out,
                                     int64_t seq0, const double* first_step_dev) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  xde_ctrl_t z;
  memset(&z, 0, sizeof(z));
  z.t0 = t_start;
  z.t1 = t_start;
  // device-resident first step (xde_initial_step): a magnitude, given the direction's sign here
  z.dt = first_step_dev ? double(p.direction) * fabs(*first_step_dev) : first_step;
  z.n_out = n_out;
  z.ratio_prev = 1e-4;
  // rows whose time equals the start time are y0 itself (the reference's `while next_t > t1` does not
  // step for them); the host fills them.
  const double dir = double(p.direction);
  int e = 1;
  while (e < n_out && dir * t_span[e] <= dir * t_start) ++e;
  z.next_out = e;
  z.out_begin = e;
  z.out_end = e;
  z.done = (e >= n_out) ? 1 : 0;
  // next_step_index = min(bisect(step_t, t0), len-1)                 base_adaptive_solver_rk.py:109-111
  int idx = 0;
  if (p.n_step_t > 0 && step_t) {
    while (idx < p.n_step_t && dir * step_t[idx] <= dir * t_start) ++idx;
    if (idx > p.n_step_t - 1) idx = p.n_step_t - 1;
  }
  z.next_step_index = idx;
  z.seq = seq0;
  if (p.time_dtype == XDE_F32)
    plan_next<float>(&z, p, step_t, t_stage_out);
  else
    plan_next<double>(&z, p, step_t, t_stage_out);
  *c = z;
}

// Hairer's initial-step heuristic, scalar part (solver/base_adaptive_solver.py:55-72), in the state dtype Y with the
// reference's op order.  phase 0: (d0, d1) -> h0, written to ctrl->dt (the Euler probe y0 + h0*f0 is a combine that
// reads dt from ctrl) and t0 + h0 for func.  phase 1: d2 = |norm((f1-f0)/scale) / h0| -> h1 -> min(100*h0, h1).
template <typename Y>
__device__ void initial_step_phase(int phase, const double* res, double* hs, const xde_ctrl_params_t& p, double t_start,
                                   void* t_probe_out, int probe_dtype, xde_ctrl_t* c) {
  if (phase == 0) {
    const Y d0 = Y(fabs(res[0])), d1 = Y(fabs(res[1]));
    Y h0;
    if (d0 < Y(1e-5) || d1 < Y(1e-5))
      h0 = Y(1e-6);
    else
      h0 = Y(0.01) * d0 / d1;
    h0 = h0 < Y(0) ? -h0 : h0;
    hs[0] = double(d0);
    hs[1] = double(d1);
    hs[2] = double(h0);
    c->dt = double(h0);
    // t0 + h0: time dtype + state dtype -> the promoted dtype
    if (probe_dtype == XDE_F32)
      *static_cast<float*>(t_probe_out) = float(t_start) + float(h0);
    else
      *static_cast<double*>(t_probe_out) = (p.time_dtype == XDE_F32 ? double(float(t_start)) : t_start) + double(h0);
  } else {
    const Y h0 = Y(hs[2]), d1 = Y(hs[1]);
    Y d2 = Y(res[0]) / h0;
    d2 = d2 < Y(0) ? -d2 : d2;
    Y h1;
    if (d1 <= Y(1e-15) && d2 <= Y(1e-15)) {
      const Y a = Y(1e-6), b = h0 * Y(1e-3);
      h1 = (b > a) ? b : a;  // Python's max(a, b)
    } else {
      // Python's max(d1, d2): returns d1 unless d2 > d1 (a NaN d2 is ignored, a NaN d1 is kept)
      const Y m = (d2 > d1) ? d2 : d1;
      const Y e = Y(1.0 / (p.order - 1.0 + 1.0));  // called with order - 1: exponent 1 / (order - 1 + 1)
      h1 = Y(pow_<Y>(Y(0.01) / m, e));
    }
    h1 = h1 < Y(0) ? -h1 : h1;
    const Y a = Y(100.0) * h0;
    const Y first = fmin__<Y>(a, h1);
    hs[3] = p.time_dtype == XDE_F32 ? double(float(first)) : double(first);
  }
}

__global__ void xde_initial_step_kernel(int phase, const double* res, double* hs, xde_ctrl_params_t p, double t_start,
                                        void* t_probe_out, int probe_dtype, xde_ctrl_t* c) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (p.state_dtype == XDE_F32)
    initial_step_phase<float>(phase, res, hs, p, t_start, t_probe_out, probe_dtype, c);
  else
    initial_step_phase<double>(phase, res, hs, p, t_start, t_probe_out, probe_dtype, c);
}

// ------------------------------------------------------------------------------------------
// K4: dense output (quartic through y0, y_mid, y1, f0, f1), coefficients never materialised
// ------------------------------------------------------------------------------------------
template <typename T, typename TT, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_dense_kernel(DenseArgs a) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  const xde_ctrl_t* c = a.ctrl;
  if (!c->accept) return;
  if (a.expect_step >= 0 && c->n_steps != a.expect_step) return;
  const int ob = c->out_begin, oe = c->out_end;
  if (oe <= ob) return;
  const int sel = a.use_sel ? (c->sel_used ? 1 : 0) : 0;
  const T* __restrict__ y0 = static_cast<const T*>(a.y0[sel]);
  const T* __restrict__ k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  const T* __restrict__ y1 = static_cast<const T*>(a.y1);
  const T* __restrict__ f1p = static_cast<const T*>(a.f1);
  T* __restrict__ out = static_cast<T*>(a.out_base);
  const T dt = T(TT(c->dt_last));  // `dt.astype(y0.dtype)`
  const TT t0 = TT(c->t0), t1 = TT(c->t1);
  const int nk = a.nk;
  const int64_t nvec = a.n / W;
  const int64_t stride = int64_t(gridDim.x) * kBlock;

  auto eval = [&](T y0v, T y1v, T f0v, T f1v, T ymid, T x) -> T {
    // interp_fit (utils/ode_utils.py:44-49) + interp_evaluate (:69-77), same op order
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
  };

  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nvec; i += stride) {
    P y0v = P::load(y0, i);
    P y1v = P::load(y1, i);
    P f0v = P::load(k0, i);
    P f1v = P::load(f1p, i);
    P acc;
    for (int j = 0; j < nk; ++j) {
      const T* kj = j == 0 ? k0 : static_cast<const T*>(a.k[j]);
      P kk = (j == 0) ? f0v : P::load(kj, i);
      T cj = dt * T(a.mid[j]);  // `dt * self.mid`
#pragma unroll
      for (int w = 0; w < W; ++w) acc.v[w] = (j == 0) ? kk.v[w] * cj : acc.v[w] + kk.v[w] * cj;
    }
    for (int r = ob; r < oe; ++r) {
      TT xt = (TT(a.t_span[r]) - t0) / (t1 - t0);
      T x = T(xt);
      P o;
#pragma unroll
      for (int w = 0; w < W; ++w) o.v[w] = eval(y0v.v[w], y1v.v[w], f0v.v[w], f1v.v[w], y0v.v[w] + acc.v[w], x);
      o.store(out + int64_t(r) * a.n, i);
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
      for (int r = ob; r < oe; ++r) {
        TT xt = (TT(a.t_span[r]) - t0) / (t1 - t0);
        out[int64_t(r) * a.n + i] = eval(y0[i], y1[i], k0[i], f1p[i], y0[i] + acc, T(xt));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// element-wise error ratio (materialised only for user-supplied norm callables)
// ------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_ratio_kernel(ErrArgs a, T* __restrict__ out, int64_t n) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  const T dt = a.ctrl ? T(a.ctrl->dt) : T(a.dt_host);
  const T* __restrict__ y0 = static_cast<const T*>(a.y0[0]);
  const T* __restrict__ y1 = static_cast<const T*>(a.y1);
  const T rtol = T(a.rtol), atol = T(a.atol);
  const int nk = a.nk;
  const int64_t nvec = n / W;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nvec; i += stride) {
    P y0v = P::load(y0, i);
    P y1v = P::load(y1, i);
    P e;
    for (int j = 0; j < nk; ++j) {
      P kk = P::load(static_cast<const T*>(a.k[j]), i);
      T cj = dt * T(a.coef[j]);
#pragma unroll
      for (int w = 0; w < W; ++w) e.v[w] = (j == 0) ? kk.v[w] * cj : e.v[w] + kk.v[w] * cj;
    }
    P o;
#pragma unroll
    for (int w = 0; w < W; ++w) o.v[w] = e.v[w] / (atol + rtol * fmax_(abs_(y0v.v[w]), abs_(y1v.v[w])));
    o.store(out, i);
  }
  if (VEC) {
    const int64_t i = nvec * W + threadIdx.x;
    if (blockIdx.x == 0 && i < n) {
      T e = T(0);
      for (int j = 0; j < nk; ++j) {
        T term = static_cast<const T*>(a.k[j])[i] * (dt * T(a.coef[j]));
        e = (j == 0) ? term : e + term;
      }
      out[i] = e / (atol + rtol * fmax_(abs_(y0[i]), abs_(y1[i])));
    }
  }
}

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

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct ProfRec {
  int kid;
  hipEvent_t start, stop;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
int g_prof_period = 1;
int64_t g_prof_launches[XDE_KID_COUNT] = {0};
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_event_pool;
double g_prof_bytes[XDE_KID_COUNT] = {0};
int64_t g_prof_counts[XDE_KID_COUNT] = {0};
double g_prof_ms[XDE_KID_COUNT] = {0};

hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

// When profiling is on, the start/stop events are handed to hipExtLaunchKernelGGL, which stamps them with
// the kernel dispatch's own begin/end timestamps (no extra marker packets on the stream).
struct ProfScope {
  bool on;
  int kid;
  hipEvent_t start = nullptr, stop = nullptr;
  ProfScope(int kid_, double bytes) : on(g_prof_on), kid(kid_) {
    if (on) on = (g_prof_launches[kid]++ % g_prof_period) == 0;  // sample every period-th launch of this kernel
    if (on) {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      start = get_event();
      stop = get_event();
      g_prof_bytes[kid] += bytes;
    }
  }
  ~ProfScope() {
    if (on) {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      g_prof_recs.push_back(ProfRec{kid, start, stop});
    }
  }
};

// plain launch unless this launch is being timed (the plain form is what stream capture records)
#define XDE_LAUNCH(kernel, grid, block, st, prof, ...)                                                       \
  do {                                                                                                       \
    if ((prof).on)                                                                                           \
      hipExtLaunchKernelGGL(kernel, grid, block, 0, st, (prof).start, (prof).stop, 0, __VA_ARGS__);         \
    else                                                                                                     \
      hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                           \
  } while (0)

int build_segmap(const xde_segments_t* segs, int width, bool vec, SegMap* m, int* nblocks_out, int cap_override = 0) {
  if (!segs || segs->n_seg < 1 || segs->n_seg > XDE_MAX_SEG) return fail(XDE_EBADARG, "segments: n_seg out of range");
  const int cap = cap_override > 0 ? cap_override : grid_cap();
  m->n_seg = segs->n_seg;
  int64_t total = 0;
  for (int s = 0; s < segs->n_seg; ++s) {
    if (segs->seg_len[s] < 0 || segs->seg_start[s] < 0) return fail(XDE_EBADARG, "segments: negative start/len");
    if (vec && (segs->seg_start[s] % width) != 0) return fail(XDE_EBADARG, "segments: start not vector aligned");
    total += segs->seg_len[s];
  }
  int b = 0;
  for (int s = 0; s < segs->n_seg; ++s) {
    m->seg_start[s] = segs->seg_start[s];
    m->seg_len[s] = segs->seg_len[s];
    m->seg_blk[s] = b;
    int64_t per_block = int64_t(kBlock) * width;
    int64_t want = (segs->seg_len[s] + per_block - 1) / per_block;
    // share the grid cap between segments proportionally to their length, at least one block each
    int64_t share = total > 0 ? (int64_t(cap) * segs->seg_len[s] + total - 1) / total : 1;
    if (share < 1) share = 1;
    if (want > share) want = share;
    if (want < 1) want = 1;
    b += int(want);
  }
  m->seg_blk[segs->n_seg] = b;
  if (b > XDE_MAX_PARTIALS) return fail(XDE_EBADARG, "segments: too many blocks");
  *nblocks_out = b;
  return XDE_OK;
}

bool segs_vec_ok(const xde_segments_t* segs, int width) {
  for (int s = 0; s < segs->n_seg; ++s)
    if (segs->seg_start[s] % width) return false;
  return true;
}

inline NormSlot* slot_ptr(void* ws, int slot) { return reinterpret_cast<NormSlot*>(ws) + slot; }
inline const NormSlot* slot_ptr(const void* ws, int slot) { return reinterpret_cast<const NormSlot*>(ws) + slot; }

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

const char* xde_last_error(void) { return g_last_error.c_str(); }
int xde_abi_version(void) { return XDE_ABI_VERSION; }
int64_t xde_sizeof_ctrl(void) { return int64_t(sizeof(xde_ctrl_t)); }
int64_t xde_workspace_bytes(void) { return int64_t(sizeof(NormSlot)) * kSlots; }

int xde_stage_combine(void* out, const void* y0, const void* y0_alt, const void* const* k, const void* k0_alt,
                      const double* coef, int nk, int mode, double scale, double dt_host, const xde_ctrl_t* ctrl,
                      int64_t n, int dtype, void* out2, const double* coef2, double damping, uint32_t nt_mask, void* stream) {
  if (!out || !y0 || !k || !coef) return fail(XDE_EBADARG, "xde_stage_combine: null pointer");
  if (nk < 1 || nk > XDE_MAX_K) return fail(XDE_EBADARG, "xde_stage_combine: nk out of range");
  if (n < 0) return fail(XDE_EBADARG, "xde_stage_combine: negative n");
  if (mode < 0 || mode > 2) return fail(XDE_EBADARG, "xde_stage_combine: bad mode");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_stage_combine: bad dtype");
  if ((y0_alt == nullptr) != (k0_alt == nullptr)) return fail(XDE_EBADARG, "xde_stage_combine: y0_alt/k0_alt must come together");
  if (y0_alt && !ctrl) return fail(XDE_EBADARG, "xde_stage_combine: operand select needs ctrl");
  if ((out2 == nullptr) != (coef2 == nullptr)) return fail(XDE_EBADARG, "xde_stage_combine: out2/coef2 must come together");
  if (out2 && mode != XDE_COMBINE_RK) return fail(XDE_EBADARG, "xde_stage_combine: second output needs mode RK");
  if (n == 0) return XDE_OK;
  CombineArgs a;
  memset(&a, 0, sizeof(a));
  a.out = out;
  a.out2 = out2;
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
  if (damping != 0.0 && mode == XDE_COMBINE_RK) return fail(XDE_EBADARG, "xde_stage_combine: damping applies to FUSE/WFUSE");
  a.damp = damping;
  // operands far larger than the Infinity Cache: stream everything; otherwise only what the caller marks as last use
  a.nt = big_operand(n, dtype) ? int(0xFFFFFFFFu) : ((nt_policy() & 4) ? int(nt_mask) : 0);
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
  ProfScope prof(XDE_KID_COMBINE, double(nk + 2 + (out2 ? 1 : 0)) * double(n) * elt);
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
#define LAUNCH_COMBINE(T, MODE)                                                     \
  do {                                                                              \
    if (vec)                                                                        \
      XDE_LAUNCH((xde_combine_kernel<T, MODE, true, false>), g, b, st, prof, a);    \
    else                                                                            \
      XDE_LAUNCH((xde_combine_kernel<T, MODE, false, false>), g, b, st, prof, a);   \
  } while (0)
#define LAUNCH_COMBINE2(T)                                                                   \
  do {                                                                                       \
    if (vec)                                                                                 \
      XDE_LAUNCH((xde_combine_kernel<T, XDE_COMBINE_RK, true, true>), g, b, st, prof, a);    \
    else                                                                                     \
      XDE_LAUNCH((xde_combine_kernel<T, XDE_COMBINE_RK, false, true>), g, b, st, prof, a);   \
  } while (0)
  if (out2) {
    if (dtype == XDE_F32) LAUNCH_COMBINE2(float);
    else LAUNCH_COMBINE2(double);
  } else if (dtype == XDE_F32) {
    if (mode == XDE_COMBINE_RK) LAUNCH_COMBINE(float, XDE_COMBINE_RK);
    else if (mode == XDE_COMBINE_FUSE) LAUNCH_COMBINE(float, XDE_COMBINE_FUSE);
    else LAUNCH_COMBINE(float, XDE_COMBINE_WFUSE);
  } else {
    if (mode == XDE_COMBINE_RK) LAUNCH_COMBINE(double, XDE_COMBINE_RK);
    else if (mode == XDE_COMBINE_FUSE) LAUNCH_COMBINE(double, XDE_COMBINE_FUSE);
    else LAUNCH_COMBINE(double, XDE_COMBINE_WFUSE);
  }
#undef LAUNCH_COMBINE
#undef LAUNCH_COMBINE2
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

static int setup_err_args(const char* who, const void* const* k, const void* k0_alt, const double* c_err, int nk, const void* y0,
                          const void* y0_alt, const void* y1, double rtol, double atol, double dt_host, const xde_ctrl_t* ctrl,
                          const xde_segments_t* segs, int norm_kind, int dtype, void* ws, const void* e_pre, ErrArgs* out,
                          bool* vec_out, int* nblocks_out, double* bytes_out, int cap_override = 0) {
  const std::string w(who);
  if (!k || !c_err || !y0 || !y1 || !ws || !segs) return fail(XDE_EBADARG, w + ": null pointer");
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
#define LAUNCH_ERR(T, NORM)                                                       \
  do {                                                                            \
    if (vec)                                                                      \
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

static int check_params(const xde_ctrl_params_t* p, const char* who);

int xde_error_norm_control(const void* const* k, const void* k0_alt, const double* c_err, int nk, const void* y0,
                           const void* y0_alt, const void* y1, const xde_segments_t* segs, int dtype, void* ws,
                           const void* e_pre, xde_ctrl_t* ctrl, const xde_ctrl_params_t* params, const double* t_span_dev,
                           const double* step_t_dev, void* t_stage_out, xde_ctrl_t* host_mirror, void* stream) {
  if (!ctrl || !t_span_dev || !t_stage_out) return fail(XDE_EBADARG, "xde_error_norm_control: null pointer");
  int rc = check_params(params, "xde_error_norm_control");
  if (rc != XDE_OK) return rc;
  if (params->n_step_t > 0 && !step_t_dev) return fail(XDE_EBADARG, "xde_error_norm_control: n_step_t > 0 without step_t_dev");
  if (!segs || segs->n_seg != params->n_seg) return fail(XDE_EBADARG, "xde_error_norm_control: segments do not match params->n_seg");
  ErrArgs a;
  bool vec = false;
  int nblocks = 0;
  double bytes = 0;
  rc = setup_err_args("xde_error_norm_control", k, k0_alt, c_err, nk, y0, y0_alt, y1, params->rtol, params->atol, 0.0, ctrl, segs,
                      params->norm_kind, dtype, ws, e_pre, &a, &vec, &nblocks, &bytes, fused_grid_cap());
  if (rc != XDE_OK) return rc;
  CtrlTail tl;
  tl.ctrl = ctrl;
  tl.p = *params;
  tl.t_span = t_span_dev;
  tl.step_t = step_t_dev;
  tl.t_stage_out = t_stage_out;
  tl.mirror = host_mirror;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_ERRNORM, bytes);
  dim3 g(nblocks), b(kBlock);
#define LAUNCH_EC(T, NORM)                                                                 \
  do {                                                                                     \
    if (vec)                                                                               \
      XDE_LAUNCH((xde_errnorm_control_kernel<T, NORM, true>), g, b, st, prof, a, tl);      \
    else                                                                                   \
      XDE_LAUNCH((xde_errnorm_control_kernel<T, NORM, false>), g, b, st, prof, a, tl);     \
  } while (0)
  if (dtype == XDE_F32) {
    if (params->norm_kind == XDE_NORM_RMS) LAUNCH_EC(float, XDE_NORM_RMS);
    else LAUNCH_EC(float, XDE_NORM_LINF);
  } else {
    if (params->norm_kind == XDE_NORM_RMS) LAUNCH_EC(double, XDE_NORM_RMS);
    else LAUNCH_EC(double, XDE_NORM_LINF);
  }
#undef LAUNCH_EC
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_error_ratio(void* out, const void* const* k, const double* c_err, int nk, const void* y0, const void* y1,
                    double rtol, double atol, double dt_host, const xde_ctrl_t* ctrl, int64_t n, int dtype, void* stream) {
  if (!out || !k || !c_err || !y0 || !y1) return fail(XDE_EBADARG, "xde_error_ratio: null pointer");
  if (nk < 1 || nk > XDE_MAX_K) return fail(XDE_EBADARG, "xde_error_ratio: nk out of range");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_error_ratio: bad dtype");
  if (n < 0) return fail(XDE_EBADARG, "xde_error_ratio: negative n");
  if (n == 0) return XDE_OK;
  ErrArgs a;
  memset(&a, 0, sizeof(a));
  a.y0[0] = a.y0[1] = y0;
  a.y1 = y1;
  bool vec = aligned16(out) && aligned16(y0) && aligned16(y1);
  for (int j = 0; j < nk; ++j) {
    if (!k[j]) return fail(XDE_EBADARG, "xde_error_ratio: null k[j]");
    a.k[j] = k[j];
    a.coef[j] = c_err[j];
    vec = vec && aligned16(k[j]);
  }
  a.k0_alt = k[0];
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
    if (vec) XDE_LAUNCH((xde_ratio_kernel<float, true>), g, b, st, prof, a, static_cast<float*>(out), n);
    else XDE_LAUNCH((xde_ratio_kernel<float, false>), g, b, st, prof, a, static_cast<float*>(out), n);
  } else {
    if (vec) XDE_LAUNCH((xde_ratio_kernel<double, true>), g, b, st, prof, a, static_cast<double*>(out), n);
    else XDE_LAUNCH((xde_ratio_kernel<double, false>), g, b, st, prof, a, static_cast<double*>(out), n);
  }
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_scaled_norm_partial(const void* av, const void* bv, const void* y0, double rtol, double atol,
                            const xde_segments_t* segs, int norm_kind, int dtype, void* ws, int slot, void* stream) {
  if (!av || !y0 || !ws || !segs) return fail(XDE_EBADARG, "xde_scaled_norm_partial: null pointer");
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

static int check_params(const xde_ctrl_params_t* p, const char* who) {
  if (!p) return fail(XDE_EBADARG, std::string(who) + ": null params");
  if (p->n_stage < 1 || p->n_stage > XDE_MAX_STAGE) return fail(XDE_EBADARG, std::string(who) + ": n_stage out of range");
  if (p->n_seg < 1 || p->n_seg > XDE_MAX_SEG) return fail(XDE_EBADARG, std::string(who) + ": n_seg out of range");
  if (p->direction != 1 && p->direction != -1) return fail(XDE_EBADARG, std::string(who) + ": direction must be +-1");
  if (p->time_dtype != XDE_F32 && p->time_dtype != XDE_F64) return fail(XDE_EBADARG, std::string(who) + ": bad time_dtype");
  if (p->state_dtype != XDE_F32 && p->state_dtype != XDE_F64) return fail(XDE_EBADARG, std::string(who) + ": bad state_dtype");
  if (!(p->order > 0)) return fail(XDE_EBADARG, std::string(who) + ": order must be positive");
  return XDE_OK;
}

int xde_rk_control(xde_ctrl_t* ctrl, const xde_ctrl_params_t* params, const void* ws, const double* sums,
                   const double* t_span_dev, const double* step_t_dev, void* t_stage_out, xde_ctrl_t* host_mirror,
                   void* stream) {
  if (!ctrl || !t_span_dev || !t_stage_out) return fail(XDE_EBADARG, "xde_rk_control: null pointer");
  if (!ws && !sums) return fail(XDE_EBADARG, "xde_rk_control: need ws or sums");
  int rc = check_params(params, "xde_rk_control");
  if (rc != XDE_OK) return rc;
  if (params->n_step_t > 0 && !step_t_dev) return fail(XDE_EBADARG, "xde_rk_control: n_step_t > 0 without step_t_dev");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_CONTROL, 0.0);
  XDE_LAUNCH(xde_control_kernel, dim3(1), dim3(kBlock), st, prof, ctrl, *params, ws ? slot_ptr(ws, 0) : nullptr, sums,
             t_span_dev, step_t_dev, t_stage_out, host_mirror);
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_ctrl_init(xde_ctrl_t* ctrl, const xde_ctrl_params_t* params, double t_start, double first_step, int32_t n_out,
                  const double* t_span_dev, const double* step_t_dev, void* t_stage_out, int64_t seq0,
                  const double* first_step_dev, void* stream) {
  if (!ctrl || !t_span_dev || !t_stage_out) return fail(XDE_EBADARG, "xde_ctrl_init: null pointer");
  int rc = check_params(params, "xde_ctrl_init");
  if (rc != XDE_OK) return rc;
  if (n_out < 1) return fail(XDE_EBADARG, "xde_ctrl_init: n_out must be >= 1");
  if (params->n_step_t > 0 && !step_t_dev) return fail(XDE_EBADARG, "xde_ctrl_init: n_step_t > 0 without step_t_dev");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(xde_ctrl_init_kernel, dim3(1), dim3(64), 0, st, ctrl, *params, t_start, first_step, n_out,
                     t_span_dev, step_t_dev, t_stage_out, seq0, first_step_dev);
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_initial_step(int phase, const double* res_dev, double* hs_dev, const xde_ctrl_params_t* params, double t_start,
                     void* t_probe_out, int probe_dtype, xde_ctrl_t* ctrl, void* stream) {
  if (!res_dev || !hs_dev || !ctrl) return fail(XDE_EBADARG, "xde_initial_step: null pointer");
  if (phase != 0 && phase != 1) return fail(XDE_EBADARG, "xde_initial_step: phase must be 0 or 1");
  if (phase == 0 && !t_probe_out) return fail(XDE_EBADARG, "xde_initial_step: phase 0 needs t_probe_out");
  if (probe_dtype != XDE_F32 && probe_dtype != XDE_F64) return fail(XDE_EBADARG, "xde_initial_step: bad probe dtype");
  int rc = check_params(params, "xde_initial_step");
  if (rc != XDE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(xde_initial_step_kernel, dim3(1), dim3(64), 0, st, phase, res_dev, hs_dev, *params, t_start, t_probe_out,
                     probe_dtype, ctrl);
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_ctrl_read(const xde_ctrl_t* ctrl_dev, xde_ctrl_t* host_out, void* stream) {
  if (!ctrl_dev || !host_out) return fail(XDE_EBADARG, "xde_ctrl_read: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(host_out, ctrl_dev, sizeof(xde_ctrl_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return XDE_OK;
}

int xde_host_alloc(int64_t bytes, void** ptr_out) {
  if (bytes <= 0 || !ptr_out) return fail(XDE_EBADARG, "xde_host_alloc: bad argument");
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, size_t(bytes), hipHostMallocMapped | hipHostMallocCoherent));
  memset(p, 0, size_t(bytes));
  *ptr_out = p;
  return XDE_OK;
}

int xde_host_free(void* ptr) {
  if (!ptr) return XDE_OK;
  HIP_TRY(hipHostFree(ptr));
  return XDE_OK;
}

int xde_ctrl_wait(const xde_ctrl_t* host_mirror, int64_t seq, double timeout_ms, xde_ctrl_t* host_out) {
  if (!host_mirror || !host_out || seq < 0) return fail(XDE_EBADARG, "xde_ctrl_wait: bad argument");
  const xde_ctrl_t* slot = host_mirror + (seq % XDE_MIRROR_SLOTS);
  const auto t_begin = std::chrono::steady_clock::now();
  uint64_t spins = 0;
  bool slow = false;
  for (;;) {
    int64_t cur = __atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE);
    if (cur == seq) break;
    if (cur > seq) return fail(XDE_EBADARG, "xde_ctrl_wait: mirror slot already overwritten by a later launch");
    if (slow || (++spins & 0x3ff) == 0) {
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
      if (ms > timeout_ms) return fail(XDE_ETIMEOUT, "xde_ctrl_wait: timed out waiting for the controller launch");
      slow = ms > 5.0;  // a long func evaluation is in flight: stop burning the core, poll every ~50 us
    }
    if (slow) {
      struct timespec ts = {0, 50000};
      nanosleep(&ts, nullptr);
      continue;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  memcpy(host_out, slot, sizeof(xde_ctrl_t));
  // the slot may have been overwritten while copying (only if the host lags >= SLOTS launches behind)
  if (__atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE) != seq)
    return fail(XDE_EBADARG, "xde_ctrl_wait: mirror slot overwritten while reading");
  return XDE_OK;
}

int xde_dense_eval(void* out_base, const void* const* k, const void* k0_alt, const double* mid, int nk, const void* y0,
                   const void* y0_alt, const void* y1, const void* f1, const xde_ctrl_t* ctrl, const double* t_span_dev,
                   int time_dtype, int64_t n, int dtype, int64_t expect_step, void* stream) {
  if (!out_base || !k || !mid || !y0 || !y1 || !f1 || !ctrl || !t_span_dev) return fail(XDE_EBADARG, "xde_dense_eval: null pointer");
  if (nk < 1 || nk > XDE_MAX_K) return fail(XDE_EBADARG, "xde_dense_eval: nk out of range");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_dense_eval: bad dtype");
  if (time_dtype != XDE_F32 && time_dtype != XDE_F64) return fail(XDE_EBADARG, "xde_dense_eval: bad time_dtype");
  if ((y0_alt == nullptr) != (k0_alt == nullptr)) return fail(XDE_EBADARG, "xde_dense_eval: y0_alt/k0_alt must come together");
  if (n < 0) return fail(XDE_EBADARG, "xde_dense_eval: negative n");
  if (n == 0) return XDE_OK;
  DenseArgs a;
  memset(&a, 0, sizeof(a));
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
  const int width = dtype == XDE_F32 ? 4 : 2;
  // every output row starts at out_base + r*n elements: rows stay 16-byte aligned only if n % width == 0
  bool vec = aligned16(out_base) && (n % width == 0) && aligned16(y0) && aligned16(a.y0[1]) && aligned16(y1) &&
             aligned16(f1) && aligned16(a.k0_alt);
  for (int j = 0; j < nk; ++j) {
    if (!k[j]) return fail(XDE_EBADARG, "xde_dense_eval: null k[j]");
    a.k[j] = k[j];
    a.mid[j] = mid[j];
    vec = vec && aligned16(k[j]);
  }
  const int64_t work = vec ? (n + width - 1) / width : n;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  // predicated launch: most launches of the speculative pipeline exit at once, so keep the grid small (a no-op
  // launch costs ~5 us with 2048 workgroups); the streaming rate does not depend on the grid between 512 and 4096
  const int64_t dense_cap = grid_cap() < 512 ? grid_cap() : 512;
  if (blocks > dense_cap) blocks = dense_cap;
  if (blocks < 1) blocks = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_DENSE, double(nk + 4) * double(n) * (dtype == XDE_F32 ? 4.0 : 8.0));
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
#define LAUNCH_DENSE(T, TT)                                                     \
  do {                                                                          \
    if (vec)                                                                    \
      XDE_LAUNCH((xde_dense_kernel<T, TT, true>), g, b, st, prof, a);      \
    else                                                                        \
      XDE_LAUNCH((xde_dense_kernel<T, TT, false>), g, b, st, prof, a);     \
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

int xde_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  g_prof_period = on > 1 ? on : 1;
  if (on) {
    for (int i = 0; i < XDE_KID_COUNT; ++i) {
      g_prof_launches[i] = 0;
      g_prof_bytes[i] = 0;
      g_prof_counts[i] = 0;
      g_prof_ms[i] = 0;
    }
    for (auto& r : g_prof_recs) {
      g_event_pool.push_back(r.start);
      g_event_pool.push_back(r.stop);
    }
    g_prof_recs.clear();
  }
  return XDE_OK;
}

int xde_prof_collect(int64_t* counts_out, double* ms_out, double* bytes_out) {
  if (!counts_out || !ms_out || !bytes_out) return fail(XDE_EBADARG, "xde_prof_collect: null pointer");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) {
    HIP_TRY(hipEventSynchronize(r.stop));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.start, r.stop));
    g_prof_counts[r.kid] += 1;
    g_prof_ms[r.kid] += double(ms);
    g_event_pool.push_back(r.start);
    g_event_pool.push_back(r.stop);
  }
  g_prof_recs.clear();
  for (int i = 0; i < XDE_KID_COUNT; ++i) {
    counts_out[i] = g_prof_counts[i];
    ms_out[i] = g_prof_ms[i];
    bytes_out[i] = g_prof_bytes[i];
  }
  return XDE_OK;
}

}  // extern "C"
