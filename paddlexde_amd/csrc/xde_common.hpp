// Shared by every translation unit of libxde_hip.so: launch geometry, environment knobs, 16-byte vector access,
// kernel argument blocks, the norm workspace layout, HIP-event sampling of launches.
#pragma once

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

namespace xde {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kSlots = 2;

extern thread_local std::string g_last_error;
int fail(int code, const std::string& msg);

#define HIP_TRY(expr)                                                               \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess)                                                           \
      return fail(XDE_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));     \
  } while (0)


inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  int x = atoi(v);
  return x > 0 ? x : dflt;
}

// 0 / 1 switches (env_int keeps its default for anything that is not a positive number)
inline bool env_flag(const char* name, bool dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  return atoi(v) != 0;
}

// XDE_NT: bit 0 = stream the dead operands of the error-norm kernel with non-temporal loads; bit 1 = stream every
// operand load of the combine / error-norm kernels when one operand is >= XDE_NT_BYTES (default 64 MiB); bit 2 = honour
// the caller's per-operand last-use mask in the stage combines (-1.6 % step time on config 2).  Default 7.
inline int nt_policy() {
  static int v = [] {
    const char* e = getenv("XDE_NT");
    return (e && *e) ? atoi(e) : 7;
  }();
  return v;
}

inline bool big_operand(int64_t n, int dtype) {
  static int64_t thr = [] {
    const char* e = getenv("XDE_NT_BYTES");
    return (e && *e) ? atoll(e) : (int64_t(64) << 20);
  }();
  return (nt_policy() & 2) && n * (dtype == XDE_F32 ? 4 : 8) >= thr;
}

inline int norm_grid_cap();

inline int grid_cap() {
  static int cap = env_int("XDE_GRID_BLOCKS", 2048);
  return cap > XDE_MAX_PARTIALS ? XDE_MAX_PARTIALS : cap;
}

// Grid of the norm launches (error norm, scaled norms).  Measured on config 2: the streaming rate of these kernels is the
// same with 512, 1024 and 2048 workgroups (each lane keeps 4-8 independent 16-byte loads in flight), while every workgroup
// costs the controller one 32-byte partial record to fetch — 512 is the default.  XDE_NORM_GRID overrides; an explicit
// XDE_GRID_BLOCKS applies to these launches too.
inline int norm_grid_cap() {
  static int cap = [] {
    const char* e = getenv("XDE_NORM_GRID");
    if (e && *e && atoi(e) > 0) return atoi(e);
    const char* g = getenv("XDE_GRID_BLOCKS");
    if (g && *g && atoi(g) > 0) return atoi(g);
    return 512;
  }();
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
  // streaming store (global_store_dwordx4 ... nt): for data nobody re-reads soon (solution rows)
  __device__ void store_nt(float* p, int64_t i) const {
    v4f_t x = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(x, reinterpret_cast<v4f_t*>(p) + i);
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
  __device__ void store_nt(double* p, int64_t i) const {
    v2d_t x = {v[0], v[1]};
    __builtin_nontemporal_store(x, reinterpret_cast<v2d_t*>(p) + i);
  }
};
template <typename T> struct Pack<T, false> {
  static constexpr int W = 1;
  T v[1];
  __device__ static Pack load(const T* p, int64_t i) { return Pack{{p[i]}}; }
  __device__ static Pack load_nt(const T* p, int64_t i) { return Pack{{__builtin_nontemporal_load(p + i)}}; }
  __device__ void store(T* p, int64_t i) const { p[i] = v[0]; }
  __device__ void store_nt(T* p, int64_t i) const { __builtin_nontemporal_store(v[0], p + i); }
};

// dt and the operand select of this launch, read from the device control block.  Both fields are requested BEFORE either is
// waited for (they sit in different cache lines of the block, and every launch finds them cold): one memory round trip
// instead of two dependent ones in front of the first vector load.
template <typename T>
__device__ __forceinline__ void read_dt_sel(const xde_ctrl_t* c, int use_sel, T& dt, int& sel) {
  const double dtd = __builtin_nontemporal_load(&c->dt);
  const int32_t acc = __builtin_nontemporal_load(&c->accept);
  dt = T(dtd);
  sel = (use_sel && acc) ? 1 : 0;
}

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
  void* out2;  // optional second output (RK mode): sum_j k_j * (dt * coef2_j), no y0 — the partial error estimate, or the NEXT stage's partial sum
  const void* pre;  // optional pre-accumulated sum (RK mode): out = y0 + ((pre + k_0 c_0) + k_1 c_1 ...) — written by the previous stage's out2
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
  int nt;  // the launch's cache policy (combine_body, NTP): 0 default, 1 derivatives streamed, 2 everything streamed
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
  // optional predicated commit fused into the launch (hipGraph pipeline): if ctrl->accept, y0 <- y1 and f0 <- f1 after the
  // rows were evaluated (the destinations ARE the y0 / k[0] operands: each lane reads an element before it overwrites it)
  void* commit_y0;
  void* commit_f0;
  int f1_is_last_k;  // f1 == k[nk-1] (an FSAL pair's dense output: the last mid operand IS f1): loaded once
  int nt;            // cache policy: bit 0 = stream the operands that die with this step (y0, k_0..), bit 1 = stream the row stores
};

// fuse(dy, dt, y0): BaseODE `dy*dt + y0` (xde/base_ode.py:58) or, with damping, BaseDDE

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct ProfRec {
  int kid;
  hipEvent_t start, stop;
  double bytes;
};
extern std::mutex g_prof_mu;
extern bool g_prof_on;
extern int g_prof_period;
extern int64_t g_prof_launches[XDE_KID_COUNT];
extern std::vector<ProfRec> g_prof_recs;
extern double g_prof_bytes[XDE_KID_COUNT];
extern int64_t g_prof_counts[XDE_KID_COUNT];
extern double g_prof_ms[XDE_KID_COUNT];
hipEvent_t get_event();          // an idle event of the CURRENT device's pool (created when the pool is empty); g_prof_mu held
void put_event(hipEvent_t e);    // back to the pool of the device it was created on; g_prof_mu held

// When profiling is on, the start/stop events are handed to hipExtLaunchKernelGGL, which stamps them with
// the kernel dispatch's own begin/end timestamps (no extra marker packets on the stream).
struct ProfScope {
  bool on;
  int kid;
  hipEvent_t start = nullptr, stop = nullptr;
  double bytes;
  ProfScope(int kid_, double bytes_) : on(g_prof_on), kid(kid_), bytes(bytes_) {
    if (on) on = (g_prof_launches[kid]++ % g_prof_period) == 0;  // sample every period-th launch of this kernel
    if (on) {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      start = get_event();
      stop = get_event();
    }
  }
  ~ProfScope() {
    if (on) {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      g_prof_recs.push_back(ProfRec{kid, start, stop, bytes});
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


// The caller's statement of xde_segments_t's layout is checked before any other field is read (a stale hand-written mirror
// must get XDE_EBADARG, not a read past its end).
inline int check_segments(const xde_segments_t* segs) {
  if (!segs) return fail(XDE_EBADARG, "segments: null");
  if (segs->struct_size != sizeof(xde_segments_t))
    return fail(XDE_EBADARG, "segments: xde_segments_t layout mismatch: the caller says struct_size=" + std::to_string(segs->struct_size) +
                                 ", this library has sizeof=" + std::to_string(sizeof(xde_segments_t)) +
                                 " (rebuild the binding against include/xde_hip.h)");
  if (segs->n_seg < 1 || segs->n_seg > XDE_MAX_SEG) return fail(XDE_EBADARG, "segments: n_seg out of range");
  return XDE_OK;
}

inline int build_segmap(const xde_segments_t* segs, int width, bool vec, SegMap* m, int* nblocks_out, int cap_override = 0) {
  if (int rc0 = check_segments(segs)) return rc0;
  const int cap = cap_override > 0 ? cap_override : norm_grid_cap();
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

inline bool segs_vec_ok(const xde_segments_t* segs, int width) {
  for (int s = 0; s < segs->n_seg; ++s)
    if (segs->seg_start[s] % width) return false;
  return true;
}

inline NormSlot* slot_ptr(void* ws, int slot) { return reinterpret_cast<NormSlot*>(ws) + slot; }
inline const NormSlot* slot_ptr(const void* ws, int slot) { return reinterpret_cast<const NormSlot*>(ws) + slot; }


}  // namespace xde
