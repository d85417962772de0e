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
// K3: the device-resident step controller, control-block init, initial-step scalars, host mirror.

#include "xde_common.hpp"
#include "xde_reduce.hpp"
#include "xde_errnorm_device.hpp"
#include "xde_control_device.hpp"

using namespace xde;

namespace {

__global__ __launch_bounds__(kBlock) void xde_control_kernel(xde_ctrl_t* c, xde_ctrl_params_t p, const NormSlot* slot,
                                                             const double* sums, const double* t_span,
                                                             const double* step_t, void* t_stage_out,
                                                             xde_ctrl_t* mirror, int flags, int partial_cap) {
  control_block<false>(c, p, slot, sums, t_span, step_t, t_stage_out, mirror, 0, 0, flags, partial_cap);
}

// Controller arguments of the fused launch
struct CtrlTail {
  xde_ctrl_t* ctrl;
  xde_ctrl_params_t p;
  const double* t_span;
  const double* step_t;
  void* t_stage_out;
  xde_ctrl_t* mirror;
  int flags;
};

// K2+K3 fused: every workgroup does the error-norm pass and publishes its partial; the workgroup whose ticket is the
// last one acquires (agent scope), reduces all partials in the same fixed order as the standalone controller, runs
// the controller and resets the ticket.  Nobody waits for anybody: no spin, no residency requirement.
template <typename T, int NORM, bool VEC>
__global__ __launch_bounds__(kBlock) void xde_errnorm_control_kernel(ErrArgs a, CtrlTail tl) {
  int sel = 0;
  T dt;
  read_dt_sel<T>(a.ctrl, a.use_sel, dt, sel);
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  const int seg = find_segment(a.map, blockIdx.x);
  const int lb = blockIdx.x - a.map.seg_blk[seg];
  const int nb = a.map.seg_blk[seg + 1] - a.map.seg_blk[seg];
  double acc = 0.0;
  int nf = 0;
  errnorm_dispatch<T, NORM, VEC>(a, y0, k0, dt, seg, lb, nb, acc, nf);
  const bool last = block_reduce_store<NORM, true>(acc, double(nf), a.slot, seg);
  if (!last) return;
  // ---- last arriver: every other workgroup's partial is in L2/memory (sc1 stores, drained before its ticket) ----
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  control_block<true>(tl.ctrl, tl.p, a.slot, nullptr, tl.t_span, tl.step_t, tl.t_stage_out, tl.mirror, int(gridDim.x), NORM, tl.flags);
  // every workgroup has arrived: re-arm the ticket words for the next launch
  if (threadIdx.x < kTicketShards)
    __hip_atomic_store(&a.slot->shard[threadIdx.x].count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) __hip_atomic_store(&a.slot->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// K2+K3 for SMALL states (launch-latency-bound: configs 3 and 5): ONE workgroup walks every segment, reduces it in LDS and
// runs the controller — no partial records, no tickets, one launch (and one hipGraph node) less per attempted step.
// The workgroup has kSingleBlock = 1024 lanes (a 32 KiB state is two 16-byte loads per lane and stream: the pass is one or two
// memory round trips, not eight); errnorm_body's indexing takes it as kSingleBlock / kBlock "virtual" workgroups side by side.
constexpr int kSingleBlock = 1024;
constexpr int kSingleWaves = kSingleBlock / 64;

template <typename T, int NORM, bool VEC>
__global__ __launch_bounds__(kSingleBlock) void xde_errnorm_control_single_kernel(ErrArgs a, CtrlTail tl, int flags) {
  __shared__ double seg_val[XDE_MAX_SEG];
  __shared__ double seg_nf[XDE_MAX_SEG];
  __shared__ double w_val[kSingleWaves][XDE_MAX_SEG];
  __shared__ double w_nf[kSingleWaves][XDE_MAX_SEG];
  __shared__ xde_ctrl_t zs;
  __shared__ TimePrefetch pfs;
  control_prologue(tl.ctrl, tl.p, tl.t_span, tl.step_t, tl.mirror, flags, &zs, &pfs);
  int sel = 0;
  T dt;
  read_dt_sel<T>(a.ctrl, a.use_sel, dt, sel);
  const T* y0 = static_cast<const T*>(a.y0[sel]);
  const T* k0 = static_cast<const T*>(sel ? a.k0_alt : a.k[0]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int seg = 0; seg < a.map.n_seg; ++seg) {
    double acc = 0.0;
    int nfi = 0;
    errnorm_dispatch<T, NORM, VEC>(a, y0, k0, dt, seg, 0, kSingleBlock / kBlock, acc, nfi);
    double nf = double(nfi);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      acc = merge_<NORM>(acc, __shfl_down(acc, off, 64));
      nf += __shfl_down(nf, off, 64);
    }
    if (lane == 0) {  // parked per segment; the cross-wave step runs once for all segments below (it ran per segment, two barriers each)
      w_val[wave][seg] = acc;
      w_nf[wave][seg] = nf;
    }
  }
  __syncthreads();
  if (int(threadIdx.x) < a.map.n_seg) {
    const int seg = threadIdx.x;
    double v = w_val[0][seg], f = w_nf[0][seg];
#pragma unroll
    for (int w = 1; w < kSingleWaves; ++w) {
      v = merge_<NORM>(v, w_val[w][seg]);
      f += w_nf[w][seg];
    }
    seg_val[seg] = v;
    seg_nf[seg] = f;
  }
  __syncthreads();
  control_tail(tl.ctrl, tl.p, seg_val, seg_nf, tl.t_span, tl.step_t, tl.t_stage_out, tl.mirror, flags, &zs, &pfs);
}

// Construction of the control block before the first step (`_RungeKuttaState(y0, f0, t[0], t[0], first_step, ...)`,
// base_adaptive_solver_rk.py:89-92, and the step_t bookkeeping :95-111); `z` is an LDS scratch copy (no private copy: no scratch memory).
// One thread.
__device__ __forceinline__ void ctrl_init_body(xde_ctrl_t* c, const xde_ctrl_params_t& p, double t_start, double first_step_signed,
                                               int32_t n_out, const double* t_span, const double* step_t, void* t_stage_out,
                                               int64_t seq0, xde_ctrl_t& z) {
  // seq0 < 0: a block that is re-armed in place (a captured interval solve) keeps counting its controller launches where it was
  const int64_t seq_start = seq0 < 0 ? c->seq : seq0;
  for (int i = 0; i < int(sizeof(xde_ctrl_t) / 8); ++i) reinterpret_cast<uint64_t*>(&z)[i] = 0;
  z.t0 = t_start;
  z.t1 = t_start;
  z.dt = first_step_signed;
  if (p.replay && p.n_replay > 0) z.dt = p.replay[0];  // prescribed step sequence: the first attempt's step
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
  z.seq = seq_start;
  if (p.time_dtype == XDE_F32)
    plan_next<float>(&z, p, step_t, t_stage_out);
  else
    plan_next<double>(&z, p, step_t, t_stage_out);
  for (int i = 0; i < int(sizeof(xde_ctrl_t) / 8); ++i) reinterpret_cast<uint64_t*>(c)[i] = reinterpret_cast<const uint64_t*>(&z)[i];
}

__global__ void xde_ctrl_init_kernel(xde_ctrl_t* c, xde_ctrl_params_t p, double t_start, double first_step,
                                     int32_t n_out, const double* t_span, const double* step_t, void* t_stage_out,
                                     int64_t seq0, const double* first_step_dev) {
  __shared__ xde_ctrl_t z;  // (LDS, not a private copy: no scratch)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // device-resident first step (xde_initial_step): a magnitude, given the direction's sign here
  const double first = first_step_dev ? double(p.direction) * fabs(*first_step_dev) : first_step;
  if (t_start != t_start) t_start = t_span[0];  // (a launch recorded in a graph: the start time is the first output time)
  ctrl_init_body(c, p, t_start, first, n_out, t_span, step_t, t_stage_out, seq0, z);
}

// New output list for a running solve (AdaptiveRKSolver.step(next_t), base_adaptive_solver_rk.py:116-127): rows of the new
// list that the last ACCEPTED step already covers are reported at once (the reference's `while next_t > rk_state.t1` is
// false for them and it interpolates in the retained step); the `max_num_steps` count restarts, as it does per step() call.
__global__ __launch_bounds__(64) void xde_ctrl_retarget_kernel(xde_ctrl_t* c, xde_ctrl_params_t p, const double* t_span,
                                                               int32_t n_out, xde_ctrl_t* mirror, int flags) {
  __shared__ xde_ctrl_t zs;
  constexpr int kWords = sizeof(xde_ctrl_t) / 8;
  if (threadIdx.x < kWords) {
    const uint64_t word = reinterpret_cast<const uint64_t*>(c)[threadIdx.x];
    reinterpret_cast<uint64_t*>(&zs)[threadIdx.x] = word;
    if (mirror && !(flags & kCtrlChecksum) && threadIdx.x == offsetof(xde_ctrl_t, seq) / 8) invalidate_slot(mirror, int64_t(word) + 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    xde_ctrl_t& z = zs;  // in place, in LDS (no scratch: see control_block)
    z.seq += 1;
    const double dir = double(p.direction);
    int e = 0;
    if (z.accept && z.n_accept > 0)
      while (e < n_out && dir * t_span[e] <= dir * z.t1) ++e;
    z.n_out = n_out;
    z.out_begin = 0;
    z.out_end = e;
    z.next_out = e;
    z.done = (e >= n_out) ? 1 : 0;
    z.steps_in_interval = 0;
    if (z.status == XDE_STATUS_MAX_STEPS) z.status = XDE_STATUS_OK;
  }
  __syncthreads();
  publish_block(c, zs, mirror, flags);
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
    // the Euler probe is taken in the direction of integration (the reference integrates forward only; in reverse time
    // this equals its heuristic on the flipped problem t -> -t, f -> -f)
    const Y hs0 = p.direction < 0 ? -h0 : h0;
    c->dt = double(hs0);
    // t0 + h0: time dtype + state dtype -> the promoted dtype
    if (probe_dtype == XDE_F32)
      *static_cast<float*>(t_probe_out) = float(t_start) + float(hs0);
    else
      *static_cast<double*>(t_probe_out) = (p.time_dtype == XDE_F32 ? double(float(t_start)) : t_start) + double(hs0);
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
  if (phase == 2) {  // phase 0 of a launch recorded in a graph: the start time is device data (res[2])
    phase = 0;
    t_start = res[2];
  }
  if (p.state_dtype == XDE_F32)
    initial_step_phase<float>(phase, res, hs, p, t_start, t_probe_out, probe_dtype, c);
  else
    initial_step_phase<double>(phase, res, hs, p, t_start, t_probe_out, probe_dtype, c);
}


// The initial-step heuristic for SMALL states as one workgroup per phase (round 4): the adjoint's backward pass starts a solve per
// output interval, and the heuristic's 3 norm passes + 3 finalize + 3 result + 2 scalar launches + ctrl_init were 12 of that solve's
// launches on a 33 021-element state.  Phase 0: norm(y0/scale) and norm(f0/scale) in ONE pass (they share `scale`), then the phase-0
// scalars (h0 -> ctrl->dt, t0 + h0).  Phase 1: norm((f1 - f0)/scale), the phase-1 scalars, and the control block's construction
// (xde_ctrl_init's arithmetic).  The Euler probe between them stays an xde_stage_combine that reads dt from ctrl.
struct InitArgs {
  const void* a;   // phase 0: f0; phase 1: f1
  const void* b;   // phase 1: f0
  const void* y0;
  SegMap map;
  double rtol, atol;
  double* hs;
  double t_start;
  void* t_probe_out;
  int probe_dtype;
  int phase;
  int32_t n_out;
  const double* t_span;
  const double* step_t;
  void* t_stage_out;
  int64_t seq0;
};

template <typename T, int NORM, bool VEC>
__global__ __launch_bounds__(kSingleBlock) void xde_initial_step_single_kernel(InitArgs g, xde_ctrl_params_t p, xde_ctrl_t* c) {
  using P = Pack<T, VEC>;
  constexpr int W = P::W;
  __shared__ double w_val[2][kSingleWaves][XDE_MAX_SEG];
  __shared__ double seg_val[2][XDE_MAX_SEG];
  __shared__ xde_ctrl_t z;
  const T* __restrict__ a = static_cast<const T*>(g.a);
  const T* __restrict__ b = static_cast<const T*>(g.b);
  const T* __restrict__ y0 = static_cast<const T*>(g.y0);
  const T rtol = T(g.rtol), atol = T(g.atol);
  const bool diff = g.phase == 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int seg = 0; seg < g.map.n_seg; ++seg) {
    const int64_t start = g.map.seg_start[seg], len = g.map.seg_len[seg];
    const int64_t nvec = len / W, vbase = start / W;
    T acc0 = T(0), acc1 = T(0);
    auto one = [&](T av, T bv, T yv) {
      const T scale = atol + abs_(yv) * rtol;
      // phase 0: r0 = |y0 / scale|, r1 = |f0 / scale|;  phase 1: r0 = |(f1 - f0) / scale|
      const T r0 = abs_((diff ? (av - bv) : yv) / scale);
      const T r1 = diff ? T(0) : abs_(av / scale);
      if (NORM == XDE_NORM_RMS) {
        acc0 = acc0 + r0 * r0;
        acc1 = acc1 + r1 * r1;
      } else {
        acc0 = (r0 != r0 || acc0 != acc0) ? (r0 != r0 ? r0 : acc0) : (r0 > acc0 ? r0 : acc0);
        acc1 = (r1 != r1 || acc1 != acc1) ? (r1 != r1 ? r1 : acc1) : (r1 > acc1 ? r1 : acc1);
      }
    };
    for (int64_t i = threadIdx.x; i < nvec; i += kSingleBlock) {  // (a state this kernel serves is <= 16 such iterations: no fp64 flush needed)
      P av = P::load(a, vbase + i);
      P yv = P::load(y0, vbase + i);
      P bv = av;
      if (diff) bv = P::load(b, vbase + i);
#pragma unroll
      for (int w = 0; w < W; ++w) one(av.v[w], bv.v[w], yv.v[w]);
    }
    if (VEC) {
      const int64_t i = start + nvec * W + threadIdx.x;
      if (i < start + len) one(a[i], diff ? b[i] : a[i], y0[i]);
    }
    double v0 = double(acc0), v1 = double(acc1);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      v0 = merge_<NORM>(v0, __shfl_down(v0, off, 64));
      v1 = merge_<NORM>(v1, __shfl_down(v1, off, 64));
    }
    if (lane == 0) {
      w_val[0][wave][seg] = v0;
      w_val[1][wave][seg] = v1;
    }
  }
  __syncthreads();
  if (int(threadIdx.x) < 2 * g.map.n_seg) {
    const int which = threadIdx.x / g.map.n_seg, seg = threadIdx.x % g.map.n_seg;
    double v = w_val[which][0][seg];
#pragma unroll
    for (int w = 1; w < kSingleWaves; ++w) v = merge_<NORM>(v, w_val[which][w][seg]);
    seg_val[which][seg] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double res[2];
  res[0] = norm_from_sums(seg_val[0], p.seg_count, p.n_seg, NORM, p.state_dtype, nullptr);
  res[1] = diff ? 0.0 : norm_from_sums(seg_val[1], p.seg_count, p.n_seg, NORM, p.state_dtype, nullptr);
  // t_start = NaN: the start time is t_span_dev[0] (a launch recorded in a graph serves every interval it is replayed for)
  const double t_start = (g.t_start != g.t_start && g.t_span) ? g.t_span[0] : g.t_start;
  if (p.state_dtype == XDE_F32)
    initial_step_phase<float>(g.phase, res, g.hs, p, t_start, g.t_probe_out, g.probe_dtype, c);
  else
    initial_step_phase<double>(g.phase, res, g.hs, p, t_start, g.t_probe_out, g.probe_dtype, c);
  if (diff) {
    g.hs[4] = res[0];  // (the third norm, for the parity tests)
    ctrl_init_body(c, p, t_start, double(p.direction) * fabs(g.hs[3]), g.n_out, g.t_span, g.step_t, g.t_stage_out, g.seq0, z);
  }
}

// The same heuristic for states ABOVE the one-workgroup kernels' reach (round 6): the norm passes stay multi-workgroup launches
// (xde_scaled_norm2_partial for phase 0's two norms in one pass, xde_scaled_norm_partial for phase 1's), and everything that came after
// each of them as a launch of its own — xde_norm_finalize, xde_norm_result, xde_initial_step, and in phase 1 xde_ctrl_init — is ONE
// one-workgroup launch per phase: 12 launches -> 4.  Same fixed-order reduction (reduce_partials), same norm_from_sums, same scalar
// phases, same ctrl_init_body: same bits as the separate launches.
struct InitTailArgs {
  const NormSlot* slot_a;  // phase 0: partials of norm(y0/scale); phase 1: of norm((f1 - f0)/scale)
  const NormSlot* slot_b;  // phase 0: partials of norm(f0/scale)
  double* hs;
  double t_start;
  void* t_probe_out;
  int probe_dtype;
  int phase;
  int32_t n_out;
  const double* t_span;
  const double* step_t;
  void* t_stage_out;
  int64_t seq0;
  xde_ctrl_t* mirror;  // phase 1: the constructed block is also published to the host mirror (slot seq0 % SLOTS), or NULL
  int flags;
};

__global__ __launch_bounds__(kBlock) void xde_initial_step_tail_kernel(InitTailArgs g, xde_ctrl_params_t p, xde_ctrl_t* c) {
  __shared__ double seg_val[2][XDE_MAX_SEG];
  __shared__ double seg_nf[2][XDE_MAX_SEG];
  __shared__ xde_ctrl_t z;
  if (threadIdx.x < XDE_MAX_SEG) {
    seg_val[0][threadIdx.x] = seg_val[1][threadIdx.x] = 0.0;
    seg_nf[0][threadIdx.x] = seg_nf[1][threadIdx.x] = 0.0;
  }
  __syncthreads();
  const bool diff = g.phase == 1;
  reduce_partials(g.slot_a, seg_val[0], seg_nf[0]);
  if (!diff) reduce_partials(g.slot_b, seg_val[1], seg_nf[1]);
  if (threadIdx.x == 0) {
    double res[2];
    res[0] = norm_from_sums(seg_val[0], p.seg_count, p.n_seg, p.norm_kind, p.state_dtype, nullptr);
    res[1] = diff ? 0.0 : norm_from_sums(seg_val[1], p.seg_count, p.n_seg, p.norm_kind, p.state_dtype, nullptr);
    // t_start = NaN: the start time is t_span_dev[0] (a launch recorded in a graph serves every interval it is replayed for)
    const double t_start = (g.t_start != g.t_start && g.t_span) ? g.t_span[0] : g.t_start;
    if (p.state_dtype == XDE_F32)
      initial_step_phase<float>(g.phase, res, g.hs, p, t_start, g.t_probe_out, g.probe_dtype, c);
    else
      initial_step_phase<double>(g.phase, res, g.hs, p, t_start, g.t_probe_out, g.probe_dtype, c);
    if (diff) {
      g.hs[4] = res[0];  // (the third norm, for the parity tests)
      ctrl_init_body(c, p, t_start, double(p.direction) * fabs(g.hs[3]), g.n_out, g.t_span, g.step_t, g.t_stage_out, g.seq0, z);
    }
  }
  if (!diff || !g.mirror) return;  // (block-uniform)
  // The speculative pipeline wants to know where the FIRST attempt lands (`t_plan` of the block just constructed) before it
  // enqueues a second one: the block goes to the host mirror the way every controller launch's does — no copy command, no event
  __syncthreads();
  publish_block(c, z, g.mirror, g.flags | kCtrlChecksum);
}

}  // namespace

extern "C" {

int xde_error_norm_control(const void* const* k, const void* k0_alt, const double* c_err, int nk, const void* y0,
                           const void* y0_alt, const void* y1, const xde_segments_t* segs, int dtype, void* ws,
                           const void* e_pre, xde_ctrl_t* ctrl, const xde_ctrl_params_t* params, const double* t_span_dev,
                           const double* step_t_dev, void* t_stage_out, xde_ctrl_t* host_mirror, void* stream) {
  if (!ctrl || !t_span_dev || !t_stage_out) return fail(XDE_EBADARG, "xde_error_norm_control: null pointer");
  int rc = check_params(params, "xde_error_norm_control");
  if (rc != XDE_OK) return rc;
  if (params->n_step_t > 0 && !step_t_dev) return fail(XDE_EBADARG, "xde_error_norm_control: n_step_t > 0 without step_t_dev");
  if (int rc0 = check_segments(segs)) return rc0;
  if (segs->n_seg != params->n_seg) return fail(XDE_EBADARG, "xde_error_norm_control: segments do not match params->n_seg");
  ErrArgs a;
  bool vec = false;
  int nblocks = 0;
  double bytes = 0;
  rc = setup_err_args("xde_error_norm_control", k, k0_alt, c_err, nk, y0, y0_alt, y1, params->rtol, params->atol, 0.0, ctrl, segs,
                      params->norm_kind, dtype, ws, e_pre, &a, &vec, &nblocks, &bytes, norm_grid_cap());
  if (rc != XDE_OK) return rc;
  CtrlTail tl;
  tl.ctrl = ctrl;
  tl.p = *params;
  tl.t_span = t_span_dev;
  tl.step_t = step_t_dev;
  tl.t_stage_out = t_stage_out;
  tl.mirror = host_mirror;
  tl.flags = ctrl_flags();
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_ERRNORM, bytes);
  {
    // small state: one workgroup does everything (65536 elements or fewer: configs 3 and 5, launch-latency-bound)
    constexpr int64_t single_max = int64_t(1) << 16;
    int64_t total = 0;
    for (int s2 = 0; s2 < segs->n_seg; ++s2) total += segs->seg_len[s2];
    if (total <= single_max) {
      dim3 g1(1), b1(kSingleBlock);
      const int fl = ctrl_flags();
#define LAUNCH_ES(T, NORM)                                                                           \
  do {                                                                                               \
    if (vec)                                                                                         \
      XDE_LAUNCH((xde_errnorm_control_single_kernel<T, NORM, true>), g1, b1, st, prof, a, tl, fl);   \
    else                                                                                             \
      XDE_LAUNCH((xde_errnorm_control_single_kernel<T, NORM, false>), g1, b1, st, prof, a, tl, fl);  \
  } while (0)
      if (dtype == XDE_F32) {
        if (params->norm_kind == XDE_NORM_RMS) LAUNCH_ES(float, XDE_NORM_RMS);
        else LAUNCH_ES(float, XDE_NORM_LINF);
      } else {
        if (params->norm_kind == XDE_NORM_RMS) LAUNCH_ES(double, XDE_NORM_RMS);
        else LAUNCH_ES(double, XDE_NORM_LINF);
      }
#undef LAUNCH_ES
      HIP_TRY(hipGetLastError());
      return XDE_OK;
    }
  }
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
  int cap = norm_grid_cap() + XDE_MAX_SEG;  // the largest grid a norm launch uses (build_segmap: the cap + one block per segment)
  if (cap > XDE_MAX_PARTIALS) cap = XDE_MAX_PARTIALS;
  XDE_LAUNCH(xde_control_kernel, dim3(1), dim3(kBlock), st, prof, ctrl, *params, ws ? slot_ptr(ws, 0) : nullptr, sums,
             t_span_dev, step_t_dev, t_stage_out, host_mirror, ctrl_flags(), cap);
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

int xde_ctrl_retarget(xde_ctrl_t* ctrl, const xde_ctrl_params_t* params, const double* t_span_dev, int32_t n_out,
                      xde_ctrl_t* host_mirror, void* stream) {
  if (!ctrl || !t_span_dev) return fail(XDE_EBADARG, "xde_ctrl_retarget: null pointer");
  int rc = check_params(params, "xde_ctrl_retarget");
  if (rc != XDE_OK) return rc;
  if (n_out < 1) return fail(XDE_EBADARG, "xde_ctrl_retarget: n_out must be >= 1");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(xde_ctrl_retarget_kernel, dim3(1), dim3(64), 0, st, ctrl, *params, t_span_dev, n_out, host_mirror, ctrl_flags());
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_initial_step(int phase, const double* res_dev, double* hs_dev, const xde_ctrl_params_t* params, double t_start,
                     void* t_probe_out, int probe_dtype, xde_ctrl_t* ctrl, void* stream) {
  if (!res_dev || !hs_dev || !ctrl) return fail(XDE_EBADARG, "xde_initial_step: null pointer");
  if (phase != 0 && phase != 1 && phase != 2) return fail(XDE_EBADARG, "xde_initial_step: phase must be 0, 1 or 2");
  if (phase != 1 && !t_probe_out) return fail(XDE_EBADARG, "xde_initial_step: phase 0 needs t_probe_out");
  if (probe_dtype != XDE_F32 && probe_dtype != XDE_F64) return fail(XDE_EBADARG, "xde_initial_step: bad probe dtype");
  int rc = check_params(params, "xde_initial_step");
  if (rc != XDE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(xde_initial_step_kernel, dim3(1), dim3(64), 0, st, phase, res_dev, hs_dev, *params, t_start, t_probe_out,
                     probe_dtype, ctrl);
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_initial_step_fused(int phase, const void* a, const void* b, const void* y0, const xde_segments_t* segs, int dtype, double* hs_dev,
                           const xde_ctrl_params_t* params, double t_start, void* t_probe_out, int probe_dtype, xde_ctrl_t* ctrl,
                           int32_t n_out, const double* t_span_dev, const double* step_t_dev, void* t_stage_out, int64_t seq0,
                           void* stream) {
  if (!a || !y0 || !hs_dev || !ctrl) return fail(XDE_EBADARG, "xde_initial_step_fused: null pointer");
  if (phase != 0 && phase != 1) return fail(XDE_EBADARG, "xde_initial_step_fused: phase must be 0 or 1");
  if (phase == 0 && !t_probe_out) return fail(XDE_EBADARG, "xde_initial_step_fused: phase 0 needs t_probe_out");
  if (phase == 1 && (!b || !t_span_dev || !t_stage_out)) return fail(XDE_EBADARG, "xde_initial_step_fused: phase 1 needs b, t_span_dev and t_stage_out");
  if (phase == 1 && n_out < 1) return fail(XDE_EBADARG, "xde_initial_step_fused: n_out must be >= 1");
  if (probe_dtype != XDE_F32 && probe_dtype != XDE_F64) return fail(XDE_EBADARG, "xde_initial_step_fused: bad probe dtype");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_initial_step_fused: bad dtype");
  int rc = check_params(params, "xde_initial_step_fused");
  if (rc != XDE_OK) return rc;
  if (int rc0 = check_segments(segs)) return rc0;
  if (segs->n_seg != params->n_seg) return fail(XDE_EBADARG, "xde_initial_step_fused: segments do not match params->n_seg");
  if (dtype != params->state_dtype) return fail(XDE_EBADARG, "xde_initial_step_fused: dtype does not match params->state_dtype");
  if (phase == 1 && params->n_step_t > 0 && !step_t_dev) return fail(XDE_EBADARG, "xde_initial_step_fused: n_step_t > 0 without step_t_dev");
  int64_t total = 0;
  for (int s2 = 0; s2 < segs->n_seg; ++s2) total += segs->seg_len[s2];
  if (total > (int64_t(1) << 20)) return fail(XDE_EBADARG, "xde_initial_step_fused: one workgroup serves small states only (<= 2^20 elements)");
  InitArgs g;
  memset(&g, 0, sizeof(g));
  g.a = a;
  g.b = b;
  g.y0 = y0;
  g.rtol = params->rtol;
  g.atol = params->atol;
  g.hs = hs_dev;
  g.t_start = t_start;
  g.t_probe_out = t_probe_out;
  g.probe_dtype = probe_dtype;
  g.phase = phase;
  g.n_out = n_out;
  g.t_span = t_span_dev;
  g.step_t = step_t_dev;
  g.t_stage_out = t_stage_out;
  g.seq0 = seq0;
  const int width = dtype == XDE_F32 ? 4 : 2;
  const bool vec = aligned16(a) && aligned16(y0) && (!b || aligned16(b)) && segs_vec_ok(segs, width);
  int nblocks = 0;
  rc = build_segmap(segs, vec ? width : 1, vec, &g.map, &nblocks);
  if (rc != XDE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_SCALEDNORM, (phase == 0 ? 2.0 : 3.0) * double(total) * (dtype == XDE_F32 ? 4.0 : 8.0));
  dim3 g1(1), b1(kSingleBlock);
#define LAUNCH_IS(T, NORM)                                                                              \
  do {                                                                                                  \
    if (vec)                                                                                            \
      XDE_LAUNCH((xde_initial_step_single_kernel<T, NORM, true>), g1, b1, st, prof, g, *params, ctrl);   \
    else                                                                                                \
      XDE_LAUNCH((xde_initial_step_single_kernel<T, NORM, false>), g1, b1, st, prof, g, *params, ctrl);  \
  } while (0)
  if (dtype == XDE_F32) {
    if (params->norm_kind == XDE_NORM_RMS) LAUNCH_IS(float, XDE_NORM_RMS);
    else LAUNCH_IS(float, XDE_NORM_LINF);
  } else {
    if (params->norm_kind == XDE_NORM_RMS) LAUNCH_IS(double, XDE_NORM_RMS);
    else LAUNCH_IS(double, XDE_NORM_LINF);
  }
#undef LAUNCH_IS
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

int xde_initial_step_tail(int phase, const void* ws, double* hs_dev, const xde_ctrl_params_t* params, double t_start, void* t_probe_out,
                          int probe_dtype, xde_ctrl_t* ctrl, int32_t n_out, const double* t_span_dev, const double* step_t_dev,
                          void* t_stage_out, int64_t seq0, xde_ctrl_t* host_mirror, void* stream) {
  if (!ws || !hs_dev || !ctrl) return fail(XDE_EBADARG, "xde_initial_step_tail: null pointer");
  if (phase != 0 && phase != 1) return fail(XDE_EBADARG, "xde_initial_step_tail: phase must be 0 or 1");
  if (phase == 0 && !t_probe_out) return fail(XDE_EBADARG, "xde_initial_step_tail: phase 0 needs t_probe_out");
  if (phase == 1 && (!t_span_dev || !t_stage_out)) return fail(XDE_EBADARG, "xde_initial_step_tail: phase 1 needs t_span_dev and t_stage_out");
  if (phase == 1 && n_out < 1) return fail(XDE_EBADARG, "xde_initial_step_tail: n_out must be >= 1");
  if (t_start != t_start && !t_span_dev) return fail(XDE_EBADARG, "xde_initial_step_tail: t_start = NaN needs t_span_dev");
  if (probe_dtype != XDE_F32 && probe_dtype != XDE_F64) return fail(XDE_EBADARG, "xde_initial_step_tail: bad probe dtype");
  int rc = check_params(params, "xde_initial_step_tail");
  if (rc != XDE_OK) return rc;
  if (phase == 1 && params->n_step_t > 0 && !step_t_dev) return fail(XDE_EBADARG, "xde_initial_step_tail: n_step_t > 0 without step_t_dev");
  if (host_mirror && (phase != 1 || seq0 < 0)) return fail(XDE_EBADARG, "xde_initial_step_tail: host_mirror goes with phase 1 and seq0 >= 0");
  if (host_mirror && !(ctrl_flags() & kCtrlChecksum))
    return fail(XDE_EBADARG, "xde_initial_step_tail: host_mirror needs the checksummed publish protocol (XDE_CTRL_FLAGS bit 8)");
  InitTailArgs g;
  memset(&g, 0, sizeof(g));
  g.slot_a = slot_ptr(ws, 0);
  g.slot_b = slot_ptr(ws, 1);
  g.hs = hs_dev;
  g.t_start = t_start;
  g.t_probe_out = t_probe_out;
  g.probe_dtype = probe_dtype;
  g.phase = phase;
  g.n_out = n_out;
  g.t_span = t_span_dev;
  g.step_t = step_t_dev;
  g.t_stage_out = t_stage_out;
  g.seq0 = seq0;
  g.mirror = host_mirror;
  g.flags = ctrl_flags();
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope prof(XDE_KID_FINALIZE, 0.0);
  XDE_LAUNCH(xde_initial_step_tail_kernel, dim3(1), dim3(kBlock), st, prof, g, *params, ctrl);
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

// whether a copy of a mirror slot is one whole published block (CHECKSUMMED protocol, publish_block)
static bool ctrl_copy_holds(const xde_ctrl_t& z) {
  const uint64_t* w = reinterpret_cast<const uint64_t*>(&z);
  uint64_t sum = 0;
  for (int i = 0; i < kCtrlWords; ++i) sum += ctrl_chk_term(w[i], i);
  return sum == z.chk;
}

int xde_ctrl_wait(const xde_ctrl_t* host_mirror, int64_t seq, double timeout_ms, xde_ctrl_t* host_out) {
  if (!host_mirror || !host_out || seq < 0) return fail(XDE_EBADARG, "xde_ctrl_wait: bad argument");
  const xde_ctrl_t* slot = host_mirror + (seq % XDE_MIRROR_SLOTS);
  const bool checksummed = (ctrl_flags() & kCtrlChecksum) != 0;
  const auto t_begin = std::chrono::steady_clock::now();
  uint64_t spins = 0;
  bool slow = false;
  for (;;) {
    int64_t cur = __atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE);
    if (cur == seq) {
      if (!checksummed) break;
      // the block's words arrive in no particular order: the copy counts only if it is one whole block (else: words still landing)
      // (into a local block first: the caller's host_out is written only with a copy that holds — after a timeout or a refusal it
      //  still has what the caller left in it, include/xde_hip.h: "such a copy is never handed back")
      xde_ctrl_t cand;
      memcpy(&cand, slot, sizeof(xde_ctrl_t));
      if (cand.seq == seq && ctrl_copy_holds(cand)) {
        memcpy(host_out, &cand, sizeof(xde_ctrl_t));
        return XDE_OK;
      }
    } else if (cur > seq && (!checksummed || cur - seq >= XDE_MIRROR_SLOTS)) {
      // (checksummed: a sequence number ahead of ours can only be a LATER launch's word in this slot, i.e. seq + k * SLOTS)
      return fail(XDE_EBADARG, "xde_ctrl_wait: mirror slot already overwritten by a later launch");
    }
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
  xde_ctrl_t cand;
  memcpy(&cand, slot, sizeof(xde_ctrl_t));
  // the slot may have been overwritten while copying (only if the host lags >= SLOTS launches behind)
  if (__atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE) != seq)
    return fail(XDE_EBADARG, "xde_ctrl_wait: mirror slot overwritten while reading");
  memcpy(host_out, &cand, sizeof(xde_ctrl_t));
  return XDE_OK;
}


}  // extern "C"
