// Device body of the step controller (accept/reject, optimal step size, stage times, output bookkeeping, mirror
// publish), shared by xde_control_kernel and the opt-in one-launch error-norm + controller kernel.
#pragma once
#include "xde_common.hpp"
#include "xde_reduce.hpp"

namespace xde {

// ------------------------------------------------------------------------------------------
// K3: controller
// ------------------------------------------------------------------------------------------
template <typename TT> __device__ inline TT pow_(TT a, TT b);
// fp32: a double-precision pow rounded once — the correctly rounded fp32 power in all but ~1e-9 of the cases, i.e. what the host's libm
// powf returns (the reference's CPU path: `error_ratio ** exponent` on fp32 tensors, utils/ode_utils.py:96) except where libm itself
// misrounds.  The device's own powf is 1-2 ulp off; one thread calls this once per attempt, so the cost is nil.
template <> __device__ inline float pow_<float>(float a, float b) { return float(pow(double(a), double(b))); }
template <> __device__ inline double pow_<double>(double a, double b) { return pow(a, b); }
template <typename TT> __device__ inline TT fmin__(TT a, TT b);
template <> __device__ inline float fmin__<float>(float a, float b) { return fminf(a, b); }
template <> __device__ inline double fmin__<double>(double a, double b) { return fmin(a, b); }
template <typename TT> __device__ inline TT fmax__(TT a, TT b);
template <> __device__ inline float fmax__<float>(float a, float b) { return fmaxf(a, b); }
template <> __device__ inline double fmax__<double>(double a, double b) { return fmax(a, b); }

// The two time-table entries the controller is about to need, fetched by spare lanes while the partials are being reduced
// (each is otherwise a dependent global load on lane 0's critical path).
struct TimePrefetch {
  int out_idx;      // t_span[out_idx], t_span[out_idx + 1]
  double out_t[2];
  int step_idx;     // step_t[step_idx], step_t[step_idx + 1]
  double step_v[2];
};

__device__ __forceinline__ double table_at(const double* table, int i, const TimePrefetch* pf, bool is_span) {
  if (pf) {
    const int base = is_span ? pf->out_idx : pf->step_idx;
    if (i == base) return is_span ? pf->out_t[0] : pf->step_v[0];
    if (i == base + 1) return is_span ? pf->out_t[1] : pf->step_v[1];
  }
  return table[i];
}

// Plan the pending attempt: step_t clipping (base_adaptive_solver_rk.py:209-215), the underflow and
// max_num_steps assertions (:200, :120-122) and the stage times of _runge_kutta_step (:159-164).
template <typename TT>
__device__ __forceinline__ void plan_next(xde_ctrl_t* c, const xde_ctrl_params_t& p, const double* step_t, void* t_stage_out,
                          const TimePrefetch* pf = nullptr) {
  const TT dir = TT(p.direction);
  TT t0 = TT(c->t1);
  TT dt = TT(c->dt);
  TT t1 = t0 + dt;
  int on = 0;
  if (p.n_step_t > 0 && step_t) {
    TT nxt = TT(table_at(step_t, c->next_step_index, pf, false));
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
__device__ __forceinline__ void control_step(xde_ctrl_t* c, const xde_ctrl_params_t& p, double ratio_d, double nonfinite,
                             const double* t_span, const double* step_t, void* t_stage_out, const TimePrefetch* pf = nullptr) {
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
  // prescribed step sequence (xde_ctrl_params_t.replay): this attempt's verdict and the next attempt's step come from
  // the table; the error ratio above is still what the kernels measured
  if (p.replay && c->n_steps < p.n_replay) {
    const int64_t i = c->n_steps;
    accept = p.replay[2 * i + 1] != 0.0 ? 1 : 0;
    if (i + 1 < p.n_replay) dt_next = TT(p.replay[2 * (i + 1)]);
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
    while (e < c->n_out && dir * TT(table_at(t_span, e, pf, true)) <= dir * t1) ++e;
  }
  c->out_begin = b;
  c->out_end = e;
  c->next_out = e;
  if (e > b) c->steps_in_interval = 0;
  c->done = (e >= c->n_out) ? 1 : 0;

  if (accept && c->on_step_t && c->next_step_index != p.n_step_t - 1) c->next_step_index += 1;  // :263-265

  plan_next<TT>(c, p, step_t, t_stage_out, pf);
}

// The pinned host mirror of a control block is a ring of XDE_MIRROR_SLOTS copies (slot[seq % SLOTS]).  Two publish protocols
// (XDE_CTRL_FLAGS bit 8 selects; both leave the same block in device memory and in the slot):
//
//   CHECKSUMMED (default, round 5): the block carries a 64-bit checksum of its other words (`chk`, salted per position).  ONE wave
//   stores all 36 words — sequence number and checksum included — to the slot in one instruction, with no ordering among them and no
//   wait: the launch's only PCIe round trip is the one its end waits for anyway.  The reader (xde_ctrl_wait) polls `seq`, copies the
//   slot and accepts the copy only if its sequence number is the expected one AND its checksum holds — a copy taken while words were
//   still landing (a mix of this block and the one published 16 launches earlier) fails the check and is taken again.
//
//   SEQLOCK (rounds 2-4): `seq` is invalidated FIRST (invalidate_slot — by one lane of the publishing wave, as early as the launch
//   knows its sequence number), the payload words follow, the new seq is stored last, each group acknowledged before the next is issued
//   (`s_waitcnt vmcnt(0)` within the one publishing wave; full system-scope fences without bit 1): three dependent PCIe writes, 0.9-1.2
//   us of the controller's 7, and — in the sharded attempt's one-launch controller — a host write in flight under the mailbox
//   exchange's system-scope fences, which then wait for it: 3.7 us of its 9.6 (profiles/r05_ctrl_decomposition.txt).
__device__ inline void invalidate_slot(xde_ctrl_t* mirror, int64_t seq_next) {
  __hip_atomic_store(&mirror[seq_next % XDE_MIRROR_SLOTS].seq, int64_t(-1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// XDE_CTRL_FLAGS bits (kernel argument `flags`; all variants publish the same block):
//   1  SEQLOCK protocol only: light ordering of the mirror stores (`s_waitcnt` instead of system-scope release fences)
//   2  header and partial records in one memory round trip (reduce_partials_speculative)
//   4  the next output time / forced step time prefetched by spare lanes (TimePrefetch)
//   8  CHECKSUMMED publish (host side: xde_ctrl_wait validates instead of trusting the order of arrival)
constexpr int kCtrlLightPublish = 1, kCtrlSpecPartials = 2, kCtrlPrefetchTimes = 4, kCtrlChecksum = 8;

// splitmix64's finaliser: every input bit reaches every output bit
__host__ __device__ inline uint64_t ctrl_mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
constexpr int kCtrlWords = sizeof(xde_ctrl_t) / 8;
constexpr int kCtrlSeqWord = offsetof(xde_ctrl_t, seq) / 8;
constexpr int kCtrlChkWord = offsetof(xde_ctrl_t, chk) / 8;
// word i's contribution to the checksum (position-salted; the checksum word itself does not take part): a SUM of contributions, so
// that one wave can form it with a butterfly and the host with a loop, in any order
__host__ __device__ inline uint64_t ctrl_chk_term(uint64_t word, int i) {
  return i == kCtrlChkWord ? 0ull : ctrl_mix64(word + 0x9E3779B97F4A7C15ull * uint64_t(i + 1));
}

// Write a finished control block back: one wave instruction to the device block, one to the mirror slot.  Called by all threads of
// the block after `zs` is complete and visible in LDS.  `pub_wave`: the wave that publishes (SEQLOCK: it must be the wave that issued
// the invalidation in control_prologue — the light ordering is program order + `s_waitcnt` WITHIN one wave).
__device__ inline void publish_block(xde_ctrl_t* c, const xde_ctrl_t& zs, xde_ctrl_t* mirror, int flags = 0, int pub_wave = 0) {
  constexpr int kWords = kCtrlWords;
  constexpr int kSeqWord = kCtrlSeqWord;
  static_assert(kWords <= 64, "the control block is published by ONE wave");
  xde_ctrl_t* ms = mirror ? mirror + (zs.seq % XDE_MIRROR_SLOTS) : nullptr;
  const int lane = int(threadIdx.x) - 64 * pub_wave;
  if (lane < 0 || lane >= 64) return;
  if (flags & kCtrlChecksum) {
    uint64_t word = lane < kWords ? reinterpret_cast<const uint64_t*>(&zs)[lane] : 0ull;
    uint64_t sum = lane < kWords ? ctrl_chk_term(word, lane) : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += (unsigned long long)__shfl_xor((long long)sum, off, 64);
    if (lane == kCtrlChkWord) word = sum;
    if (lane < kWords) {
      reinterpret_cast<uint64_t*>(c)[lane] = word;
      if (mirror) __hip_atomic_store(reinterpret_cast<uint64_t*>(ms) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const bool light = (flags & kCtrlLightPublish) != 0;
  if (mirror) {  // the invalidation (issued by this wave long ago) is ordered before the payload stores
    if (light) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else __threadfence_system();
  }
  if (lane < kWords) {
    const uint64_t word = reinterpret_cast<const uint64_t*>(&zs)[lane];
    reinterpret_cast<uint64_t*>(c)[lane] = word;
    if (mirror && lane != kSeqWord)
      __hip_atomic_store(reinterpret_cast<uint64_t*>(ms) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (mirror) {  // the wave that wrote the words waits for them to be acknowledged, then publishes seq
    if (light) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&ms->seq, zs.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      __threadfence_system();
      if (lane == 0) __hip_atomic_store(&ms->seq, zs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The controller workgroup, in three parts so that the single-workgroup error-norm + controller launch can put its own
// reduction in the middle:
//   control_prologue  the first lanes fetch the control block into LDS, invalidate the mirror slot, prefetch the time tables;
//   (reduction)       per-segment sums into seg_val / seg_nf (LDS), ending in a __syncthreads;
//   control_tail      lane 0 runs the controller on the LDS block in place, one wave publishes it.
__device__ __forceinline__ void control_prologue(const xde_ctrl_t* c, const xde_ctrl_params_t& p, const double* t_span,
                                                 const double* step_t, xde_ctrl_t* mirror, int flags, xde_ctrl_t* zs,
                                                 TimePrefetch* pfs, int pub_wave = 0) {
  constexpr int kWords = sizeof(xde_ctrl_t) / 8;
  constexpr int kSeqWord = offsetof(xde_ctrl_t, seq) / 8;
  const bool prefetch = (flags & kCtrlPrefetchTimes) != 0;
  if (prefetch && threadIdx.x == 64) {  // a lane of wave 1: two dependent loads, off lane 0's critical path
    const int i = c->next_out;
    pfs->out_idx = i;
    pfs->out_t[0] = i < c->n_out ? t_span[i] : 0.0;
    pfs->out_t[1] = i + 1 < c->n_out ? t_span[i + 1] : 0.0;
  }
  if (prefetch && threadIdx.x == 128) {
    const int i = c->next_step_index;
    pfs->step_idx = i;
    const bool has = p.n_step_t > 0 && step_t;
    pfs->step_v[0] = has && i < p.n_step_t ? step_t[i] : 0.0;
    pfs->step_v[1] = has && i + 1 < p.n_step_t ? step_t[i + 1] : 0.0;
  }
  // the control block is fetched by the first lanes while the reduction runs (it is written only by controller launches,
  // i.e. before this launch started)
  if (threadIdx.x < kWords) {
    const uint64_t word = reinterpret_cast<const uint64_t*>(c)[threadIdx.x];
    reinterpret_cast<uint64_t*>(zs)[threadIdx.x] = word;
    if (mirror && !(flags & kCtrlChecksum) && pub_wave == 0 && threadIdx.x == kSeqWord) invalidate_slot(mirror, int64_t(word) + 1);
  }
  // (a publishing wave other than wave 0 fetches the sequence number for itself: the invalidation must come from the wave that
  //  publishes, see publish_block)
  if (mirror && !(flags & kCtrlChecksum) && pub_wave != 0 && int(threadIdx.x) == 64 * pub_wave)
    invalidate_slot(mirror, reinterpret_cast<const int64_t*>(c)[kSeqWord] + 1);
}

// Call after a __syncthreads that made zs / pfs / seg_val / seg_nf visible.
__device__ __forceinline__ void control_tail(xde_ctrl_t* c, const xde_ctrl_params_t& p, const double* seg_val, const double* seg_nf,
                                             const double* t_span, const double* step_t, void* t_stage_out, xde_ctrl_t* mirror,
                                             int flags, xde_ctrl_t* zs, const TimePrefetch* pfs, int pub_wave = 0) {
  if (threadIdx.x == 0) {
    // The controller works on the LDS copy in place.  (A private copy of the 288-byte block — it has arrays indexed at run
    // time — lives in SCRATCH memory: every field access is a memory round trip and, worse, a dispatch that needs scratch
    // costs microseconds more to launch.  These kernels use no scratch.)
    xde_ctrl_t& z = *zs;
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
      const TimePrefetch* pf = (flags & kCtrlPrefetchTimes) ? pfs : nullptr;
      if (p.time_dtype == XDE_F32)
        control_step<float>(&z, p, ratio, nf, t_span, step_t, t_stage_out, pf);
      else
        control_step<double>(&z, p, ratio, nf, t_span, step_t, t_stage_out, pf);
    }
  }
  __syncthreads();
  publish_block(c, *zs, mirror, flags, pub_wave);
}

// FUSED = called by the last workgroup of the ticketed error-norm launch (partials were published write-through inside it).
template <bool FUSED>
__device__ void control_block(xde_ctrl_t* c, const xde_ctrl_params_t& p, const NormSlot* slot, const double* sums,
                              const double* t_span, const double* step_t, void* t_stage_out, xde_ctrl_t* mirror,
                              int nblocks, int norm_kind, int flags = 0, int partial_cap = XDE_MAX_PARTIALS) {
  __shared__ double seg_val[XDE_MAX_SEG];
  __shared__ double seg_nf[XDE_MAX_SEG];
  __shared__ xde_ctrl_t zs;
  __shared__ TimePrefetch pfs;
  control_prologue(c, p, t_span, step_t, mirror, flags, &zs, &pfs);
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
    else if (flags & kCtrlSpecPartials)
      reduce_partials_speculative(slot, seg_val, seg_nf, partial_cap);
    else
      reduce_partials<false>(slot, seg_val, seg_nf);
  }
  control_tail(c, p, seg_val, seg_nf, t_span, step_t, t_stage_out, mirror, flags, &zs, &pfs);
}


// XDE_CTRL_FLAGS: see publish_block (host side: one value per process — kernels and xde_ctrl_wait must agree on the publish protocol)
inline int ctrl_flags() {
  static int v = [] {
    const char* e = getenv("XDE_CTRL_FLAGS");
    return (e && *e) ? atoi(e) : 15;
  }();
  return v;
}

inline int check_params(const xde_ctrl_params_t* p, const char* who) {
  if (!p) return fail(XDE_EBADARG, std::string(who) + ": null params");
  // the caller's statement of the layout comes first: nothing past these two words is read from a struct that does not match
  if (p->struct_size != sizeof(xde_ctrl_params_t) || p->abi_version != XDE_ABI_VERSION)
    return fail(XDE_EBADARG, std::string(who) + ": xde_ctrl_params_t layout mismatch: the caller says struct_size=" + std::to_string(p->struct_size) +
                                 ", abi_version=" + std::to_string(p->abi_version) + "; this library has sizeof=" +
                                 std::to_string(sizeof(xde_ctrl_params_t)) + ", XDE_ABI_VERSION=" + std::to_string(XDE_ABI_VERSION) +
                                 " (rebuild the binding against include/xde_hip.h)");
  if (p->n_stage < 1 || p->n_stage > XDE_MAX_STAGE) return fail(XDE_EBADARG, std::string(who) + ": n_stage out of range");
  if (p->n_seg < 1 || p->n_seg > XDE_MAX_SEG) return fail(XDE_EBADARG, std::string(who) + ": n_seg out of range");
  if (p->direction != 1 && p->direction != -1) return fail(XDE_EBADARG, std::string(who) + ": direction must be +-1");
  if (p->time_dtype != XDE_F32 && p->time_dtype != XDE_F64) return fail(XDE_EBADARG, std::string(who) + ": bad time_dtype");
  if (p->state_dtype != XDE_F32 && p->state_dtype != XDE_F64) return fail(XDE_EBADARG, std::string(who) + ": bad state_dtype");
  if (!(p->order > 0)) return fail(XDE_EBADARG, std::string(who) + ": order must be positive");
  return XDE_OK;
}


}  // namespace xde
