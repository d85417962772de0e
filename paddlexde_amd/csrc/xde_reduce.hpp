// Workgroup reduction to one partial record, and the fixed-order reduction of all records by one workgroup.
#pragma once
#include "xde_common.hpp"

namespace xde {

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
// fixed-order reduction of block partials → per-segment sums (one workgroup)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void reduce_records(const double* rv, const double* rf, const int* rs, int n_seg, bool rms,
                                               double* seg_val, double* seg_nf);

template <bool SC1 = false>
__device__ void reduce_partials(const NormSlot* slot, double* seg_val, double* seg_nf, int nblocks_known = -1,
                                int n_seg_known = 0, int norm_kind_known = 0) {
  // Called by all kBlock threads of ONE block; results land in seg_val/seg_nf (shared, XDE_MAX_SEG each).
  // Every thread first pulls its (up to) XDE_MAX_PARTIALS/kBlock records into registers with independent
  // 32-byte loads (one memory round trip), then the per-segment reductions run on registers:
  // strided per-thread order -> wave64 shuffle tree -> 4 wave results summed in order.  Fixed order, so
  // the value is bit-reproducible from launch to launch and identical on every rank.
  constexpr int kPer = XDE_MAX_PARTIALS / kBlock;
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
  reduce_records(rv, rf, rs, n_seg, rms, seg_val, seg_nf);
}

// `speculative` form of the load phase: the header (nblocks) and the records are fetched in ONE memory round trip — every
// lane pulls its records of the whole partial array (it is allocated at full size; stale records are masked afterwards)
// instead of waiting for nblocks first.  Same reduction order, same result.
__device__ __forceinline__ void reduce_partials_speculative(const NormSlot* slot, double* seg_val, double* seg_nf, int cap) {
  constexpr int kPer = XDE_MAX_PARTIALS / kBlock;
  double rv[kPer], rf[kPer];
  int rs[kPer];
  const int per = (cap + kBlock - 1) / kBlock;  // wave-uniform: records beyond the largest grid any launch uses are skipped
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    if (i < per) {
      Partial rec = slot->p[threadIdx.x + i * kBlock];
      rv[i] = rec.val;
      rf[i] = rec.nf;
      rs[i] = rec.seg;
    } else {
      rv[i] = 0.0;
      rf[i] = 0.0;
      rs[i] = -1;
    }
  }
  const int nblocks = slot->nblocks;
  const int n_seg = slot->n_seg;
  const bool rms = slot->norm_kind == XDE_NORM_RMS;
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    if (threadIdx.x + i * kBlock >= nblocks) {
      rv[i] = 0.0;
      rf[i] = 0.0;
      rs[i] = -1;
    }
  }
  reduce_records(rv, rf, rs, n_seg, rms, seg_val, seg_nf);
}

__device__ __forceinline__ void reduce_records(const double* rv, const double* rf, const int* rs, int n_seg, bool rms,
                                               double* seg_val, double* seg_nf) {
  constexpr int kPer = XDE_MAX_PARTIALS / kBlock;
  // Every segment's wave-level results are parked in LDS first; ONE barrier; then lane s finishes segment s.  Round 2 ran the
  // cross-wave step (two barriers) once per segment: 8.6 us for the adjoint's 7-segment state.  The arithmetic order per segment is
  // unchanged (strided per-thread terms -> wave64 shuffle tree -> the waves' results summed in wave order): same bits.
  __shared__ double w_val[kWaves][XDE_MAX_SEG];
  __shared__ double w_nf[kWaves][XDE_MAX_SEG];
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
      w_val[wave][s] = v;
      w_nf[wave][s] = f;
    }
  }
  __syncthreads();
  if (int(threadIdx.x) < n_seg) {
    const int s = threadIdx.x;
    double tv = w_val[0][s], tf = w_nf[0][s];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
      tv = rms ? tv + w_val[w][s] : nanmax_(tv, w_val[w][s]);
      tf += w_nf[w][s];
    }
    seg_val[s] = tv;
    seg_nf[s] = tf;
  }
  __syncthreads();
}


__device__ inline double round_to(double x, int dtype) { return dtype == XDE_F32 ? double(float(x)) : x; }

// norm value from finalised sums: max over segments of sqrt(sum/count) (RMS) or of max (LINF),
// in the state dtype like `_rms_norm` (utils/ode_utils.py:8-9).  NaN propagates.
__device__ inline double norm_from_sums(const double* seg_val, const double* seg_count, int n_seg, int norm_kind,
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

}  // namespace xde
