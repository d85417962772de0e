// libxde_hip.so — multi-tensor pack: the members of a tuple state written into ONE flat buffer with 16-byte-aligned segments.
//
// odeint_adjoint's augmented dynamics returns a tuple (vjp_t, f, vjp_y, vjp_theta...) on every evaluation (reference:
// functional/odeint_adjoint.py:89-124); the kernels integrate a flat state (the reference's tuple support — utils/misc.py
// flat_to_shape and the commented-out concat in functional/odeint.py — was removed, its intent is one concatenated buffer).  Packing
// with framework ops is one fill + one copy launch PER MEMBER: 8 launches per evaluation for config 3's state (adj_t, y, adj_y and
// four parameter tensors), 9 evaluations per output interval — launch-bound work on our side of the boundary.  Here: one launch.

#include "xde_common.hpp"

using namespace xde;

namespace {

struct PackArgs {
  const void* src[XDE_MAX_PACK];
  int64_t start[XDE_MAX_PACK + 1];  // element offset of each segment in the flat buffer (multiples of the vector width); [n] = total
  int64_t len[XDE_MAX_PACK];        // elements of each source
  double scale[XDE_MAX_PACK];       // member s is written as src * scale[s]; exactly 1 = a plain copy
  void* out;
  int64_t total;
  int n;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void xde_pack_kernel(PackArgs a) {
  using P = Pack<T, true>;
  constexpr int W = P::W;
  T* __restrict__ out = static_cast<T*>(a.out);
  const int64_t nvec = a.total / W;  // total is a multiple of W (every segment is padded to 16 bytes)
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nvec; i += stride) {
    const int64_t e0 = i * W;
    int s = 0;  // segment of this vector: the last one whose start is <= e0 (a handful of segments: a linear walk)
    while (s + 1 < a.n && a.start[s + 1] <= e0) ++s;
    const T* __restrict__ src = static_cast<const T*>(a.src[s]);
    const int64_t off = e0 - a.start[s];
    P v;
    if (off + W <= a.len[s] && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0)) {
      v = P::load(src, off / W);  // (off is a multiple of W: segment starts are)
    } else {
#pragma unroll
      for (int w = 0; w < W; ++w) v.v[w] = (off + w < a.len[s]) ? src[off + w] : T(0);  // the pad up to the next segment is zero
    }
    if (a.scale[s] != 1.0) {
      const T f = T(a.scale[s]);
#pragma unroll
      for (int w = 0; w < W; ++w) v.v[w] = v.v[w] * f;
    }
    v.store(out, i);
  }
}

}  // namespace

extern "C" {

int xde_pack_segments(void* flat_out, const void* const* srcs, const int64_t* starts, const int64_t* lens, const double* scales,
                      int n_seg, int64_t total, int dtype, void* stream) {
  if (!flat_out || !srcs || !starts || !lens) return fail(XDE_EBADARG, "xde_pack_segments: null pointer");
  if (n_seg < 1 || n_seg > XDE_MAX_PACK) return fail(XDE_EBADARG, "xde_pack_segments: n_seg out of range");
  if (dtype != XDE_F32 && dtype != XDE_F64) return fail(XDE_EBADARG, "xde_pack_segments: bad dtype");
  const int width = dtype == XDE_F32 ? 4 : 2;
  if (total < 0 || total % width) return fail(XDE_EBADARG, "xde_pack_segments: total must be a multiple of the 16-byte vector width");
  if (!aligned16(flat_out)) return fail(XDE_EBADARG, "xde_pack_segments: the flat buffer must be 16-byte aligned");
  if (total == 0) return XDE_OK;
  PackArgs a;
  memset(&a, 0, sizeof(a));
  int64_t prev_end = 0;
  for (int s = 0; s < n_seg; ++s) {
    if (lens[s] < 0 || starts[s] < prev_end || starts[s] % width) return fail(XDE_EBADARG, "xde_pack_segments: segments must be ordered, disjoint and 16-byte aligned");
    if (lens[s] > 0 && !srcs[s]) return fail(XDE_EBADARG, "xde_pack_segments: null source");
    a.src[s] = srcs[s] ? srcs[s] : flat_out;  // (never dereferenced when len == 0)
    a.start[s] = starts[s];
    a.len[s] = lens[s];
    a.scale[s] = scales ? scales[s] : 1.0;
    prev_end = starts[s] + lens[s];
  }
  if (starts[0] != 0) return fail(XDE_EBADARG, "xde_pack_segments: the first segment starts at 0");
  if (prev_end > total) return fail(XDE_EBADARG, "xde_pack_segments: segments exceed total");
  a.start[n_seg] = total;
  a.out = flat_out;
  a.total = total;
  a.n = n_seg;
  int64_t blocks = (total / width + kBlock - 1) / kBlock;
  if (blocks > grid_cap()) blocks = grid_cap();
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
  if (dtype == XDE_F32)
    hipLaunchKernelGGL(xde_pack_kernel<float>, g, b, 0, st, a);
  else
    hipLaunchKernelGGL(xde_pack_kernel<double>, g, b, 0, st, a);
  HIP_TRY(hipGetLastError());
  return XDE_OK;
}

}  // extern "C"
