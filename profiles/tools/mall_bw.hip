// Read bandwidth of a streaming sum kernel as a function of the footprint, same buffer re-read back to back: does data that is
// resident in the 256 MiB Infinity Cache stream faster than HBM?  (Decides whether config 4's 16 MiB-operand shard, whose whole
// working set is cache resident, has headroom above the ~6.3 TB/s HBM copy ceiling.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_bw profiles/tools/mall_bw.hip && /tmp/mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void read_sum(const float4* __restrict__ p, size_t nvec, float* out, int nt) {
  float acc = 0.f;
  size_t stride = size_t(gridDim.x) * 256;
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < nvec; i += stride) {
    float4 v;
    if (nt) {
      typedef float v4 __attribute__((ext_vector_type(4)));
      v4 x = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p) + i);
      v = make_float4(x.x, x.y, x.z, x.w);
    } else {
      v = p[i];
    }
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;  // never true: keeps the loads alive
}

__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                             float4* __restrict__ o, size_t nvec) {
  size_t stride = size_t(gridDim.x) * 256;
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < nvec; i += stride) {
    float4 x = a[i], y = b[i], z = c[i];
    o[i] = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
  }
}

int main() {
  const size_t MB = 1 << 20;
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<size_t> sizes = {8, 16, 32, 64, 128, 192, 256, 384, 512, 1024};
  printf("# read-only sum, one buffer re-read 20x back to back\n# MiB  grid  policy  GB/s  us/launch\n");
  for (size_t mb : sizes) {
    float4* p;
    hipMalloc(&p, mb * MB);
    hipMemset(p, 0, mb * MB);
    size_t nvec = mb * MB / 16;
    for (int grid : {1024, 2048, 4096}) {
      for (int nt = 0; nt < 2; ++nt) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(read_sum, dim3(grid), dim3(256), 0, 0, p, nvec, out, nt);
        hipEventRecord(e0);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_sum, dim3(grid), dim3(256), 0, 0, p, nvec, out, nt);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%5zu %5d %s %8.0f %8.2f\n", mb, grid, nt ? "nt " : "def", double(mb * MB) * reps / (ms * 1e-3) / 1e9, 1e3 * ms / reps);
      }
    }
    hipFree(p);
  }
  printf("# 3 reads + 1 write (a 2-operand stage combine's shape), 4 buffers of the given size each, 20x\n# MiB/buffer  GB/s  us/launch\n");
  for (size_t mb : {4, 8, 16, 32, 64, 128}) {
    float4 *a, *b, *c, *o;
    hipMalloc(&a, mb * MB); hipMalloc(&b, mb * MB); hipMalloc(&c, mb * MB); hipMalloc(&o, mb * MB);
    hipMemset(a, 0, mb * MB); hipMemset(b, 0, mb * MB); hipMemset(c, 0, mb * MB);
    size_t nvec = mb * MB / 16;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, a, b, c, o, nvec);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, a, b, c, o, nvec);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%5zu %8.0f %8.2f\n", mb, double(4 * mb * MB) * reps / (ms * 1e-3) / 1e9, 1e3 * ms / reps);
    hipFree(a); hipFree(b); hipFree(c); hipFree(o);
  }
  return 0;
}
