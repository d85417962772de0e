// Microbenchmark: out = y0 + sum_j k_j*c_j with NK input streams, variants of block->data mapping, unroll, nt policy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int NKMAX = 8;
struct Args { const float* k[NKMAX]; const float* y0; float* out; float c[NKMAX]; long nvec; };

template <int NK, int UNROLL, bool NT, bool CHUNK>
__global__ __launch_bounds__(256) void kern(Args a, int chunk_iters) {
  const v4* y0 = (const v4*)a.y0; v4* out = (v4*)a.out;
  const long stride = (long)gridDim.x * 256;
  auto body = [&](long i) {
    v4 acc = NT ? __builtin_nontemporal_load(y0 + i) : y0[i];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      v4 kk = NT ? __builtin_nontemporal_load((const v4*)a.k[j] + i) : ((const v4*)a.k[j])[i];
      acc += kk * a.c[j];
    }
    if (NT) __builtin_nontemporal_store(acc, out + i); else out[i] = acc;
  };
  if (!CHUNK) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < a.nvec; i += UNROLL * stride) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) body(i + u * stride);
    }
    for (; i < a.nvec; i += stride) body(i);
  } else {
    // block-contiguous: block b owns [b*chunk, (b+1)*chunk) vectors per pass, chunk = chunk_iters*256
    const long chunk = (long)chunk_iters * 256;
    for (long base = (long)blockIdx.x * chunk; base < a.nvec; base += (long)gridDim.x * chunk) {
      for (int it = 0; it < chunk_iters; ++it) { long i = base + (long)it * 256 + threadIdx.x; if (i < a.nvec) body(i); }
    }
  }
}

template <int NK, int UNROLL, bool NT, bool CHUNK>
float run(Args a, int grid, int chunk_iters, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kern<NK, UNROLL, NT, CHUNK>), dim3(grid), dim3(256), 0, 0, a, chunk_iters);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kern<NK, UNROLL, NT, CHUNK>), dim3(grid), dim3(256), 0, 0, a, chunk_iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : (1L << 25);
  Args a; a.nvec = n / 4;
  for (int j = 0; j < NKMAX; ++j) { float* p; hipMalloc(&p, n * 4); hipMemset(p, 0, n * 4); a.k[j] = p; a.c[j] = 0.1f * (j + 1); }
  float* p; hipMalloc(&p, n * 4); hipMemset(p, 0, n * 4); a.y0 = p; hipMalloc(&p, n * 4); a.out = p;
  const int reps = 20;
  auto gb = [&](int nk, float ms) { return (nk + 2.0) * n * 4 / (ms * 1e-3) / 1e9; };
  printf("n=%ld (%.0f MiB per stream)\n", n, n * 4.0 / (1 << 20));
#define R(NK, U, NT, CH, grid, ci) { float ms = run<NK, U, NT, CH>(a, grid, ci, reps); printf("nk=%d unroll=%d nt=%d chunk=%d grid=%d ci=%d : %.1f us  %.0f GB/s\n", NK, U, NT, CH, grid, ci, ms * 1e3, gb(NK, ms)); }
  R(5, 1, false, false, 2048, 0) R(5, 2, false, false, 2048, 0) R(5, 1, true, false, 2048, 0) R(5, 2, true, false, 2048, 0)
  R(5, 1, false, false, 1024, 0) R(5, 1, false, false, 4096, 0) R(5, 1, false, false, 8192, 0)
  R(5, 1, false, true, 2048, 4) R(5, 1, false, true, 2048, 16) R(5, 1, false, true, 1024, 16) R(5, 1, true, true, 2048, 16) R(5, 1, false, true, 2048, 64)
  R(1, 1, false, false, 2048, 0) R(1, 2, false, false, 2048, 0) R(1, 4, false, false, 2048, 0) R(1, 1, true, false, 2048, 0) R(1, 1, false, true, 2048, 16)
  R(3, 1, false, false, 2048, 0) R(3, 2, false, false, 2048, 0) R(3, 1, true, false, 2048, 0)
  R(7, 1, false, false, 2048, 0) R(7, 1, true, false, 2048, 0) R(7, 1, false, true, 2048, 16)
  return 0;
}
