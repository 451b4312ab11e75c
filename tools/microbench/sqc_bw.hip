// Microbenchmark: scalar-cache (K$) bandwidth per CU on gfx950.
// Each wave issues `iters` x 4 s_load_dwordx16 from a small buffer (resident in the scalar cache) and
// xors the results so that the loads stay alive.  Reports bytes/clk/CU for several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) k(const int* __restrict__ buf, int iters, int stride_words, int* out) {
  const int* p = buf + (blockIdx.x % 4) * 64;  // a few distinct lines
  int acc = 0;
  for (int i = 0; i < iters; ++i) {
    const int* q = p + (size_t)(i & 15) * stride_words;
    // 4 x 64 bytes (like one POP op)
#pragma unroll
    for (int j = 0; j < 64; ++j) acc ^= q[j];
  }
  if (acc == 0x12345678) out[0] = acc;
}

int main() {
  int *buf, *out;
  const int words = 1 << 20;
  hipMalloc(&buf, words * 4);
  hipMalloc(&out, 4);
  hipMemset(buf, 1, words * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  for (int stride : {64, 4096}) {        // 64 words = 256 B (16 lines x 256 B = 4 KB footprint: hits) ; 16 KB stride: misses
    for (int waves_per_cu : {1, 2, 4, 8, 16, 32}) {
      const int grid = 256 * waves_per_cu;
      hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, buf, 10, stride, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, buf, iters, stride, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)grid * iters * 256.0;
      printf("stride %5d B  waves/CU %2d  %.3f ms  %.1f GB/s total  %.2f B/clk/CU (at 2.4 GHz)  %.1f ns per x16 load per wave\n",
             stride * 4, waves_per_cu, ms, bytes / ms / 1e6, bytes / (ms * 1e-3) / 256 / 2.4e9,
             ms * 1e6 / (iters * 4.0));
    }
  }
  return 0;
}
