// Microbenchmark: FP64 matrix-core rate on gfx950 for the 4x4x4 (4 blocks) shape, and whether it runs beside
// FP64 vector FMAs of other waves.  mode 0: all waves MFMA; 1: all waves VALU FMA; 2: even waves MFMA, odd
// waves VALU (same per-wave instruction counts as modes 0 / 1).  Workgroups of 512 threads: waves w and w + 4
// share a SIMD, so in mode 2 (waves 0-3 MFMA, waves 4-7 VALU) every SIMD hosts one wave of each kind.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f64.hip -o /tmp/mfma_f64 && /tmp/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(512) k(int iters, int mode, double* out) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {  // 8 independent accumulators
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    }
  } else {
    for (int i = 0; i < iters; ++i) {  // 8 x 4 dependent-free FMAs per iteration (= the flops of 2 MFMAs)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c0 = fma(a, b, c0);
        c1 = fma(a, b, c1);
        c2 = fma(a, b, c2);
        c3 = fma(a, b, c3);
        c4 = fma(a, b, c4);
        c5 = fma(a, b, c5);
        c6 = fma(a, b, c6);
        c7 = fma(a, b, c7);
      }
    }
  }
  const double s = ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7));
  if (s == 12345.678) out[0] = s;
}

int main() {
  double* out;
  hipMalloc(&out, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int wgs_per_cu : {1, 2}) {
    for (int mode : {0, 1, 2}) {
      const int grid = 256 * wgs_per_cu;
      hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, 10, mode, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, iters, mode, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      // flops: MFMA 4x4x4 x 4 blocks = 512 per instruction, 8 per iteration; VALU wave: 32 FMAs x 64 lanes x 2
      const double waves = (double)grid * 8;
      const double mf = mode == 0 ? waves : mode == 2 ? waves / 2 : 0, vf = mode == 1 ? waves : mode == 2 ? waves / 2 : 0;
      const double flops = (mf * 8 * 512.0 + vf * 32 * 64 * 2.0) * iters;
      printf("WGs/CU %d mode %d: %.3f ms  %.1f TFLOP/s  (cycles per wave-iteration at 2.4 GHz: %.1f)\n", wgs_per_cu, mode,
             ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / iters);
    }
  }
  return 0;
}
