// Microbenchmark: does a v_fma_f64 / v_mul_f64 that takes one operand from SGPRs (K1's P-matrix entries) issue at
// the same rate as one with vector operands only?  8 independent chains per wave, 32 instructions per iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/fma64_sgpr.hip -o tools/microbench/fma64_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, double s0, double s1, double s2, double s3, double* out) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 0) {  // vector operands only
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c0) : "v"(a), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c1) : "v"(a), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c2) : "v"(a), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c3) : "v"(a), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c4) : "v"(a), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c5) : "v"(a), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c6) : "v"(a), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c7) : "v"(a), "v"(b));
      } else if (MODE == 1) {  // one scalar operand
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c0) : "s"(s0), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c1) : "s"(s1), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c2) : "s"(s2), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c3) : "s"(s3), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c4) : "s"(s0), "v"(a));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c5) : "s"(s1), "v"(a));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c6) : "s"(s2), "v"(a));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c7) : "s"(s3), "v"(a));
      } else if (MODE == 2) {  // multiplies with one scalar operand
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c0) : "s"(s0), "v"(b));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c1) : "s"(s1), "v"(b));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c2) : "s"(s2), "v"(b));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c3) : "s"(s3), "v"(b));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c4) : "s"(s0), "v"(a));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c5) : "s"(s1), "v"(a));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c6) : "s"(s2), "v"(a));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c7) : "s"(s3), "v"(a));
      } else if (MODE == 3) {  // K1's mat-vec shape: rows of 4 dependent steps (mul, fma, fma, fma), 8 rows in flight
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c0) : "s"(s0), "v"(b));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c1) : "s"(s1), "v"(b));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c2) : "s"(s2), "v"(b));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c3) : "s"(s3), "v"(b));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c0) : "s"(s1), "v"(a));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c1) : "s"(s2), "v"(a));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c2) : "s"(s3), "v"(a));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c3) : "s"(s0), "v"(a));
      } else if (MODE == 4) {  // v_fma_f64 VOP3 with vector operands: d = a * b + c (separate destination)
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c4) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c5) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c6) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c7) : "v"(a), "v"(b));
      }
    }
  }
  const double s = ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7));
  if (s == 12345.678) out[0] = s;
}

template <int MODE>
static float run(int grid, int iters, double* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, 10, 1.0, 0.5, 0.25, 0.125, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, iters, 1.0, 0.5, 0.25, 0.125, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out;
  (void)hipMalloc(&out, 64);
  const int iters = 20000;
  for (int wgs : {1, 2, 3}) {
    const int grid = 256 * wgs;
    const float t[5] = {run<0>(grid, iters, out), run<1>(grid, iters, out), run<2>(grid, iters, out),
                        run<3>(grid, iters, out), run<4>(grid, iters, out)};
    const char* name[5] = {"fmac vector operands", "fmac one SGPR operand", "mul one SGPR operand",
                           "mat-vec rows (mul+fmac, SGPR)", "fma VOP3 vector operands"};
    for (int m = 0; m < 5; ++m) {
      const double n_inst = (m == 3 ? 8.0 : 8.0) * 4 * iters;  // per wave
      const double flops = (double)grid * 8 * 64 * n_inst * ((m == 2) ? 1 : (m == 3 ? 1.5 : 2));
      printf("%d waves/SIMD  %-32s %.3f ms  %.1f TFLOP/s  %.2f cycles per instruction per SIMD at 2.4 GHz\n", 2 * wgs,
             name[m], t[m], flops / t[m] / 1e9, t[m] * 1e-3 * 2.4e9 / (n_inst * 2 * wgs));
    }
  }
  return 0;
}
