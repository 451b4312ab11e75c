// Microbenchmark: a 4x4 FP64 mat-vec whose matrix is GATHERED PER LANE from LDS (128 bytes per
// lane: eight ds_read_b128, every lane its own matrix -- what a per-pattern collapsed-tree walk of K1 would need, where
// neighbouring lanes sit on different tree nodes) against the form K1 uses today (the matrix is wave-uniform and comes
// through scalar loads into SGPRs, operands of v_fma_f64), at 4 / 6 / 8 resident waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_matvec.hip -o tools/microbench/lds_matvec
// Output: time, mat-vecs per second and the ratio gathered / SGPR (decision rule: build the collapsed-tree kernel only
// if the gathered form sustains >= 0.5x the SGPR form).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kMat = 200;  // matrices per workgroup table (a 101-tip tree has 199 branches): 25.6 KB of LDS

typedef const double __attribute__((address_space(4))) * cptr;

// MODE 0: wave-uniform matrix through scalar loads (global memory, L2-resident table)
// MODE 1: per-lane matrix gathered from LDS (random matrix per lane and step)
// MODE 2: per-lane matrix gathered from LDS, but all lanes of a wave take the SAME matrix (LDS broadcast: what the
//         bandwidth limit is not)
template <int MODE, int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k(int iters, const double* __restrict__ table, double* out) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < kMat * 16; i += blockDim.x) lds[i] = table[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double a[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[s][i] = 0.25 + 1e-3 * (lane + s + i);
  unsigned h = 0x9e3779b9u * (blockIdx.x * 256 + threadIdx.x + 1);
  const cptr ct = (cptr)table;
  for (int it = 0; it < iters; ++it) {
    double x[2][4];
    if (MODE == 0) {
      const int m = __builtin_amdgcn_readfirstlane((it * 7 + blockIdx.x) % kMat);
      const cptr p = ct + m * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          x[s][i] = fma(p[i * 4 + 3], a[s][3], fma(p[i * 4 + 2], a[s][2], fma(p[i * 4 + 1], a[s][1], p[i * 4] * a[s][0])));
    } else {
      h = h * 1664525u + 1013904223u;
      const int m = MODE == 1 ? (int)((h >> 8) % kMat) : (it * 7 + blockIdx.x) % kMat;
      const double2* q = reinterpret_cast<const double2*>(lds + m * 16);
      double p[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double2 v = q[j];
        p[2 * j] = v.x;
        p[2 * j + 1] = v.y;
      }
      // the same matrix serves both sites of the lane (as it would: a lane's two patterns would share a node only by
      // luck, so this is the favourable case for the gathered form)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          x[s][i] = fma(p[i * 4 + 3], a[s][3], fma(p[i * 4 + 2], a[s][2], fma(p[i * 4 + 1], a[s][1], p[i * 4] * a[s][0])));
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) a[s][i] = x[s][i];
  }
  double sum = 0;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += a[s][i];
  if (sum == 12345.678) out[0] = sum;
}

template <int MODE, int WAVES>
static double run(int iters, const double* table, double* out) {
  // 256 CUs x 4 SIMDs x WAVES waves = 256 x WAVES workgroups of 4 waves; LDS sized so that exactly WAVES workgroups fit a CU
  const int grid = 256 * WAVES;
  const size_t lds = (160 * 1024 / WAVES) & ~(size_t)1023;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(grid), dim3(256), lds, 0, 10, table, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(grid), dim3(256), lds, 0, iters, table, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int WAVES>
static void sweep(int iters, const double* table, double* out) {
  const double n = 256.0 * WAVES * 256 * 2 * iters;  // mat-vecs (lanes x two sites x steps)
  const double t0 = run<0, WAVES>(iters, table, out), t1 = run<1, WAVES>(iters, table, out), t2 = run<2, WAVES>(iters, table, out);
  printf("%d waves/SIMD  SGPR matrix (wave-uniform)      %8.3f ms  %7.1f G mat-vec/s  %5.1f TFLOP/s\n", WAVES, t0, n / t0 * 1e-6, n * 28 / t0 * 1e-9);
  printf("%d waves/SIMD  LDS gather, a matrix per lane   %8.3f ms  %7.1f G mat-vec/s  %5.1f TFLOP/s  = %.2fx the SGPR form\n", WAVES, t1,
         n / t1 * 1e-6, n * 28 / t1 * 1e-9, t0 / t1);
  printf("%d waves/SIMD  LDS read, one matrix per wave   %8.3f ms  %7.1f G mat-vec/s  %5.1f TFLOP/s  = %.2fx the SGPR form\n", WAVES, t2,
         n / t2 * 1e-6, n * 28 / t2 * 1e-9, t0 / t2);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  std::vector<double> h(kMat * 16);
  for (int m = 0; m < kMat; ++m)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) h[m * 16 + i * 4 + j] = i == j ? 0.97 - 1e-4 * m : 0.01 + 1e-4 * m / 3;  // row-stochastic
  double *table, *out;
  (void)hipMalloc(&table, h.size() * sizeof(double));
  (void)hipMalloc(&out, 64);
  (void)hipMemcpy(table, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
  printf("# 4x4 FP64 mat-vec, two sites per lane, %d steps; mat-vec = 4 mul + 12 FMA (28 flop)\n", iters);
  sweep<4>(iters, table, out);
  sweep<6>(iters, table, out);
  sweep<8>(iters, table, out);
  return 0;
}
