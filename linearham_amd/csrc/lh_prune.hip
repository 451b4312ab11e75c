// K1: Felsenstein pruning over the clonal tree for all alignment columns (gfx950).
//
// Replaces Partition::TraversalUpdate(root, FULL) + Partition::LogLikelihood(root, per_site)
// (src/PhyloHMM.cpp:224-226; libpll's pll_update_partials / pll_compute_edge_loglikelihood [3P]).
//
// Design ("K1"):
//  * The reference evaluates every xMSA column, i.e. every (naive base, MSA site) pair, as an
//    independent alignment column.  All xMSA columns of one MSA site differ only in the state of
//    the `naive` tip, so the tree is rooted at naive's neighbour: the CLV of that node is
//    computed ONCE per MSA site and the five possible naive states (A,C,G,T,N) are closed in the
//    epilogue with the naive branch's P-matrix.  By reversibility of GTR the result is the
//    libpll value for every xMSA column.
//  * One workgroup = (site tile, rate category, tree sample); one lane = one MSA site.  The
//    traversal is a wave-uniform schedule (lh_schedule_tree): every lane executes the same op, so
//    the 4x4 P-matrices of the op are wave-uniform and are fetched with scalar loads into SGPRs
//    (no LDS / VGPR cost), and child CLVs never leave the chip: the running CLV lives in VGPRs
//    and pending siblings live in a register-resident stack of statically indexed slots.
//  * Tip children need no mat-vec: P * onehot(state) is a column of P.  Those columns (plus the
//    row sums for N) are staged once per workgroup in LDS as tiptab[tip][state][4] and gathered
//    with two ds_read_b128 per lane.
//  * K0b stores the P-matrices of a (sample, rate) in schedule order, so the scalar loads of the loop
//    stream through memory and the waves of a workgroup share every fetched line.
//  * HBM traffic is therefore ~T bytes of tip states per site (L2-resident, shared by all samples)
//    and 5 doubles out, instead of the 2*I*R*32 bytes per column of a CLV-streaming kernel.
#include <cstdlib>

#include "lh_device.h"

namespace lh {

#define LH_STACK_CASE(d)        \
  case d:                       \
    s##d##_0 = a0;              \
    s##d##_1 = a1;              \
    s##d##_2 = a2;              \
    s##d##_3 = a3;              \
    break;
#define LH_POP_CASE(d)          \
  case d:                       \
    y0 = s##d##_0;              \
    y1 = s##d##_1;              \
    y2 = s##d##_2;              \
    y3 = s##d##_3;              \
    break;
#define LH_DECL_SLOT(d) double s##d##_0 = 0, s##d##_1 = 0, s##d##_2 = 0, s##d##_3 = 0;

template <int kDepth>
__device__ __forceinline__ void prune_body(int compute_threads, int ahead, const uint8_t* __restrict__ msa, int L, int T, int n_ops,
                 const int32_t* __restrict__ ops, const double* __restrict__ pmat,
                 const double* __restrict__ tipvec, const double* __restrict__ pi,
                 double* __restrict__ site_lik, int32_t* __restrict__ site_scal) {
  extern __shared__ double2 smem2[];
  double* tiptab = reinterpret_cast<double*>(smem2);  // [T][5][4]

  const int tid = threadIdx.x;
  const int R = gridDim.y;
  const int rate = blockIdx.y;
  const int sample = blockIdx.z;
  const int site_raw = blockIdx.x * compute_threads + tid;
  const int site = site_raw < L ? site_raw : L - 1;

  {  // stage this (sample, rate)'s tip table in LDS
    const double2* src =
        reinterpret_cast<const double2*>(tipvec + ((size_t)sample * R + rate) * (size_t)T * 20);
    for (int i = tid; i < T * 10; i += blockDim.x) smem2[i] = src[i];
  }
  int* progress = reinterpret_cast<int*>(smem2 + T * 10);  // furthest op any compute wave has reached
  if (tid == 0) *progress = 0;
  __syncthreads();

  const int4* __restrict__ op_ptr = reinterpret_cast<const int4*>(ops) + (size_t)sample * n_ops;
  // P-matrices in schedule order: op k's accumulator-child matrix at [k][0], its popped-child
  // matrix at [k][1] (written by K0b), so the scalar loads walk memory sequentially and their
  // addresses do not depend on the op descriptor.
  const double* __restrict__ pm = pmat + ((size_t)sample * R + rate) * (size_t)(T - 2) * 32;
  const unsigned usite = (unsigned)site;  // MSA byte offsets are 32-bit: (tip row) * L + site

  if (tid >= compute_threads) {
    // Prefetcher wave.  The P-matrix scalar loads of the compute waves miss the 16 KB scalar cache
    // (four workgroups stream 25 KB each through it); measured, those misses were 27 % of the kernel.
    // A touch issued by a compute wave would not help -- scalar loads return out of order, so its next
    // s_waitcnt lgkmcnt(0) would wait for the touch as well -- but this extra wave can take the
    // misses instead: it walks the same P-matrix stream a few ops ahead of the compute waves (paced by
    // the progress word in LDS) and pulls every 64-byte line into the scalar cache.  (Plain loads
    // whose values feed a never-true store: inline asm here doubled the kernel's VGPR allocation.)
    const int* __restrict__ pwords = reinterpret_cast<const int*>(pm);
    const int* __restrict__ owords = reinterpret_cast<const int*>(op_ptr);
    int pf = 0, sink = 0;
    for (int spins = 0; pf < n_ops && spins < (1 << 24); ++spins) {
      int target = __builtin_amdgcn_readfirstlane(
                       __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) + ahead;
      target = target < n_ops ? target : n_ops;
      if (pf >= target) {
        __builtin_amdgcn_s_sleep(4);
        continue;
      }
      for (; pf < target; ++pf) {
        const int* q = pwords + (size_t)pf * 64;  // 256 bytes = four 64-byte lines per op
        sink ^= q[0] ^ q[16] ^ q[32] ^ q[48];
        sink ^= owords[(size_t)(pf + 4 < n_ops ? pf + 4 : pf) * 4];
      }
    }
    if (sink == 0x5a17c0de) site_scal[0] = sink;  // practically never; keeps the touches alive
    return;
  }

  const bool lane0 = (tid & 63) == 0;
  double a0 = 1.0, a1 = 1.0, a2 = 1.0, a3 = 1.0;
  int scal = 0;
  LH_DECL_SLOT(0) LH_DECL_SLOT(1) LH_DECL_SLOT(2) LH_DECL_SLOT(3)
  LH_DECL_SLOT(4) LH_DECL_SLOT(5) LH_DECL_SLOT(6) LH_DECL_SLOT(7)
  LH_DECL_SLOT(8) LH_DECL_SLOT(9) LH_DECL_SLOT(10) LH_DECL_SLOT(11)
  LH_DECL_SLOT(12) LH_DECL_SLOT(13) LH_DECL_SLOT(14) LH_DECL_SLOT(15)

  // Software pipeline across iterations: op k's descriptor and tip states were requested during
  // iteration k-1, so an iteration starts with everything but its P-matrices at hand (and those come
  // from addresses that depend on k only).
  int4 op = op_ptr[0];
  int sa = 0, sb = 0;
  {
    const int kd = op.x & 15;
    if (kd != OP_POP_ACC) sa = msa[(unsigned)((op.y - 1) * L) + usite];
    if (kd == OP_CHERRY) sb = msa[(unsigned)((op.z - 1) * L) + usite];
  }
  for (int k = 0; k < n_ops; ++k) {
    const int4 op_next = op_ptr[k + 1 < n_ops ? k + 1 : k];
    // paces the prefetcher wave (a relaxed LDS store, no atomic max: the waves of a workgroup run within
    // an op or two of each other)
    if (lane0) __hip_atomic_store(progress, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int kind = op.x & 15;
    if (op.x & OP_PUSH_FLAG) {
      // Independent wave-uniform branches, one per slot: each leaves every other slot's registers alone
      // (a single switch made the compiler shuffle whole slots through temporaries at its merge points).
#define LH_PUSH_IF(d)             \
  if constexpr (kDepth > d) {     \
    if (op.w == d) {              \
      s##d##_0 = a0;              \
      s##d##_1 = a1;              \
      s##d##_2 = a2;              \
      s##d##_3 = a3;              \
    }                             \
  }
      LH_PUSH_IF(0) LH_PUSH_IF(1) LH_PUSH_IF(2) LH_PUSH_IF(3)
      if constexpr (kDepth > 4) {
        if (op.w >= 4) {
          LH_PUSH_IF(4) LH_PUSH_IF(5) LH_PUSH_IF(6) LH_PUSH_IF(7)
          LH_PUSH_IF(8) LH_PUSH_IF(9) LH_PUSH_IF(10) LH_PUSH_IF(11)
          LH_PUSH_IF(12) LH_PUSH_IF(13) LH_PUSH_IF(14) LH_PUSH_IF(15)
        }
      }
#undef LH_PUSH_IF
    }
    if (kind == OP_CHERRY) {
      const double2* ta = reinterpret_cast<const double2*>(tiptab + op.y * 20 + sa * 4);
      const double2* tb = reinterpret_cast<const double2*>(tiptab + op.z * 20 + sb * 4);
      const double2 ta0 = ta[0], ta1 = ta[1], tb0 = tb[0], tb1 = tb[1];
      a0 = ta0.x * tb0.x;
      a1 = ta0.y * tb0.y;
      a2 = ta1.x * tb1.x;
      a3 = ta1.y * tb1.y;
    } else {
      const double* __restrict__ pb = pm + (size_t)k * 32;
      const double x0 = fma(pb[3], a3, fma(pb[2], a2, fma(pb[1], a1, pb[0] * a0)));
      const double x1 = fma(pb[7], a3, fma(pb[6], a2, fma(pb[5], a1, pb[4] * a0)));
      const double x2 = fma(pb[11], a3, fma(pb[10], a2, fma(pb[9], a1, pb[8] * a0)));
      const double x3 = fma(pb[15], a3, fma(pb[14], a2, fma(pb[13], a1, pb[12] * a0)));
      if (kind == OP_TIP_ACC) {
        const double2* ta = reinterpret_cast<const double2*>(tiptab + op.y * 20 + sa * 4);
        const double2 ta0 = ta[0], ta1 = ta[1];
        a0 = ta0.x * x0;
        a1 = ta0.y * x1;
        a2 = ta1.x * x2;
        a3 = ta1.y * x3;
      } else {  // OP_POP_ACC
        // The popped sibling is multiplied straight out of its slot (one copy of the mat-vec per shallow
        // slot) instead of being copied into common registers first.
        const double* __restrict__ pa = pm + (size_t)k * 32 + 16;
        double z0, z1, z2, z3;
#define LH_POP_MATVEC(y0, y1, y2, y3)                                      \
  z0 = fma(pa[3], y3, fma(pa[2], y2, fma(pa[1], y1, pa[0] * y0)));         \
  z1 = fma(pa[7], y3, fma(pa[6], y2, fma(pa[5], y1, pa[4] * y0)));         \
  z2 = fma(pa[11], y3, fma(pa[10], y2, fma(pa[9], y1, pa[8] * y0)));       \
  z3 = fma(pa[15], y3, fma(pa[14], y2, fma(pa[13], y1, pa[12] * y0)));
#define LH_POP_SLOT(d) LH_POP_MATVEC(s##d##_0, s##d##_1, s##d##_2, s##d##_3)
        if (op.w == 0) {
          LH_POP_SLOT(0)
        } else if (op.w == 1) {
          LH_POP_SLOT(1)
        } else if (kDepth > 2 && op.w == 2) {
          LH_POP_SLOT(2)
        } else if (kDepth > 3 && op.w == 3) {
          LH_POP_SLOT(3)
        } else {
          double y0 = 0, y1 = 0, y2 = 0, y3 = 0;
          if constexpr (kDepth > 4) {
            switch (op.w) {
              LH_POP_CASE(4) LH_POP_CASE(5) LH_POP_CASE(6) LH_POP_CASE(7)
              default:
                if constexpr (kDepth > 8) {
                  switch (op.w) {
                    LH_POP_CASE(8) LH_POP_CASE(9) LH_POP_CASE(10) LH_POP_CASE(11)
                    LH_POP_CASE(12) LH_POP_CASE(13) LH_POP_CASE(14) LH_POP_CASE(15)
                  }
                }
            }
          }
          LH_POP_MATVEC(y0, y1, y2, y3)
        }
#undef LH_POP_SLOT
#undef LH_POP_MATVEC
        a0 = z0 * x0;
        a1 = z1 * x1;
        a2 = z2 * x2;
        a3 = z3 * x3;
      }
    }
    // per-site, per-rate 2^256 rescaling (libpll PLL_ATTRIB_RATE_SCALERS semantics): the single
    // running counter is valid for the whole tree because scalers are additive along the traversal.
    op = op_next;
    {
      const int kd = op.x & 15;
      if (kd != OP_POP_ACC) sa = msa[(unsigned)((op.y - 1) * L) + usite];
      if (kd == OP_CHERRY) sb = msa[(unsigned)((op.z - 1) * L) + usite];
    }
    // CLV entries are non-negative, so the largest has the largest high word; it is below 2^-256
    // exactly when that word is below 0x2FF00000 (integer compares instead of 7 FP64 max/compare).
    const unsigned hw = max(max((unsigned)__double2hiint(a0), (unsigned)__double2hiint(a1)),
                            max((unsigned)__double2hiint(a2), (unsigned)__double2hiint(a3)));
    const bool tiny = hw < 0x2FF00000u && hw != 0u;
    if (__builtin_expect(__ballot(tiny) != 0, 0) && tiny) {  // rare: skip the whole block wave-wide
      a0 *= kScaleFactor;
      a1 *= kScaleFactor;
      a2 *= kScaleFactor;
      a3 *= kScaleFactor;
      ++scal;
    }
  }

  // epilogue: close the naive branch for each possible naive state b (A,C,G,T,N):
  //   L_b = sum_i pi_i * clv_root[i] * P_naive[i][b]        (N: row sums of P_naive)
  const double* __restrict__ p4 = pi + (size_t)sample * 4;
  const double w0 = p4[0] * a0, w1 = p4[1] * a1, w2 = p4[2] * a2, w3 = p4[3] * a3;
  if (site_raw < L) {
    double* out = site_lik + (((size_t)sample * R + rate) * 5) * (size_t)L + site;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const double* tv = tiptab + b * 4;  // tip 0 = naive
      out[(size_t)b * L] = fma(w3, tv[3], fma(w2, tv[2], fma(w1, tv[1], w0 * tv[0])));
    }
    site_scal[((size_t)sample * R + rate) * (size_t)L + site] = scal;
  }
}

#define LH_PRUNE_PARAMS                                                                                       \
  int compute_threads, int ahead, const uint8_t *__restrict__ msa, int L, int T, int n_ops,                  \
      const int32_t *__restrict__ ops, const double *__restrict__ pmat, const double *__restrict__ tipvec,   \
      const double *__restrict__ pi, double *__restrict__ site_lik, int32_t *__restrict__ site_scal
#define LH_PRUNE_ARGS compute_threads, ahead, msa, L, T, n_ops, ops, pmat, tipvec, pi, site_lik, site_scal

// Shallow stacks (the common case): 64 VGPRs allow 8 waves per SIMD, but only if the wave also stays
// within 96 SGPRs (the two P-matrices of an op alone are 64) -- the cap trades a few scalar spills for
// a fourth resident workgroup per CU.
template <int kDepth>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_sgpr(96))) prune_kernel(LH_PRUNE_PARAMS) {
  prune_body<kDepth>(LH_PRUNE_ARGS);
}

// Deep stacks are VGPR-limited anyway: no SGPR cap.
template <int kDepth>
__global__ void __launch_bounds__(512) prune_kernel_deep(LH_PRUNE_PARAMS) {
  prune_body<kDepth>(LH_PRUNE_ARGS);
}

void launch_prune(const DevFamily& fam, int n, int R, int T, int max_depth, const int32_t* ops,
                  const double* pmat, const double* tipvec, const double* pi, double* site_lik,
                  int32_t* site_scal, hipStream_t stream) {
  const int L = fam.n_sites;
  // up to 7 compute waves (448 sites) + 1 prefetcher wave per workgroup
  int compute = ((L + 63) / 64) * 64;
  if (compute > 448) compute = 448;
  const int tiles = (L + compute - 1) / compute;
  compute = (((L + tiles - 1) / tiles) + 63) / 64 * 64;  // rebalance so the tiles are equally full
  const size_t lds = (size_t)T * 20 * sizeof(double) + 16;
  dim3 grid(tiles, R, n), block(compute + 64);
  const int n_ops = T - 2;
  static const int ahead = getenv("LH_K1_AHEAD") ? atoi(getenv("LH_K1_AHEAD")) : 8;
#define LH_LAUNCH_K(K)                                                                                    \
  {                                                                                                       \
    if (lds > 64 * 1024)                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                \
    hipLaunchKernelGGL(K, grid, block, lds, stream, compute, ahead, fam.msa, L, T, n_ops, ops, pmat, tipvec, pi, \
                       site_lik, site_scal);                                                              \
  }
  if (max_depth <= 3)
    LH_LAUNCH_K(prune_kernel<3>)
  else if (max_depth <= 4)
    LH_LAUNCH_K(prune_kernel<4>)
  else if (max_depth <= 8)
    LH_LAUNCH_K(prune_kernel_deep<8>)
  else
    LH_LAUNCH_K(prune_kernel_deep<16>)
#undef LH_LAUNCH_K
}

}  // namespace lh
