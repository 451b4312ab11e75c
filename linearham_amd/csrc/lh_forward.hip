// K2: emission assembly and the scaled forward sweep of the V/D/J HMM (gfx950).
//
// Replaces, per tree sample (citations into matsengrp/linearham):
//   K2a  PhyloHMM::FillXmsaEmission tail        src/PhyloHMM.cpp:226-237 (rate mix, log, -log pi, exp)
//   K2b  PhyloHMM::InitializeEmission           src/PhyloHMM.cpp:94-114  (5x FillGermlinePaddingEmission,
//                                               :158-193, 2x FillJunctionEmission, :202-215)
//        HMM::RunForwardAlgorithm               src/HMM.cpp:254-287
//          ComputeInitialForwardProbabilities   src/HMM.cpp:291-319
//          ComputeJunctionForwardProbabilities  src/HMM.cpp:1107-1139
//          ComputeGermlineForwardProbabilities  src/HMM.cpp:1160-1177
//        HMM::LogLikelihood                     src/HMM.cpp:345-354
//        ScaleMatrix                            src/utils.cpp:135-144
//
// The reference multiplies a 1xS row by dense SxS junction matrices that are >99% zeros.  Here the
// same products are applied in structured form (lh_junction): per junction row only the states that
// can emit at that site are live -- one germline position per gene plus the four NTI states of every
// right-hand gene -- and the cross-gene block is rank one (sum_l f_l*landing_out_l) * gene_prob *
// landing_in.  One workgroup = one tree sample; genes are spread over the 256 lanes; the state
// vector, the per-column emissions and the partial sums live in LDS.
#include "lh_device.h"

namespace lh {

constexpr int kFwdThreads = 256;

__device__ static inline double block_sum(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const double r = ((red[0] + red[1]) + red[2]) + red[3];
  __syncthreads();
  return r;
}

// smallest strictly positive value over the block (+inf if none)
__device__ static inline double block_minpos(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const double r = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
  __syncthreads();
  return r;
}

__device__ static inline int block_max_int(int v, int* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_down(v, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const int r = max(max(red[0], red[1]), max(red[2], red[3]));
  __syncthreads();
  return r;
}

// ScaleMatrix (src/utils.cpp:135-144) on a vector spread over the block: the loop
// "while any 0 < m < 2^-256: m *= 2^256" runs exactly as often as it takes the smallest positive
// entry to reach the threshold (multiplication by 2^256 is exact).
__device__ static inline int scale_count(double minpos) {
  int k = 0;
  while (minpos < kScaleThreshold) {  // minpos = +inf when there is no positive entry
    minpos *= kScaleFactor;
    ++k;
  }
  return k;
}

__device__ static inline double posmin(double a, double v) { return (v > 0.0) ? fmin(a, v) : a; }

__device__ static inline double pow_scale(int d) {  // std::pow(SCALE_FACTOR, d), src/PhyloHMM.cpp:191
  return d <= 0 ? 1.0 : d == 1 ? 0x1p256 : d == 2 ? 0x1p512 : d == 3 ? 0x1p768 : __builtin_inf();
}

// ---- K2a ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
    xmsa_emission_kernel(const int32_t* __restrict__ xmsa_site, const uint8_t* __restrict__ xmsa_base,
                         int C, int L, int R, const double* __restrict__ site_lik,
                         const int32_t* __restrict__ site_scal, const double* __restrict__ pi,
                         double* __restrict__ em) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (c >= C) return;
  const int site = xmsa_site[c];
  const int b = xmsa_base[c];
  int smin = 0x7fffffff;
  for (int r = 0; r < R; ++r) smin = min(smin, site_scal[((size_t)s * R + r) * L + site]);
  double acc = 0.0;
  const double w = 1.0 / R;  // equal category weights
  for (int r = 0; r < R; ++r) {
    double v = site_lik[(((size_t)s * R + r) * 5 + b) * (size_t)L + site];
    const int d = site_scal[((size_t)s * R + r) * L + site] - smin;
    for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
    acc += w * v;
  }
  double lnl = log(acc) - smin * kLogScaleFactor;            // Partition::LogLikelihood per-site value
  if (b != 4) lnl -= log(pi[(size_t)s * 4 + b]);              // src/PhyloHMM.cpp:231-234
  em[(size_t)s * C + c] = exp(lnl);                           // src/PhyloHMM.cpp:237
}

// ---- K2b ---------------------------------------------------------------------------------------------

struct FwdShared {
  double* em;      // [C]
  double* gA;      // [max_genes] germline-region forward (ping)
  double* gB;      // [max_genes] (pong)
  double* e1;      // [max_genes] emission products of the current region
  double* e2;      // [max_genes] padding emission products
  double *fL0, *fL1;   // [max_left]
  double *fN0, *fN1;   // [max_right*4]
  double *fR0, *fR1;   // [max_right]
  double* red;     // [4]
  int* redi;       // [4]
  int* cnt;        // [max_genes]
};

// FillGermlinePaddingEmission (src/PhyloHMM.cpp:158-193): out[g] = equalised running products,
// returns the region's max scaler count.
__device__ static int fill_segments(const DevSegments& seg, const double* em, double* out, int* cnt,
                                    int* redi) {
  int local_max = 0;
  for (int g = threadIdx.x; g < seg.n_genes; g += kFwdThreads) {
    double v = 1.0;
    int c = 0;
    const int32_t* __restrict__ col = seg.inds_t + g;
    const int stride = seg.n_genes;
    for (int j0 = 0; j0 < seg.n_rows; j0 += 8) {
      int idx[8];
      double e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) idx[u] = col[(size_t)(j0 + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = em[idx[u]];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // padded factors are exactly 1.0: no effect on (v, c)
        v *= e[u];
        while (v > 0.0 && v < kScaleThreshold) {
          v *= kScaleFactor;
          ++c;
        }
      }
    }
    out[g] = v;
    cnt[g] = c;
    local_max = max(local_max, c);
  }
  const int mx = block_max_int(local_max, redi);
  for (int g = threadIdx.x; g < seg.n_genes; g += kFwdThreads) out[g] *= pow_scale(mx - cnt[g]);
  __syncthreads();
  return mx;
}

// One junction region + the germline region to its right.
//   g_in[nL]: forward of the left germline region, count_in its scaler count.
//   germ_em[nR] (LDS), pad_trans (global, may be null = ones), pad_em (LDS, may be null = ones)
//   g_out[nR]; returns the right germline region's scaler count *excluding* its emission counts.
__device__ static int junction_forward(const DevJunction& J, const FwdShared& sh, const double* g_in,
                                       int count_in, const double* germ_em, const double* pad_trans,
                                       const double* pad_em, double* g_out, double* fwd_out,
                                       int32_t* scal_out) {
  const int W = J.n_rows, nL = J.n_left, nR = J.n_right;
  const double* em = sh.em;
  int count = count_in;
  // explicit pointer ping-pong (indexing a pointer array with a run-time value would go to scratch)
  double *fLc = sh.fL0, *fLp = sh.fL1, *fNc = sh.fN0, *fNp = sh.fN1, *fRc = sh.fR0, *fRp = sh.fR1;
  const size_t row_stride = (size_t)nL + 5 * (size_t)nR;
  for (int i = 0; i < W; ++i) {
    // rank-one cross-gene term: A = sum_l f_prev[l] * landing_out_l
    double part = 0.0;
    for (int t = threadIdx.x; t < nL; t += kFwdThreads) {
      const double f = (i == 0) ? g_in[t] : fLp[t];
      const double lo = (i == 0) ? J.enter_lo[t] : J.left_lo[(size_t)(i - 1) * nL + t];
      part += f * lo;
    }
    const double A = block_sum(part, sh.red);
    double mp = __builtin_inf();
    for (int t = threadIdx.x; t < nL; t += kFwdThreads) {
      const double f = (i == 0) ? g_in[t] : fLp[t];
      const double tr = (i == 0) ? J.enter_trans[t] : J.left_trans[(size_t)i * nL + t];
      const int idx = J.left_xmsa[(size_t)i * nL + t];
      const double v = (f * tr) * (idx >= 0 ? em[idx] : 0.0);
      fLc[t] = v;
      mp = posmin(mp, v);
    }
    for (int t = threadIdx.x; t < nR; t += kFwdThreads) {
      double n0 = 0, n1 = 0, n2 = 0, n3 = 0, fr = 0;
      if (i > 0) {
        n0 = fNp[t * 4 + 0];
        n1 = fNp[t * 4 + 1];
        n2 = fNp[t * 4 + 2];
        n3 = fNp[t * 4 + 3];
        fr = fRp[t];
      }
      const double* ntt = J.right_ntt + (size_t)t * 16;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double s = ((n0 * ntt[b] + n1 * ntt[4 + b]) + n2 * ntt[8 + b]) + n3 * ntt[12 + b];
        s += A * J.right_gp_nli[(size_t)t * 4 + b];
        const double v = s * em[J.nti_xmsa[((size_t)i * nR + t) * 4 + b]];
        fNc[t * 4 + b] = v;
        mp = posmin(mp, v);
      }
      const double* nlo = J.right_nlo + ((size_t)i * nR + t) * 4;
      double s = ((n0 * nlo[0] + n1 * nlo[1]) + n2 * nlo[2]) + n3 * nlo[3];
      s += fr * J.right_trans[(size_t)i * nR + t];
      s += A * J.right_gp_li[(size_t)i * nR + t];
      const int idx = J.right_xmsa[(size_t)i * nR + t];
      const double v = s * (idx >= 0 ? em[idx] : 0.0);
      fRc[t] = v;
      mp = posmin(mp, v);
    }
    const int k = scale_count(block_minpos(mp, sh.red));  // also orders the writes above
    if (k > 0) {
      for (int q = 0; q < k; ++q) {
        for (int t = threadIdx.x; t < nL; t += kFwdThreads) fLc[t] *= kScaleFactor;
        for (int t = threadIdx.x; t < nR * 4; t += kFwdThreads) fNc[t] *= kScaleFactor;
        for (int t = threadIdx.x; t < nR; t += kFwdThreads) fRc[t] *= kScaleFactor;
      }
      __syncthreads();
    }
    count += k;
    if (fwd_out) {
      double* o = fwd_out + (size_t)i * row_stride;
      for (int t = threadIdx.x; t < nL; t += kFwdThreads) o[t] = fLc[t];
      for (int t = threadIdx.x; t < nR * 4; t += kFwdThreads) o[nL + t] = fNc[t];
      for (int t = threadIdx.x; t < nR; t += kFwdThreads) o[nL + 4 * (size_t)nR + t] = fRc[t];
    }
    if (scal_out && threadIdx.x == 0) scal_out[i] = count;
    double* t;
    t = fLc; fLc = fLp; fLp = t;
    t = fNc; fNc = fNp; fNp = t;
    t = fRc; fRc = fRp; fRp = t;
  }
  double part = 0.0;
  for (int t = threadIdx.x; t < nL; t += kFwdThreads)
    part += fLp[t] * J.left_lo[(size_t)(W - 1) * nL + t];
  const double A = block_sum(part, sh.red);
  double mp = __builtin_inf();
  for (int t = threadIdx.x; t < nR; t += kFwdThreads) {
    const double* xn = J.exit_nlo + (size_t)t * 4;
    const double* fn = fNp + t * 4;
    double s = ((fn[0] * xn[0] + fn[1] * xn[1]) + fn[2] * xn[2]) + fn[3] * xn[3];
    s += fRp[t] * J.exit_trans[t];
    s += A * J.exit_gp_li[t];
    double v = s * germ_em[t];
    if (pad_trans) v *= pad_trans[t];
    if (pad_em) v *= pad_em[t];
    g_out[t] = v;
    mp = posmin(mp, v);
  }
  const int k = scale_count(block_minpos(mp, sh.red));
  if (k > 0) {
    for (int q = 0; q < k; ++q)
      for (int t = threadIdx.x; t < nR; t += kFwdThreads) g_out[t] *= kScaleFactor;
    __syncthreads();
  }
  return count + k;
}

__global__ void __launch_bounds__(kFwdThreads)
    forward_kernel(const DevFamily* __restrict__ famp, int max_left, int max_right, int max_genes,
                   const double* __restrict__ em_all, double* __restrict__ loglik,
                   double* __restrict__ fwd_all, int32_t* __restrict__ scal_all) {
  extern __shared__ double lds[];
  const DevFamily& fam = *famp;
  const int s = blockIdx.x;
  const int C = fam.n_xmsa;

  FwdShared sh;
  double* p = lds;
  sh.em = p;            p += (C + 2) & ~1;   // + sentinel em[C] = 1.0
  sh.gA = p;            p += max_genes;
  sh.gB = p;            p += max_genes;
  sh.e1 = p;            p += max_genes;
  sh.e2 = p;            p += max_genes;
  sh.fL0 = p;           p += max_left;
  sh.fL1 = p;           p += max_left;
  sh.fN0 = p;           p += 4 * max_right;
  sh.fN1 = p;           p += 4 * max_right;
  sh.fR0 = p;           p += max_right;
  sh.fR1 = p;           p += max_right;
  sh.red = p;           p += 4;
  sh.redi = reinterpret_cast<int*>(p);  p += 2;
  sh.cnt = reinterpret_cast<int*>(p);

  const double* em_g = em_all + (size_t)s * C;
  for (int c = threadIdx.x; c < C; c += kFwdThreads) sh.em[c] = em_g[c];
  if (threadIdx.x == 0) sh.em[C] = 1.0;
  __syncthreads();

  double* fwd = fwd_all ? fwd_all + (size_t)s * fam.forward_size : nullptr;
  int32_t* sco = scal_all ? scal_all + (size_t)s * fam.scaler_size : nullptr;

  // V padding + V germline emissions, initial forward (src/HMM.cpp:291-319)
  const int nV = fam.vgerm.n_genes;
  int vcount = fill_segments(fam.vpadding, sh.em, sh.e2, sh.cnt, sh.redi);
  vcount += fill_segments(fam.vgerm, sh.em, sh.e1, sh.cnt, sh.redi);
  double mp = __builtin_inf();
  for (int g = threadIdx.x; g < nV; g += kFwdThreads) {
    double v = fam.vgerm_gene_prob[g];
    v *= fam.vpadding_transition[g];
    v *= sh.e2[g];
    v *= fam.vgerm_trans_prod[g];
    v *= sh.e1[g];
    sh.gA[g] = v;
    mp = posmin(mp, v);
  }
  {
    const int k = scale_count(block_minpos(mp, sh.red));
    if (k > 0) {
      for (int q = 0; q < k; ++q)
        for (int g = threadIdx.x; g < nV; g += kFwdThreads) sh.gA[g] *= kScaleFactor;
      __syncthreads();
    }
    vcount += k;
  }
  if (fwd) {
    for (int g = threadIdx.x; g < nV; g += kFwdThreads) fwd[g] = sh.gA[g];
    fwd += nV;
  }
  if (sco) {
    if (threadIdx.x == 0) sco[0] = vcount;
    sco += 1;
  }

  int jcount;
  const double* gJ;
  if (fam.has_d) {
    // D germline emissions; V-D junction; D germline forward
    const int nD = fam.dgerm.n_genes;
    int dcount = fill_segments(fam.dgerm, sh.em, sh.e1, sh.cnt, sh.redi);
    dcount += junction_forward(fam.vd, sh, sh.gA, vcount, sh.e1, nullptr, nullptr, sh.gB, fwd, sco);
    if (fwd) {
      fwd += (size_t)fam.vd.n_rows * (fam.vd.n_left + 5 * (size_t)fam.vd.n_right);
      for (int g = threadIdx.x; g < nD; g += kFwdThreads) fwd[g] = sh.gB[g];
      fwd += nD;
    }
    if (sco) {
      sco += fam.vd.n_rows;
      if (threadIdx.x == 0) sco[0] = dcount;
      sco += 1;
    }
    __syncthreads();
    // J germline + J padding emissions; D-J junction; J germline forward
    jcount = fill_segments(fam.jgerm, sh.em, sh.e1, sh.cnt, sh.redi);
    jcount += fill_segments(fam.jpadding, sh.em, sh.e2, sh.cnt, sh.redi);
    jcount += junction_forward(fam.dj, sh, sh.gB, dcount, sh.e1, fam.jpadding_transition, sh.e2, sh.gA,
                               fwd, sco);
    if (fwd) fwd += (size_t)fam.dj.n_rows * (fam.dj.n_left + 5 * (size_t)fam.dj.n_right);
    if (sco) sco += fam.dj.n_rows;
    gJ = sh.gA;
  } else {
    jcount = fill_segments(fam.jgerm, sh.em, sh.e1, sh.cnt, sh.redi);
    jcount += fill_segments(fam.jpadding, sh.em, sh.e2, sh.cnt, sh.redi);
    jcount += junction_forward(fam.vd, sh, sh.gA, vcount, sh.e1, fam.jpadding_transition, sh.e2, sh.gB,
                               fwd, sco);
    if (fwd) fwd += (size_t)fam.vd.n_rows * (fam.vd.n_left + 5 * (size_t)fam.vd.n_right);
    if (sco) sco += fam.vd.n_rows;
    gJ = sh.gB;
  }
  const int nJ = fam.jgerm.n_genes;
  if (fwd)
    for (int g = threadIdx.x; g < nJ; g += kFwdThreads) fwd[g] = gJ[g];
  if (sco && threadIdx.x == 0) sco[0] = jcount;

  // HMM::LogLikelihood (src/HMM.cpp:352-353)
  double part = 0.0;
  for (int g = threadIdx.x; g < nJ; g += kFwdThreads) part += gJ[g];
  const double tot = block_sum(part, sh.red);
  if (threadIdx.x == 0) loglik[s] = log(tot) - jcount * kLogScaleFactor;
}

void launch_xmsa_emission(const DevFamily& fam, int n, int R, const double* site_lik,
                          const int32_t* site_scal, const double* pi, double* em, hipStream_t stream) {
  dim3 grid((fam.n_xmsa + 255) / 256, n), block(256);
  hipLaunchKernelGGL(xmsa_emission_kernel, grid, block, 0, stream, fam.xmsa_site, fam.xmsa_naive_base,
                     fam.n_xmsa, fam.n_sites, R, site_lik, site_scal, pi, em);
}

size_t forward_lds_bytes(const DevFamily& fam, int max_left, int max_right) {
  size_t d = ((size_t)fam.n_xmsa + 2) & ~(size_t)1;
  d += 4 * (size_t)fam.max_genes + 2 * (size_t)max_left + 10 * (size_t)max_right + 4 + 2;
  return d * sizeof(double) + (size_t)fam.max_genes * sizeof(int) + 16;
}

void launch_forward(const DevFamily* fam_dev, const DevFamily& fam, int n, const double* em,
                    double* loglik, double* forward_out, int32_t* scaler_out, hipStream_t stream) {
  int max_left = fam.vd.n_left, max_right = fam.vd.n_right;
  if (fam.has_d) {
    max_left = max_left > fam.dj.n_left ? max_left : fam.dj.n_left;
    max_right = max_right > fam.dj.n_right ? max_right : fam.dj.n_right;
  }
  const size_t lds = forward_lds_bytes(fam, max_left, max_right);
  if (lds > 64 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(forward_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(forward_kernel, dim3(n), dim3(kFwdThreads), lds, stream, fam_dev, max_left, max_right,
                     fam.max_genes, em, loglik, forward_out, scaler_out);
}

}  // namespace lh
