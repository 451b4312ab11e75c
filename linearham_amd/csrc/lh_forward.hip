// K2: emission assembly and the scaled forward sweep of the V/D/J HMM (gfx950).
//
// Replaces, per tree sample (citations into matsengrp/linearham):
//   PhyloHMM::FillXmsaEmission tail           src/PhyloHMM.cpp:226-237 (rate mix, log, -log pi, exp)
//   PhyloHMM::InitializeEmission              src/PhyloHMM.cpp:94-114  (5x FillGermlinePaddingEmission,
//                                             :158-193, 2x FillJunctionEmission, :202-215)
//   HMM::RunForwardAlgorithm                  src/HMM.cpp:254-287
//     ComputeInitialForwardProbabilities      src/HMM.cpp:291-319
//     ComputeJunctionForwardProbabilities     src/HMM.cpp:1107-1139
//     ComputeGermlineForwardProbabilities     src/HMM.cpp:1160-1177
//   HMM::LogLikelihood                        src/HMM.cpp:345-354
//   ScaleMatrix                               src/utils.cpp:135-144
//
// The reference multiplies a 1xS row by dense SxS junction matrices that are >99% zeros.  Here the
// same products are applied in structured form (lh_junction): per junction row only the states that
// can emit at that site are live -- one germline position per gene plus the four NTI states of every
// right-hand gene -- and the cross-gene block is rank one (sum_l f_l*landing_out_l) * gene_prob *
// landing_in.
//
// Mapping: one 256-lane workgroup = one tree sample.  Gene g of a region lives in lane g % 256, register
// slot g / 256, so the whole HMM state (forward values of the live states) stays in VGPRs; LDS holds
// the sample's per-column emission vector (gathered by index) and the reduction scratch.  Per junction
// row there is ONE combined block reduction (the rank-one sum for the next row and the smallest
// positive entry that drives ScaleMatrix); the row's 2^256 scaling is applied lazily, as an exact
// power-of-two factor, when the row is consumed and when it is written out.
#include "lh_device.h"

namespace lh {

constexpr int kFwdThreads = 256;
constexpr int kFwdWaves = kFwdThreads / 64;

// Block-wide (sum, smallest positive) in one pass: shuffles inside the wave, one LDS exchange, one
// barrier (red must hold 2 * 2 * kFwdWaves doubles; `phase` alternates its two halves so that a
// reduction never overwrites values another wave is still reading).
__device__ static inline void block_sum_min(double& sum, double& mn, double* red, int phase) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_xor(sum, off, 64);
    mn = fmin(mn, __shfl_xor(mn, off, 64));
  }
  double* r = red + phase * 2 * kFwdWaves;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    r[wave] = sum;
    r[kFwdWaves + wave] = mn;
  }
  __syncthreads();
  sum = ((r[0] + r[1]) + r[2]) + r[3];
  mn = fmin(fmin(r[kFwdWaves], r[kFwdWaves + 1]), fmin(r[kFwdWaves + 2], r[kFwdWaves + 3]));
}

__device__ static inline int block_max_int(int v, int* red, int phase) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  int* r = red + phase * kFwdWaves;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
  __syncthreads();
  return max(max(r[0], r[1]), max(r[2], r[3]));
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

// FillGermlinePaddingEmission (src/PhyloHMM.cpp:158-193): per gene the running product of its
// columns' emissions with ScaleMatrix after every factor, then the 2^(256*d) equalisation to the
// region's largest scaler count (returned).  Thread `tid` owns genes tid + 256*q.
template <int kG>
__device__ static int fill_segments(const DevSegments& seg, const double* em, int C, int tid, double (&out)[kG],
                                    int* redi, int phase) {
  double v[kG];
  int c[kG];
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    v[q] = 1.0;
    c[q] = 0;
  }
  const int n = seg.n_genes;
  for (int j0 = 0; j0 < seg.n_rows; j0 += 8) {
#pragma unroll
    for (int q = 0; q < kG; ++q) {
      const int g = tid + kFwdThreads * q;
      int idx[8];
      double e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) idx[u] = g < n ? seg.inds_t[(size_t)(j0 + u) * n + g] : C;  // em[C] = 1
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = em[idx[u]];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // padded factors are exactly 1.0: no effect on (v, c)
        v[q] *= e[u];
        while (v[q] > 0.0 && v[q] < kScaleThreshold) {
          v[q] *= kScaleFactor;
          ++c[q];
        }
      }
    }
  }
  int local_max = 0;
#pragma unroll
  for (int q = 0; q < kG; ++q) local_max = max(local_max, c[q]);
  const int mx = block_max_int(local_max, redi, phase);
#pragma unroll
  for (int q = 0; q < kG; ++q) out[q] = v[q] * pow_scale(mx - c[q]);
  return mx;
}

__device__ static inline double scale_by(double v, int k) {  // v * (2^256)^k, exactly as k multiplications
  for (int t = 0; t < k; ++t) v *= kScaleFactor;
  return v;
}

// One junction region + the germline region to its right; all state in registers.
//   g[q]      in: forward of the left germline region (genes tid+256q);  out: forward of the right one
//   germ_em   emission products of the right germline region, pad_trans (global, may be null = ones),
//   pad_em    padding emission products (may be null = ones)
// Returns the right region's scaler count excluding its emission counts.
//
// Row i is first computed "raw" (without its own ScaleMatrix factor); one combined reduction then
// yields k_i (from the smallest positive raw entry) and the raw rank-one sum for row i+1.  Every
// later use multiplies by 2^(256*k_i), which is exact, so all values equal the reference's.
// Per-row table entries of the genes a thread owns (family constants, independent of the sample),
// fetched in one batch at the top of a row.  (Fetching them a row ahead, and keeping the row-invariant
// NTI tables in registers, was measured: the extra ~40 VGPRs cost more occupancy than the overlap won.)
template <int kG>
struct RowTables {
  double ltr[kG], llo[kG];          // left gene: transition into the row's state, its landing_out
  int lidx[kG];                     // left gene: emission column or -1
  double nlo[kG][4], rtr[kG], rli[kG];  // right gene: N->germline, germline->germline, cross-gene landing
  int ridx[kG], nx[kG][4];          // right gene: emission columns (germline state or -1, NTI states)
};

template <int kG>
__device__ static inline void load_row(const DevJunction& J, int i, int tid, RowTables<kG>& t) {
  const int nL = J.n_left, nR = J.n_right;
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    const int l = tid + kFwdThreads * q;
    if (l < nL) {
      t.ltr[q] = (i == 0) ? J.enter_trans[l] : J.left_trans[(size_t)i * nL + l];
      t.llo[q] = J.left_lo[(size_t)i * nL + l];
      t.lidx[q] = J.left_xmsa[(size_t)i * nL + l];
    }
    const int r = tid + kFwdThreads * q;
    if (r < nR) {
      const double* nlo = J.right_nlo + ((size_t)i * nR + r) * 4;
      const int32_t* nx = J.nti_xmsa + ((size_t)i * nR + r) * 4;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        t.nlo[q][b] = nlo[b];
        t.nx[q][b] = nx[b];
      }
      t.rtr[q] = J.right_trans[(size_t)i * nR + r];
      t.rli[q] = J.right_gp_li[(size_t)i * nR + r];
      t.ridx[q] = J.right_xmsa[(size_t)i * nR + r];
    }
  }
}

template <int kG>
__device__ static int junction_forward(const DevJunction& J, const double* em, int tid, double (&g)[kG],
                                       int count_in, const double (&germ_em)[kG], const double* pad_trans,
                                       const double* pad_em, double* fwd_out, int32_t* scal_out, double* red) {
  const int W = J.n_rows, nL = J.n_left, nR = J.n_right;
  int count = count_in;
  double fL[kG], fN[kG][4], fR[kG];  // raw values of the previous row
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    fL[q] = 0.0;
    fR[q] = 0.0;
    fN[q][0] = fN[q][1] = fN[q][2] = fN[q][3] = 0.0;
  }
  const size_t row_stride = (size_t)nL + 5 * (size_t)nR;
  // rank-one term of row 0: A = sum_l g[l] * landing_out_l[last germline-region index]
  double A;
  {
    double part = 0.0, dummy = __builtin_inf();
#pragma unroll
    for (int q = 0; q < kG; ++q) {
      const int l = tid + kFwdThreads * q;
      if (l < nL) part += g[q] * J.enter_lo[l];
    }
    block_sum_min(part, dummy, red, 0);
    A = part;
  }
  int k_prev = 0;  // ScaleMatrix count of the previous row, not yet applied to fL/fN/fR
  for (int i = 0; i < W; ++i) {
    RowTables<kG> cur;
    load_row<kG>(J, i, tid, cur);
    double mp = __builtin_inf(), part = 0.0;
#pragma unroll
    for (int q = 0; q < kG; ++q) {
      const int l = tid + kFwdThreads * q;
      if (l < nL) {
        const double f = (i == 0) ? g[q] : scale_by(fL[q], k_prev);
        const double v = (f * cur.ltr[q]) * (cur.lidx[q] >= 0 ? em[cur.lidx[q]] : 0.0);
        fL[q] = v;
        mp = posmin(mp, v);
        part += v * cur.llo[q];  // raw contribution to the next row's rank-one term
      }
    }
#pragma unroll
    for (int q = 0; q < kG; ++q) {
      const int r = tid + kFwdThreads * q;
      if (r < nR) {
        // previous row (zeros at i == 0), with its scaling applied now
        const double n0 = scale_by(fN[q][0], k_prev), n1 = scale_by(fN[q][1], k_prev);
        const double n2 = scale_by(fN[q][2], k_prev), n3 = scale_by(fN[q][3], k_prev);
        const double fr = scale_by(fR[q], k_prev);
        const double* ntt = J.right_ntt + (size_t)r * 16;
        const double* nli = J.right_gp_nli + (size_t)r * 4;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          double s = ((n0 * ntt[b] + n1 * ntt[4 + b]) + n2 * ntt[8 + b]) + n3 * ntt[12 + b];
          s += A * nli[b];
          const double v = s * em[cur.nx[q][b]];
          fN[q][b] = v;
          mp = posmin(mp, v);
        }
        double s = ((n0 * cur.nlo[q][0] + n1 * cur.nlo[q][1]) + n2 * cur.nlo[q][2]) + n3 * cur.nlo[q][3];
        s += fr * cur.rtr[q];
        s += A * cur.rli[q];
        const double v = s * (cur.ridx[q] >= 0 ? em[cur.ridx[q]] : 0.0);
        fR[q] = v;
        mp = posmin(mp, v);
      }
    }
    block_sum_min(part, mp, red, (i + 1) & 1);
    const int k = scale_count(mp);
    A = scale_by(part, k);  // = sum_l (row i scaled)[l] * landing_out_l
    count += k;
    k_prev = k;
    if (fwd_out) {
      double* o = fwd_out + (size_t)i * row_stride;
#pragma unroll
      for (int q = 0; q < kG; ++q) {
        const int t = tid + kFwdThreads * q;
        if (t < nL) o[t] = scale_by(fL[q], k);
        if (t < nR) {
          o[nL + 4 * (size_t)t + 0] = scale_by(fN[q][0], k);
          o[nL + 4 * (size_t)t + 1] = scale_by(fN[q][1], k);
          o[nL + 4 * (size_t)t + 2] = scale_by(fN[q][2], k);
          o[nL + 4 * (size_t)t + 3] = scale_by(fN[q][3], k);
          o[nL + 4 * (size_t)nR + t] = scale_by(fR[q], k);
        }
      }
    }
    if (scal_out && tid == 0) scal_out[i] = count;
  }
  // hand-off into the right germline region (A already holds the last row's rank-one sum)
  double mp = __builtin_inf(), dummy = 0.0;
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    const int r = tid + kFwdThreads * q;
    double v = 0.0;
    if (r < nR) {
      const double* xn = J.exit_nlo + (size_t)r * 4;
      const double n0 = scale_by(fN[q][0], k_prev), n1 = scale_by(fN[q][1], k_prev);
      const double n2 = scale_by(fN[q][2], k_prev), n3 = scale_by(fN[q][3], k_prev);
      double s = ((n0 * xn[0] + n1 * xn[1]) + n2 * xn[2]) + n3 * xn[3];
      s += scale_by(fR[q], k_prev) * J.exit_trans[r];
      s += A * J.exit_gp_li[r];
      v = s * germ_em[q];
      if (pad_trans) v *= pad_trans[r];
      if (pad_em) v *= pad_em[q];
      mp = posmin(mp, v);
    }
    g[q] = v;
  }
  block_sum_min(dummy, mp, red, (W + 1) & 1);
  const int k = scale_count(mp);
#pragma unroll
  for (int q = 0; q < kG; ++q) g[q] = scale_by(g[q], k);
  return count + k;
}

template <int kG, bool kFromSiteLik>
__global__ void __launch_bounds__(kFwdThreads)
    forward_kernel(const DevFamily fam, int R, const double* __restrict__ site_lik,
                   const int32_t* __restrict__ site_scal, const double* __restrict__ pi,
                   const double* __restrict__ em_in, double* __restrict__ em_out, double* __restrict__ loglik,
                   double* __restrict__ fwd_all, int32_t* __restrict__ scal_all) {
  extern __shared__ double em[];  // [C + 1] emissions (em[C] = 1.0 sentinel) | reduction scratch
  const int s = blockIdx.x;
  const int tid = threadIdx.x;
  const int C = fam.n_xmsa;
  double* red = em + ((C + 2) & ~1);                      // 4 * kFwdWaves doubles
  int* redi = reinterpret_cast<int*>(red + 4 * kFwdWaves);  // 2 * kFwdWaves ints

  if constexpr (kFromSiteLik) {
    // PhyloHMM::FillXmsaEmission tail: mix the rate categories (equal weights, scalers aligned to the
    // smallest one), log, naive correction, exp -- in the reference's formula order.
    const int L = fam.n_sites;
    const double w = 1.0 / R;
    for (int c = tid; c < C; c += kFwdThreads) {
      const int site = fam.xmsa_site[c];
      const int b = fam.xmsa_naive_base[c];
      int smin = 0x7fffffff;
      for (int r = 0; r < R; ++r) smin = min(smin, site_scal[((size_t)s * R + r) * L + site]);
      double acc = 0.0;
      for (int r = 0; r < R; ++r) {
        double v = site_lik[(((size_t)s * R + r) * 5 + b) * (size_t)L + site];
        const int d = site_scal[((size_t)s * R + r) * L + site] - smin;
        for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
        acc += w * v;
      }
      // The reference forms exp(log(site_lik) - smin*log(2^256) - log(pi_b)) (src/PhyloHMM.cpp:226-237);
      // the same quantity is computed here without the log/exp round trip (two FP64 transcendentals
      // per column): site_lik / pi_b scaled down by 2^(256*smin), which also underflows to 0 like exp().
      double e = acc;
      if (b != 4) e /= pi[(size_t)s * 4 + b];
      for (int q = 0; q < smin && e != 0.0; ++q) e *= kScaleThreshold;
      em[c] = e;
      if (em_out) em_out[(size_t)s * C + c] = e;
    }
  } else {
    for (int c = tid; c < C; c += kFwdThreads) em[c] = em_in[(size_t)s * C + c];
  }
  if (tid == 0) em[C] = 1.0;
  __syncthreads();

  double* fwd = fwd_all ? fwd_all + (size_t)s * fam.forward_size : nullptr;
  int32_t* sco = scal_all ? scal_all + (size_t)s * fam.scaler_size : nullptr;

  // V padding + V germline emissions, initial forward (src/HMM.cpp:291-319)
  const int nV = fam.vgerm.n_genes;
  double e1[kG], e2[kG], g[kG];
  int vcount = fill_segments<kG>(fam.vpadding, em, C, tid, e2, redi, 0);
  vcount += fill_segments<kG>(fam.vgerm, em, C, tid, e1, redi, 1);
  double mp = __builtin_inf(), dummy = 0.0;
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    const int t = tid + kFwdThreads * q;
    double v = 0.0;
    if (t < nV) {
      v = fam.vgerm_gene_prob[t];
      v *= fam.vpadding_transition[t];
      v *= e2[q];
      v *= fam.vgerm_trans_prod[t];
      v *= e1[q];
      mp = posmin(mp, v);
    }
    g[q] = v;
  }
  {
    block_sum_min(dummy, mp, red, 1);
    const int k = scale_count(mp);
#pragma unroll
    for (int q = 0; q < kG; ++q) g[q] = scale_by(g[q], k);
    vcount += k;
  }
  if (fwd) {
#pragma unroll
    for (int q = 0; q < kG; ++q)
      if (tid + kFwdThreads * q < nV) fwd[tid + kFwdThreads * q] = g[q];
    fwd += nV;
  }
  if (sco) {
    if (tid == 0) sco[0] = vcount;
    sco += 1;
  }

  int jcount;
  if (fam.has_d) {
    const int nD = fam.dgerm.n_genes;
    int dcount = fill_segments<kG>(fam.dgerm, em, C, tid, e1, redi, 0);
    dcount += junction_forward<kG>(fam.vd, em, tid, g, vcount, e1, nullptr, nullptr, fwd, sco, red);
    if (fwd) {
      fwd += (size_t)fam.vd.n_rows * (fam.vd.n_left + 5 * (size_t)fam.vd.n_right);
#pragma unroll
      for (int q = 0; q < kG; ++q)
        if (tid + kFwdThreads * q < nD) fwd[tid + kFwdThreads * q] = g[q];
      fwd += nD;
    }
    if (sco) {
      sco += fam.vd.n_rows;
      if (tid == 0) sco[0] = dcount;
      sco += 1;
    }
    jcount = fill_segments<kG>(fam.jgerm, em, C, tid, e1, redi, 1);
    jcount += fill_segments<kG>(fam.jpadding, em, C, tid, e2, redi, 0);
    jcount += junction_forward<kG>(fam.dj, em, tid, g, dcount, e1, fam.jpadding_transition, e2, fwd, sco, red);
    if (fwd) fwd += (size_t)fam.dj.n_rows * (fam.dj.n_left + 5 * (size_t)fam.dj.n_right);
    if (sco) sco += fam.dj.n_rows;
  } else {
    jcount = fill_segments<kG>(fam.jgerm, em, C, tid, e1, redi, 0);
    jcount += fill_segments<kG>(fam.jpadding, em, C, tid, e2, redi, 1);
    jcount += junction_forward<kG>(fam.vd, em, tid, g, vcount, e1, fam.jpadding_transition, e2, fwd, sco, red);
    if (fwd) fwd += (size_t)fam.vd.n_rows * (fam.vd.n_left + 5 * (size_t)fam.vd.n_right);
    if (sco) sco += fam.vd.n_rows;
  }
  const int nJ = fam.jgerm.n_genes;
  if (fwd) {
#pragma unroll
    for (int q = 0; q < kG; ++q)
      if (tid + kFwdThreads * q < nJ) fwd[tid + kFwdThreads * q] = g[q];
  }
  if (sco && tid == 0) sco[0] = jcount;

  // HMM::LogLikelihood (src/HMM.cpp:352-353)
  double part = 0.0, dm = __builtin_inf();
#pragma unroll
  for (int q = 0; q < kG; ++q)
    if (tid + kFwdThreads * q < nJ) part += g[q];
  __syncthreads();  // the last reduction's scratch may still be being read
  block_sum_min(part, dm, red, 0);
  if (tid == 0) loglik[s] = log(part) - jcount * kLogScaleFactor;
}

size_t forward_lds_bytes(const DevFamily& fam) {
  return (((size_t)fam.n_xmsa + 2) & ~(size_t)1) * sizeof(double) + 4 * kFwdWaves * sizeof(double) +
         2 * kFwdWaves * sizeof(int);
}

template <int kG>
static void launch_forward_g(const DevFamily& fam, int n, int R, const double* site_lik, const int32_t* site_scal,
                             const double* pi, const double* em_in, double* em_out, double* loglik,
                             double* forward_out, int32_t* scaler_out, hipStream_t stream) {
  const size_t lds = forward_lds_bytes(fam);
  if (site_lik) {
    if (lds > 64 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(forward_kernel<kG, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((forward_kernel<kG, true>), dim3(n), dim3(kFwdThreads), lds, stream, fam, R, site_lik,
                       site_scal, pi, em_in, em_out, loglik, forward_out, scaler_out);
  } else {
    if (lds > 64 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(forward_kernel<kG, false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((forward_kernel<kG, false>), dim3(n), dim3(kFwdThreads), lds, stream, fam, R, site_lik,
                       site_scal, pi, em_in, em_out, loglik, forward_out, scaler_out);
  }
}

// site_lik != null: emissions are assembled from K1's output (em_out optional);
// site_lik == null: emissions are taken from em_in (SimpleHMM / lh_forward_batch).
void launch_forward(const DevFamily& fam, int n, int R, const double* site_lik, const int32_t* site_scal,
                    const double* pi, const double* em_in, double* em_out, double* loglik, double* forward_out,
                    int32_t* scaler_out, hipStream_t stream) {
  const int slots = (fam.max_genes + kFwdThreads - 1) / kFwdThreads;
#define LH_ARGS fam, n, R, site_lik, site_scal, pi, em_in, em_out, loglik, forward_out, scaler_out, stream
  if (slots <= 1)
    launch_forward_g<1>(LH_ARGS);
  else if (slots <= 2)
    launch_forward_g<2>(LH_ARGS);
  else
    launch_forward_g<4>(LH_ARGS);
#undef LH_ARGS
}

}  // namespace lh
