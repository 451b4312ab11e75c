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
// Two kernels, because the two halves want opposite shapes:
//
//   K2a emission_kernel   one 256-lane workgroup per sample.  Wide and shallow: builds the sample's
//       per-column emission vector in LDS, then the germline/padding emission products (one gene per
//       lane, a gather from LDS per factor).  Leaves per sample: the five emission-product vectors, their
//       three scaler counts, and the emissions of the few hundred columns the junction rows touch.
//
//   K2b junction_kernel   one WAVE per sample (four samples per workgroup).  Narrow and deep: ~W
//       strictly sequential junction rows, each needing a reduction over the live states.  With a wave
//       per sample the reductions are cross-lane only -- no barriers, no LDS exchange -- the HMM state
//       (gene g in lane g % 64, register slot g / 64) stays in VGPRs, and a CU holds 16 samples instead
//       of 4, which is what hides the per-row table-load latency.  The row's 2^256 scaling is applied
//       lazily, as an exact power-of-two factor, when the row is consumed and when it is written out.
#include "lh_device.h"

namespace lh {

constexpr int kFwdThreads = 256;
constexpr int kFwdWaves = kFwdThreads / 64;
constexpr int kJunctionWaves = 4;  // samples per K2b workgroup

__device__ static inline int block_max_int(int v, int* red, int phase) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  int* r = red + phase * kFwdWaves;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
  __syncthreads();
  return max(max(r[0], r[1]), max(r[2], r[3]));
}

// ScaleMatrix (src/utils.cpp:135-144) on a vector spread over lanes: the loop
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

__device__ static inline double scale_by(double v, int k) {  // v * (2^256)^k, exactly as k multiplications
  for (int t = 0; t < k; ++t) v *= kScaleFactor;
  return v;
}

// ---------------------------------------------------------------------------------------------------
// K2a
// ---------------------------------------------------------------------------------------------------

// FillGermlinePaddingEmission (src/PhyloHMM.cpp:158-193): per gene the running product of its
// columns' emissions with ScaleMatrix after every factor, then the 2^(256*d) equalisation to the
// region's largest scaler count (returned).  Thread `tid` owns genes tid + 256*q; the products go to
// out[gene].
template <int kG>
__device__ static int fill_segments(const DevSegments& seg, const double* em, int C, int tid,
                                    double* __restrict__ out, int* redi, int phase) {
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
  for (int q = 0; q < kG; ++q) {
    const int g = tid + kFwdThreads * q;
    if (g < n) out[g] = v[q] * pow_scale(mx - c[q]);
  }
  return mx;
}

template <int kG, bool kFromSiteLik>
__global__ void __launch_bounds__(kFwdThreads)
    emission_kernel(const DevFamily fam, int R, const double* __restrict__ site_lik,
                    const int32_t* __restrict__ site_scal, const double* __restrict__ pi,
                    const double* __restrict__ em_in, double* __restrict__ em_out, double* __restrict__ gem_all,
                    int32_t* __restrict__ gcnt_all, double* __restrict__ jem_all) {
  extern __shared__ double em[];  // [C + 1] emissions (em[C] = 1.0 sentinel) | reduction scratch
  const int s = blockIdx.x;
  const int tid = threadIdx.x;
  const int C = fam.n_xmsa;
  int* redi = reinterpret_cast<int*>(em + ((C + 2) & ~1));  // 2 * kFwdWaves ints

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

  // emissions of the columns the junction rows touch, compacted for K2b
  {
    double* jem = jem_all + (size_t)s * fam.n_jcols;
    for (int j = tid; j < fam.n_jcols; j += kFwdThreads) jem[j] = em[fam.jcols[j]];
  }

  // [vpadding nV | vgerm nV | dgerm nD | jgerm nJ | jpadding nJ]
  const int nV = fam.vgerm.n_genes, nD = fam.dgerm.n_genes, nJ = fam.jgerm.n_genes;
  double* gem = gem_all + (size_t)s * fam.gem_size;
  int cv = fill_segments<kG>(fam.vpadding, em, C, tid, gem, redi, 0);
  cv += fill_segments<kG>(fam.vgerm, em, C, tid, gem + nV, redi, 1);
  int cd = 0, cj;
  if (fam.has_d) {
    cd = fill_segments<kG>(fam.dgerm, em, C, tid, gem + 2 * (size_t)nV, redi, 0);
    cj = fill_segments<kG>(fam.jgerm, em, C, tid, gem + 2 * (size_t)nV + nD, redi, 1);
    cj += fill_segments<kG>(fam.jpadding, em, C, tid, gem + 2 * (size_t)nV + nD + nJ, redi, 0);
  } else {
    cj = fill_segments<kG>(fam.jgerm, em, C, tid, gem + 2 * (size_t)nV, redi, 0);
    cj += fill_segments<kG>(fam.jpadding, em, C, tid, gem + 2 * (size_t)nV + nJ, redi, 1);
  }
  if (tid == 0) {
    gcnt_all[(size_t)s * 3 + 0] = cv;
    gcnt_all[(size_t)s * 3 + 1] = cd;
    gcnt_all[(size_t)s * 3 + 2] = cj;
  }
}

// ---------------------------------------------------------------------------------------------------
// K2b
// ---------------------------------------------------------------------------------------------------

// Wave-wide (sum, smallest positive) by butterfly: every lane ends with both results.
__device__ static inline void wave_sum_min(double& sum, double& mn) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_xor(sum, off, 64);
    mn = fmin(mn, __shfl_xor(mn, off, 64));
  }
}

__device__ static inline double wave_min(double mn) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_xor(mn, off, 64));
  return mn;
}

// One junction region + the germline region to its right; all state in registers of one wave.
//   f_in[q]   forward of the left germline region (genes lane + 64q)
//   g_out[q]  forward of the right germline region
//   germ_em   emission products of the right germline region (per sample, global),
//   pad_trans / pad_em   padding transition (family) and padding emission products (sample); null = ones
// Returns count_in plus every ScaleMatrix count taken inside (junction rows and the hand-off).
//
// Row i is first computed "raw" (without its own ScaleMatrix factor); one combined reduction then
// yields k_i (from the smallest positive raw entry) and the raw rank-one sum for row i+1.  Every
// later use multiplies by 2^(256*k_i), which is exact, so all values equal the reference's.
template <int GL, int GR>
__device__ static int junction_wave(const DevJunction& J, const double* jem, int lane, const double (&f_in)[GL],
                                    int count_in, const double* __restrict__ germ_em,
                                    const double* __restrict__ pad_trans, const double* __restrict__ pad_em,
                                    double (&g_out)[GR], double* __restrict__ fwd_out,
                                    int32_t* __restrict__ scal_out) {
  const int W = J.n_rows, nL = J.n_left, nR = J.n_right;
  int count = count_in;
  double fL[GL], fN[GR][4], fR[GR];  // raw values of the previous row
  double ntt[GR][16], nli[GR][4];    // row-invariant NTI tables of the right genes this lane owns
#pragma unroll
  for (int q = 0; q < GL; ++q) fL[q] = f_in[q];
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    fR[q] = 0.0;
    fN[q][0] = fN[q][1] = fN[q][2] = fN[q][3] = 0.0;
    const int r = lane + 64 * q;
    if (r < nR) {
#pragma unroll
      for (int u = 0; u < 16; ++u) ntt[q][u] = J.right_ntt[(size_t)r * 16 + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) nli[q][u] = J.right_gp_nli[(size_t)r * 4 + u];
    }
  }
  const size_t row_stride = (size_t)nL + 5 * (size_t)nR;
  // rank-one term of row 0: A = sum_l f_in[l] * landing_out_l[last germline-region index]
  double A;
  {
    double part = 0.0, dummy = __builtin_inf();
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      const int l = lane + 64 * q;
      if (l < nL) part += f_in[q] * J.enter_lo[l];
    }
    wave_sum_min(part, dummy);
    A = part;
  }
  int k_prev = 0;  // ScaleMatrix count of the previous row, not yet applied to fL/fN/fR
  for (int i = 0; i < W; ++i) {
    // this row's table entries for the genes the lane owns (family constants)
    double ltr[GL], llo[GL];
    int lidx[GL];
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      const int l = lane + 64 * q;
      if (l < nL) {
        ltr[q] = (i == 0) ? J.enter_trans[l] : J.left_trans[(size_t)i * nL + l];
        llo[q] = J.left_lo[(size_t)i * nL + l];
        lidx[q] = J.left_xmsa[(size_t)i * nL + l];
      }
    }
    double nlo[GR][4], rtr[GR], rli[GR];
    int ridx[GR], nx[GR][4];
#pragma unroll
    for (int q = 0; q < GR; ++q) {
      const int r = lane + 64 * q;
      if (r < nR) {
        const double* pn = J.right_nlo + ((size_t)i * nR + r) * 4;
        const int32_t* px = J.nti_xmsa + ((size_t)i * nR + r) * 4;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          nlo[q][b] = pn[b];
          nx[q][b] = px[b];
        }
        rtr[q] = J.right_trans[(size_t)i * nR + r];
        rli[q] = J.right_gp_li[(size_t)i * nR + r];
        ridx[q] = J.right_xmsa[(size_t)i * nR + r];
      }
    }
    double mp = __builtin_inf(), part = 0.0;
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      const int l = lane + 64 * q;
      if (l < nL) {
        const double f = scale_by(fL[q], k_prev);  // k_prev = 0 on row 0: the germline forward itself
        const double v = (f * ltr[q]) * (lidx[q] >= 0 ? jem[lidx[q]] : 0.0);
        fL[q] = v;
        mp = posmin(mp, v);
        part += v * llo[q];  // raw contribution to the next row's rank-one term
      }
    }
#pragma unroll
    for (int q = 0; q < GR; ++q) {
      const int r = lane + 64 * q;
      if (r < nR) {
        // previous row (zeros at i == 0), with its scaling applied now
        const double n0 = scale_by(fN[q][0], k_prev), n1 = scale_by(fN[q][1], k_prev);
        const double n2 = scale_by(fN[q][2], k_prev), n3 = scale_by(fN[q][3], k_prev);
        const double fr = scale_by(fR[q], k_prev);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          double s = ((n0 * ntt[q][b] + n1 * ntt[q][4 + b]) + n2 * ntt[q][8 + b]) + n3 * ntt[q][12 + b];
          s += A * nli[q][b];
          const double v = s * jem[nx[q][b]];
          fN[q][b] = v;
          mp = posmin(mp, v);
        }
        double s = ((n0 * nlo[q][0] + n1 * nlo[q][1]) + n2 * nlo[q][2]) + n3 * nlo[q][3];
        s += fr * rtr[q];
        s += A * rli[q];
        const double v = s * (ridx[q] >= 0 ? jem[ridx[q]] : 0.0);
        fR[q] = v;
        mp = posmin(mp, v);
      }
    }
    wave_sum_min(part, mp);
    const int k = scale_count(mp);
    A = scale_by(part, k);  // = sum_l (row i scaled)[l] * landing_out_l
    count += k;
    k_prev = k;
    if (fwd_out) {
      double* o = fwd_out + (size_t)i * row_stride;
#pragma unroll
      for (int q = 0; q < GL; ++q) {
        const int t = lane + 64 * q;
        if (t < nL) o[t] = scale_by(fL[q], k);
      }
#pragma unroll
      for (int q = 0; q < GR; ++q) {
        const int t = lane + 64 * q;
        if (t < nR) {
          o[nL + 4 * (size_t)t + 0] = scale_by(fN[q][0], k);
          o[nL + 4 * (size_t)t + 1] = scale_by(fN[q][1], k);
          o[nL + 4 * (size_t)t + 2] = scale_by(fN[q][2], k);
          o[nL + 4 * (size_t)t + 3] = scale_by(fN[q][3], k);
          o[nL + 4 * (size_t)nR + t] = scale_by(fR[q], k);
        }
      }
    }
    if (scal_out && lane == 0) scal_out[i] = count;
  }
  // hand-off into the right germline region (A already holds the last row's rank-one sum)
  double mp = __builtin_inf();
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const int r = lane + 64 * q;
    double v = 0.0;
    if (r < nR) {
      const double* xn = J.exit_nlo + (size_t)r * 4;
      const double n0 = scale_by(fN[q][0], k_prev), n1 = scale_by(fN[q][1], k_prev);
      const double n2 = scale_by(fN[q][2], k_prev), n3 = scale_by(fN[q][3], k_prev);
      double s = ((n0 * xn[0] + n1 * xn[1]) + n2 * xn[2]) + n3 * xn[3];
      s += scale_by(fR[q], k_prev) * J.exit_trans[r];
      s += A * J.exit_gp_li[r];
      v = s * germ_em[r];
      if (pad_trans) v *= pad_trans[r];
      if (pad_em) v *= pad_em[r];
      mp = posmin(mp, v);
    }
    g_out[q] = v;
  }
  const int k = scale_count(wave_min(mp));
#pragma unroll
  for (int q = 0; q < GR; ++q) g_out[q] = scale_by(g_out[q], k);
  return count + k;
}

// GA: register slots for the V genes (ceil(nV / 64)); GB: slots for the D and J genes.
template <int GA, int GB>
__global__ void __launch_bounds__(64 * kJunctionWaves)
    junction_kernel(const DevFamily fam, int n, const double* __restrict__ gem_all,
                    const int32_t* __restrict__ gcnt_all, const double* __restrict__ jem_all,
                    double* __restrict__ loglik, double* __restrict__ fwd_all, int32_t* __restrict__ scal_all) {
  extern __shared__ double jem_lds[];  // kJunctionWaves slices of n_jcols doubles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x * kJunctionWaves + wave;
  if (s >= n) return;  // whole waves leave; nothing below synchronises across waves
  const int NJ = fam.n_jcols;
  double* jem = jem_lds + (size_t)wave * NJ;
  {
    const double* src = jem_all + (size_t)s * NJ;
    for (int j = lane; j < NJ; j += 64) jem[j] = src[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int nV = fam.vgerm.n_genes, nD = fam.dgerm.n_genes, nJ = fam.jgerm.n_genes;
  const double* gem = gem_all + (size_t)s * fam.gem_size;
  const int cv = gcnt_all[(size_t)s * 3 + 0], cd = gcnt_all[(size_t)s * 3 + 1], cj = gcnt_all[(size_t)s * 3 + 2];
  double* fwd = fwd_all ? fwd_all + (size_t)s * fam.forward_size : nullptr;
  int32_t* sco = scal_all ? scal_all + (size_t)s * fam.scaler_size : nullptr;

  // initial forward over the V germline region (src/HMM.cpp:291-319)
  double gV[GA];
  double mp = __builtin_inf();
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int t = lane + 64 * q;
    double v = 0.0;
    if (t < nV) {
      v = fam.vgerm_gene_prob[t];
      v *= fam.vpadding_transition[t];
      v *= gem[t];
      v *= fam.vgerm_trans_prod[t];
      v *= gem[nV + t];
      mp = posmin(mp, v);
    }
    gV[q] = v;
  }
  int vcount = cv;
  {
    const int k = scale_count(wave_min(mp));
#pragma unroll
    for (int q = 0; q < GA; ++q) gV[q] = scale_by(gV[q], k);
    vcount += k;
  }
  if (fwd) {
#pragma unroll
    for (int q = 0; q < GA; ++q)
      if (lane + 64 * q < nV) fwd[lane + 64 * q] = gV[q];
    fwd += nV;
  }
  if (sco) {
    if (lane == 0) sco[0] = vcount;
    sco += 1;
  }

  double gJ[GB];
  int jcount;
  if (fam.has_d) {
    double gD[GB];
    const double* dgerm_em = gem + 2 * (size_t)nV;
    const double* jgerm_em = dgerm_em + nD;
    const double* jpad_em = jgerm_em + nJ;
    const int dcount =
        cd + junction_wave<GA, GB>(fam.vd, jem, lane, gV, vcount, dgerm_em, nullptr, nullptr, gD, fwd, sco);
    if (fwd) {
      fwd += (size_t)fam.vd.n_rows * (fam.vd.n_left + 5 * (size_t)fam.vd.n_right);
#pragma unroll
      for (int q = 0; q < GB; ++q)
        if (lane + 64 * q < nD) fwd[lane + 64 * q] = gD[q];
      fwd += nD;
    }
    if (sco) {
      sco += fam.vd.n_rows;
      if (lane == 0) sco[0] = dcount;
      sco += 1;
    }
    jcount = cj + junction_wave<GB, GB>(fam.dj, jem, lane, gD, dcount, jgerm_em, fam.jpadding_transition, jpad_em,
                                        gJ, fwd, sco);
    if (fwd) fwd += (size_t)fam.dj.n_rows * (fam.dj.n_left + 5 * (size_t)fam.dj.n_right);
    if (sco) sco += fam.dj.n_rows;
  } else {
    const double* jgerm_em = gem + 2 * (size_t)nV;
    const double* jpad_em = jgerm_em + nJ;
    jcount = cj + junction_wave<GA, GB>(fam.vd, jem, lane, gV, vcount, jgerm_em, fam.jpadding_transition, jpad_em,
                                        gJ, fwd, sco);
    if (fwd) fwd += (size_t)fam.vd.n_rows * (fam.vd.n_left + 5 * (size_t)fam.vd.n_right);
    if (sco) sco += fam.vd.n_rows;
  }
  if (fwd) {
#pragma unroll
    for (int q = 0; q < GB; ++q)
      if (lane + 64 * q < nJ) fwd[lane + 64 * q] = gJ[q];
  }
  if (sco && lane == 0) sco[0] = jcount;

  // HMM::LogLikelihood (src/HMM.cpp:352-353)
  double part = 0.0, dm = __builtin_inf();
#pragma unroll
  for (int q = 0; q < GB; ++q)
    if (lane + 64 * q < nJ) part += gJ[q];
  wave_sum_min(part, dm);
  if (lane == 0) loglik[s] = log(part) - jcount * kLogScaleFactor;
}

size_t forward_lds_bytes(const DevFamily& fam) {
  const size_t a = (((size_t)fam.n_xmsa + 2) & ~(size_t)1) * sizeof(double) + 2 * kFwdWaves * sizeof(int);
  const size_t b = (size_t)kJunctionWaves * fam.n_jcols * sizeof(double);
  return a > b ? a : b;
}

template <int kG>
static void launch_emission_g(const DevFamily& fam, int n, int R, const double* site_lik, const int32_t* site_scal,
                              const double* pi, const double* em_in, double* em_out, double* gem, int32_t* gcnt,
                              double* jem, hipStream_t stream) {
  const size_t lds = (((size_t)fam.n_xmsa + 2) & ~(size_t)1) * sizeof(double) + 2 * kFwdWaves * sizeof(int);
  if (site_lik) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(emission_kernel<kG, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((emission_kernel<kG, true>), dim3(n), dim3(kFwdThreads), lds, stream, fam, R, site_lik,
                       site_scal, pi, em_in, em_out, gem, gcnt, jem);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(emission_kernel<kG, false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((emission_kernel<kG, false>), dim3(n), dim3(kFwdThreads), lds, stream, fam, R, site_lik,
                       site_scal, pi, em_in, em_out, gem, gcnt, jem);
  }
}

template <int GA, int GB>
static void launch_junction_g(const DevFamily& fam, int n, const double* gem, const int32_t* gcnt, const double* jem,
                              double* loglik, double* forward_out, int32_t* scaler_out, hipStream_t stream) {
  const size_t lds = (size_t)kJunctionWaves * fam.n_jcols * sizeof(double);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(junction_kernel<GA, GB>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((junction_kernel<GA, GB>), dim3((n + kJunctionWaves - 1) / kJunctionWaves),
                     dim3(64 * kJunctionWaves), lds, stream, fam, n, gem, gcnt, jem, loglik, forward_out, scaler_out);
}

template <int GA>
static void launch_junction_a(int gb, const DevFamily& fam, int n, const double* gem, const int32_t* gcnt,
                              const double* jem, double* loglik, double* forward_out, int32_t* scaler_out,
                              hipStream_t stream) {
  if (gb <= 1)
    launch_junction_g<GA, 1>(fam, n, gem, gcnt, jem, loglik, forward_out, scaler_out, stream);
  else if (gb <= 2)
    launch_junction_g<GA, 2>(fam, n, gem, gcnt, jem, loglik, forward_out, scaler_out, stream);
  else
    launch_junction_g<GA, 4>(fam, n, gem, gcnt, jem, loglik, forward_out, scaler_out, stream);
}

// site_lik != null: emissions are assembled from K1's output (em_out optional);
// site_lik == null: emissions are taken from em_in (SimpleHMM / lh_forward_batch).
// gem [n][gem_size], gcnt [n][3], jem [n][n_jcols]: per-sample hand-off buffers between K2a and K2b.
void launch_forward(const DevFamily& fam, int n, int R, const double* site_lik, const int32_t* site_scal,
                    const double* pi, const double* em_in, double* em_out, double* gem, int32_t* gcnt, double* jem,
                    double* loglik, double* forward_out, int32_t* scaler_out, hipStream_t stream) {
  const int slots = (fam.max_genes + kFwdThreads - 1) / kFwdThreads;
#define LH_ARGS fam, n, R, site_lik, site_scal, pi, em_in, em_out, gem, gcnt, jem, stream
  if (slots <= 1)
    launch_emission_g<1>(LH_ARGS);
  else if (slots <= 2)
    launch_emission_g<2>(LH_ARGS);
  else
    launch_emission_g<4>(LH_ARGS);
#undef LH_ARGS
  const int ga = (fam.vgerm.n_genes + 63) / 64;
  const int gb = (std::max(fam.dgerm.n_genes, fam.jgerm.n_genes) + 63) / 64;
#define LH_ARGS gb, fam, n, gem, gcnt, jem, loglik, forward_out, scaler_out, stream
  if (ga <= 1)
    launch_junction_a<1>(LH_ARGS);
  else if (ga <= 2)
    launch_junction_a<2>(LH_ARGS);
  else if (ga <= 4)
    launch_junction_a<4>(LH_ARGS);
  else if (ga <= 8)
    launch_junction_a<8>(LH_ARGS);
  else
    launch_junction_a<16>(LH_ARGS);
#undef LH_ARGS
}

}  // namespace lh
