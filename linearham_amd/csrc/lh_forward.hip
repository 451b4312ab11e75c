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

// Block-wide maximum of kS counters at once: shuffles inside the wave, one LDS exchange, one barrier
// (red holds 2 * kFwdWaves * kS ints; `phase` alternates its halves so that a reduction never
// overwrites values another wave is still reading).
template <int kS>
__device__ static inline void block_max_ints(int (&v)[kS], int* red, int phase) {
#pragma unroll
  for (int i = 0; i < kS; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] = max(v[i], __shfl_xor(v[i], off, 64));
  }
  int* r = red + phase * kFwdWaves * kS;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < kS; ++i) r[(threadIdx.x >> 6) * kS + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kS; ++i) v[i] = max(max(r[i], r[kS + i]), max(r[2 * kS + i], r[3 * kS + i]));
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

// The emission vectors of the kS samples a workgroup handles are interleaved in LDS: em[c * kS + i].
template <int kS>
__device__ static inline void load_em(const double* em, int idx, double (&e)[kS]) {
  if constexpr (kS == 4) {
    const double2* p = reinterpret_cast<const double2*>(em + (size_t)idx * 4);
    const double2 a = p[0], b = p[1];
    e[0] = a.x;
    e[1] = a.y;
    e[2] = b.x;
    e[3] = b.y;
  } else if constexpr (kS == 2) {
    const double2 a = *reinterpret_cast<const double2*>(em + (size_t)idx * 2);
    e[0] = a.x;
    e[1] = a.y;
  } else {
    e[0] = em[idx];
  }
}

// FillGermlinePaddingEmission (src/PhyloHMM.cpp:158-193): per gene the running product of its
// columns' emissions with ScaleMatrix after every factor, then the 2^(256*d) equalisation to the
// region's largest scaler count (added to cnt[i]).  Thread `tid` owns genes tid + 256*q; the products of
// sample i go to out[i][gene].  The index stream (one 16-byte load = eight factors of one gene) is
// shared by the kS samples: it is the dominant L2 traffic of this kernel.
template <int kG, int kS>
__device__ static void fill_segments(const DevSegments& seg, const double* em, int tid, double* const (&out)[kS],
                                     int* redi, int phase, int (&cnt)[kS]) {
  double v[kG][kS];
  int c[kG][kS];
#pragma unroll
  for (int q = 0; q < kG; ++q) {
#pragma unroll
    for (int i = 0; i < kS; ++i) {
      v[q][i] = 1.0;
      c[q][i] = 0;
    }
  }
  const int n = seg.n_genes;
  if (n > 0) {
#pragma unroll
    for (int q = 0; q < kG; ++q) {
      // lanes beyond the last gene shadow gene n-1 (loads stay unpredicated; their result is dropped and
      // cannot change the maximum); a wave made only of such lanes skips the walk.
      const int g = min(tid + kFwdThreads * q, n - 1);
      if ((tid & ~63) + kFwdThreads * q >= n) continue;
      const uint4* chunk = seg.inds_c + g;
      for (int j = 0; j < seg.n_chunks; ++j) {
        const uint4 w = chunk[(size_t)j * n];
        const unsigned packed[4] = {w.x, w.y, w.z, w.w};
        double e[8][kS];
#pragma unroll
        for (int u = 0; u < 8; ++u) load_em<kS>(em, (packed[u >> 1] >> (16 * (u & 1))) & 0xffffu, e[u]);
#pragma unroll
        for (int i = 0; i < kS; ++i) {
          // The eight factors are applied without looking at the threshold, tracking the smallest
          // prefix product m.  ScaleMatrix multiplies by 2^256 (exact) until the value is back above
          // 2^-256, so after the chunk the reference holds p * 2^(256 k) with k = the number of
          // rescalings the smallest prefix needs -- provided no unscaled prefix came near the
          // subnormal range, which m also tells.
          const double v0 = v[q][i];
          double p = v0, m = v0;
#pragma unroll
          for (int u = 0; u < 8; ++u) {  // padded factors are exactly 1.0: no effect on (v, c)
            p *= e[u][i];
            m = fmin(m, p);
          }
          if (m >= 0x1p-768) {
            const int k = (m < kScaleThreshold) + (m < 0x1p-512);
            v[q][i] = p * (k == 0 ? 1.0 : k == 1 ? 0x1p256 : 0x1p512);
            c[q][i] += k;
          } else {  // a zero, or a drop of more than 2^-512 inside one chunk: step by step
            double x = v0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              x *= e[u][i];
              while (x > 0.0 && x < kScaleThreshold) {
                x *= kScaleFactor;
                ++c[q][i];
              }
            }
            v[q][i] = x;
          }
        }
      }
    }
  }
  int mx[kS];
#pragma unroll
  for (int i = 0; i < kS; ++i) {
    mx[i] = 0;
#pragma unroll
    for (int q = 0; q < kG; ++q)
      if (tid + kFwdThreads * q < n) mx[i] = max(mx[i], c[q][i]);
  }
  block_max_ints<kS>(mx, redi, phase);
#pragma unroll
  for (int i = 0; i < kS; ++i) {
#pragma unroll
    for (int q = 0; q < kG; ++q) {
      const int g = tid + kFwdThreads * q;
      if (g < n) out[i][g] = v[q][i] * pow_scale(mx[i] - c[q][i]);
    }
    cnt[i] += mx[i];
  }
}

// kS consecutive samples per workgroup (the last workgroup repeats sample n-1 and rewrites identical
// values).
template <int kG, int kS, bool kFromSiteLik>
__global__ void __launch_bounds__(kFwdThreads)
    emission_kernel(const DevFamily fam, int n, int R, const double* __restrict__ site_lik,
                    const int32_t* __restrict__ site_scal, const double* __restrict__ pi,
                    const double* __restrict__ em_in, double* __restrict__ em_out, double* __restrict__ gem_all,
                    int32_t* __restrict__ gcnt_all, double* __restrict__ jem_all) {
  extern __shared__ double em[];  // [(C + 1) * kS] interleaved emissions (column C = 1.0) | reduction scratch
  const int tid = threadIdx.x;
  const int C = fam.n_xmsa;
  int* redi = reinterpret_cast<int*>(em + (size_t)(C + 1) * kS);  // 2 * kFwdWaves * kS ints
  size_t smp[kS];
#pragma unroll
  for (int i = 0; i < kS; ++i) smp[i] = (size_t)min(blockIdx.x * kS + i, n - 1);

  if constexpr (kFromSiteLik) {
    // PhyloHMM::FillXmsaEmission tail: mix the rate categories (equal weights, scalers aligned to the
    // smallest one) and apply the naive correction.
    const int L = fam.n_sites;
    const double w = 1.0 / R;
    for (int t = tid; t < C; t += kFwdThreads) {
      const int c = fam.xmsa_col[t];
      const int site = fam.xmsa_site[t];
      const int b = fam.xmsa_naive_base[t];
#pragma unroll
      for (int i = 0; i < kS; ++i) {
        const size_t s = smp[i];
        int smin = 0x7fffffff;
        for (int r = 0; r < R; ++r) smin = min(smin, site_scal[(s * R + r) * L + site]);
        double acc = 0.0;
        for (int r = 0; r < R; ++r) {
          double v = site_lik[((s * R + r) * 5 + b) * (size_t)L + site];
          const int d = site_scal[(s * R + r) * L + site] - smin;
          for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
          acc += w * v;
        }
        // The reference forms exp(log(site_lik) - smin*log(2^256) - log(pi_b)) (src/PhyloHMM.cpp:226-237);
        // the same quantity is computed here without the log/exp round trip (two FP64 transcendentals
        // per column): site_lik / pi_b scaled down by 2^(256*smin), which also underflows to 0 like exp().
        double e = acc;
        if (b != 4) e /= pi[s * 4 + b];
        for (int q = 0; q < smin && e != 0.0; ++q) e *= kScaleThreshold;
        em[(size_t)c * kS + i] = e;
        if (em_out) em_out[s * C + c] = e;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < kS; ++i)
      for (int c = tid; c < C; c += kFwdThreads) em[(size_t)c * kS + i] = em_in[smp[i] * C + c];
  }
  if (tid < kS) em[(size_t)C * kS + tid] = 1.0;
  __syncthreads();

  // emissions of the columns the junction rows touch, compacted for K2b
  for (int j = tid; j < fam.n_jcols; j += kFwdThreads) {
    const int c = fam.jcols[j];
#pragma unroll
    for (int i = 0; i < kS; ++i) jem_all[smp[i] * fam.n_jcols + j] = em[(size_t)c * kS + i];
  }

  // [vpadding nV | vgerm nV | dgerm nD | jgerm nJ | jpadding nJ]
  const int nV = fam.vgerm.n_genes, nD = fam.dgerm.n_genes, nJ = fam.jgerm.n_genes;
  double* o[kS];
  int cv[kS], cd[kS], cj[kS];
#pragma unroll
  for (int i = 0; i < kS; ++i) {
    o[i] = gem_all + smp[i] * fam.gem_size;
    cv[i] = cd[i] = cj[i] = 0;
  }
  auto advance = [&](int by) {
#pragma unroll
    for (int i = 0; i < kS; ++i) o[i] += by;
  };
  fill_segments<kG, kS>(fam.vpadding, em, tid, o, redi, 0, cv);
  advance(nV);
  fill_segments<kG, kS>(fam.vgerm, em, tid, o, redi, 1, cv);
  advance(nV);
  if (fam.has_d) {
    fill_segments<kG, kS>(fam.dgerm, em, tid, o, redi, 0, cd);
    advance(nD);
    fill_segments<kG, kS>(fam.jgerm, em, tid, o, redi, 1, cj);
    advance(nJ);
    fill_segments<kG, kS>(fam.jpadding, em, tid, o, redi, 0, cj);
  } else {
    fill_segments<kG, kS>(fam.jgerm, em, tid, o, redi, 0, cj);
    advance(nJ);
    fill_segments<kG, kS>(fam.jpadding, em, tid, o, redi, 1, cj);
  }
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < kS; ++i) {
      gcnt_all[smp[i] * 3 + 0] = cv[i];
      gcnt_all[smp[i] * 3 + 1] = cd[i];
      gcnt_all[smp[i] * 3 + 2] = cj[i];
    }
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

__device__ static inline double wave_sum(double sum) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  return sum;
}

__device__ static inline double wave_min(double mn) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_xor(mn, off, 64));
  return mn;
}

// ScaleMatrix on a wave-uniform basis.  `minpos` (identical in all lanes, +inf if nothing is positive)
// is the smallest positive entry of the vector; the reference multiplies the whole vector by 2^256
// until that entry reaches 2^-256, i.e. k = #{j in 1..4 : minpos < 2^(-256 j)} times (a double is never
// below 2^-1074, so k <= 4).  k is derived from the exponent field with scalar instructions, and the
// k multiplications -- each exact -- collapse into one by 2^(256 min(k,3)) plus, for k = 4 only
// (subnormal minpos), one more by 2^256.
struct RowScale {
  int k;
  double factor;  // 2^(256 * min(k, 3))
  bool extra;     // k == 4
  __device__ double apply(double v) const {
    v *= factor;
    if (extra) v *= kScaleFactor;
    return v;
  }
};

__device__ static inline RowScale row_scale(double minpos) {
  const unsigned hw = __builtin_amdgcn_readfirstlane(__double2hiint(minpos));
  const unsigned e = (hw >> 20) & 0x7ffu;
  RowScale s;
  s.extra = (e == 0) && ((hw & 0xfffffu) < (1u << 18));  // minpos < 2^-1024
  const int k3 = (e < 767u) + (e < 511u) + (e < 255u);
  s.k = k3 + (s.extra ? 1 : 0);
  s.factor = __hiloint2double((1023 + 256 * k3) << 20, 0);
  return s;
}

// One junction region + the germline region to its right; all state in registers of one wave.
//   f_in[q]   forward of the left germline region (genes lane + 64q; zero beyond the last gene)
//   g_out[q]  forward of the right germline region (likewise)
//   germ_em   emission products of the right germline region (per sample, global),
//   pad_trans / pad_em   padding transition (family) and padding emission products (sample); null = ones
// Returns count_in plus every ScaleMatrix count taken inside (junction rows and the hand-off).
//
// Row i is first computed "raw" (without its own ScaleMatrix factor); one combined reduction then
// yields k_i (from the smallest positive raw entry) and the raw rank-one sum for row i+1.  Every
// later use multiplies by 2^(256*k_i), which is exact, so all values equal the reference's.
// The tables are padded to whole waves with entries that produce zeros (see DevJunction), so the row
// body has no per-lane predicates.
template <int GL, int GR>
__device__ static int junction_wave(const DevJunction& J, const double* jem, const double* ntt_lds, int lane,
                                    const double (&f_in)[GL], int count_in, const double* __restrict__ germ_em,
                                    const double* __restrict__ pad_trans, const double* __restrict__ pad_em,
                                    double (&g_out)[GR], double* __restrict__ fwd_out,
                                    int32_t* __restrict__ scal_out) {
  const int W = J.n_rows, nL = J.n_left, nR = J.n_right;
  const unsigned pL = J.left_pad, pR = J.right_pad;
  int count = count_in;
  double fL[GL], fN[GR][4], fR[GR];  // raw values of the previous row
  double nli[GR][4];                 // row-invariant NTI landing table of the right genes this lane owns
#pragma unroll
  for (int q = 0; q < GL; ++q) fL[q] = f_in[q];
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    fR[q] = 0.0;
    fN[q][0] = fN[q][1] = fN[q][2] = fN[q][3] = 0.0;
    const double2* p = reinterpret_cast<const double2*>(J.right_gp_nli) + 2u * (lane + 64u * q);
    const double2 a = p[0], b = p[1];
    nli[q][0] = a.x;
    nli[q][1] = a.y;
    nli[q][2] = b.x;
    nli[q][3] = b.y;
  }
  const size_t row_stride = (size_t)nL + 5 * (size_t)nR;
  // rank-one term of row 0: A = sum_l f_in[l] * landing_out_l[last germline-region index]
  double A;
  {
    double part = 0.0;
#pragma unroll
    for (int q = 0; q < GL; ++q) part += f_in[q] * J.enter_lo[lane + 64u * q];
    A = wave_sum(part);
  }
  RowScale prev{0, 1.0, false};  // ScaleMatrix factor of the previous row, not yet applied to fL/fN/fR
  for (int i = 0; i < W; ++i) {
    // this row's table entries for the genes the lane owns (family constants)
    double ltr[GL], llo[GL];
    int lidx[GL];
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      const unsigned o = (unsigned)i * pL + lane + 64u * q;
      ltr[q] = J.left_trans[o];
      llo[q] = J.left_lo[o];
      lidx[q] = J.left_xmsa[o];
    }
    double nlo[GR][4], rtr[GR], rli[GR];
    int ridx[GR];
    int4 nx[GR];
#pragma unroll
    for (int q = 0; q < GR; ++q) {
      const unsigned o = (unsigned)i * pR + lane + 64u * q;
      const double2* pn = reinterpret_cast<const double2*>(J.right_nlo) + 2u * o;
      const double2 a = pn[0], b = pn[1];
      nlo[q][0] = a.x;
      nlo[q][1] = a.y;
      nlo[q][2] = b.x;
      nlo[q][3] = b.y;
      nx[q] = reinterpret_cast<const int4*>(J.nti_xmsa)[o];
      rtr[q] = J.right_trans[o];
      rli[q] = J.right_gp_li[o];
      ridx[q] = J.right_xmsa[o];
    }
    double mp = __builtin_inf(), part = 0.0;
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      const double f = prev.apply(fL[q]);  // factor 1 on row 0: the germline forward itself
      const double v = (f * ltr[q]) * jem[lidx[q]];
      fL[q] = v;
      mp = posmin(mp, v);
      part += v * llo[q];  // raw contribution to the next row's rank-one term
    }
#pragma unroll
    for (int q = 0; q < GR; ++q) {
      // previous row (zeros at i == 0), with its scaling applied now
      const double n0 = prev.apply(fN[q][0]), n1 = prev.apply(fN[q][1]);
      const double n2 = prev.apply(fN[q][2]), n3 = prev.apply(fN[q][3]);
      const double fr = prev.apply(fR[q]);
      // NTI->NTI block of gene r, transposed in LDS: [b * 4 + a] = transition a -> b
      const double2* tt = reinterpret_cast<const double2*>(ntt_lds) + 8u * (lane + 64u * q);
      const int nxs[4] = {nx[q].x, nx[q].y, nx[q].z, nx[q].w};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double2 t01 = tt[2 * b], t23 = tt[2 * b + 1];
        double s = ((n0 * t01.x + n1 * t01.y) + n2 * t23.x) + n3 * t23.y;
        s += A * nli[q][b];
        const double v = s * jem[nxs[b]];
        fN[q][b] = v;
        mp = posmin(mp, v);
      }
      double s = ((n0 * nlo[q][0] + n1 * nlo[q][1]) + n2 * nlo[q][2]) + n3 * nlo[q][3];
      s += fr * rtr[q];
      s += A * rli[q];
      const double v = s * jem[ridx[q]];
      fR[q] = v;
      mp = posmin(mp, v);
    }
    wave_sum_min(part, mp);
    const RowScale cur = row_scale(mp);
    A = cur.apply(part);  // = sum_l (row i scaled)[l] * landing_out_l
    count += cur.k;
    prev = cur;
    if (fwd_out) {
      double* o = fwd_out + (size_t)i * row_stride;
#pragma unroll
      for (int q = 0; q < GL; ++q) {
        const int t = lane + 64 * q;
        if (t < nL) o[t] = cur.apply(fL[q]);
      }
#pragma unroll
      for (int q = 0; q < GR; ++q) {
        const int t = lane + 64 * q;
        if (t < nR) {
          o[nL + 4 * (size_t)t + 0] = cur.apply(fN[q][0]);
          o[nL + 4 * (size_t)t + 1] = cur.apply(fN[q][1]);
          o[nL + 4 * (size_t)t + 2] = cur.apply(fN[q][2]);
          o[nL + 4 * (size_t)t + 3] = cur.apply(fN[q][3]);
          o[nL + 4 * (size_t)nR + t] = cur.apply(fR[q]);
        }
      }
    }
    if (scal_out && lane == 0) scal_out[i] = count;
  }
  // hand-off into the right germline region (A already holds the last row's rank-one sum)
  double mp = __builtin_inf();
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const unsigned r = lane + 64u * q;
    const double2* xn = reinterpret_cast<const double2*>(J.exit_nlo) + 2u * r;
    const double2 x01 = xn[0], x23 = xn[1];
    const double n0 = prev.apply(fN[q][0]), n1 = prev.apply(fN[q][1]);
    const double n2 = prev.apply(fN[q][2]), n3 = prev.apply(fN[q][3]);
    double s = ((n0 * x01.x + n1 * x01.y) + n2 * x23.x) + n3 * x23.y;
    s += prev.apply(fR[q]) * J.exit_trans[r];
    s += A * J.exit_gp_li[r];
    double v = 0.0;
    if ((int)r < nR) {
      v = s * germ_em[r];
      if (pad_trans) v *= pad_trans[r];
      if (pad_em) v *= pad_em[r];
    }
    mp = posmin(mp, v);
    g_out[q] = v;
  }
  const RowScale last = row_scale(wave_min(mp));
#pragma unroll
  for (int q = 0; q < GR; ++q) g_out[q] = last.apply(g_out[q]);
  return count + last.k;
}

// GA: register slots for the V genes (ceil(nV / 64)); GB: slots for the D and J genes.
template <int GA, int GB>
__global__ void __launch_bounds__(64 * kJunctionWaves)
    junction_kernel(const DevFamily fam, int n, const double* __restrict__ gem_all,
                    const int32_t* __restrict__ gcnt_all, const double* __restrict__ jem_all,
                    double* __restrict__ loglik, double* __restrict__ fwd_all, int32_t* __restrict__ scal_all) {
  // [NTI->NTI blocks of the vd right genes | same for dj | kJunctionWaves slices of n_jcols + 1 doubles]
  extern __shared__ double jlds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x * kJunctionWaves + wave;
  const int NJ = fam.n_jcols;
  double* ntt_vd = jlds;
  double* ntt_dj = ntt_vd + 16 * (size_t)fam.vd.right_pad;
  double* jem = ntt_dj + (fam.has_d ? 16 * (size_t)fam.dj.right_pad : 0) + (size_t)wave * (NJ + 1);
  for (int t = threadIdx.x; t < 16 * fam.vd.right_pad; t += 64 * kJunctionWaves) ntt_vd[t] = fam.vd.right_ntt[t];
  if (fam.has_d)
    for (int t = threadIdx.x; t < 16 * fam.dj.right_pad; t += 64 * kJunctionWaves) ntt_dj[t] = fam.dj.right_ntt[t];
  __syncthreads();
  if (s >= n) return;  // whole waves leave; nothing below synchronises across waves
  {
    const double* src = jem_all + (size_t)s * NJ;
    for (int j = lane; j < NJ; j += 64) jem[j] = src[j];
    if (lane == 0) jem[NJ] = 0.0;  // what a state that cannot emit at a site looks up
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
    const RowScale sc = row_scale(wave_min(mp));
#pragma unroll
    for (int q = 0; q < GA; ++q) gV[q] = sc.apply(gV[q]);
    vcount += sc.k;
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
    const int dcount = cd + junction_wave<GA, GB>(fam.vd, jem, ntt_vd, lane, gV, vcount, dgerm_em, nullptr, nullptr,
                                                  gD, fwd, sco);
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
    jcount = cj + junction_wave<GB, GB>(fam.dj, jem, ntt_dj, lane, gD, dcount, jgerm_em, fam.jpadding_transition,
                                        jpad_em, gJ, fwd, sco);
    if (fwd) fwd += (size_t)fam.dj.n_rows * (fam.dj.n_left + 5 * (size_t)fam.dj.n_right);
    if (sco) sco += fam.dj.n_rows;
  } else {
    const double* jgerm_em = gem + 2 * (size_t)nV;
    const double* jpad_em = jgerm_em + nJ;
    jcount = cj + junction_wave<GA, GB>(fam.vd, jem, ntt_vd, lane, gV, vcount, jgerm_em, fam.jpadding_transition,
                                        jpad_em, gJ, fwd, sco);
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
  double part = 0.0;
#pragma unroll
  for (int q = 0; q < GB; ++q) part += gJ[q];  // zero beyond the last J gene
  part = wave_sum(part);
  if (lane == 0) loglik[s] = log(part) - jcount * kLogScaleFactor;
}

static size_t junction_lds_bytes(const DevFamily& fam) {
  return ((size_t)kJunctionWaves * (fam.n_jcols + 1) +
          16 * ((size_t)fam.vd.right_pad + (fam.has_d ? fam.dj.right_pad : 0))) *
         sizeof(double);
}

static size_t emission_lds_bytes(const DevFamily& fam, int ks) {
  return ((size_t)fam.n_xmsa + 1) * ks * sizeof(double) + 2 * kFwdWaves * ks * sizeof(int);
}

// Samples per K2a workgroup: as many (4, 2, 1) as keep three workgroups resident per CU.
static int emission_samples_per_group(const DevFamily& fam) {
  constexpr size_t kBudget = 52 * 1024;
  return emission_lds_bytes(fam, 4) <= kBudget ? 4 : emission_lds_bytes(fam, 2) <= kBudget ? 2 : 1;
}

size_t forward_lds_bytes(const DevFamily& fam) {
  const size_t a = emission_lds_bytes(fam, emission_samples_per_group(fam));
  const size_t b = junction_lds_bytes(fam);
  return a > b ? a : b;
}

template <int kG, int kS>
static void launch_emission_gs(const DevFamily& fam, int n, int R, const double* site_lik, const int32_t* site_scal,
                               const double* pi, const double* em_in, double* em_out, double* gem, int32_t* gcnt,
                               double* jem, hipStream_t stream) {
  const size_t lds = emission_lds_bytes(fam, kS);
  const dim3 grid((n + kS - 1) / kS);
  if (site_lik) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(emission_kernel<kG, kS, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((emission_kernel<kG, kS, true>), grid, dim3(kFwdThreads), lds, stream, fam, n, R, site_lik,
                       site_scal, pi, em_in, em_out, gem, gcnt, jem);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(emission_kernel<kG, kS, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((emission_kernel<kG, kS, false>), grid, dim3(kFwdThreads), lds, stream, fam, n, R, site_lik,
                       site_scal, pi, em_in, em_out, gem, gcnt, jem);
  }
}

template <int kG>
static void launch_emission_g(const DevFamily& fam, int n, int R, const double* site_lik, const int32_t* site_scal,
                              const double* pi, const double* em_in, double* em_out, double* gem, int32_t* gcnt,
                              double* jem, hipStream_t stream) {
  // more gene slots per lane leave fewer registers for samples (kG * kS chains are live at once)
  int ks = std::min(emission_samples_per_group(fam), kG == 1 ? 4 : kG == 2 ? 2 : 1);
  if (getenv("LH_K2A_KS")) ks = std::min(ks, atoi(getenv("LH_K2A_KS")));
#define LH_ARGS fam, n, R, site_lik, site_scal, pi, em_in, em_out, gem, gcnt, jem, stream
  if constexpr (kG == 1) {
    if (ks == 4) return launch_emission_gs<kG, 4>(LH_ARGS);
  }
  if constexpr (kG <= 2) {
    if (ks >= 2) return launch_emission_gs<kG, 2>(LH_ARGS);
  }
  launch_emission_gs<kG, 1>(LH_ARGS);
#undef LH_ARGS
}

template <int GA, int GB>
static void launch_junction_g(const DevFamily& fam, int n, const double* gem, const int32_t* gcnt, const double* jem,
                              double* loglik, double* forward_out, int32_t* scaler_out, hipStream_t stream) {
  const size_t lds = junction_lds_bytes(fam);
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
