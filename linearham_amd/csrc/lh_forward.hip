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
#include <cstdio>
#include <cstdlib>

#include "lh_device.h"

namespace lh {

constexpr int kFwdThreads = 256;
constexpr int kFwdWaves = kFwdThreads / 64;
constexpr int kJunctionWaves = 4;  // samples per K2b workgroup

// Block-wide maximum: shuffles inside the wave, one LDS exchange, one barrier (red holds
// 2 * kFwdWaves ints; `phase` alternates its halves so that a reduction never overwrites values
// another wave is still reading).
__device__ static inline int block_max_int(int v, int* red, int phase) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  int* r = red + phase * kFwdWaves;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
  __syncthreads();
  return max(max(r[0], r[1]), max(r[2], r[3]));
}

__device__ static inline int block_min_int(int v, int* red, int phase) {
  return -block_max_int(-v, red, phase);
}

__device__ static inline double pow_scale(int d) {  // std::pow(SCALE_FACTOR, d), src/PhyloHMM.cpp:191
  return d <= 0 ? 1.0 : d == 1 ? 0x1p256 : d == 2 ? 0x1p512 : d == 3 ? 0x1p768 : __builtin_inf();
}

// ---------------------------------------------------------------------------------------------------
// K2a
// ---------------------------------------------------------------------------------------------------

// FillGermlinePaddingEmission (src/PhyloHMM.cpp:158-193): per gene the running product of its
// columns' emissions with ScaleMatrix after every factor, then the 2^(256*d) equalisation to the
// region's largest scaler count (returned).  Thread `tid` owns genes tid + 256*q; the products go to
// out[gene].  One 16-byte load brings a gene's next eight column indices.
template <int kG, bool kByteOff>
__device__ static int fill_segments(const DevSegments& seg, const double* em, int tid, double* __restrict__ out,
                                    int* redi, int phase) {
  double v[kG];
  int c[kG];
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    v[q] = 1.0;
    c[q] = 0;
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
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned x = (packed[u >> 1] >> (16 * (u & 1))) & 0xffffu;
          // families with at most 8190 distinct columns store byte offsets (index * 8): no shift here
          e[u] = kByteOff ? *reinterpret_cast<const double*>(reinterpret_cast<const char*>(em) + x) : em[x];
        }
        // The eight factors are applied without looking at the threshold, tracking the smallest prefix
        // product m.  ScaleMatrix multiplies by 2^256 (exact) until the value is back above 2^-256, so
        // after the chunk the reference holds p * 2^(256 k) with k = the number of rescalings the
        // smallest prefix needs -- provided no unscaled prefix came near the subnormal range, which m
        // also tells.
        const double v0 = v[q];
        double p = v0, m = v0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // padded factors are exactly 1.0: no effect on (v, c)
          p *= e[u];
          m = fmin(m, p);
        }
        if (__ballot(!(m >= kScaleThreshold)) == 0) {
          v[q] = p;  // no gene of this wave crossed the threshold in this chunk (the usual case)
        } else if (m >= 0x1p-768) {
          const int k = (m < kScaleThreshold) + (m < 0x1p-512);
          v[q] = p * (k == 0 ? 1.0 : k == 1 ? 0x1p256 : 0x1p512);
          c[q] += k;
        } else {  // a zero, or a drop of more than 2^-512 inside one chunk: step by step
          double x = v0;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            x *= e[u];
            while (x > 0.0 && x < kScaleThreshold) {
              x *= kScaleFactor;
              ++c[q];
            }
          }
          v[q] = x;
        }
      }
    }
  }
  int local_max = 0;
#pragma unroll
  for (int q = 0; q < kG; ++q)
    if (tid + kFwdThreads * q < n) local_max = max(local_max, c[q]);
  const int mx = block_max_int(local_max, redi, phase);
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    const int g = tid + kFwdThreads * q;
    if (g < n) out[g] = v[q] * pow_scale(mx - c[q]);
  }
  return mx;
}

// fill_segments for a set of at most 64 genes, run by ONE wave (gene = lane): the largest count is taken inside
// the wave, no LDS, no barrier -- the three small sets of a heavy-chain family (D, J, J padding) then walk side by
// side on three waves instead of one after the other on all four.
template <bool kByteOff>
__device__ static int fill_segments_wave(const DevSegments& seg, const double* em, int lane, double* __restrict__ out) {
  const int n = seg.n_genes;
  double v = 1.0;
  int c = 0;
  if (n > 0) {
    const int g = min(lane, n - 1);  // lanes beyond the last gene shadow gene n-1 (result dropped)
    const uint4* chunk = seg.inds_c + g;
    const int nc = seg.n_chunks;
    uint4 w_next = nc > 0 ? chunk[0] : uint4{0, 0, 0, 0};
    for (int j = 0; j < nc; ++j) {
      const uint4 w = w_next;
      if (j + 1 < nc) w_next = chunk[(size_t)(j + 1) * n];  // the next chunk's indices travel while this one is applied
      const unsigned packed[4] = {w.x, w.y, w.z, w.w};
      double e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned x = (packed[u >> 1] >> (16 * (u & 1))) & 0xffffu;
        e[u] = kByteOff ? *reinterpret_cast<const double*>(reinterpret_cast<const char*>(em) + x) : em[x];
      }
      const double v0 = v;
      double p = v0, m = v0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // (see fill_segments for the three cases)
        p *= e[u];
        m = fmin(m, p);
      }
      if (__ballot(!(m >= kScaleThreshold)) == 0) {
        v = p;
      } else if (m >= 0x1p-768) {
        const int k = (m < kScaleThreshold) + (m < 0x1p-512);
        v = p * (k == 0 ? 1.0 : k == 1 ? 0x1p256 : 0x1p512);
        c += k;
      } else {
        double x = v0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          x *= e[u];
          while (x > 0.0 && x < kScaleThreshold) {
            x *= kScaleFactor;
            ++c;
          }
        }
        v = x;
      }
    }
  }
  int mx = lane < n ? c : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
  if (lane < n) out[lane] = v * pow_scale(mx - c);
  return mx;
}

// Extended-range mode (lh_family_set_extended_range): every factor is a pair (em[x], ems[x]) = value and
// 2^-256 count of the column's emission; the running product carries the counts along, and the region is
// equalised to its SMALLEST count -- alleles more than 2^-1024 below the best one underflow to 0 -- instead of
// to the largest as the reference does (where the likely alleles overflow to inf, src/PhyloHMM.cpp:190-192).
template <int kG, bool kByteOff>
__device__ static int fill_segments_ext(const DevSegments& seg, const double* em, const int* ems, int tid,
                                        double* __restrict__ out, int* redi, int phase) {
  double v[kG];
  int c[kG];
  const int n = seg.n_genes;
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    v[q] = 1.0;
    c[q] = 0;
    const int g = tid + kFwdThreads * q;
    if (g >= n) continue;
    const uint4* chunk = seg.inds_c + g;
    double x = 1.0;
    int k = 0;
    for (int j = 0; j < seg.n_chunks; ++j) {
      const uint4 w = chunk[(size_t)j * n];
      const unsigned packed[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned idx = ((packed[u >> 1] >> (16 * (u & 1))) & 0xffffu) >> (kByteOff ? 3 : 0);
        x *= em[idx];
        k += ems[idx];
        while (x > 0.0 && x < kScaleThreshold) {
          x *= kScaleFactor;
          ++k;
        }
      }
    }
    v[q] = x;
    c[q] = k;
  }
  int local_min = 0x3fffffff;
#pragma unroll
  for (int q = 0; q < kG; ++q)
    if (tid + kFwdThreads * q < n && v[q] > 0.0) local_min = min(local_min, c[q]);
  int mn = block_min_int(local_min, redi, phase);
  if (mn == 0x3fffffff) mn = 0;  // every product is zero (or the set is empty)
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    const int g = tid + kFwdThreads * q;
    if (g < n) {
      double x = v[q];
      for (int d = c[q] - mn; d > 0 && x != 0.0; --d) x *= kScaleThreshold;
      out[g] = x;
    }
  }
  return mn;
}

// capacity of the consensus form's LDS arrays: the largest set's sites + 2 (the prefix past the last site and
// the reciprocal of the padding position), even
__host__ __device__ static inline int cons_capacity(const DevFamily& fam) {
  int m = fam.vpadding.cons_sites;
  m = m > fam.vgerm.cons_sites ? m : fam.vgerm.cons_sites;
  m = m > fam.dgerm.cons_sites ? m : fam.dgerm.cons_sites;
  m = m > fam.jgerm.cons_sites ? m : fam.jgerm.cons_sites;
  m = m > fam.jpadding.cons_sites ? m : fam.jpadding.cons_sites;
  return m > 0 ? (m + 3) & ~1 : 0;
}

// (v, k) = the reference's running product after ScaleMatrix: value v * 2^(-256 k), v kept in [2^-256, 1]
struct ScaledProd {
  double v;
  int k;
};
__device__ static inline ScaledProd sp_mul(ScaledProd a, ScaledProd b) {
  ScaledProd r{a.v * b.v, a.k + b.k};  // both factors >= 2^-256: no underflow
  if (r.v < kScaleThreshold) {
    r.v *= kScaleFactor;
    ++r.k;
  }
  return r;
}

__device__ static inline ScaledProd sp_from(double e) {  // e in (0, 1]
  ScaledProd r{e, 0};
  while (r.v < kScaleThreshold) {
    r.v *= kScaleFactor;
    ++r.k;
  }
  return r;
}

// FillGermlinePaddingEmission through the set's consensus (DevSegments::cons_*).  All emissions are in (0, 1]
// here (checked by the caller), so a gene's running product falls monotonically and the reference's pair
// (value, ScaleMatrix count) is a function of the product alone: count = the fewest 2^256 factors that bring it
// back to >= 2^-256.  The product itself is formed as
//     prefix(last site + 1) / prefix(first site)  x  prod over the gene's departures  em[own] / em[consensus]
// with prefix = exclusive product scan of the consensus emissions over the set's sites, carried as (v, k)
// pairs.  A fifth to a twentieth of the factor-by-factor walk's multiplications for allele sets as alike as the
// candidates of one rearrangement.  lds: cons_inv[cap] | cons_pv[cap + 4] | cons_pk[cap + 4], cap = the largest
// set's sites + 2, even (emission_lds_bytes).
template <int kG, bool kByteOff>
__device__ static int fill_consensus(const DevSegments& seg, const double* em, int tid, double* __restrict__ out,
                                     int* redi, int phase, double* cons_inv, double* cons_pv, int* cons_pk,
                                     int cap) {  // cap: capacity of the prefix arrays; 4 more slots follow
  const int ns = seg.cons_sites, n = seg.n_genes;
  auto em_at = [&](unsigned x) {
    return kByteOff ? *reinterpret_cast<const double*>(reinterpret_cast<const char*>(em) + x) : em[x];
  };
  // exclusive prefix products: thread t owns sites 2t, 2t+1
  ScaledProd x0{1.0, 0}, x1{1.0, 0};
  const int j0 = 2 * tid;
  if (j0 < ns) {
    const double e = em_at(seg.cons_col[j0]);
    cons_inv[j0] = fast_rcp(e);
    x0 = sp_from(e);
  }
  if (j0 + 1 < ns) {
    const double e = em_at(seg.cons_col[j0 + 1]);
    cons_inv[j0 + 1] = fast_rcp(e);
    x1 = sp_from(e);
  }
  if (tid == 0) cons_inv[ns] = 1.0;  // the diff lists' padding position
  ScaledProd incl = sp_mul(x0, x1);  // inclusive scan of the per-thread products across the block
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double ov = __shfl_up(incl.v, off, 64);
    const int ok = __shfl_up(incl.k, off, 64);
    if (lane >= off) incl = sp_mul(ScaledProd{ov, ok}, incl);
  }
  // wave totals through the four slots behind the prefixes
  if (lane == 63) {
    cons_pv[cap + wave] = incl.v;
    cons_pk[cap + wave] = incl.k;
  }
  __syncthreads();
  ScaledProd excl = incl;  // -> exclusive: shift by one thread
  {
    const double ov = __shfl_up(incl.v, 1, 64);
    const int ok = __shfl_up(incl.k, 1, 64);
    excl = lane == 0 ? ScaledProd{1.0, 0} : ScaledProd{ov, ok};
    for (int w = 0; w < wave; ++w) excl = sp_mul(ScaledProd{cons_pv[cap + w], cons_pk[cap + w]}, excl);
  }
  if (j0 <= ns) {
    cons_pv[j0] = excl.v;
    cons_pk[j0] = excl.k;
  }
  if (j0 + 1 <= ns) {
    const ScaledProd p1 = sp_mul(excl, x0);
    cons_pv[j0 + 1] = p1.v;
    cons_pk[j0 + 1] = p1.k;
  }
  __syncthreads();

  double v[kG];
  int c[kG];
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    v[q] = 1.0;
    c[q] = 0;
    const int g = tid + kFwdThreads * q;
    if (g < n) {
      const unsigned rng = seg.cons_rng[g];  // first | (last + 1) << 16 | rounds of eight departures << 25
      const int a = rng & 0xffffu, b = (rng >> 16) & 0x1ffu;
      double r = cons_pv[b] * fast_rcp(cons_pv[a]);  // in (2^-256, 2^256)
      int k = cons_pk[b] - cons_pk[a];     // value = r * 2^(-256 k)
      const uint32_t* dp = seg.cons_dif + g;
      const int n_dif = 8 * (int)(rng >> 25);  // this gene's own departures (the table is padded to the set's longest list)
      uint32_t w_next[8];  // the next round's entries travel while this round's are applied
#pragma unroll
      for (int i = 0; i < 8; ++i) w_next[i] = n_dif > 0 ? dp[(size_t)i * n] : 0u;
      for (int d = 0; d < n_dif; d += 8) {  // eight departures at a time: their loads overlap
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = w_next[i];
        if (d + 8 < n_dif) {
#pragma unroll
          for (int i = 0; i < 8; ++i) w_next[i] = dp[(size_t)(d + 8 + i) * n];
        }
        double fct[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) fct[i] = em_at(w[i] >> 16) * cons_inv[w[i] & 0xffffu];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          r *= fct[i];
          // keep r within [2^-256, 2^256): a ratio of two emissions can be far from 1
          if (r < kScaleThreshold) {
            r *= kScaleFactor;
            ++k;
          } else if (r >= kScaleFactor) {
            r *= kScaleThreshold;
            --k;
          }
        }
      }
      // the reference's form: the fewest rescalings k >= 0 with value * 2^(256 k) >= 2^-256
      while (r < kScaleThreshold) {
        r *= kScaleFactor;
        ++k;
      }
      while (k > 0 && r >= 1.0) {
        r *= kScaleThreshold;
        --k;
      }
      while (k < 0) {  // cannot happen for products of factors <= 1; keeps the pair consistent anyway
        r *= kScaleFactor;
        ++k;
      }
      v[q] = r;
      c[q] = k;
    }
  }
  int local_max = 0;
#pragma unroll
  for (int q = 0; q < kG; ++q)
    if (tid + kFwdThreads * q < n) local_max = max(local_max, c[q]);
  const int mx = block_max_int(local_max, redi, phase);
#pragma unroll
  for (int q = 0; q < kG; ++q) {
    const int g = tid + kFwdThreads * q;
    if (g < n) out[g] = v[q] * pow_scale(mx - c[q]);
  }
  return mx;
}

template <int kG, bool kFromSiteLik, bool kByteOff, bool kExt = false>
__global__ void __launch_bounds__(kFwdThreads) __attribute__((amdgpu_num_sgpr(80)))  // 80: eight workgroups per CU
    emission_kernel(const DevFamily* __restrict__ fam_dev, int R, const double* __restrict__ site_lik,
                    const int32_t* __restrict__ site_scal, const double* __restrict__ pi,
                    const double* __restrict__ em_in, double* __restrict__ em_out, double* __restrict__ gem_all,
                    int32_t* __restrict__ gcnt_all, double* __restrict__ jem_all, int32_t* __restrict__ jrs_all) {
  // (the family descriptor is read from its device copy where a field is needed: passed by value it took more
  // scalar registers than the 96 that still admit seven workgroups per CU)
  const DevFamily& fam = *fam_dev;
  extern __shared__ double em[];  // [C + 1] emissions (em[C] = 1.0 sentinel) | reduction scratch
  const size_t s = blockIdx.x;
  const int tid = threadIdx.x;
  const int C = fam.n_ucol;  // u-columns (lh_device.h); the caller's columns only appear in em_in / em_out
  double* inv_pi = em + ((C + 2) & ~1);                      // [6]: 1 / pi_b of this sample, 1 for b = N
  int* redi = reinterpret_cast<int*>(inv_pi + 6);            // 2 * kFwdWaves ints, then one flag word
  int* em_bad = redi + 2 * kFwdWaves;                        // some emission outside (0, 1]
  const int cons_cap = cons_capacity(fam);                                  // 0: no set in consensus form
  double* cons_inv = reinterpret_cast<double*>(redi + 2 * kFwdWaves + 2);  // [cap]
  double* cons_pv = cons_inv + cons_cap;                                    // [cap + 4]
  int* cons_pk = reinterpret_cast<int*>(cons_pv + cons_cap + 4);            // [cap + 4]
  int* ems = cons_pk + (cons_cap ? cons_cap + 4 : 0);                       // [C + 1] (kExt only): 2^-256 counts
  if (tid == 0) *em_bad = 0;
  bool my_bad = false;

  if constexpr (kFromSiteLik) {
    // PhyloHMM::FillXmsaEmission tail: mix the rate categories (equal weights, scalers aligned to the
    // smallest one) and apply the naive correction, once per (naive base, pattern) pair.
    // u-columns are numbered by their place in K1's planes (base * NP + pattern): a thread takes a pattern, asks for
    // its 6 R plane entries at once -- no look-up in front of the loads -- and writes the pattern's five emissions.
    // The first pattern's loads are in flight while the sample's 1 / pi are formed (the naive correction divides a
    // column's likelihood by pi of its naive base: one reciprocal per base and sample instead of a division per
    // column).
    const int NP = fam.n_prune;
    const double w = 1.0 / R;
    struct Mixed {
      double acc[5];
      int smin;
      unsigned used;  // bit b: some xMSA column is the pair (b, pattern)
    };
    auto mix = [&](int pat) {
      Mixed m;
      m.smin = 0x7fffffff;
      m.used = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b) m.used |= (fam.u_base[b * NP + pat] != 0xff ? 1u : 0u) << b;
      for (int r = 0; r < R; ++r) m.smin = min(m.smin, site_scal[(s * R + r) * NP + pat]);
#pragma unroll
      for (int b = 0; b < 5; ++b) m.acc[b] = 0.0;
      for (int r = 0; r < R; ++r) {
        const int d = site_scal[(s * R + r) * NP + pat] - m.smin;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          double v = site_lik[((s * R + r) * 5 + b) * (size_t)NP + pat];
          for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
          m.acc[b] += w * v;
        }
      }
      return m;
    };
    auto finish = [&](int pat, const Mixed& m) {
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int u = b * NP + pat;
        // The reference forms exp(log(site_lik) - smin*log(2^256) - log(pi_b)) (src/PhyloHMM.cpp:226-237);
        // the same quantity is computed here without the log/exp round trip (two FP64 transcendentals
        // per column): site_lik / pi_b scaled down by 2^(256*smin), which also underflows to 0 like exp().
        double e = m.acc[b] * inv_pi[b];
        if constexpr (kExt) {
          ems[u] = m.smin;  // the emission is e * 2^(-256 smin); the count travels beside the value
        } else {
          for (int q = 0; q < m.smin && e != 0.0; ++q) e *= kScaleThreshold;
        }
        em[u] = e;
        // (a pair no xMSA column has is never read below: it cannot send the sample down the slow path)
        my_bad |= ((m.used >> b) & 1u) && !(e >= 0x1p-1000 && e <= 1.0 + 1e-9);
      }
    };
    Mixed first;
    if (tid < NP) first = mix(tid);
    if (tid < 5) inv_pi[tid] = tid < 4 ? 1.0 / pi[s * 4 + tid] : 1.0;
    __syncthreads();
    if (tid < NP) finish(tid, first);
    for (int pat = tid + kFwdThreads; pat < NP; pat += kFwdThreads) finish(pat, mix(pat));
    if (tid < 5) {
      // the all-N padding pattern: likelihood pi_b, emission 1 -- and for the naive base N the sum of the
      // sample's pi, which is 1 only as far as its digits go
      const double* q = pi + s * 4;
      em[5 * NP + tid] = tid == 4 ? ((q[0] + q[1]) + q[2]) + q[3] : 1.0;
      if constexpr (kExt) ems[5 * NP + tid] = 0;
    }
  } else {
    __syncthreads();  // (the flag word is cleared before anyone sets it)
    for (int u = tid; u < C; u += kFwdThreads) {
      const int c = fam.col_of_ucol[u];
      const double e = c >= 0 ? em_in[s * fam.n_xmsa + c] : 1.0;  // (no column: the pair is never read)
      em[u] = e;
      if constexpr (kExt) ems[u] = 0;
      my_bad |= !(e >= 0x1p-1000 && e <= 1.0 + 1e-9);
    }
  }
  if constexpr (kExt) {
    if (tid == 0) ems[C] = 0;
  }
  // The consensus form of the germline products needs every emission in (0, 1] (see fill_consensus) and its
  // reciprocal finite; a sample with a zero or nearly subnormal emission (underflow), a NaN or an emission above 1
  // walks its products factor by factor as the reference does.
  if (my_bad) *em_bad = 1;
  if (tid == 0) em[C] = 1.0;
  __syncthreads();
  if (em_out) {  // the caller's view: one value per xMSA column
    const int CX = fam.n_xmsa;
    for (int c = tid; c < CX; c += kFwdThreads) {
      double e = em[fam.ucol_of_col[c]];
      if constexpr (kExt)  // PhyloHMM::xmsa_emission_ is the plain value (it may underflow; the path below does not use it)
        for (int q = ems[fam.ucol_of_col[c]]; q > 0 && e != 0.0; --q) e *= kScaleThreshold;
      em_out[s * CX + c] = e;
    }
  }

  // emissions of the columns the junction rows touch, compacted for K2b
  {
    double* jem = jem_all + s * fam.n_jcols;
    for (int j = tid; j < fam.n_jcols; j += kFwdThreads) jem[j] = em[fam.jcols[j]];
  }
  if constexpr (kExt) {
    // ... and the 2^-256 count of each junction row: all columns of a row belong to one alignment site, whose
    // count (the smallest over the rate categories, as above) depends on the site's pattern only
    const int W1 = fam.vd.n_rows, W2 = fam.has_d ? fam.dj.n_rows : 0, NP = fam.n_prune;
    int32_t* jrs = jrs_all + s * (size_t)(W1 + W2);
    for (int r = tid; r < W1 + W2; r += kFwdThreads) {
      const int pat = r < W1 ? fam.vd.row_pat[r] : fam.dj.row_pat[r - W1];
      int smin = 0;
      if (kFromSiteLik && pat < NP) {
        smin = 0x7fffffff;
        for (int q = 0; q < R; ++q) smin = min(smin, site_scal[(s * R + q) * NP + pat]);
      }
      jrs[r] = smin;
    }
  }

  // [vpadding nV | vgerm nV | dgerm nD | jgerm nJ | jpadding nJ]
  const int nV = fam.vgerm.n_genes, nD = fam.dgerm.n_genes, nJ = fam.jgerm.n_genes;
  double* gem = gem_all + s * fam.gem_size;
  const bool direct = *em_bad != 0;  // (written before the barrier that followed the emission assembly)
  auto fill = [&](const DevSegments& seg, double* out, int phase) {
    if constexpr (kExt) return fill_segments_ext<kG, kByteOff>(seg, em, ems, tid, out, redi, phase);
    if (seg.cons_sites > 0 && !direct)
      return fill_consensus<kG, kByteOff>(seg, em, tid, out, redi, phase, cons_inv, cons_pv, cons_pk, cons_cap);
    return fill_segments<kG, kByteOff>(seg, em, tid, out, redi, phase);
  };
  int cv = fill(fam.vpadding, gem, 0);
  cv += fill(fam.vgerm, gem + nV, 1);
  int cd = 0, cj;
  if (!kExt && nD <= 64 && nJ <= 64 &&
      (direct || (fam.dgerm.cons_sites == 0 && fam.jgerm.cons_sites == 0 && fam.jpadding.cons_sites == 0))) {
    // the small sets (D, J, J padding; light chains: J, J padding), a wave each, side by side (fill_segments_wave);
    // the counts meet in LDS
    const int wave = tid >> 6, lane = tid & 63;
    double* jg = gem + 2 * (size_t)nV + (fam.has_d ? nD : 0);
    int m = 0;
    if (wave == 1 && fam.has_d) m = fill_segments_wave<kByteOff>(fam.dgerm, em, lane, gem + 2 * (size_t)nV);
    if (wave == 2) m = fill_segments_wave<kByteOff>(fam.jgerm, em, lane, jg);
    if (wave == 3) m = fill_segments_wave<kByteOff>(fam.jpadding, em, lane, jg + nJ);
    if (lane == 0) redi[wave] = m;  // (the first half of redi: last read before vgerm's barriers)
    __syncthreads();
    cd = redi[1];
    cj = redi[2] + redi[3];
  } else if (fam.has_d) {
    cd = fill(fam.dgerm, gem + 2 * (size_t)nV, 0);
    cj = fill(fam.jgerm, gem + 2 * (size_t)nV + nD, 1);
    cj += fill(fam.jpadding, gem + 2 * (size_t)nV + nD + nJ, 0);
  } else {
    cj = fill(fam.jgerm, gem + 2 * (size_t)nV, 0);
    cj += fill(fam.jpadding, gem + 2 * (size_t)nV + nJ, 1);
  }
  if (tid == 0) {
    gcnt_all[s * 3 + 0] = cv;
    gcnt_all[s * 3 + 1] = cd;
    gcnt_all[s * 3 + 2] = cj;
  }
}

// ---------------------------------------------------------------------------------------------------
// K2b
// ---------------------------------------------------------------------------------------------------

// Wave-wide reductions without LDS traffic: four DPP steps fold each row of 16 lanes (quad swaps, then
// the two mirror patterns), four v_readlane pick the row totals.  A __shfl_xor butterfly costs six
// dependent ds_bpermute round trips per value, which was the longest single item of a junction row.
template <int kCtrl>
__device__ static inline double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

__device__ static inline double read_lane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}

constexpr int kDppQuadSwap1 = 0xB1;    // quad_perm [1,0,3,2]
constexpr int kDppQuadSwap2 = 0x4E;    // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // lane i <-> 7 - i within each 8
constexpr int kDppMirror = 0x140;      // lane i <-> 15 - i within each 16

__device__ static inline double wave_sum(double v) {
  v += dpp_move<kDppQuadSwap1>(v);
  v += dpp_move<kDppQuadSwap2>(v);
  v += dpp_move<kDppHalfMirror>(v);
  v += dpp_move<kDppMirror>(v);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

// ScaleMatrix needs, per vector, only the binade of its smallest positive entry.  Every entry is >= 0,
// so the high word of a double orders like the value; key(v) = high word - 1 sends zeros to the far
// end, and the smallest key over the wave identifies that binade with 32-bit integer minima (one VALU
// op per step, DPP-fused) instead of 64-bit compare/select/min chains.  (A positive entry below
// 2^-1042 has a zero high word and is skipped like a zero -- a value no forward sweep produces next to
// normal-range neighbours without the reference overflowing itself.)
__device__ static inline unsigned scale_key(double v) { return (unsigned)__double2hiint(v) - 1u; }

template <int kCtrl>
__device__ static inline unsigned dpp_min_u32(unsigned v) {
  return min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, 0xf, 0xf, true));
}

__device__ static inline unsigned wave_min_key(unsigned v) {
  v = dpp_min_u32<kDppQuadSwap1>(v);
  v = dpp_min_u32<kDppQuadSwap2>(v);
  v = dpp_min_u32<kDppHalfMirror>(v);
  v = dpp_min_u32<kDppMirror>(v);
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}

template <int kCtrl>
__device__ static inline unsigned dpp_max_u32(unsigned v) {
  return max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, 0xf, 0xf, true));
}

__device__ static inline unsigned wave_max_u32(unsigned v) {
  v = dpp_max_u32<kDppQuadSwap1>(v);
  v = dpp_max_u32<kDppQuadSwap2>(v);
  v = dpp_max_u32<kDppHalfMirror>(v);
  v = dpp_max_u32<kDppMirror>(v);
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}

// The entry that decides a vector's rescaling: its smallest positive one (ScaleMatrix, the reference) or, in the
// extended-range mode, its largest -- the vector is then brought back above 2^-256 at its top, entries far
// below it may underflow, and no entry can overflow.  Keys as scale_key (min) / plain high words (max).
template <bool kExt>
__device__ static inline unsigned key_start() {
  return kExt ? 0u : 0xffffffffu;
}
template <bool kExt>
__device__ static inline unsigned key_add(unsigned key, double v) {
  return kExt ? max(key, (unsigned)__double2hiint(v)) : min(key, scale_key(v));
}

// ScaleMatrix on a wave-uniform basis: the reference multiplies the whole vector by 2^256 until its
// smallest positive entry reaches 2^-256, i.e. k = #{j in 1..4 : minpos < 2^(-256 j)} times (a double
// is never below 2^-1074, so k <= 4).  k follows from the exponent field with scalar instructions, and
// the k multiplications -- each exact -- collapse into one by 2^(256 min(k,3)) plus, for k = 4 only
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

__device__ static inline RowScale row_scale(unsigned min_key) {  // min_key: wave-uniform
  const unsigned hw = min_key + 1u;  // high word of the smallest positive entry; 0 = there is none
  const unsigned e = (hw >> 20) & 0x7ffu;
  RowScale s;
  s.extra = hw != 0 && e == 0 && (hw & 0xfffffu) < (1u << 18);  // minpos < 2^-1024
  const int k3 = hw == 0 ? 0 : (e < 767u) + (e < 511u) + (e < 255u);
  s.k = k3 + (s.extra ? 1 : 0);
  s.factor = __hiloint2double((1023 + 256 * k3) << 20, 0);
  return s;
}

template <bool kExt>
__device__ static inline RowScale wave_row_scale(unsigned key) {
  return row_scale(kExt ? wave_max_u32(key) - 1u : wave_min_key(key));
}

// One junction row's table entries for the genes a lane owns (family constants, independent of the
// sample and of the HMM state).
template <int GL, int GR>
struct RowTables {
  double ltr[GL], llo[GL];
  int lidx[GL];
  double nlo[GR][4], rtr[GR], rli[GR];
  int ridx[GR];
  int4 nx[GR];
};

template <int GL, int GR>
__device__ static inline void load_row(const DevJunction& J, int i, unsigned lane, RowTables<GL, GR>& t) {
  // wave-uniform row bases + a lane offset that fits the instruction's immediate: no vector address math
  const size_t ol = (size_t)i * J.left_pad, orr = (size_t)i * J.right_pad;
  const double* lt = J.left_trans + ol;
  const double* ll = J.left_lo + ol;
  const int32_t* lx = J.left_xmsa + ol;
#pragma unroll
  for (int q = 0; q < GL; ++q) {
    t.ltr[q] = lt[lane + 64u * q];
    t.llo[q] = ll[lane + 64u * q];
    t.lidx[q] = lx[lane + 64u * q];
  }
  const double2* pn = reinterpret_cast<const double2*>(J.right_nlo) + 2 * orr;
  const int4* px = reinterpret_cast<const int4*>(J.nti_xmsa) + orr;
  const double* rt = J.right_trans + orr;
  const double* rl = J.right_gp_li + orr;
  const int32_t* rx = J.right_xmsa + orr;
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const unsigned r = lane + 64u * q;
    const double2 a = pn[2u * r], b = pn[2u * r + 1u];
    t.nlo[q][0] = a.x;
    t.nlo[q][1] = a.y;
    t.nlo[q][2] = b.x;
    t.nlo[q][3] = b.y;
    t.nx[q] = px[r];
    t.rtr[q] = rt[r];
    t.rli[q] = rl[r];
    t.ridx[q] = rx[r];
  }
}

// kExt (extended-range mode): row i's emissions carry the 2^-256 count jrs[i], added to the running count, and
// rows are rescaled by their largest entry (see key_start).
template <int GL, int GR, bool kExt>
__device__ static int junction_wave(const DevJunction& J, const double* jem, const double* ntt_lds, int lane,
                                    const double (&f_in)[GL], int count_in, const double* __restrict__ germ_em,
                                    const double* __restrict__ pad_trans, const double* __restrict__ pad_em,
                                    double (&g_out)[GR], double* __restrict__ fwd_out,
                                    int32_t* __restrict__ scal_out, const int32_t* __restrict__ jrs) {
  const int W = J.n_rows, nL = J.n_left, nR = J.n_right;
  int count = count_in;
  double fL[GL], fN[GR][4], fR[GR];  // the previous row, ScaleMatrix already applied
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

  // Row i: compute every live state from row i-1, one reduction for (rank-one sum, smallest binade),
  // then -- only on the rows where ScaleMatrix fires -- rescale the row in place.
  auto step = [&](int i, const RowTables<GL, GR>& t) __attribute__((always_inline)) {
    unsigned key = key_start<kExt>();
    double part = 0.0;
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      const double v = (fL[q] * t.ltr[q]) * jem[t.lidx[q]];
      fL[q] = v;
      key = key_add<kExt>(key, v);
      part += v * t.llo[q];  // contribution to the next row's rank-one term
    }
#pragma unroll
    for (int q = 0; q < GR; ++q) {
      const double n0 = fN[q][0], n1 = fN[q][1], n2 = fN[q][2], n3 = fN[q][3];
      // NTI->NTI block of gene r, transposed in LDS: [b * 4 + a] = transition a -> b
      const double2* tt = reinterpret_cast<const double2*>(ntt_lds) + 8u * (lane + 64u * q);
      const int nxs[4] = {t.nx[q].x, t.nx[q].y, t.nx[q].z, t.nx[q].w};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double2 t01 = tt[2 * b], t23 = tt[2 * b + 1];
        double s = ((n0 * t01.x + n1 * t01.y) + n2 * t23.x) + n3 * t23.y;
        s += A * nli[q][b];
        const double v = s * jem[nxs[b]];
        fN[q][b] = v;
        key = key_add<kExt>(key, v);
      }
      double s = ((n0 * t.nlo[q][0] + n1 * t.nlo[q][1]) + n2 * t.nlo[q][2]) + n3 * t.nlo[q][3];
      s += fR[q] * t.rtr[q];
      s += A * t.rli[q];
      const double v = s * jem[t.ridx[q]];
      fR[q] = v;
      key = key_add<kExt>(key, v);
    }
    A = wave_sum(part);
    const RowScale sc = wave_row_scale<kExt>(key);
    if constexpr (kExt) count += jrs[i];
    if (sc.k != 0) {  // wave-uniform, a few rows per junction
      A = sc.apply(A);
#pragma unroll
      for (int q = 0; q < GL; ++q) fL[q] = sc.apply(fL[q]);
#pragma unroll
      for (int q = 0; q < GR; ++q) {
        fR[q] = sc.apply(fR[q]);
#pragma unroll
        for (int b = 0; b < 4; ++b) fN[q][b] = sc.apply(fN[q][b]);
      }
      count += sc.k;
    }
    if (fwd_out) {
      double* o = fwd_out + (size_t)i * row_stride;
#pragma unroll
      for (int q = 0; q < GL; ++q) {
        const int g = lane + 64 * q;
        if (g < nL) o[g] = fL[q];
      }
#pragma unroll
      for (int q = 0; q < GR; ++q) {
        const int g = lane + 64 * q;
        if (g < nR) {
          o[nL + 4 * (size_t)g + 0] = fN[q][0];
          o[nL + 4 * (size_t)g + 1] = fN[q][1];
          o[nL + 4 * (size_t)g + 2] = fN[q][2];
          o[nL + 4 * (size_t)g + 3] = fN[q][3];
          o[nL + 4 * (size_t)nR + g] = fR[q];
        }
      }
    }
    if (scal_out && lane == 0) scal_out[i] = count;
  };

  // (Fetching the next row's table entries while a row computes was measured: the second table set costs
  // 90 VGPRs, i.e. the third wave per SIMD, and that wave hides the fetch better than the prefetch did.)
  for (int i = 0; i < W; ++i) {
    RowTables<GL, GR> t;
    load_row<GL, GR>(J, i, lane, t);
    step(i, t);
  }

  // hand-off into the right germline region (A already holds the last row's rank-one sum)
  unsigned key = key_start<kExt>();
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const unsigned r = lane + 64u * q;
    const double2* xn = reinterpret_cast<const double2*>(J.exit_nlo) + 2u * r;
    const double2 x01 = xn[0], x23 = xn[1];
    double s = ((fN[q][0] * x01.x + fN[q][1] * x01.y) + fN[q][2] * x23.x) + fN[q][3] * x23.y;
    s += fR[q] * J.exit_trans[r];
    s += A * J.exit_gp_li[r];
    double v = 0.0;
    if ((int)r < nR) {
      v = s * germ_em[r];
      if (pad_trans) v *= pad_trans[r];
      if (pad_em) v *= pad_em[r];
    }
    key = key_add<kExt>(key, v);
    g_out[q] = v;
  }
  const RowScale last = wave_row_scale<kExt>(key);
#pragma unroll
  for (int q = 0; q < GR; ++q) g_out[q] = last.apply(g_out[q]);
  return count + last.k;
}

// ---- two samples per wave (D-J junction of small D / J sets) ------------------------------------------------
// A D-J junction row keeps one lane per D gene and one per J gene busy: with at most 32 of each, 42 of 64 lanes
// of a wave-per-sample sweep idle on configs[2] (30 D, 12 J).  The pair form gives lanes 0-31 to one sample and
// lanes 32-63 to another; gene g of either sample sits in lane g of its half.  Reductions are taken per half
// (the DPP steps fold rows of 16 lanes; two v_readlane per half finish the job), everything that was
// wave-uniform per sample -- the rank-one sum, the ScaleMatrix count -- becomes a per-lane value that is equal
// across a half.

struct HalfPair {
  double a, b;  // totals of lanes 0-31 / 32-63
};

__device__ static inline HalfPair half_sums(double v) {
  v += dpp_move<kDppQuadSwap1>(v);
  v += dpp_move<kDppQuadSwap2>(v);
  v += dpp_move<kDppHalfMirror>(v);
  v += dpp_move<kDppMirror>(v);
  return HalfPair{read_lane(v, 0) + read_lane(v, 16), read_lane(v, 32) + read_lane(v, 48)};
}

struct HalfKeys {
  unsigned a, b;
};

template <bool kExt>
__device__ static inline HalfKeys half_keys(unsigned v) {  // min of scale keys, or max of high words (kExt)
  if (kExt) {
    v = dpp_max_u32<kDppQuadSwap1>(v);
    v = dpp_max_u32<kDppQuadSwap2>(v);
    v = dpp_max_u32<kDppHalfMirror>(v);
    v = dpp_max_u32<kDppMirror>(v);
  } else {
    v = dpp_min_u32<kDppQuadSwap1>(v);
    v = dpp_min_u32<kDppQuadSwap2>(v);
    v = dpp_min_u32<kDppHalfMirror>(v);
    v = dpp_min_u32<kDppMirror>(v);
  }
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return kExt ? HalfKeys{max(a, b) - 1u, max(c, d) - 1u} : HalfKeys{min(a, b), min(c, d)};
}

// The D-J junction sweep and the hand-off into the J germline region for the two samples of a wave
// (junction_wave<1, 1> twice over).  `hi` = this lane belongs to the second sample; every pointer is the lane's
// own sample's.  Returns the lane's sample's J scaler count; g_out = forward probability of J gene (lane & 31).
template <bool kExt>
__device__ static int junction_pair(const DevJunction& J, const double* jem, const double* ntt_lds, int lane, bool hi,
                                    double f_in, int count_in, const double* __restrict__ germ_em,
                                    const double* __restrict__ pad_trans, const double* __restrict__ pad_em,
                                    double& g_out, double* __restrict__ fwd_out, int32_t* __restrict__ scal_out,
                                    const int32_t* __restrict__ jrs, bool valid) {
  const int W = J.n_rows, nL = J.n_left, nR = J.n_right;
  const unsigned g = lane & 31;  // gene of this lane, left and right alike
  int count = count_in;
  double fL = f_in, fR = 0.0, fN[4] = {0.0, 0.0, 0.0, 0.0}, nli[4];
  {
    const double2* p = reinterpret_cast<const double2*>(J.right_gp_nli) + 2u * g;
    const double2 a = p[0], b = p[1];
    nli[0] = a.x, nli[1] = a.y, nli[2] = b.x, nli[3] = b.y;
  }
  const size_t row_stride = (size_t)nL + 5 * (size_t)nR;
  auto mine = [&](const HalfPair& p) { return hi ? p.b : p.a; };
  double A = mine(half_sums(f_in * J.enter_lo[g]));
  for (int i = 0; i < W; ++i) {
    const size_t ol = (size_t)i * J.left_pad, orr = (size_t)i * J.right_pad;
    const double ltr = J.left_trans[ol + g], llo = J.left_lo[ol + g];
    const int lidx = J.left_xmsa[ol + g];
    const double2* pn = reinterpret_cast<const double2*>(J.right_nlo) + 2 * (orr + g);
    const double2 n01 = pn[0], n23 = pn[1];
    const int4 nx = reinterpret_cast<const int4*>(J.nti_xmsa)[orr + g];
    const double rtr = J.right_trans[orr + g], rli = J.right_gp_li[orr + g];
    const int ridx = J.right_xmsa[orr + g];

    unsigned key = key_start<kExt>();
    {
      const double v = (fL * ltr) * jem[lidx];
      fL = v;
      key = key_add<kExt>(key, v);
    }
    const double part = fL * llo;
    const double n0 = fN[0], n1 = fN[1], n2 = fN[2], n3 = fN[3];
    const double2* tt = reinterpret_cast<const double2*>(ntt_lds) + 8u * g;
    const int nxs[4] = {nx.x, nx.y, nx.z, nx.w};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double2 t01 = tt[2 * b], t23 = tt[2 * b + 1];
      double sacc = ((n0 * t01.x + n1 * t01.y) + n2 * t23.x) + n3 * t23.y;
      sacc += A * nli[b];
      const double v = sacc * jem[nxs[b]];
      fN[b] = v;
      key = key_add<kExt>(key, v);
    }
    {
      double sacc = ((n0 * n01.x + n1 * n01.y) + n2 * n23.x) + n3 * n23.y;
      sacc += fR * rtr;
      sacc += A * rli;
      const double v = sacc * jem[ridx];
      fR = v;
      key = key_add<kExt>(key, v);
    }
    A = mine(half_sums(part));
    const HalfKeys hk = half_keys<kExt>(key);
    const RowScale sa = row_scale(hk.a), sb = row_scale(hk.b);
    if constexpr (kExt) count += jrs[i];
    if ((sa.k | sb.k) != 0) {  // wave-uniform, a few rows per junction
      const RowScale& sc = hi ? sb : sa;  // (per-lane choice between two wave-uniform scalings)
      const double factor = hi ? sb.factor : sa.factor;
      const bool extra = hi ? sb.extra : sa.extra;
      auto apply = [&](double v) {
        v *= factor;
        if (extra) v *= kScaleFactor;
        return v;
      };
      A = apply(A);
      fL = apply(fL);
      fR = apply(fR);
#pragma unroll
      for (int b = 0; b < 4; ++b) fN[b] = apply(fN[b]);
      count += sc.k;
    }
    if (fwd_out && valid) {
      double* o = fwd_out + (size_t)i * row_stride;
      if ((int)g < nL) o[g] = fL;
      if ((int)g < nR) {
        o[nL + 4 * (size_t)g + 0] = fN[0];
        o[nL + 4 * (size_t)g + 1] = fN[1];
        o[nL + 4 * (size_t)g + 2] = fN[2];
        o[nL + 4 * (size_t)g + 3] = fN[3];
        o[nL + 4 * (size_t)nR + g] = fR;
      }
    }
    if (scal_out && valid && g == 0) scal_out[i] = count;
  }
  unsigned key = key_start<kExt>();
  {
    const double2* xn = reinterpret_cast<const double2*>(J.exit_nlo) + 2u * g;
    const double2 x01 = xn[0], x23 = xn[1];
    double sacc = ((fN[0] * x01.x + fN[1] * x01.y) + fN[2] * x23.x) + fN[3] * x23.y;
    sacc += fR * J.exit_trans[g];
    sacc += A * J.exit_gp_li[g];
    double v = 0.0;
    if ((int)g < nR) {
      v = sacc * germ_em[g];
      if (pad_trans) v *= pad_trans[g];
      if (pad_em) v *= pad_em[g];
    }
    key = key_add<kExt>(key, v);
    g_out = v;
  }
  const HalfKeys hk = half_keys<kExt>(key);
  const RowScale sa = row_scale(hk.a), sb = row_scale(hk.b);
  const RowScale& last = hi ? sb : sa;
  {
    double v = g_out * (hi ? sb.factor : sa.factor);
    if (hi ? sb.extra : sa.extra) v *= kScaleFactor;
    g_out = v;
  }
  return count + last.k;
}

// ---- two samples per wave on the V-D junction --------------------------------------------------------------
// The V-D sweep's lanes are full on its left (V) genes but not on its right (D) genes: 30 of 64 on configs[2], and
// the right genes are where most of a row's arithmetic is (five states per gene).  Two samples per wave: the left
// genes of both samples go through the same lanes one after the other on ONE set of row-table registers (the
// tables are family constants -- half the table loads per sample), the right genes of sample A sit in lanes
// 0-31 and those of sample B in lanes 32-63 as in junction_pair.  The rank-one sums are whole-wave sums per
// sample; the ScaleMatrix key of a sample covers its left entries on all lanes and its right entries on its half.
// Per sample the operations and their order are those of junction_wave<GL, 1>.
template <int GL, bool kExt>
__device__ static void junction_vd_pair(const DevJunction& J, const double* jemA, const double* jemB,
                                        const double* ntt_lds, int lane, bool hi, const double (&fA_in)[GL],
                                        const double (&fB_in)[GL], int countA_in, int countB_in,
                                        const double* __restrict__ germ_em_mine, double& g_out, int& count_mine,
                                        double* __restrict__ fwdA, double* __restrict__ fwdB,
                                        int32_t* __restrict__ scoA, int32_t* __restrict__ scoB,
                                        const int32_t* __restrict__ jrsA, const int32_t* __restrict__ jrsB) {
  const int W = J.n_rows, nL = J.n_left, nR = J.n_right;
  const unsigned g = lane & 31;  // right gene of this lane (its half's sample)
  const double* jemM = hi ? jemB : jemA;
  int countA = countA_in, countB = countB_in;
  double fA[GL], fB[GL], fN[4] = {0.0, 0.0, 0.0, 0.0}, fR = 0.0, nli[4];
#pragma unroll
  for (int q = 0; q < GL; ++q) {
    fA[q] = fA_in[q];
    fB[q] = fB_in[q];
  }
  {
    const double2* p = reinterpret_cast<const double2*>(J.right_gp_nli) + 2u * g;
    const double2 a = p[0], b = p[1];
    nli[0] = a.x, nli[1] = a.y, nli[2] = b.x, nli[3] = b.y;
  }
  const size_t row_stride = (size_t)nL + 5 * (size_t)nR;
  double AA, AB;
  {
    double pa = 0.0, pb = 0.0;
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      const double lo = J.enter_lo[lane + 64u * q];
      pa += fA_in[q] * lo;
      pb += fB_in[q] * lo;
    }
    AA = wave_sum(pa);
    AB = wave_sum(pb);
  }
  auto combine = [](unsigned a, unsigned b) { return kExt ? max(a, b) : min(a, b); };
  for (int i = 0; i < W; ++i) {
    const size_t ol = (size_t)i * J.left_pad, orr = (size_t)i * J.right_pad;
    unsigned keyA = key_start<kExt>(), keyB = key_start<kExt>();
    double partA = 0.0, partB = 0.0;
    {
      const double* lt = J.left_trans + ol;
      const double* ll = J.left_lo + ol;
      const int32_t* lx = J.left_xmsa + ol;
#pragma unroll
      for (int q = 0; q < GL; ++q) {
        const double ltr = lt[lane + 64u * q], llo = ll[lane + 64u * q];
        const int lidx = lx[lane + 64u * q];
        const double va = (fA[q] * ltr) * jemA[lidx];
        const double vb = (fB[q] * ltr) * jemB[lidx];
        fA[q] = va;
        fB[q] = vb;
        keyA = key_add<kExt>(keyA, va);
        keyB = key_add<kExt>(keyB, vb);
        partA += va * llo;
        partB += vb * llo;
      }
    }
    {
      const double AM = hi ? AB : AA;
      const double2* pn = reinterpret_cast<const double2*>(J.right_nlo) + 2 * (orr + g);
      const double2 n01 = pn[0], n23 = pn[1];
      const int4 nx = reinterpret_cast<const int4*>(J.nti_xmsa)[orr + g];
      const double rtr = J.right_trans[orr + g], rli = J.right_gp_li[orr + g];
      const int ridx = J.right_xmsa[orr + g];
      unsigned keyR = key_start<kExt>();
      const double n0 = fN[0], n1 = fN[1], n2 = fN[2], n3 = fN[3];
      const double2* tt = reinterpret_cast<const double2*>(ntt_lds) + 8u * g;
      const int nxs[4] = {nx.x, nx.y, nx.z, nx.w};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double2 t01 = tt[2 * b], t23 = tt[2 * b + 1];
        double sacc = ((n0 * t01.x + n1 * t01.y) + n2 * t23.x) + n3 * t23.y;
        sacc += AM * nli[b];
        const double v = sacc * jemM[nxs[b]];
        fN[b] = v;
        keyR = key_add<kExt>(keyR, v);
      }
      {
        double sacc = ((n0 * n01.x + n1 * n01.y) + n2 * n23.x) + n3 * n23.y;
        sacc += fR * rtr;
        sacc += AM * rli;
        const double v = sacc * jemM[ridx];
        fR = v;
        keyR = key_add<kExt>(keyR, v);
      }
      if (hi)
        keyB = combine(keyB, keyR);
      else
        keyA = combine(keyA, keyR);
    }
    AA = wave_sum(partA);
    AB = wave_sum(partB);
    const RowScale sa = wave_row_scale<kExt>(keyA), sb = wave_row_scale<kExt>(keyB);
    if constexpr (kExt) {
      countA += jrsA[i];
      countB += jrsB[i];
    }
    if ((sa.k | sb.k) != 0) {  // wave-uniform, a few rows per junction
      AA = sa.apply(AA);
      AB = sb.apply(AB);
#pragma unroll
      for (int q = 0; q < GL; ++q) {
        fA[q] = sa.apply(fA[q]);
        fB[q] = sb.apply(fB[q]);
      }
      const double factor = hi ? sb.factor : sa.factor;
      const bool extra = hi ? sb.extra : sa.extra;
      auto apply = [&](double v) {
        v *= factor;
        if (extra) v *= kScaleFactor;
        return v;
      };
      fR = apply(fR);
#pragma unroll
      for (int b = 0; b < 4; ++b) fN[b] = apply(fN[b]);
      countA += sa.k;
      countB += sb.k;
    }
    if (fwdA) {  // (fwdB is null when the second sample does not exist)
      double* oa = fwdA + (size_t)i * row_stride;
      double* ob = fwdB ? fwdB + (size_t)i * row_stride : nullptr;
#pragma unroll
      for (int q = 0; q < GL; ++q) {
        const int t = lane + 64 * q;
        if (t < nL) {
          oa[t] = fA[q];
          if (ob) ob[t] = fB[q];
        }
      }
      double* om = hi ? ob : oa;
      if (om && (int)g < nR) {
        om[nL + 4 * (size_t)g + 0] = fN[0];
        om[nL + 4 * (size_t)g + 1] = fN[1];
        om[nL + 4 * (size_t)g + 2] = fN[2];
        om[nL + 4 * (size_t)g + 3] = fN[3];
        om[nL + 4 * (size_t)nR + g] = fR;
      }
    }
    if (lane == 0) {
      if (scoA) scoA[i] = countA;
      if (scoB) scoB[i] = countB;
    }
  }
  // hand-off into the D germline region, each half for its sample
  unsigned key = key_start<kExt>();
  {
    const double AM = hi ? AB : AA;
    const double2* xn = reinterpret_cast<const double2*>(J.exit_nlo) + 2u * g;
    const double2 x01 = xn[0], x23 = xn[1];
    double sacc = ((fN[0] * x01.x + fN[1] * x01.y) + fN[2] * x23.x) + fN[3] * x23.y;
    sacc += fR * J.exit_trans[g];
    sacc += AM * J.exit_gp_li[g];
    double v = 0.0;
    if ((int)g < nR) v = sacc * germ_em_mine[g];
    key = key_add<kExt>(key, v);
    g_out = v;
  }
  const HalfKeys hk = half_keys<kExt>(key);
  const RowScale la = row_scale(hk.a), lb = row_scale(hk.b);
  {
    double v = g_out * (hi ? lb.factor : la.factor);
    if (hi ? lb.extra : la.extra) v *= kScaleFactor;
    g_out = v;
  }
  count_mine = (hi ? countB + lb.k : countA + la.k);
}

// The pair form (igh families with at most 32 D and 32 J alleles) as two kernels, because the two halves want
// different launch shapes: junction_vd_kernel sweeps the V-D junction with a wave per sample (the V genes fill
// its lanes, 156 VGPRs, three waves per SIMD) and leaves the D-germline forward vector and its scaler count in
// dxf / dxc; junction_dj_kernel sweeps the D-J junctions of TWO samples per wave (junction_pair; few registers,
// every wave slot of the CU in use).  Run inside one kernel -- waves 2 and 3 of a workgroup retiring after the
// V-D half -- the freed slots could not be refilled before the whole workgroup had finished (K2 0.81 -> 0.80 ms
// only).
// (113 VGPRs, four waves per SIMD; a 96-register budget for five waves: 0.28 -> 0.32 ms, 80 for six: 0.75 ms)
template <int GA, bool kExt>
__global__ void __launch_bounds__(64 * kJunctionWaves)
    junction_vd_kernel(const DevFamily fam, int n, const double* __restrict__ gem_all,
                       const int32_t* __restrict__ gcnt_all, const double* __restrict__ jem_all,
                       const int32_t* __restrict__ jrs_all, double* __restrict__ fwd_all,
                       int32_t* __restrict__ scal_all, double* __restrict__ dxf, int32_t* __restrict__ dxc) {
  extern __shared__ double jlds[];  // [NTI->NTI blocks of the vd right genes | kJunctionWaves jem slices]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x * kJunctionWaves + wave;
  const int NJ = fam.n_jcols;
  double* ntt_vd = jlds;
  double* jem = ntt_vd + 16 * (size_t)fam.vd.right_pad + (size_t)wave * (NJ + 1);
  for (int t = threadIdx.x; t < 16 * fam.vd.right_pad; t += 64 * kJunctionWaves) ntt_vd[t] = fam.vd.right_ntt[t];
  __syncthreads();
  if (s >= n) return;  // whole waves leave; nothing below synchronises across waves
  const int nV = fam.vgerm.n_genes, nD = fam.dgerm.n_genes;
  const size_t vd_fwd = (size_t)fam.vd.n_rows * (fam.vd.n_left + 5 * (size_t)fam.vd.n_right);
  {
    const double* src = jem_all + (size_t)s * NJ;
    for (int j = lane; j < NJ; j += 64) jem[j] = src[j];
    if (lane == 0) jem[NJ] = 0.0;  // what a state that cannot emit at a site looks up
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const double* gem = gem_all + (size_t)s * fam.gem_size;
  const int cv = gcnt_all[(size_t)s * 3 + 0], cd = gcnt_all[(size_t)s * 3 + 1];
  double* fwd = fwd_all ? fwd_all + (size_t)s * fam.forward_size : nullptr;
  int32_t* sco = scal_all ? scal_all + (size_t)s * fam.scaler_size : nullptr;
  // initial forward over the V germline region (src/HMM.cpp:291-319)
  double gV[GA];
  unsigned key = key_start<kExt>();
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
    }
    key = key_add<kExt>(key, v);
    gV[q] = v;
  }
  int vcount = cv;
  {
    const RowScale sc = wave_row_scale<kExt>(key);
#pragma unroll
    for (int q = 0; q < GA; ++q) gV[q] = sc.apply(gV[q]);
    vcount += sc.k;
  }
  if (fwd) {
#pragma unroll
    for (int q = 0; q < GA; ++q)
      if (lane + 64 * q < nV) fwd[lane + 64 * q] = gV[q];
  }
  if (sco && lane == 0) sco[0] = vcount;
  double gD[1];
  const int dcount =
      cd + junction_wave<GA, 1, kExt>(fam.vd, jem, ntt_vd, lane, gV, vcount, gem + 2 * (size_t)nV, nullptr, nullptr, gD,
                                      fwd ? fwd + nV : nullptr, sco ? sco + 1 : nullptr,
                                      kExt ? jrs_all + (size_t)s * (fam.vd.n_rows + fam.dj.n_rows) : nullptr);
  if (fwd && lane < nD) fwd[nV + vd_fwd + lane] = gD[0];
  if (sco && lane == 0) sco[1 + fam.vd.n_rows] = dcount;
  if (lane < 32) dxf[(size_t)s * 32 + lane] = gD[0];  // zero beyond the last D gene
  if (lane == 0) dxc[s] = dcount;
}

// V-D half of the pair form with two samples per wave (junction_vd_pair); same outputs as junction_vd_kernel.
constexpr int kVdPairWaves = 4;  // waves per workgroup (eight samples)
// (167 VGPRs, three waves = six samples per SIMD: 0.327 ms per 49 152 against 0.567 for junction_vd_kernel; forced to
// four waves it spills: 0.358, two waves: 0.393)
template <int GA, bool kExt>
__global__ void __launch_bounds__(64 * kVdPairWaves)
    junction_vd2_kernel(const DevFamily fam, int n, const double* __restrict__ gem_all,
                        const int32_t* __restrict__ gcnt_all, const double* __restrict__ jem_all,
                        const int32_t* __restrict__ jrs_all, double* __restrict__ fwd_all,
                        int32_t* __restrict__ scal_all, double* __restrict__ dxf, int32_t* __restrict__ dxc) {
  extern __shared__ double jlds[];  // [NTI->NTI blocks of the vd right genes | 2 * kVdPairWaves jem slices]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NJ = fam.n_jcols;
  double* ntt_vd = jlds;
  double* jem0 = ntt_vd + 16 * (size_t)fam.vd.right_pad + (size_t)(2 * wave) * (NJ + 1);
  for (int t = threadIdx.x; t < 16 * fam.vd.right_pad; t += 64 * kVdPairWaves) ntt_vd[t] = fam.vd.right_ntt[t];
  __syncthreads();
  const int p = blockIdx.x * kVdPairWaves + wave;
  if (2 * p >= n) return;  // neither sample exists; nothing below synchronises across waves
  const bool hi = lane >= 32;
  const int sA = 2 * p, sB = min(2 * p + 1, n - 1);
  const bool validB = 2 * p + 1 < n;  // an absent second sample reads the first one's inputs and writes nothing
  for (int h = 0; h < 2; ++h) {
    const double* src = jem_all + (size_t)(h ? sB : sA) * NJ;
    double* dst = jem0 + (size_t)h * (NJ + 1);
    for (int j = lane; j < NJ; j += 64) dst[j] = src[j];
    if (lane == 0) dst[NJ] = 0.0;  // what a state that cannot emit at a site looks up
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int nV = fam.vgerm.n_genes, nD = fam.dgerm.n_genes;
  const size_t vd_fwd = (size_t)fam.vd.n_rows * (fam.vd.n_left + 5 * (size_t)fam.vd.n_right);
  const double* gemA = gem_all + (size_t)sA * fam.gem_size;
  const double* gemB = gem_all + (size_t)sB * fam.gem_size;
  double* fwdA = fwd_all ? fwd_all + (size_t)sA * fam.forward_size : nullptr;
  double* fwdB = fwd_all && validB ? fwd_all + (size_t)sB * fam.forward_size : nullptr;
  int32_t* scoA = scal_all ? scal_all + (size_t)sA * fam.scaler_size : nullptr;
  int32_t* scoB = scal_all && validB ? scal_all + (size_t)sB * fam.scaler_size : nullptr;
  // initial forward over the V germline region (src/HMM.cpp:291-319), both samples
  double gA[GA], gB[GA];
  unsigned keyA = key_start<kExt>(), keyB = key_start<kExt>();
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int t = lane + 64 * q;
    double va = 0.0, vb = 0.0;
    if (t < nV) {
      const double c0 = fam.vgerm_gene_prob[t], c1 = fam.vpadding_transition[t], c2 = fam.vgerm_trans_prod[t];
      va = c0;
      va *= c1;
      va *= gemA[t];
      va *= c2;
      va *= gemA[nV + t];
      vb = c0;
      vb *= c1;
      vb *= gemB[t];
      vb *= c2;
      vb *= gemB[nV + t];
    }
    keyA = key_add<kExt>(keyA, va);
    keyB = key_add<kExt>(keyB, vb);
    gA[q] = va;
    gB[q] = vb;
  }
  int countA = gcnt_all[(size_t)sA * 3 + 0], countB = gcnt_all[(size_t)sB * 3 + 0];
  {
    const RowScale sa = wave_row_scale<kExt>(keyA), sb = wave_row_scale<kExt>(keyB);
#pragma unroll
    for (int q = 0; q < GA; ++q) {
      gA[q] = sa.apply(gA[q]);
      gB[q] = sb.apply(gB[q]);
    }
    countA += sa.k;
    countB += sb.k;
  }
  if (fwdA) {
#pragma unroll
    for (int q = 0; q < GA; ++q)
      if (lane + 64 * q < nV) {
        fwdA[lane + 64 * q] = gA[q];
        if (fwdB) fwdB[lane + 64 * q] = gB[q];
      }
  }
  if (lane == 0) {
    if (scoA) scoA[0] = countA;
    if (scoB) scoB[0] = countB;
  }
  const size_t jr = (size_t)fam.vd.n_rows + fam.dj.n_rows;
  double gD;
  int count_mine;
  junction_vd_pair<GA, kExt>(fam.vd, jem0, jem0 + (NJ + 1), ntt_vd, lane, hi, gA, gB, countA, countB,
                             (hi ? gemB : gemA) + 2 * (size_t)nV, gD, count_mine, fwdA ? fwdA + nV : nullptr,
                             fwdB ? fwdB + nV : nullptr, scoA ? scoA + 1 : nullptr, scoB ? scoB + 1 : nullptr,
                             kExt ? jrs_all + (size_t)sA * jr : nullptr, kExt ? jrs_all + (size_t)sB * jr : nullptr);
  const unsigned g = lane & 31;
  const int dcount = gcnt_all[(size_t)(hi ? sB : sA) * 3 + 1] + count_mine;
  if (!hi || validB) {
    const int sM = hi ? sB : sA;
    double* fwdM = hi ? fwdB : fwdA;
    int32_t* scoM = hi ? scoB : scoA;
    if (fwdM && (int)g < nD) fwdM[nV + vd_fwd + g] = gD;
    if (scoM && g == 0) scoM[1 + fam.vd.n_rows] = dcount;
    dxf[(size_t)sM * 32 + g] = gD;  // zero beyond the last D gene
    if (g == 0) dxc[sM] = dcount;
  }
}

constexpr int kPairWaves = 4;  // waves per junction_dj_kernel workgroup (eight samples)

// (122 VGPRs, four waves per SIMD: capped at 80 / 64 registers the kernel spills and takes 2.9x / 4.9x as long)
template <bool kExt>
__global__ void __launch_bounds__(64 * kPairWaves)
    junction_dj_kernel(const DevFamily fam, int n, const double* __restrict__ gem_all,
                       const int32_t* __restrict__ gcnt_all, const double* __restrict__ jem_all,
                       const int32_t* __restrict__ jrs_all, const double* __restrict__ dxf,
                       const int32_t* __restrict__ dxc, double* __restrict__ loglik, double* __restrict__ fwd_all,
                       int32_t* __restrict__ scal_all) {
  extern __shared__ double jlds[];  // [NTI->NTI blocks of the dj right genes | 2 * kPairWaves jem slices]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NJ = fam.n_jcols;
  double* ntt_dj = jlds;
  double* jem0 = ntt_dj + 16 * (size_t)fam.dj.right_pad + (size_t)(2 * wave) * (NJ + 1);
  for (int t = threadIdx.x; t < 16 * fam.dj.right_pad; t += 64 * kPairWaves) ntt_dj[t] = fam.dj.right_ntt[t];
  __syncthreads();
  // this wave's two samples: lanes 0-31 -> sample 2 p, lanes 32-63 -> sample 2 p + 1
  const int p = blockIdx.x * kPairWaves + wave;
  if (2 * p >= n) return;  // neither sample exists
  const bool hi = lane >= 32;
  const int s2 = 2 * p + (hi ? 1 : 0);
  const bool valid = s2 < n;
  const int sr = valid ? s2 : n - 1;  // an absent second sample reads the last one's inputs and writes nothing
  for (int h = 0; h < 2; ++h) {
    const int sh = min(2 * p + h, n - 1);
    const double* src = jem_all + (size_t)sh * NJ;
    double* dst = jem0 + (size_t)h * (NJ + 1);
    for (int j = lane; j < NJ; j += 64) dst[j] = src[j];
    if (lane == 0) dst[NJ] = 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int nV = fam.vgerm.n_genes, nD = fam.dgerm.n_genes, nJ = fam.jgerm.n_genes;
  const size_t vd_fwd = (size_t)fam.vd.n_rows * (fam.vd.n_left + 5 * (size_t)fam.vd.n_right);
  const unsigned g = lane & 31;
  const double* gem = gem_all + (size_t)sr * fam.gem_size;
  const double* jgerm_em = gem + 2 * (size_t)nV + nD;
  const double* jpad_em = jgerm_em + nJ;
  const int cj = gcnt_all[(size_t)sr * 3 + 2];
  double* fwd = fwd_all ? fwd_all + (size_t)sr * fam.forward_size + nV + vd_fwd + nD : nullptr;
  int32_t* sco = scal_all ? scal_all + (size_t)sr * fam.scaler_size + 2 + fam.vd.n_rows : nullptr;
  const double f_in = valid ? dxf[(size_t)s2 * 32 + g] : 0.0;
  const int dcount = valid ? dxc[s2] : 0;
  double gJ;
  const int jcount =
      cj + junction_pair<kExt>(fam.dj, jem0 + (size_t)(hi ? 1 : 0) * (NJ + 1), ntt_dj, lane, hi, f_in, dcount, jgerm_em,
                               fam.jpadding_transition, jpad_em, gJ, fwd, sco,
                               kExt ? jrs_all + (size_t)sr * (fam.vd.n_rows + fam.dj.n_rows) + fam.vd.n_rows : nullptr,
                               valid);
  if (fwd && valid && (int)g < nJ) fwd[(size_t)fam.dj.n_rows * (fam.dj.n_left + 5 * (size_t)fam.dj.n_right) + g] = gJ;
  if (sco && valid && g == 0) sco[fam.dj.n_rows] = jcount;
  // HMM::LogLikelihood (src/HMM.cpp:352-353)
  const HalfPair tot = half_sums(gJ);  // zero beyond the last J gene
  const double part = hi ? tot.b : tot.a;
  if (valid && g == 0) loglik[s2] = log(part) - jcount * kLogScaleFactor;
}

// GA: register slots for the V genes (ceil(nV / 64)); GB: slots for the D and J genes.
template <int GA, int GB, bool kExt = false>
__global__ void __launch_bounds__(64 * kJunctionWaves)
    junction_kernel(const DevFamily fam, int n, const double* __restrict__ gem_all,
                    const int32_t* __restrict__ gcnt_all, const double* __restrict__ jem_all,
                    const int32_t* __restrict__ jrs_all, double* __restrict__ loglik, double* __restrict__ fwd_all,
                    int32_t* __restrict__ scal_all) {
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
  unsigned key = key_start<kExt>();
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
    }
    key = key_add<kExt>(key, v);
    gV[q] = v;
  }
  int vcount = cv;
  const int32_t* jrs = kExt ? jrs_all + (size_t)s * (fam.vd.n_rows + (fam.has_d ? fam.dj.n_rows : 0)) : nullptr;
  {
    const RowScale sc = wave_row_scale<kExt>(key);
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
    const int dcount = cd + junction_wave<GA, GB, kExt>(fam.vd, jem, ntt_vd, lane, gV, vcount, dgerm_em, nullptr,
                                                        nullptr, gD, fwd, sco, jrs);
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
    jcount = cj + junction_wave<GB, GB, kExt>(fam.dj, jem, ntt_dj, lane, gD, dcount, jgerm_em,
                                              fam.jpadding_transition, jpad_em, gJ, fwd, sco,
                                              kExt ? jrs + fam.vd.n_rows : nullptr);
    if (fwd) fwd += (size_t)fam.dj.n_rows * (fam.dj.n_left + 5 * (size_t)fam.dj.n_right);
    if (sco) sco += fam.dj.n_rows;
  } else {
    const double* jgerm_em = gem + 2 * (size_t)nV;
    const double* jpad_em = jgerm_em + nJ;
    jcount = cj + junction_wave<GA, GB, kExt>(fam.vd, jem, ntt_vd, lane, gV, vcount, jgerm_em,
                                              fam.jpadding_transition, jpad_em, gJ, fwd, sco, jrs);
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
  // (the pair-form kernels keep two jem slices per wave and one junction's NTI blocks: never more than this)
  return ((size_t)2 * kJunctionWaves * (fam.n_jcols + 1) +
          16 * ((size_t)fam.vd.right_pad + (fam.has_d ? fam.dj.right_pad : 0))) *
         sizeof(double);
}

// Two samples per wave on the D-J junction (junction_kernel_pair): igh families with at most 32 D and J alleles.
static bool junction_pair_form(const DevFamily& fam) {
  static const bool off = debug_options().k2b_no_pair;  // test hook: the one-sample-per-wave form
  return !off && fam.has_d && fam.dgerm.n_genes <= 32 && fam.jgerm.n_genes <= 32;
}

static size_t emission_lds_bytes(const DevFamily& fam, bool ext) {
  const size_t cap = (size_t)cons_capacity(fam);
  return ((((size_t)fam.n_ucol + 2) & ~(size_t)1) + 6) * sizeof(double) + (2 * kFwdWaves + 2) * sizeof(int) +
         (cap ? (2 * cap + 4) * sizeof(double) + (cap + 4) * sizeof(int) : 0) +
         (ext ? ((size_t)fam.n_ucol + 2) * sizeof(int) : 0);
}

size_t forward_lds_bytes(const DevFamily& fam) {
  const size_t a = emission_lds_bytes(fam, true), b = junction_lds_bytes(fam);
  return a > b ? a : b;
}

template <int kG, bool kSite, bool kByteOff, bool kExt>
static void launch_emission_k(const DevFamily& fam, const DevFamily* fam_dev, int n, int R, const double* site_lik,
                              const int32_t* site_scal, const double* pi, const double* em_in, double* em_out,
                              double* gem, int32_t* gcnt, double* jem, int32_t* jrs, hipStream_t stream) {
  const size_t lds = emission_lds_bytes(fam, kExt);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(emission_kernel<kG, kSite, kByteOff, kExt>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((emission_kernel<kG, kSite, kByteOff, kExt>), dim3(n), dim3(kFwdThreads), lds, stream, fam_dev, R,
                     site_lik, site_scal, pi, em_in, em_out, gem, gcnt, jem, jrs);
}

template <int kG>
static void launch_emission_g(const DevFamily& fam, const DevFamily* fam_dev, int n, int R, const double* site_lik,
                              const int32_t* site_scal, const double* pi, const double* em_in, double* em_out,
                              double* gem, int32_t* gcnt, double* jem, int32_t* jrs, bool ext, hipStream_t stream) {
#define LH_ARGS fam, fam_dev, n, R, site_lik, site_scal, pi, em_in, em_out, gem, gcnt, jem, jrs, stream
  if (ext) {  // opt-in mode: one index form is enough
    if (site_lik) {
      if (fam.idx_byte_offsets)
        launch_emission_k<kG, true, true, true>(LH_ARGS);
      else
        launch_emission_k<kG, true, false, true>(LH_ARGS);
    } else {
      if (fam.idx_byte_offsets)
        launch_emission_k<kG, false, true, true>(LH_ARGS);
      else
        launch_emission_k<kG, false, false, true>(LH_ARGS);
    }
  } else if (site_lik) {
    if (fam.idx_byte_offsets)
      launch_emission_k<kG, true, true, false>(LH_ARGS);
    else
      launch_emission_k<kG, true, false, false>(LH_ARGS);
  } else {
    if (fam.idx_byte_offsets)
      launch_emission_k<kG, false, true, false>(LH_ARGS);
    else
      launch_emission_k<kG, false, false, false>(LH_ARGS);
  }
#undef LH_ARGS
}

template <int GA, int GB>
static void launch_junction_g(const DevFamily& fam, int n, const double* gem, const int32_t* gcnt, const double* jem,
                              const int32_t* jrs, double* dxf, int32_t* dxc, double* loglik, double* forward_out,
                              int32_t* scaler_out, bool ext, hipStream_t stream) {
  if (GB == 1 && junction_pair_form(fam)) {
    const size_t lds_vd = ((size_t)kJunctionWaves * (fam.n_jcols + 1) + 16 * (size_t)fam.vd.right_pad) * sizeof(double);
    const size_t lds_dj = ((size_t)2 * kPairWaves * (fam.n_jcols + 1) + 16 * (size_t)fam.dj.right_pad) * sizeof(double);
    const dim3 grid_vd((n + kJunctionWaves - 1) / kJunctionWaves), grid_dj(((n + 1) / 2 + kPairWaves - 1) / kPairWaves);
    static const bool vd_single = debug_options().k2b_vd_single;  // test hook: one sample per V-D wave
    if (!vd_single && GA <= 4) {  // (beyond 256 V alleles the second sample's left-gene registers no longer fit)
      const size_t lds_vd2 = ((size_t)2 * kVdPairWaves * (fam.n_jcols + 1) + 16 * (size_t)fam.vd.right_pad) * sizeof(double);
      const dim3 grid_vd2(((n + 1) / 2 + kVdPairWaves - 1) / kVdPairWaves);
#define LH_PAIR2_LAUNCH(E)                                                                                            \
  {                                                                                                                   \
    if (lds_vd2 > 64 * 1024)                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(junction_vd2_kernel<GA, E>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_vd2);                            \
    if (lds_dj > 64 * 1024)                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(junction_dj_kernel<E>),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dj);                             \
    hipLaunchKernelGGL((junction_vd2_kernel<GA, E>), grid_vd2, dim3(64 * kVdPairWaves), lds_vd2, stream, fam, n, gem, \
                       gcnt, jem, jrs, forward_out, scaler_out, dxf, dxc);                                            \
    hipLaunchKernelGGL((junction_dj_kernel<E>), grid_dj, dim3(64 * kPairWaves), lds_dj, stream, fam, n, gem, gcnt,    \
                       jem, jrs, dxf, dxc, loglik, forward_out, scaler_out);                                          \
  }
      if (ext)
        LH_PAIR2_LAUNCH(true)
      else
        LH_PAIR2_LAUNCH(false)
#undef LH_PAIR2_LAUNCH
      return;
    }
#define LH_PAIR_LAUNCH(E)                                                                                             \
  {                                                                                                                   \
    if (lds_vd > 64 * 1024)                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(junction_vd_kernel<GA, E>),                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_vd);                             \
    if (lds_dj > 64 * 1024)                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(junction_dj_kernel<E>),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dj);                             \
    hipLaunchKernelGGL((junction_vd_kernel<GA, E>), grid_vd, dim3(64 * kJunctionWaves), lds_vd, stream, fam, n, gem,  \
                       gcnt, jem, jrs, forward_out, scaler_out, dxf, dxc);                                            \
    hipLaunchKernelGGL((junction_dj_kernel<E>), grid_dj, dim3(64 * kPairWaves), lds_dj, stream, fam, n, gem, gcnt,    \
                       jem, jrs, dxf, dxc, loglik, forward_out, scaler_out);                                          \
  }
    if (ext)
      LH_PAIR_LAUNCH(true)
    else
      LH_PAIR_LAUNCH(false)
#undef LH_PAIR_LAUNCH
    return;
  }
  const size_t lds = junction_lds_bytes(fam);
  const dim3 grid((n + kJunctionWaves - 1) / kJunctionWaves), block(64 * kJunctionWaves);
  if (ext) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(junction_kernel<GA, GB, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((junction_kernel<GA, GB, true>), grid, block, lds, stream, fam, n, gem, gcnt, jem, jrs, loglik,
                       forward_out, scaler_out);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(junction_kernel<GA, GB, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((junction_kernel<GA, GB, false>), grid, block, lds, stream, fam, n, gem, gcnt, jem, jrs, loglik,
                       forward_out, scaler_out);
  }
}

template <int GA>
static void launch_junction_a(int gb, const DevFamily& fam, int n, const double* gem, const int32_t* gcnt,
                              const double* jem, const int32_t* jrs, double* dxf, int32_t* dxc, double* loglik,
                              double* forward_out, int32_t* scaler_out, bool ext, hipStream_t stream) {
  if (gb <= 1)
    launch_junction_g<GA, 1>(fam, n, gem, gcnt, jem, jrs, dxf, dxc, loglik, forward_out, scaler_out, ext, stream);
  else if (gb <= 2)
    launch_junction_g<GA, 2>(fam, n, gem, gcnt, jem, jrs, dxf, dxc, loglik, forward_out, scaler_out, ext, stream);
  else
    launch_junction_g<GA, 4>(fam, n, gem, gcnt, jem, jrs, dxf, dxc, loglik, forward_out, scaler_out, ext, stream);
}

// site_lik != null: emissions are assembled from K1's output (em_out optional);
// site_lik == null: emissions are taken from em_in (SimpleHMM / lh_forward_batch).
// gem [n][gem_size], gcnt [n][3], jem [n][n_jcols]: per-sample hand-off buffers between K2a and K2b
// (+ jrs [n][junction rows] in the extended-range mode); dxf [n][32], dxc [n]: between the two K2b kernels of the
// pair form.
void launch_forward(const DevFamily& fam, const DevFamily* fam_dev, int n, int R, const double* site_lik,
                    const int32_t* site_scal, const double* pi, const double* em_in, double* em_out, double* gem,
                    int32_t* gcnt, double* jem, int32_t* jrs, double* dxf, int32_t* dxc, double* loglik,
                    double* forward_out, int32_t* scaler_out, bool ext, hipStream_t stream) {
  const int slots = (fam.max_genes + kFwdThreads - 1) / kFwdThreads;
#define LH_ARGS fam, fam_dev, n, R, site_lik, site_scal, pi, em_in, em_out, gem, gcnt, jem, jrs, ext, stream
  if (slots <= 1)
    launch_emission_g<1>(LH_ARGS);
  else if (slots <= 2)
    launch_emission_g<2>(LH_ARGS);
  else
    launch_emission_g<4>(LH_ARGS);
#undef LH_ARGS
  const int ga = (fam.vgerm.n_genes + 63) / 64;
  const int gb = (std::max(fam.dgerm.n_genes, fam.jgerm.n_genes) + 63) / 64;
#define LH_ARGS gb, fam, n, gem, gcnt, jem, jrs, dxf, dxc, loglik, forward_out, scaler_out, ext, stream
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
