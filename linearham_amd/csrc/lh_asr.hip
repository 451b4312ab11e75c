// K3: ancestral-sequence sampling on the clonal tree (gfx950).
//
// Replaces the per-tree body of scripts/run_bootstrap_asr_ess.R:48-104 (R, phylomd::phylo.likelihood +
// phylomd::asr.sim, one core per tree): for every alignment site of a tree sample
//   1. the likelihood of the column (sampled naive base on the `naive` tip) on each rate-scaled tree -- these
//      are K1's per-rate planes site_lik[n][R][5][n_prune], taken for the site's naive base (:79-81);
//   2. one rate category drawn with those weights (:82);
//   3. one joint draw of all inner-node states given the tips on that tree (:84): root state from
//      pi_i * L_root(i) * P_naive[i][naive base], then every inner node given its parent's state from
//      P(parent -> child)[s_parent][c] * L_child(c), L = partial likelihood of the data below the node.
//
// Design:
//  * One workgroup = (rate category, tree sample); it handles the sites whose drawn category is its own, so
//    that a wave's P-matrices are wave-uniform.  Every workgroup of a sample repeats the (cheap) category
//    draw of all sites -- the draws are a pure function of (seed, sample, site) -- and compacts its own
//    sites into a list; the list position plus the number of sites of lower categories is the site's slot
//    in the sample's CLV area, so the R workgroups of a sample share one [T-2][4][L] area without holes.
//  * Upward pass: the K1 schedule (lh_schedule_tree), one distinct alignment pattern of the category per lane
//    (the CLVs do not depend on the naive base: naive hangs off the root).  Every inner CLV is needed again by
//    the downward pass, so it is stored: clv[n][op][2][slot] of 16-byte entries, slot fastest -- a wave writes
//    two contiguous runs of whole 128-byte lines per op.  That makes the register stack of K1 unnecessary: a
//    popped sibling is re-read from the CLV area.  Cherry nodes are not stored (product of two tip-table
//    columns, formed again where needed).  This kernel is the CLV-streaming kernel of SURVEY 8(d)'s byte
//    model (32 B written and 32 B read per (inner node, site)), minus what patterns and cherries save.
//  * P-matrices of the workgroup's (sample, rate) live in LDS: tip branches as the K1 tip table
//    tiptab[tip][state][4] (column `state` of P), inner branches row-major, indexed by the op that produced
//    the child.  The upward mat-vec reads them as wave-uniform (broadcast) ds_reads, the downward pass
//    gathers row s_parent per lane.
//  * Downward pass: the schedule in reverse; a lane carries the state of the accumulator child in a
//    register, states of popped siblings wait in a per-lane LDS byte stack (same slot numbers as K1's
//    register stack).  CLVs are rescaled freely on the way up (only ratios within a node matter).
//  * Random numbers: Philox4x32-10, counter = (site, draw, sample), key = seed; draw 0 = rate category,
//    1 = root, 2 + (v - T) = inner node v.  oracle/asr_oracle.py restates the same stream, so GPU and oracle
//    agree draw by draw.
#include <cstdlib>

#include "lh_device.h"

namespace lh {

namespace {

__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
  const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
  c0 = n0;
  c1 = lo1;
  c2 = n2;
  c3 = lo0;
}

// 53-bit uniform in [0, 1) of cell (sample, site, draw) of stream `seed`
__device__ __forceinline__ double asr_uniform(uint64_t seed, uint64_t sample, uint32_t site, uint32_t draw) {
  uint32_t c0 = site, c1 = draw, c2 = (uint32_t)sample, c3 = (uint32_t)(sample >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const uint64_t bits = (uint64_t)(c0 >> 5) * 67108864ull + (uint64_t)(c1 >> 6);
  return (double)bits * (1.0 / 9007199254740992.0);
}

// first category whose running sum exceeds u * total; the last one if none does (all weights zero)
__device__ __forceinline__ int draw4(const double (&w)[4], double u) {
  const double c0 = w[0], c1 = c0 + w[1], c2 = c1 + w[2], c3 = c2 + w[3];
  const double t = u * c3;
  return (int)(c0 <= t) + (int)(c1 <= t) + (int)(c2 <= t);
}

__device__ __forceinline__ void tip_col(const double* tiptab, int tip, int st, double (&c)[4]) {
  const double* t = tiptab + tip * 16;
  if (st < 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = t[st * 4 + i];
  } else {  // N: row sums of P
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = ((t[i] + t[4 + i]) + t[8 + i]) + t[12 + i];
  }
}

// x = P a, P row-major in LDS at a wave-uniform address
__device__ __forceinline__ void matvec_lds(const double* p, const double (&a)[4], double (&x)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = fma(p[i * 4 + 3], a[3], fma(p[i * 4 + 2], a[2], fma(p[i * 4 + 1], a[1], p[i * 4] * a[0])));
}

}  // namespace

// slots per sample in the CLV area: every category's region starts on a multiple of 8 slots
size_t asr_slots(int L, int R) { return (((size_t)L + 7) & ~(size_t)7) + 8 * (size_t)R + 8; }

size_t asr_lds_bytes(int T, int L, int R, int n_prune) {
  const size_t n_ops = (size_t)T - 2, NP = (size_t)n_prune + 1;
  size_t b = 0;
  b += (size_t)T * 16 * sizeof(double);          // tiptab
  b += n_ops * 16 * sizeof(double);              // pin
  b += n_ops * sizeof(int4);                     // ops
  b += n_ops * 2 * sizeof(int32_t);              // popped_op, node_of_op
  b += (size_t)L * sizeof(int32_t);              // list (sites of the category)
  b += NP * 2 * sizeof(int32_t);                 // plist, pslot
  b += 16 * 256;                                 // state stack [16][256]
  b += (((size_t)R * NP + 15) & ~(size_t)15);    // flags[R][NP]
  b += 16;                                       // counters
  return b;
}

// K3a: the rate category of every (sample, site): one thread each.  Weights = K1's per-rate column
// likelihoods for the site's naive base, scalers aligned to the smallest (the arithmetic of K2a's mixture).
__global__ void __launch_bounds__(256) asr_rate_kernel(int n, int R, int L, int n_prune,
                                                       const int32_t* __restrict__ site_pat,
                                                       const double* __restrict__ site_lik,
                                                       const int32_t* __restrict__ site_scal,
                                                       const uint8_t* __restrict__ naive, uint64_t seed,
                                                       uint64_t sample0, uint8_t* __restrict__ choice) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)n * L) return;
  const int sample = (int)(gid / L), j = (int)(gid - (long long)sample * L);
  const int pat = site_pat[j];
  const int b = naive[gid];
  const double u = asr_uniform(seed, sample0 + (uint64_t)sample, (uint32_t)j, 0u);
  int pick = R - 1;
  if (pat >= n_prune) {  // all-N column: every category has the same likelihood
    const double t = u * (double)R;
    for (int k = R - 1; k >= 0; --k)
      if (t < (double)(k + 1)) pick = k;
  } else {
    const int32_t* sc = site_scal + (size_t)sample * R * n_prune + pat;
    const double* lk = site_lik + ((size_t)sample * R * 5 + b) * n_prune + pat;
    int smin = 0x7fffffff;
    for (int k = 0; k < R; ++k) smin = min(smin, sc[(size_t)k * n_prune]);
    double total = 0.0;
    for (int k = 0; k < R; ++k) {
      double v = lk[(size_t)k * 5 * n_prune];
      const int d = sc[(size_t)k * n_prune] - smin;
      for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
      total += v;
    }
    const double t = u * total;
    double cum = 0.0;
    bool found = false;
    for (int k = 0; k < R; ++k) {
      double v = lk[(size_t)k * 5 * n_prune];
      const int d = sc[(size_t)k * n_prune] - smin;
      for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
      cum += v;
      if (!found && t < cum) {
        pick = k;
        found = true;
      }
    }
  }
  choice[gid] = (uint8_t)pick;
}

__global__ void __launch_bounds__(256) asr_kernel(int R, int T, int L, int n_prune, const uint8_t* __restrict__ msa,
                                                  const int32_t* __restrict__ site_pat,
                                                  const int32_t* __restrict__ ops, const double* __restrict__ brlen,
                                                  const double* __restrict__ rates, const double* __restrict__ eig,
                                                  const double* __restrict__ pi,
                                                  const uint8_t* __restrict__ choice_g,
                                                  const uint8_t* __restrict__ naive, uint64_t seed, uint64_t sample0,
                                                  double2* clv, int Lp, uint8_t* __restrict__ anc, int dbg_mode) {
  extern __shared__ double2 asr_smem[];
  const int n_ops = T - 2;
  const int NP = n_prune + 1;  // patterns, the all-N one (id n_prune) included
  double* tiptab = reinterpret_cast<double*>(asr_smem);           // [T][4][4]
  double* pin = tiptab + (size_t)T * 16;                          // [n_ops][4][4] (entry n_ops-1 unused)
  int4* ops_s = reinterpret_cast<int4*>(pin + (size_t)n_ops * 16);  // [n_ops]
  int32_t* popped_op = reinterpret_cast<int32_t*>(ops_s + n_ops);  // [n_ops]
  int32_t* node_of_op = popped_op + n_ops;                         // [n_ops]
  int32_t* list = node_of_op + n_ops;                              // [L]  sites of this category
  int32_t* plist = list + L;                                       // [NP] distinct patterns of those sites
  int32_t* pslot = plist + NP;                                     // [NP] pattern -> position in plist
  uint8_t* st_stack = reinterpret_cast<uint8_t*>(pslot + NP);      // [16][256]
  uint8_t* flags = st_stack + 16 * 256;                            // [R][NP] pattern present in category
  int32_t* misc = reinterpret_cast<int32_t*>(flags + (((size_t)R * NP + 15) & ~(size_t)15));  // counters

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_waves = blockDim.x >> 6;
  const int rate = blockIdx.x;
  const int sample = blockIdx.y;
  const uint64_t sample_id = sample0 + (uint64_t)sample;
  const int4* __restrict__ op_ptr = reinterpret_cast<const int4*>(ops) + (size_t)sample * n_ops;
  const uint8_t* __restrict__ ch = choice_g + (size_t)sample * L;

  for (int k = tid; k < n_ops; k += blockDim.x) ops_s[k] = op_ptr[k];
  for (int i = tid; i < R * NP; i += blockDim.x) flags[i] = 0;
  __syncthreads();
  // which patterns occur among the sites of each category (same value from every writer)
  for (int j = tid; j < L; j += blockDim.x) flags[(int)ch[j] * NP + min(site_pat[j], n_prune)] = 1;
  __syncthreads();

  // ---- last wave: the site list of this category (site order) and its distinct patterns (pattern order).
  // The CLVs of the upward pass depend on (category, pattern) only, so they are computed and stored once per
  // distinct pattern; the category's region of the sample's slot space starts on a multiple of 8 slots (one
  // 128-byte line of a 16-byte-per-slot plane) after the regions of the lower categories.
  if (wave == n_waves - 1) {
    int run = 0;
    for (int j0 = 0; j0 < L; j0 += 64) {
      const int j = j0 + lane;
      const int c = j < L ? (int)ch[j] : 255;
      const unsigned long long mine = __ballot(c == rate);
      if (c == rate) list[run + __popcll(mine & ((1ull << lane) - 1ull))] = j;
      run += __popcll(mine);
    }
    int base = 0, prun = 0;
    for (int k = 0; k <= rate; ++k) {
      int cntk = 0;
      for (int p0 = 0; p0 < NP; p0 += 64) {
        const int p = p0 + lane;
        const bool f = p < NP && flags[k * NP + p] != 0;
        const unsigned long long m = __ballot(f);
        if (k == rate && f) {
          const int pos = cntk + __popcll(m & ((1ull << lane) - 1ull));
          plist[pos] = p;
          pslot[p] = pos;
        }
        cntk += __popcll(m);
      }
      if (k < rate)
        base += (cntk + 7) & ~7;
      else
        prun = cntk;
    }
    if (lane == 0) {
      misc[0] = run;
      misc[1] = base;
      misc[2] = prun;
    }
  }
  __syncthreads();
  const int cnt = misc[0], base = misc[1], cntp = misc[2];
  if (cnt == 0) return;  // no site of this sample drew this category (uniform over the workgroup)

  // ---- schedule bookkeeping (one thread): which op produced the sibling a pop op takes from the stack, and
  // which tree node every op produces.  The 16 stack slots' op numbers sit in four 64-bit registers.
  if (tid == 0) {
    unsigned long long so[4] = {0, 0, 0, 0};
    long long named = 0;
    for (int k = 0; k < n_ops; ++k) {
      const int4 op = ops_s[k];
      const int kind = op.x & 15;
      const int sh = (op.w & 3) * 16, wi = (op.w >> 2) & 3;
      if (op.x & OP_PUSH_FLAG) {
        const unsigned long long v = (unsigned long long)(unsigned)(k - 1) << sh, m = ~(0xffffull << sh);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i == wi) so[i] = (so[i] & m) | v;
      }
      if (kind == OP_TIP_ACC) {
        node_of_op[k - 1] = op.z;
        named += op.z;
      } else if (kind == OP_POP_ACC) {
        unsigned long long word = so[0];
#pragma unroll
        for (int i = 1; i < 4; ++i)
          if (i == wi) word = so[i];
        const int q = (int)((word >> sh) & 0xffffull);
        popped_op[k] = q;
        node_of_op[k - 1] = op.z;
        node_of_op[q] = op.y;
        named += op.z + op.y;
      }
    }
    // the root (naive's neighbour) is the one inner node no op names as a child
    const long long all = (long long)n_ops * T + (long long)n_ops * (n_ops - 1) / 2;
    node_of_op[n_ops - 1] = (int)(all - named);
  }

  // ---- P-matrices of the tip branches of this (sample, rate) -> tip table (columns of P)
  const double* __restrict__ e = eig + (size_t)sample * 36;
  const double rt = rates[(size_t)sample * R + rate];
  const double* __restrict__ bl = brlen + (size_t)sample * (2 * (size_t)T - 2);
  {
    double P[4][4];
    for (int j = tid; j < T; j += blockDim.x) {
      compute_pmatrix(e, bl[j] * rt, P);
      double* o = tiptab + j * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int st = 0; st < 4; ++st) o[st * 4 + i] = P[i][st];
    }
  }
  __syncthreads();

  // ---- inner-branch P-matrices, indexed by the op that produced the child
  {
    double P[4][4];
    for (int k = tid; k < n_ops; k += blockDim.x) {
      const int4 op = ops_s[k];
      const int kind = op.x & 15;
      if (kind == OP_CHERRY) continue;
      compute_pmatrix(e, bl[op.z] * rt, P);
      double* o = pin + (size_t)(k - 1) * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) o[i * 4 + q] = P[i][q];
      if (kind == OP_POP_ACC) {
        compute_pmatrix(e, bl[op.y] * rt, P);
        double* o2 = pin + (size_t)popped_op[k] * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) o2[i * 4 + q] = P[i][q];
      }
    }
  }
  __syncthreads();

  const double* __restrict__ p4 = pi + (size_t)sample * 4;
  const uint8_t* __restrict__ nv = naive + (size_t)sample * L;
  // CLV area of the sample: [op][2][Lp] double2 -- components (0,1) and (2,3) of a slot are 16-byte entries of
  // two planes, so a wave's store is one contiguous run of whole 128-byte lines per plane.  Cherry nodes are
  // not stored: their CLV is the product of two tip-table columns and is formed again where it is needed.
  const size_t plane = (size_t)Lp;
  double2* clv_s = clv + (size_t)sample * n_ops * 2 * plane;
  uint8_t* anc_s = anc + (size_t)sample * n_ops * (size_t)L;
  uint8_t* my_stack = st_stack + tid;

  if (dbg_mode == 1) return;
  // CLV of the node that op j produced, for the pattern `upat` (all_n: the all-N padding column) in slot gslot
  auto node_clv = [&](int j, unsigned upat, bool all_n, int gslot, double (&c)[4]) {
    const int4 oj = ops_s[j];
    const int kj = __builtin_amdgcn_readfirstlane(oj.x & 15);
    if (kj == OP_CHERRY) {
      const int jy = __builtin_amdgcn_readfirstlane(oj.y), jz = __builtin_amdgcn_readfirstlane(oj.z);
      const int s1 = all_n ? 4 : (int)msa[(unsigned)((jy - 1) * n_prune) + upat];
      const int s2 = all_n ? 4 : (int)msa[(unsigned)((jz - 1) * n_prune) + upat];
      double u[4], v[4];
      tip_col(tiptab, jy, s1, u);
      tip_col(tiptab, jz, s2, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = u[i] * v[i];
    } else {
      const double2* cj = clv_s + (size_t)j * 2 * plane + gslot;
      const double2 lo = cj[0], hi = cj[plane];
      c[0] = lo.x, c[1] = lo.y, c[2] = hi.x, c[3] = hi.y;
    }
  };

  // ---- upward pass over the category's distinct patterns, spread evenly over ALL waves of the workgroup (the
  // walk is bound by its dependent memory round trips, not by lanes).  What op k + 1 needs from memory (its
  // tips' states, or the CLV of the sibling it pops, stored at least two ops earlier) is requested while op k
  // computes.
  {
    const int per = min(64, (((cntp + n_waves - 1) / n_waves) + 15) & ~15);
    for (int s0 = wave * per; s0 < cntp; s0 += n_waves * per) {
      const int slot = s0 + lane;
      const bool active = lane < per && slot < cntp;
      const int pat = plist[active ? slot : cntp - 1];
      const int gslot = base + (active ? slot : cntp - 1);
      const bool all_n = pat >= n_prune;
      const unsigned upat = all_n ? 0u : (unsigned)pat;
      double a[4] = {1.0, 1.0, 1.0, 1.0};
      int sa, sb = 4;
      double y[4] = {0.0, 0.0, 0.0, 0.0};
      {
        const int4 op0 = ops_s[0];  // the first op is a cherry
        sa = all_n ? 4 : (int)msa[(unsigned)((op0.y - 1) * n_prune) + upat];
        sb = all_n ? 4 : (int)msa[(unsigned)((op0.z - 1) * n_prune) + upat];
      }
      for (int k = 0; k < n_ops; ++k) {
        const int4 op = ops_s[k];
        const int kind = __builtin_amdgcn_readfirstlane(op.x & 15);
        const int oy = __builtin_amdgcn_readfirstlane(op.y), oz = __builtin_amdgcn_readfirstlane(op.z);
        int sa_n = 4, sb_n = 4;
        double y_n[4] = {0.0, 0.0, 0.0, 0.0};
        if (k + 1 < n_ops) {
          const int4 on = ops_s[k + 1];
          const int kn = __builtin_amdgcn_readfirstlane(on.x & 15);
          const int ny = __builtin_amdgcn_readfirstlane(on.y), nz = __builtin_amdgcn_readfirstlane(on.z);
          if (kn == OP_POP_ACC) {
            node_clv(__builtin_amdgcn_readfirstlane(popped_op[k + 1]), upat, all_n, gslot, y_n);
          } else {
            if (!all_n) sa_n = (int)msa[(unsigned)((ny - 1) * n_prune) + upat];
            if (kn == OP_CHERRY && !all_n) sb_n = (int)msa[(unsigned)((nz - 1) * n_prune) + upat];
          }
        }
        double u[4], v[4];
        if (kind == OP_CHERRY) {
          tip_col(tiptab, oy, sa, u);
          tip_col(tiptab, oz, sb, v);
        } else {
          matvec_lds(pin + (size_t)(k - 1) * 16, a, v);
          if (kind == OP_TIP_ACC) {
            tip_col(tiptab, oy, sa, u);
          } else {
            const int q = __builtin_amdgcn_readfirstlane(popped_op[k]);
            matvec_lds(pin + (size_t)q * 16, y, u);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = u[i] * v[i];
        if (kind != OP_CHERRY) {
          if (fmax(fmax(a[0], a[1]), fmax(a[2], a[3])) < kScaleThreshold) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] *= kScaleFactor;
          }
          if (active) {
            double2* ck = clv_s + (size_t)k * 2 * plane + gslot;
            ck[0] = make_double2(a[0], a[1]);
            ck[plane] = make_double2(a[2], a[3]);
          }
        }
        sa = sa_n;
        sb = sb_n;
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = y_n[i];
      }
    }
  }
  if (dbg_mode == 2) return;
  // the sites' lanes below read CLVs that other lanes and waves of this workgroup stored above
  __threadfence_block();
  __syncthreads();

  // ---- downward pass over the category's sites: the schedule in reverse; the children's CLVs of op k - 1 are
  // requested while op k draws (their addresses do not depend on the states)
  {
    const int per = min(64, (((cnt + n_waves - 1) / n_waves) + 15) & ~15);
    for (int s0 = wave * per; s0 < cnt; s0 += n_waves * per) {
      const int slot = s0 + lane;
      const bool active = lane < per && slot < cnt;
      const int site = list[active ? slot : cnt - 1];
      const int pat = min(site_pat[site], n_prune);
      const bool all_n = pat >= n_prune;
      const unsigned upat = all_n ? 0u : (unsigned)pat;
      const int gslot = base + pslot[pat];
      const int b_naive = nv[site];
      int s_acc = 0;
      double c_acc[4] = {0.0, 0.0, 0.0, 0.0}, c_pop[4] = {0.0, 0.0, 0.0, 0.0}, c_root[4];
      node_clv(n_ops - 1, upat, all_n, gslot, c_root);
      {
        const int k = n_ops - 1;
        const int kind = __builtin_amdgcn_readfirstlane(ops_s[k].x & 15);
        if (kind != OP_CHERRY) node_clv(k - 1, upat, all_n, gslot, c_acc);
        if (kind == OP_POP_ACC) node_clv(__builtin_amdgcn_readfirstlane(popped_op[k]), upat, all_n, gslot, c_pop);
      }
      for (int k = n_ops - 1; k >= 0; --k) {
        const int4 op = ops_s[k];
        const int kind = __builtin_amdgcn_readfirstlane(op.x & 15);
        double n_acc[4] = {0.0, 0.0, 0.0, 0.0}, n_pop[4] = {0.0, 0.0, 0.0, 0.0};
        if (k > 0) {
          const int kp = __builtin_amdgcn_readfirstlane(ops_s[k - 1].x & 15);
          if (kp != OP_CHERRY) node_clv(k - 2, upat, all_n, gslot, n_acc);
          if (kp == OP_POP_ACC)
            node_clv(__builtin_amdgcn_readfirstlane(popped_op[k - 1]), upat, all_n, gslot, n_pop);
        }
        int s_cur;
        if (k == n_ops - 1) {
          // root = naive's neighbour: pi_i * L_root(i) * P_naive[i][naive base]
          double down[4], w[4];
          tip_col(tiptab, 0, b_naive, down);
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = p4[i] * c_root[i] * down[i];
          s_cur = draw4(w, asr_uniform(seed, sample_id, (uint32_t)site, 1u));
          if (active) anc_s[(size_t)(node_of_op[k] - T) * L + site] = (uint8_t)s_cur;
        } else {
          const int4 nxt = ops_s[k + 1];
          const int nx = __builtin_amdgcn_readfirstlane(nxt.x), nw = __builtin_amdgcn_readfirstlane(nxt.w);
          s_cur = (nx & OP_PUSH_FLAG) ? (int)my_stack[nw * 256] : s_acc;
        }
        if (kind != OP_CHERRY) {
          const int j = k - 1;  // the accumulator child
          const int node = __builtin_amdgcn_readfirstlane(node_of_op[j]);
          const double* prow = pin + (size_t)j * 16 + s_cur * 4;
          double w[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = prow[i] * c_acc[i];
          s_acc = draw4(w, asr_uniform(seed, sample_id, (uint32_t)site, 2u + (uint32_t)(node - T)));
          if (active) anc_s[(size_t)(node - T) * L + site] = (uint8_t)s_acc;
        }
        if (kind == OP_POP_ACC) {
          const int q = __builtin_amdgcn_readfirstlane(popped_op[k]);
          const int node = __builtin_amdgcn_readfirstlane(node_of_op[q]);
          const int ow = __builtin_amdgcn_readfirstlane(op.w);
          const double* prow = pin + (size_t)q * 16 + s_cur * 4;
          double w[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = prow[i] * c_pop[i];
          const int sq = draw4(w, asr_uniform(seed, sample_id, (uint32_t)site, 2u + (uint32_t)(node - T)));
          if (active) anc_s[(size_t)(node - T) * L + site] = (uint8_t)sq;
          my_stack[ow * 256] = (uint8_t)sq;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          c_acc[i] = n_acc[i];
          c_pop[i] = n_pop[i];
        }
      }
    }
  }
}

int launch_asr(const DevFamily& fam, int n, int R, int T, const int32_t* ops, const double* brlen, const double* rates,
               const double* eig, const double* pi, const double* site_lik, const int32_t* site_scal,
               const uint8_t* naive, uint64_t seed, uint64_t sample0, double* clv, uint8_t* anc,
               uint8_t* rate_choice, hipStream_t stream) {
  const int L = fam.n_sites;
  const size_t lds = asr_lds_bytes(T, L, R, fam.n_prune);
  if (lds > 160 * 1024 || L < 1) return 1;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(asr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  const long long cells = (long long)n * L;
  hipLaunchKernelGGL(asr_rate_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, n, R, L,
                     fam.n_prune, fam.site_pat, site_lik, site_scal, naive, seed, sample0, rate_choice);
  hipLaunchKernelGGL(asr_kernel, dim3(R, n), dim3(256), lds, stream, R, T, L, fam.n_prune, fam.msa, fam.site_pat, ops,
                     brlen, rates, eig, pi, (const uint8_t*)rate_choice, naive, seed, sample0,
                     reinterpret_cast<double2*>(clv), (int)asr_slots(L, R), anc,
                     getenv("LH_ASR_DBG") ? atoi(getenv("LH_ASR_DBG")) : 0);
  return 0;
}

}  // namespace lh
