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
//  * Upward pass: the K1 schedule (lh_schedule_tree), one site per lane.  Every inner CLV is needed again
//    by the downward pass, so it is stored: clv[n][op][4][slot], component planes, slot fastest -- a wave
//    writes four 512-byte runs per op.  That makes the register stack of K1 unnecessary: a popped sibling is
//    re-read from the CLV area.  This kernel is the CLV-streaming kernel of SURVEY 8(d)'s byte model:
//    32 B written and 32 B read per (inner node, site).
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

size_t asr_lds_bytes(int T, int L) {
  const size_t n_ops = (size_t)T - 2;
  size_t b = 0;
  b += (size_t)T * 16 * sizeof(double);  // tiptab
  b += n_ops * 16 * sizeof(double);      // pin
  b += n_ops * sizeof(int4);             // ops
  b += n_ops * 2 * sizeof(int32_t);      // popped_op, node_of_op
  b += (size_t)L * sizeof(int32_t);      // list
  b += 16 * 256;                         // state stack [16][256]
  b += (((size_t)L + 15) & ~(size_t)15); // choice
  b += 128;                              // counters + slot_op
  return b;
}

__global__ void __launch_bounds__(256) asr_kernel(int R, int T, int L, int n_prune, const uint8_t* __restrict__ msa,
                                                  const int32_t* __restrict__ site_pat,
                                                  const int32_t* __restrict__ ops, const double* __restrict__ brlen,
                                                  const double* __restrict__ rates, const double* __restrict__ eig,
                                                  const double* __restrict__ pi, const double* __restrict__ site_lik,
                                                  const int32_t* __restrict__ site_scal,
                                                  const uint8_t* __restrict__ naive, uint64_t seed, uint64_t sample0,
                                                  double* clv, uint8_t* __restrict__ anc,
                                                  uint8_t* __restrict__ rate_choice) {
  extern __shared__ double2 asr_smem[];
  const int n_ops = T - 2;
  double* tiptab = reinterpret_cast<double*>(asr_smem);           // [T][4][4]
  double* pin = tiptab + (size_t)T * 16;                          // [n_ops][4][4] (entry n_ops-1 unused)
  int4* ops_s = reinterpret_cast<int4*>(pin + (size_t)n_ops * 16);  // [n_ops]
  int32_t* popped_op = reinterpret_cast<int32_t*>(ops_s + n_ops);  // [n_ops]
  int32_t* node_of_op = popped_op + n_ops;                         // [n_ops]
  int32_t* list = node_of_op + n_ops;                              // [L]
  uint8_t* st_stack = reinterpret_cast<uint8_t*>(list + L);        // [16][256]
  uint8_t* choice = st_stack + 16 * 256;                           // [L]
  int32_t* misc = reinterpret_cast<int32_t*>(choice + ((L + 15) & ~15));  // cnt, base, slot_op[...]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_waves = blockDim.x >> 6;
  const int rate = blockIdx.x;
  const int sample = blockIdx.y;
  const uint64_t sample_id = sample0 + (uint64_t)sample;
  const int4* __restrict__ op_ptr = reinterpret_cast<const int4*>(ops) + (size_t)sample * n_ops;

  for (int k = tid; k < n_ops; k += blockDim.x) ops_s[k] = op_ptr[k];
  __syncthreads();

  // ---- schedule bookkeeping (one thread; ~10 cycles per op out of LDS): which op produced the sibling a
  // pop op takes from the stack, and which tree node every op produces
  if (tid == 0) {
    int32_t* slot_op = misc + 2;  // [16]
    long long named = 0;
    for (int k = 0; k < n_ops; ++k) {
      const int4 op = ops_s[k];
      const int kind = op.x & 15;
      if (op.x & OP_PUSH_FLAG) slot_op[op.w] = k - 1;
      if (kind == OP_TIP_ACC) {
        node_of_op[k - 1] = op.z;
        named += op.z;
      } else if (kind == OP_POP_ACC) {
        const int q = slot_op[op.w];
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

  // ---- rate category of every site of the sample (every workgroup of the sample computes the same)
  {
    const uint8_t* __restrict__ nv = naive + (size_t)sample * L;
    for (int j = tid; j < L; j += blockDim.x) {
      const int pat = site_pat[j];
      const int b = nv[j];
      const double u = asr_uniform(seed, sample_id, (uint32_t)j, 0u);
      int pick = R - 1;
      if (pat >= n_prune) {  // all-N column: every category has the same likelihood
        const double t = u * (double)R;
        for (int k = R - 1; k >= 0; --k)
          if (t < (double)(k + 1)) pick = k;
      } else {
        int smin = 0x7fffffff;
        for (int k = 0; k < R; ++k) smin = min(smin, site_scal[((size_t)sample * R + k) * n_prune + pat]);
        double total = 0.0;
        for (int k = 0; k < R; ++k) {
          double v = site_lik[(((size_t)sample * R + k) * 5 + b) * n_prune + pat];
          const int d = site_scal[((size_t)sample * R + k) * n_prune + pat] - smin;
          for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
          total += v;
        }
        const double t = u * total;
        double cum = 0.0;
        bool found = false;
        for (int k = 0; k < R; ++k) {
          double v = site_lik[(((size_t)sample * R + k) * 5 + b) * n_prune + pat];
          const int d = site_scal[((size_t)sample * R + k) * n_prune + pat] - smin;
          for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
          cum += v;
          if (!found && t < cum) {
            pick = k;
            found = true;
          }
        }
      }
      choice[j] = (uint8_t)pick;
      if (rate == 0 && rate_choice) rate_choice[(size_t)sample * L + j] = (uint8_t)pick;
    }
  }
  __syncthreads();

  // ---- inner-branch P-matrices (indexed by the op that produced the child), and the site list of this rate
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
  if (wave == n_waves - 1) {  // one wave compacts: deterministic slots (site order within a category)
    int run = 0, lower = 0;
    for (int j0 = 0; j0 < L; j0 += 64) {
      const int j = j0 + lane;
      const int c = j < L ? (int)choice[j] : 255;
      const unsigned long long mine = __ballot(c == rate);
      const unsigned long long low = __ballot(c < rate);
      if (c == rate) list[run + __popcll(mine & ((1ull << lane) - 1ull))] = j;
      run += __popcll(mine);
      lower += __popcll(low);
    }
    if (lane == 0) {
      misc[0] = run;
      misc[1] = lower;
    }
  }
  __syncthreads();

  const int cnt = misc[0], base = misc[1];
  const double* __restrict__ p4 = pi + (size_t)sample * 4;
  const uint8_t* __restrict__ nv = naive + (size_t)sample * L;
  const size_t plane = (size_t)L;  // doubles between the component planes of one (sample, op)
  double* clv_s = clv + (size_t)sample * n_ops * 4 * plane;
  uint8_t* anc_s = anc + (size_t)sample * n_ops * (size_t)L;
  uint8_t* my_stack = st_stack + tid;

  for (int s0 = wave * 64; s0 < cnt; s0 += n_waves * 64) {
    const int slot = s0 + lane;
    const bool active = slot < cnt;
    const int site = list[active ? slot : cnt - 1];
    const int gslot = base + (active ? slot : cnt - 1);
    const int pat = site_pat[site];
    const bool all_n = pat >= n_prune;
    const unsigned upat = all_n ? 0u : (unsigned)pat;

    // ---- upward pass: CLV of every op's node, stored for the way down
    double a[4] = {1.0, 1.0, 1.0, 1.0};
    for (int k = 0; k < n_ops; ++k) {
      const int4 op = ops_s[k];
      const int kind = __builtin_amdgcn_readfirstlane(op.x & 15);
      const int oy = __builtin_amdgcn_readfirstlane(op.y), oz = __builtin_amdgcn_readfirstlane(op.z);
      double u[4], v[4];
      if (kind == OP_CHERRY) {
        const int sa = all_n ? 4 : (int)msa[(unsigned)((oy - 1) * n_prune) + upat];
        const int sb = all_n ? 4 : (int)msa[(unsigned)((oz - 1) * n_prune) + upat];
        tip_col(tiptab, oy, sa, u);
        tip_col(tiptab, oz, sb, v);
      } else {
        matvec_lds(pin + (size_t)(k - 1) * 16, a, v);
        if (kind == OP_TIP_ACC) {
          const int sa = all_n ? 4 : (int)msa[(unsigned)((oy - 1) * n_prune) + upat];
          tip_col(tiptab, oy, sa, u);
        } else {
          const int q = __builtin_amdgcn_readfirstlane(popped_op[k]);
          double y[4];
          const double* cq = clv_s + (size_t)q * 4 * plane + gslot;
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = cq[(size_t)i * plane];
          matvec_lds(pin + (size_t)q * 16, y, u);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = u[i] * v[i];
      if (fmax(fmax(a[0], a[1]), fmax(a[2], a[3])) < kScaleThreshold) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] *= kScaleFactor;
      }
      if (active) {
        double* ck = clv_s + (size_t)k * 4 * plane + gslot;
#pragma unroll
        for (int i = 0; i < 4; ++i) ck[(size_t)i * plane] = a[i];
      }
    }

    // ---- downward pass
    int s_acc = 0;
    for (int k = n_ops - 1; k >= 0; --k) {
      const int4 op = ops_s[k];
      const int kind = __builtin_amdgcn_readfirstlane(op.x & 15);
      int s_cur;
      if (k == n_ops - 1) {
        // root = naive's neighbour: pi_i * L_root(i) * P_naive[i][naive base]
        const int b = nv[site];
        double down[4], w[4];
        tip_col(tiptab, 0, b, down);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = p4[i] * a[i] * down[i];
        s_cur = draw4(w, asr_uniform(seed, sample_id, (uint32_t)site, 1u));
        if (active) anc_s[(size_t)(node_of_op[k] - T) * L + site] = (uint8_t)s_cur;
      } else {
        const int4 nxt = ops_s[k + 1];
        const int nx = __builtin_amdgcn_readfirstlane(nxt.x), nw = __builtin_amdgcn_readfirstlane(nxt.w);
        s_cur = (nx & OP_PUSH_FLAG) ? (int)my_stack[nw * 256] : s_acc;
      }
      if (kind == OP_CHERRY) continue;
      {
        const int j = k - 1;  // the accumulator child
        const int node = __builtin_amdgcn_readfirstlane(node_of_op[j]);
        const double* cj = clv_s + (size_t)j * 4 * plane + gslot;
        const double* prow = pin + (size_t)j * 16 + s_cur * 4;
        double w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = prow[i] * cj[(size_t)i * plane];
        s_acc = draw4(w, asr_uniform(seed, sample_id, (uint32_t)site, 2u + (uint32_t)(node - T)));
        if (active) anc_s[(size_t)(node - T) * L + site] = (uint8_t)s_acc;
      }
      if (kind == OP_POP_ACC) {
        const int q = __builtin_amdgcn_readfirstlane(popped_op[k]);
        const int node = __builtin_amdgcn_readfirstlane(node_of_op[q]);
        const int ow = __builtin_amdgcn_readfirstlane(op.w);
        const double* cq = clv_s + (size_t)q * 4 * plane + gslot;
        const double* prow = pin + (size_t)q * 16 + s_cur * 4;
        double w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = prow[i] * cq[(size_t)i * plane];
        const int sq = draw4(w, asr_uniform(seed, sample_id, (uint32_t)site, 2u + (uint32_t)(node - T)));
        if (active) anc_s[(size_t)(node - T) * L + site] = (uint8_t)sq;
        my_stack[ow * 256] = (uint8_t)sq;
      }
    }
  }
}

int launch_asr(const DevFamily& fam, int n, int R, int T, const int32_t* ops, const double* brlen, const double* rates,
               const double* eig, const double* pi, const double* site_lik, const int32_t* site_scal,
               const uint8_t* naive, uint64_t seed, uint64_t sample0, double* clv, uint8_t* anc,
               uint8_t* rate_choice, hipStream_t stream) {
  const int L = fam.n_sites;
  const size_t lds = asr_lds_bytes(T, L);
  if (lds > 160 * 1024) return 1;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(asr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL(asr_kernel, dim3(R, n), dim3(256), lds, stream, R, T, L, fam.n_prune, fam.msa, fam.site_pat, ops,
                     brlen, rates, eig, pi, site_lik, site_scal, naive, seed, sample0, clv, anc, rate_choice);
  return 0;
}

}  // namespace lh
