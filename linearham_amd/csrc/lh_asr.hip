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
  // full 64-bit products: one v_mad_u64_u32 each instead of a mul_hi / mul_lo pair (32-bit integer multiplies
  // issue at a quarter of the VALU rate, and the draws are most of the downward pass's instructions)
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
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
  b += (size_t)L * sizeof(int32_t);              // list (sites of the category)
  b += NP * 2 * sizeof(int32_t);                 // plist, pslot
  b += 16 * 256;                                 // state stack [16][256]
  b += (((size_t)R * NP + 15) & ~(size_t)15);    // flags[R][NP]
  b += 16;                                       // counters
  return b;
}

// One schedule op as the sampling kernel reads it (two int4, fetched with scalar loads): everything that K3b
// would otherwise have to derive per op from the K1 schedule with dependent look-ups.
//   a.x  kind | stack slot of the op << 4 | (slot + 1 the NEXT op pushes the accumulator to, or 0) << 8
//   a.y  MSA row of tip child y (cherry, tip-acc), else 0        a.z  MSA row of tip child z (cherry), else 0
//   a.w  op that produced the sibling a pop op takes from the stack, else 0
//   b.x  tip id y (tip table row)   b.y  tip id z   b.z  inner node (minus T) this op produces
//   b.w  inner node (minus T) of the popped sibling (pop), else 0
struct AsrOp {
  int4 a, b;
};

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

// K3s: per sample, the schedule in the form K3b walks it (one thread does the stack bookkeeping out of LDS,
// then a thread per op writes its descriptor).
__global__ void __launch_bounds__(64) asr_sched_kernel(int T, const int32_t* __restrict__ ops,
                                                       const int4* __restrict__ hdr, AsrOp* __restrict__ desc) {
  extern __shared__ double2 sched_smem[];
  if (hdr[blockIdx.x].w != 0) return;  // K0c (lh_prune.hip) rejected this sample's schedule; K3b skips it too
  const int n_ops = T - 2;
  int4* ops_s = reinterpret_cast<int4*>(sched_smem);               // [n_ops]
  int32_t* popped_op = reinterpret_cast<int32_t*>(ops_s + n_ops);  // [n_ops]
  int32_t* node_of_op = popped_op + n_ops;                         // [n_ops]
  const int tid = threadIdx.x;
  const int sample = blockIdx.x;
  const int4* __restrict__ op_ptr = reinterpret_cast<const int4*>(ops) + (size_t)sample * n_ops;
  for (int k = tid; k < n_ops; k += blockDim.x) {
    ops_s[k] = op_ptr[k];
    popped_op[k] = 0;
  }
  __syncthreads();
  // which op produced the sibling a pop op takes from the stack, and which tree node every op produces; the 16
  // stack slots' op numbers sit in four 64-bit registers
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
  __syncthreads();
  AsrOp* out = desc + (size_t)sample * n_ops;
  for (int k = tid; k < n_ops; k += blockDim.x) {
    const int4 op = ops_s[k];
    const int kind = op.x & 15;
    int take = 0;
    if (k + 1 < n_ops && (ops_s[k + 1].x & OP_PUSH_FLAG)) take = ops_s[k + 1].w + 1;
    AsrOp d;
    d.a.x = kind | ((op.w & 15) << 4) | (take << 8);
    d.a.y = kind != OP_POP_ACC ? op.y - 1 : 0;
    d.a.z = kind == OP_CHERRY ? op.z - 1 : 0;
    d.a.w = kind == OP_POP_ACC ? popped_op[k] : 0;
    d.b.x = kind != OP_POP_ACC ? op.y : 0;
    d.b.y = kind == OP_CHERRY ? op.z : 0;
    d.b.z = node_of_op[k] - T;
    d.b.w = kind == OP_POP_ACC ? node_of_op[popped_op[k]] - T : 0;
    out[k] = d;
  }
}

__global__ void __launch_bounds__(256) asr_kernel(int R, int T, int L, int n_prune, const uint8_t* __restrict__ msa,
                                                  const int32_t* __restrict__ site_pat,
                                                  const AsrOp* __restrict__ desc, const double* __restrict__ brlen,
                                                  const double* __restrict__ rates, const double* __restrict__ eig,
                                                  const double* __restrict__ pi,
                                                  uint8_t* __restrict__ choice_g,
                                                  const uint8_t* __restrict__ naive, uint64_t seed, uint64_t sample0,
                                                  double2* clv, int Lp, uint8_t* __restrict__ anc,
                                                  const int4* __restrict__ hdr) {
  extern __shared__ double2 asr_smem[];
  const int n_ops = T - 2;
  if (hdr[blockIdx.y].w != 0) {
    // malformed schedule (reported through lh_family_status): states are bytes and have no NaN, so the sample's rows get
    // the sentinel 0xff -- no stale byte of the caller's buffers may look like a draw (include/linearham_amd.h)
    if (blockIdx.x == 0) {
      uint8_t* a = anc + (size_t)blockIdx.y * n_ops * (size_t)L;
      for (size_t i = threadIdx.x; i < (size_t)n_ops * L; i += blockDim.x) a[i] = 0xff;
      for (int i = threadIdx.x; i < L; i += blockDim.x) choice_g[(size_t)blockIdx.y * L + i] = 0xff;
    }
    return;
  }
  const int NP = n_prune + 1;  // patterns, the all-N one (id n_prune) included
  double* tiptab = reinterpret_cast<double*>(asr_smem);           // [T][4][4]
  double* pin = tiptab + (size_t)T * 16;                          // [n_ops][4][4] (entry n_ops-1 unused)
  int32_t* list = reinterpret_cast<int32_t*>(pin + (size_t)n_ops * 16);  // [L]  sites of this category
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
  const AsrOp* __restrict__ dsc = desc + (size_t)sample * n_ops;
  const uint8_t* __restrict__ ch = choice_g + (size_t)sample * L;

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

  // ---- P-matrices of this (sample, rate): tip branches -> tip table (columns of P); inner branches row-major,
  // indexed by the op that produced the child
  const double* __restrict__ e = eig + (size_t)sample * 36;
  const double rt = rates[(size_t)sample * R + rate];
  const double* __restrict__ bl = brlen + (size_t)sample * (2 * (size_t)T - 2);
  {
    double P[4][4];
    for (int j = tid; j < T + n_ops - 1; j += blockDim.x) {
      if (j < T) {
        compute_pmatrix(e, bl[j] * rt, P);
        double* o = tiptab + j * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int st = 0; st < 4; ++st) o[st * 4 + i] = P[i][st];
      } else {
        const int k = j - T;  // op k < n_ops - 1 produced inner node T + b.z, whose branch this is
        compute_pmatrix(e, bl[T + dsc[k].b.z] * rt, P);
        double* o = pin + (size_t)k * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) o[i * 4 + q] = P[i][q];
      }
    }
  }
  __syncthreads();
  const int cnt = misc[0], base = misc[1], cntp = misc[2];
  if (cnt == 0) return;  // no site of this sample drew this category (uniform)

  const double* __restrict__ p4 = pi + (size_t)sample * 4;
  const uint8_t* __restrict__ nv = naive + (size_t)sample * L;
  // CLV area of the sample: [op][2][Lp] double2 -- components (0,1) and (2,3) of a slot are 16-byte entries of
  // two planes, so a wave's store is one contiguous run of whole 128-byte lines per plane.
  const size_t plane = (size_t)Lp;
  double2* clv_s = clv + (size_t)sample * n_ops * 2 * plane;
  uint8_t* anc_s = anc + (size_t)sample * n_ops * (size_t)L;
  uint8_t* my_stack = st_stack + tid;

  // ---- upward pass over the category's distinct patterns, full waves first.
  // Software pipeline: everything op k + 1 needs from memory -- its tips' states and the CLV of the sibling it
  // pops (stored at least two ops earlier) -- is requested while op k computes, with loads that do not
  // depend on the op's kind (descriptor fields an op does not use point at valid dummies), so that no branch
  // stands between a load and the next iteration.
  {
    const int per = 64;
    for (int s0 = wave * per; s0 < cntp; s0 += n_waves * per) {
      // Lanes past the end of the wave's share repeat its last pattern: they compute and store the very same
      // values, so no store needs a predicate (a predicated store is a branch, and a branch between the
      // prefetch loads and their use makes the wait for them a wait for everything outstanding).
      const int slot = min(s0 + min(lane, per - 1), cntp - 1);
      const int pat = plist[slot];
      const int gslot = base + slot;
      const bool all_n = pat >= n_prune;
      const unsigned upat = all_n ? 0u : (unsigned)pat;
      const double2* cbase = clv_s + gslot;
      double a[4] = {1.0, 1.0, 1.0, 1.0};
      // Software pipeline.  A ring of four entries holds, per op, its descriptor and its tips' states, requested
      // four ops ahead (the alignment is constant); the CLV of a popped sibling can only be requested one op
      // ahead (it may have been stored two ops before it is popped) and alternates between two registers
      // sets.  The loop is unrolled by four, so ring positions are compile-time and no entry is ever copied: a
      // register copy would be a use, and the wait for the loads would land where they were issued.
      struct Up {
        int4 da, db;
        int sa, sb;
      };
      Up ring[4];
      double2 ylo[2], yhi[2];
      auto tips = [&](Up& p) {
        p.sa = msa[(unsigned)(p.da.y * n_prune) + upat];
        p.sb = msa[(unsigned)(p.da.z * n_prune) + upat];
      };
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = i < n_ops ? i : n_ops - 1;
        ring[i].da = dsc[kk].a;
        ring[i].db = dsc[kk].b;
        tips(ring[i]);
      }
      ylo[0] = yhi[0] = ylo[1] = yhi[1] = make_double2(0.0, 0.0);
      for (int k = 0; k < n_ops; k += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = k + i;
          if (kk < n_ops) {
            Up& c = ring[i];
            if ((ring[(i + 1) & 3].da.x & 15) == OP_POP_ACC) {  // the sibling op kk + 1 pops
              const double2* cq = cbase + (size_t)ring[(i + 1) & 3].da.w * 2 * plane;
              ylo[(i + 1) & 1] = cq[0];
              yhi[(i + 1) & 1] = cq[plane];
            }
            const int kf = kk + 4 < n_ops ? kk + 4 : n_ops - 1;
            const int4 fa = dsc[kf].a, fb = dsc[kf].b;
            const int kind = c.da.x & 15;
            double u[4], v[4];
            if (kind == OP_CHERRY) {
              tip_col(tiptab, c.db.x, all_n ? 4 : c.sa, u);
              tip_col(tiptab, c.db.y, all_n ? 4 : c.sb, v);
            } else {
              matvec_lds(pin + (size_t)(kk - 1) * 16, a, v);
              if (kind == OP_TIP_ACC) {
                tip_col(tiptab, c.db.x, all_n ? 4 : c.sa, u);
              } else {
                const double y[4] = {ylo[i & 1].x, ylo[i & 1].y, yhi[i & 1].x, yhi[i & 1].y};
                matvec_lds(pin + (size_t)c.da.w * 16, y, u);
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = u[q] * v[q];
            if (fmax(fmax(a[0], a[1]), fmax(a[2], a[3])) < kScaleThreshold) {
#pragma unroll
              for (int q = 0; q < 4; ++q) a[q] *= kScaleFactor;
            }
            double2* ck = clv_s + (size_t)kk * 2 * plane + gslot;
            ck[0] = make_double2(a[0], a[1]);
            ck[plane] = make_double2(a[2], a[3]);
            c.da = fa;
            c.db = fb;
            tips(c);
          }
        }
      }
    }
  }
  // the sites' lanes below read CLVs that other lanes and waves of this workgroup stored above
  __threadfence_block();
  __syncthreads();

  // ---- downward pass over the category's sites: the schedule in reverse.  The CLVs of op k - 1's children
  // are requested while op k draws (their addresses depend on the schedule only, not on the states).
  {
    const int per = 64;
    for (int s0 = wave * per; s0 < cnt; s0 += n_waves * per) {
      const int slot = min(s0 + min(lane, per - 1), cnt - 1);  // surplus lanes repeat the last site (same draws)
      const int site = list[slot];
      const int pat = min(site_pat[site], n_prune);
      const double2* cbase = clv_s + base + pslot[pat];
      const int b_naive = nv[site];
      // Children of op k: the accumulator child is op k - 1, the popped one op da.w (dummies: op 0).  All CLVs
      // are final here, so everything an op needs is requested four ops ahead: a ring of four entries at
      // compile-time positions (loop unrolled by four), as on the way up.
      struct Dn {
        int4 da, db;
        int node_acc;
        double2 alo, ahi, plo, phi;
      };
      auto fetch = [&](int k, Dn& p) {
        p.da = dsc[k].a;
        p.db = dsc[k].b;
        const int j = k > 0 ? k - 1 : 0;
        p.node_acc = dsc[j].b.z;  // the accumulator child is the node op k - 1 produced
        // (loads that do not depend on the op's kind: with branches around them the waits got coarser and the
        // pass 30 % slower, although a third of these loads fetch a dummy)
        const double2* ca = cbase + (size_t)j * 2 * plane;
        p.alo = ca[0];
        p.ahi = ca[plane];
        const double2* cp = cbase + (size_t)p.da.w * 2 * plane;
        p.plo = cp[0];
        p.phi = cp[plane];
      };
      Dn ring[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fetch(n_ops - 1 - i > 0 ? n_ops - 1 - i : 0, ring[i]);
      int s_acc;
      {
        // root = naive's neighbour: pi_i * L_root(i) * P_naive[i][naive base]; the root op pushes nothing, so
        // its state travels to step n_ops - 1 as "the accumulator's state"
        const double2* cr = cbase + (size_t)(n_ops - 1) * 2 * plane;
        const double2 rlo = cr[0], rhi = cr[plane];
        double down[4];
        tip_col(tiptab, 0, b_naive, down);
        const double w[4] = {p4[0] * rlo.x * down[0], p4[1] * rlo.y * down[1], p4[2] * rhi.x * down[2],
                             p4[3] * rhi.y * down[3]};
        s_acc = draw4(w, asr_uniform(seed, sample_id, (uint32_t)site, 1u));
        anc_s[(size_t)ring[0].db.z * L + site] = (uint8_t)s_acc;
      }
      for (int k = n_ops - 1; k >= 0; k -= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = k - i;
          if (kk >= 0) {
            Dn& c = ring[i];
            const int kind = c.da.x & 15;
            const int take = (c.da.x >> 8) & 31;
            const int s_cur = take ? (int)my_stack[(take - 1) * 256] : s_acc;
            if (kind != OP_CHERRY) {
              const double* prow = pin + (size_t)(kk - 1) * 16 + s_cur * 4;
              const double w[4] = {prow[0] * c.alo.x, prow[1] * c.alo.y, prow[2] * c.ahi.x, prow[3] * c.ahi.y};
              s_acc = draw4(w, asr_uniform(seed, sample_id, (uint32_t)site, 2u + (uint32_t)c.node_acc));
              anc_s[(size_t)c.node_acc * L + site] = (uint8_t)s_acc;
            }
            if (kind == OP_POP_ACC) {
              const int node = c.db.w;
              const double* prow = pin + (size_t)c.da.w * 16 + s_cur * 4;
              const double w[4] = {prow[0] * c.plo.x, prow[1] * c.plo.y, prow[2] * c.phi.x, prow[3] * c.phi.y};
              const int sq = draw4(w, asr_uniform(seed, sample_id, (uint32_t)site, 2u + (uint32_t)node));
              anc_s[(size_t)node * L + site] = (uint8_t)sq;
              my_stack[((c.da.x >> 4) & 15) * 256] = (uint8_t)sq;
            }
            fetch(kk - 4 > 0 ? kk - 4 : 0, c);
          }
        }
      }
    }
  }
}

size_t asr_desc_bytes(int T) { return sizeof(AsrOp) * (size_t)(T - 2); }

int launch_asr(const DevFamily& fam, int n, int R, int T, const int32_t* ops, const double* brlen, const double* rates,
               const double* eig, const double* pi, const double* site_lik, const int32_t* site_scal,
               const uint8_t* naive, uint64_t seed, uint64_t sample0, double* clv, void* desc, uint8_t* anc,
               uint8_t* rate_choice, const int4* hdr, hipStream_t stream) {
  const int L = fam.n_sites;
  const size_t lds = asr_lds_bytes(T, L, R, fam.n_prune);
  if (lds > 160 * 1024 || L < 1) return 1;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(asr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  const long long cells = (long long)n * L;
  hipLaunchKernelGGL(asr_rate_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, n, R, L,
                     fam.n_prune, fam.site_pat, site_lik, site_scal, naive, seed, sample0, rate_choice);
  const size_t sched_lds = (size_t)(T - 2) * (sizeof(int4) + 2 * sizeof(int32_t));
  hipLaunchKernelGGL(asr_sched_kernel, dim3(n), dim3(64), sched_lds, stream, T, ops, hdr, static_cast<AsrOp*>(desc));
  hipLaunchKernelGGL(asr_kernel, dim3(R, n), dim3(256), lds, stream, R, T, L, fam.n_prune, fam.msa, fam.site_pat,
                     static_cast<const AsrOp*>(desc), brlen, rates, eig, pi, rate_choice, naive, seed,
                     sample0, reinterpret_cast<double2*>(clv), (int)asr_slots(L, R), anc, hdr);
  return 0;
}

}  // namespace lh
