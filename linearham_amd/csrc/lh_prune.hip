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
//  * One workgroup = (site tile, rate category, tree sample).  The traversal is a wave-uniform
//    schedule (lh_schedule_tree): every lane executes the same op, so the 4x4 P-matrices of the op
//    are wave-uniform and are fetched with scalar loads into SGPRs (no LDS / VGPR cost), and child
//    CLVs never leave the chip: the running CLV lives in VGPRs and pending siblings live in a
//    register-resident stack of statically indexed slots.
//  * A lane carries TWO sites (64 apart) through the schedule.  With one site per lane the scalar data
//    cache saturates first: measured (tools/microbench/sqc_bw.hip) it delivers ~3.2 bytes/clk per CU, a one-site
//    workgroup pulls 7 waves x 14 KB of P-matrices through it, which alone is 1.7 ms per 8192 samples.
//    Two sites per lane halve the waves, hence the scalar bytes, per site, and give each wave two
//    independent dependency chains.  A tile's last few sites (fewer than 64) ride in a one-site wave, so
//    no wave executes for empty lanes.
//  * Tip children need no mat-vec: P * onehot(state) is a column of P.  Those columns (plus the
//    row sums for N, formed on demand) live in LDS as tiptab[tip][state][4] and are gathered
//    with two ds_read_b128 per lane and site.
//  * The workgroup first computes its (sample, rate)'s P-matrices itself (no separate kernel, no HBM
//    round trip): inner-branch matrices in schedule order into a global scratch area that its own
//    scalar loads read back from L2, tip-branch matrices straight into the LDS tip table.
//  * HBM traffic is therefore ~T bytes of tip states per site (L2-resident, shared by all samples)
//    and 5 doubles out, instead of the 2*I*R*32 bytes per column of a CLV-streaming kernel.
#include <cstdlib>

#include <cstdio>

#include "lh_device.h"

namespace lh {

namespace {

// Statically indexed register stack: slot d is its own array st<d>[S][4] (separate objects: one
// st[kDepth][S][4] array gets its slot switch folded into a variable index and lands in scratch
// memory).  The slot number of an op is wave-uniform.  Push: independent scalar branches, one per slot,
// each leaving every other slot's registers alone (a single switch made the compiler shuffle whole
// slots through temporaries at its merge points).  Pop: the sibling is multiplied straight out of its
// slot, one copy of the mat-vec per shallow slot, instead of being copied to common registers first.
#define LH_SLOT_COPY(dst, src)                       \
  _Pragma("unroll") for (int s_ = 0; s_ < S; ++s_) { \
    dst[s_][0] = src[s_][0];                         \
    dst[s_][1] = src[s_][1];                         \
    dst[s_][2] = src[s_][2];                         \
    dst[s_][3] = src[s_][3];                         \
  }
// The push tests one bit of a one-hot slot mask that is opaque to the optimiser (it comes through
// v_readfirstlane): equality tests on the slot number get merged back into one compare tree whose joins
// route whole slots through temporaries and scratch memory (57 v_mov_b64 in the loop body).
#define LH_PUSH_IF(d)                                   \
  if constexpr (kDepth > d) {                           \
    if (push_mask & (1u << d)) { LH_SLOT_COPY(st##d, a) } \
  }
#define LH_POP_CASE(d)          \
  case d:                       \
    if constexpr (kDepth > d) { \
      LH_SLOT_COPY(y, st##d)    \
    }                           \
    break;
#define LH_POP_SLOT(d) \
  _Pragma("unroll") for (int s_ = 0; s_ < S; ++s_) matvec(pa, st##d[s_], z[s_]);
#define LH_DECL(d) double st##d[S][4];

// The walk's P-matrices were written to the scratch area by THIS workgroup's prologue (vector stores, drained
// and followed by the workgroup barrier) and are wanted as SCALAR operands.  The compiler selects scalar loads
// only for memory it may assume the kernel does not change, so the walk reads the scratch area through a
// pointer to the constant address space -- and that pointer does not exist before the barrier: it is the
// OUTPUT of a volatile asm statement placed after the barrier.  Every P-matrix load is data-dependent on that
// statement, so none can be scheduled above the barrier, whatever the optimiser assumes about aliasing
// (nothing is declared __restrict__ or read-only that is written here); from that point on the lines are
// indeed constant for the rest of the kernel.  The compiler issues the loads and counts their completion
// itself, which is what lets it start them well ahead of their use.
// (The scalar cache cannot hold a stale copy: it is invalidated at kernel boundaries, and inside a launch a
// scratch line is never read before the workgroup that reads it has written it.)
typedef const double __attribute__((address_space(4))) * pmat_ptr;

// -DLH_EXP_K1_STAMPS: latency of the first P-matrix load of every non-cherry op and the length of the walk, per wave,
// for 128 workgroups in the middle of the grid (an instrument; it perturbs the schedule it measures)
#ifdef LH_DEBUG_WALK
__device__ int lh_dbg_max_ops = 1 << 30;
#endif
#ifdef LH_EXP_CT_STAMPS
__device__ unsigned long long ct_stamps[1024][10];
__device__ unsigned long long ct_phase[128][8];
#define LH_CT_PHASE(i) if (threadIdx.x == 0 && blockIdx.z >= 20000 && blockIdx.z < 20128) ct_phase[blockIdx.z - 20000][i] = __builtin_readcyclecounter();
#else
#define LH_CT_PHASE(i)
#endif
#ifdef LH_EXP_K1_STAMPS
__device__ unsigned long long k1_stamps[1024][4];
__device__ unsigned long long k1_phase[128][8];
#define LH_K1_PHASE(i) if (threadIdx.x == 0 && blockIdx.z >= 20000 && blockIdx.z < 20128) k1_phase[blockIdx.z - 20000][i] = __builtin_readcyclecounter();
#else
#define LH_K1_PHASE(i)
#endif

__device__ __forceinline__ pmat_ptr pmat_after_barrier(const double* p) {
  // the address is the same in every lane; say so in a form the register allocator has to honour
  // (the builtin returns a signed int: widen through uint32_t, or a low word with bit 31 set smears into the high one)
  const uint64_t bits = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32));
  uint64_t v = (uint64_t)lo | ((uint64_t)hi << 32);
  asm volatile("; lh: P-matrix scratch base %0 -- scalar loads of the walk depend on this statement" : "+s"(v) : : "memory");
  return (pmat_ptr)v;
}

// x = P * a for a wave-uniform row-major 4x4 P (scalar operands)
__device__ __forceinline__ void matvec(pmat_ptr p, const double (&a)[4], double (&x)[4]) {
  x[0] = fma(p[3], a[3], fma(p[2], a[2], fma(p[1], a[1], p[0] * a[0])));
  x[1] = fma(p[7], a[3], fma(p[6], a[2], fma(p[5], a[1], p[4] * a[0])));
  x[2] = fma(p[11], a[3], fma(p[10], a[2], fma(p[9], a[1], p[8] * a[0])));
  x[3] = fma(p[15], a[3], fma(p[14], a[2], fma(p[13], a[1], p[12] * a[0])));
}

// Column `st` of a tip branch's P (= P * onehot(st)) from the LDS table tiptab[tip][4][4]; a tip whose
// state is N (4) contributes the row sums of P, formed on the spot from the four columns -- only in the
// kN instantiation, which families whose alignment mixes N with bases run (the table has no fifth row:
// 128 instead of 160 bytes per tip is what lets a CU hold ten workgroups of configs[2] instead of nine).
template <bool kN>
__device__ __forceinline__ void tip_column(const double* tiptab, int tip, int st, double (&c)[4]) {
  const double* t = tiptab + tip * 16;
  const double2* q = reinterpret_cast<const double2*>(t + (kN ? (st & 3) : st) * 4);
  const double2 q0 = q[0], q1 = q[1];
  c[0] = q0.x, c[1] = q0.y, c[2] = q1.x, c[3] = q1.y;
  if constexpr (kN) {
    if (st == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = ((t[i] + t[4 + i]) + t[8 + i]) + t[12 + i];
    }
  }
}

// Large trees: a table of all T tip matrices (128 B per tip) would take the LDS of a CU for one or two
// workgroups.  The walk visits the tips in schedule order, so the table is built a SEGMENT of the schedule at a
// time: kSegOps consecutive ops, two slots per op (slot 2 (k - k0) + c for child c of op k, used when that
// child is a tip).  At a segment boundary -- the same op for every wave of the workgroup, the schedule is
// wave-uniform -- the workgroup meets at a barrier, thread j computes the matrix of slot j of the next
// segment, and a second barrier releases the walk.  12 KB of LDS whatever the tree size.
constexpr int kSegOps = 48;

struct SegCtx {
  const double* e;       // the sample's eigen-decomposition (36 doubles)
  const double* bl;      // its branch lengths
  double rt;             // the rate of this workgroup
  double* tab;           // LDS: [2 * kSegOps][4][4] slots
  int rtid, nthr;        // this thread's number among the threads that fill the table, and their count
};

// Fill the slots of segment [k0, k0 + kSegOps): thread j takes slot j (P = I + U expm1(lambda t r) U^-1 of the
// tip's branch, stored column by column as the walk gathers it).
__device__ __forceinline__ void seg_fill(const SegCtx& c, const int4* __restrict__ op_ptr, int n_ops, int k0) {
  for (int j = c.rtid; j < 2 * kSegOps; j += c.nthr) {
    const int k = k0 + (j >> 1);
    if (k >= n_ops) continue;
    const int4 op = op_ptr[k];
    const int kind = op.x & 15;
    int tip = -1;
    if (kind == OP_CHERRY) tip = (j & 1) ? op.z : op.y;
    if (kind == OP_TIP_ACC && !(j & 1)) tip = op.y;
    if (tip < 0) continue;
    double P[4][4];
    compute_pmatrix(c.e, c.bl[tip] * c.rt, P);
    double* o = c.tab + j * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int st = 0; st < 4; ++st) o[st * 4 + i] = P[i][st];
  }
}

// The schedule walk of one wave: S sites per lane (site0 + 64*s), all lanes active (sites past the end
// of the tile are clamped to a valid one and not written back).
// tiptab: the tip table ([T][4][4], or the segment slots when kSeg); naive_tab: the naive tip's entry.
template <int kDepth, int S, bool kN, bool kSeg = false>
__device__ __forceinline__ void prune_wave(int site0, int site_end, const uint8_t* __restrict__ msa, int L,
                                           int n_ops, const int4* __restrict__ op_ptr,
                                           pmat_ptr pm, const double* tiptab, const double* naive_tab,
                                           const double* __restrict__ p4, double (&lik)[S][5], int (&scl)[S],
                                           const SegCtx& seg) {
  unsigned usite[S];  // MSA byte offsets are 32-bit: (tip row) * L + site
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int site = site0 + 64 * s;
    usite[s] = (unsigned)(site < site_end ? site : site_end - 1);
  }
  double a[S][4];
  int scal[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    a[s][0] = a[s][1] = a[s][2] = a[s][3] = 1.0;
    scal[s] = 0;
  }
  LH_DECL(0) LH_DECL(1) LH_DECL(2) LH_DECL(3) LH_DECL(4) LH_DECL(5) LH_DECL(6) LH_DECL(7)
  LH_DECL(8) LH_DECL(9) LH_DECL(10) LH_DECL(11) LH_DECL(12) LH_DECL(13) LH_DECL(14) LH_DECL(15)

  // Software pipeline across iterations: op k's descriptor and tip states were requested during
  // iteration k-1, so an iteration starts with everything but its P-matrices at hand (and those come
  // from addresses that depend on k only).
  int4 op = op_ptr[0];
  int sa[S], sb[S];
#pragma unroll
  for (int s = 0; s < S; ++s) sa[s] = sb[s] = 0;
  {
    const int kd = op.x & 15;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      if (kd != OP_POP_ACC) sa[s] = msa[(unsigned)((op.y - 1) * L) + usite[s]];
      if (kd == OP_CHERRY) sb[s] = msa[(unsigned)((op.z - 1) * L) + usite[s]];
    }
  }
#ifdef LH_EXP_K1_STAMPS
  unsigned long long st_wait = 0, st_n = 0;
  const unsigned long long st_begin = __builtin_readcyclecounter();
#endif
  int seg_k0 = 0;  // first op of the segment whose tip matrices are in the table (kSeg)
  for (int k = 0; k < n_ops; ++k) {
    const int4 op_next = op_ptr[k + 1 < n_ops ? k + 1 : k];
    const int kind = op.x & 15;
    if constexpr (kSeg) {
      if (k - seg_k0 == kSegOps) {  // every wave of the workgroup arrives here at the same op
        __syncthreads();            // nobody reads the old segment any more
        seg_k0 = k;
        seg_fill(seg, op_ptr, n_ops, k);
        __syncthreads();
      }
    }
    // table rows of the op's tip children
    const int row_y = kSeg ? 2 * (k - seg_k0) : op.y, row_z = kSeg ? 2 * (k - seg_k0) + 1 : op.z;
    if (op.x & OP_PUSH_FLAG) {
      const unsigned push_mask = __builtin_amdgcn_readfirstlane(1u << op.w);
      LH_PUSH_IF(0) LH_PUSH_IF(1) LH_PUSH_IF(2) LH_PUSH_IF(3)
      if constexpr (kDepth > 4) {
        if (op.w >= 4) {
          LH_PUSH_IF(4) LH_PUSH_IF(5) LH_PUSH_IF(6) LH_PUSH_IF(7)
          LH_PUSH_IF(8) LH_PUSH_IF(9) LH_PUSH_IF(10) LH_PUSH_IF(11)
          LH_PUSH_IF(12) LH_PUSH_IF(13) LH_PUSH_IF(14) LH_PUSH_IF(15)
        }
      }
    }
    // Every op ends in the same element-wise product a = u * v: (tip column, tip column) for a cherry,
    // (tip column, P_b a) for a tip joining the accumulator, (P_a sibling, P_b a) for a pop.  Keeping that
    // product common to the three branches lets them leave their factors wherever the loads / FMAs
    // produced them (with a per-branch product the compiler moved the whole CLV at the join).
    double u[S][4], v[S][4];
    if (kind == OP_CHERRY) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        tip_column<kN>(tiptab, row_y, sa[s], u[s]);
        tip_column<kN>(tiptab, row_z, sb[s], v[s]);
      }
    } else {
      const pmat_ptr pb = pm + (size_t)k * 32;
#ifdef LH_EXP_K1_STAMPS
      {
        const unsigned long long t0 = __builtin_readcyclecounter();
        double p0 = pb[0];
        asm volatile("" : "+s"(p0));
        const unsigned long long t1 = __builtin_readcyclecounter();
        st_wait += t1 - t0;
        ++st_n;
      }
#endif
#pragma unroll
      for (int s = 0; s < S; ++s) matvec(pb, a[s], v[s]);
      if (kind == OP_TIP_ACC) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          tip_column<kN>(tiptab, row_y, sa[s], u[s]);
        }
      } else {  // OP_POP_ACC
        const pmat_ptr pa = pm + (size_t)k * 32 + 16;
        double(&z)[S][4] = u;
        if (op.w == 0) {
          LH_POP_SLOT(0)
        } else if (op.w == 1) {
          LH_POP_SLOT(1)
        } else if (kDepth > 2 && op.w == 2) {
          LH_POP_SLOT(2)
        } else if (kDepth > 3 && op.w == 3) {
          LH_POP_SLOT(3)
        } else {
          double y[S][4];
#pragma unroll
          for (int s = 0; s < S; ++s) y[s][0] = y[s][1] = y[s][2] = y[s][3] = 0.0;
          if constexpr (kDepth > 4) {
            switch (op.w) {
              LH_POP_CASE(4) LH_POP_CASE(5) LH_POP_CASE(6) LH_POP_CASE(7) LH_POP_CASE(8) LH_POP_CASE(9)
              LH_POP_CASE(10) LH_POP_CASE(11) LH_POP_CASE(12) LH_POP_CASE(13) LH_POP_CASE(14) LH_POP_CASE(15)
            }
          }
#pragma unroll
          for (int s = 0; s < S; ++s) matvec(pa, y[s], z[s]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      a[s][0] = u[s][0] * v[s][0];
      a[s][1] = u[s][1] * v[s][1];
      a[s][2] = u[s][2] * v[s][2];
      a[s][3] = u[s][3] * v[s][3];
    }
    op = op_next;
    {
      const int kd = op.x & 15;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (kd != OP_POP_ACC) sa[s] = msa[(unsigned)((op.y - 1) * L) + usite[s]];
        if (kd == OP_CHERRY) sb[s] = msa[(unsigned)((op.z - 1) * L) + usite[s]];
      }
    }
    // Per-site, per-rate 2^256 rescaling (libpll PLL_ATTRIB_RATE_SCALERS semantics): the single running
    // counter is valid for the whole tree because scalers are additive along the traversal.  CLV
    // entries are non-negative, so the largest has the largest high word; it is below 2^-256 exactly
    // when that word is below 0x2FF00000 (integer compares instead of 7 FP64 max/compare).
    // (An all-zero CLV -- impossible data -- counts as small too: rescaling it changes nothing but its
    // counter, and K2a aligns the rates' counters before it mixes them.)
    unsigned hw[S];
#pragma unroll
    for (int s = 0; s < S; ++s)
      hw[s] = max(max((unsigned)__double2hiint(a[s][0]), (unsigned)__double2hiint(a[s][1])),
                  max((unsigned)__double2hiint(a[s][2]), (unsigned)__double2hiint(a[s][3])));
    unsigned hmin = hw[0];
#pragma unroll
    for (int s = 1; s < S; ++s) hmin = min(hmin, hw[s]);
    if (__builtin_expect(__ballot(hmin < 0x2FF00000u) != 0, 0)) {  // rare: skipped wave-wide
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (hw[s] < 0x2FF00000u) {
          a[s][0] *= kScaleFactor;
          a[s][1] *= kScaleFactor;
          a[s][2] *= kScaleFactor;
          a[s][3] *= kScaleFactor;
          ++scal[s];
        }
      }
    }
  }

#ifdef LH_EXP_K1_STAMPS
  if (S == 2 && blockIdx.z >= 20000 && blockIdx.z < 20128 && (threadIdx.x & 63) == 0) {
    const int w = (blockIdx.z - 20000) * 8 + (threadIdx.x >> 6);
    k1_stamps[w][0] = st_wait;
    k1_stamps[w][1] = st_n;
    k1_stamps[w][2] = __builtin_readcyclecounter() - st_begin;
  }
#endif
  // epilogue: close the naive branch for each possible naive state b (A,C,G,T,N):
  //   L_b = sum_i pi_i * clv_root[i] * P_naive[i][b]        (N: row sums of P_naive)
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const double w0 = p4[0] * a[s][0], w1 = p4[1] * a[s][1], w2 = p4[2] * a[s][2], w3 = p4[3] * a[s][3];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      double tv[4];  // tip 0 = naive; its possible states are the five naive bases of the xMSA
      if (b < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = naive_tab[b * 4 + i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = ((naive_tab[i] + naive_tab[4 + i]) + naive_tab[8 + i]) + naive_tab[12 + i];
      }
      lik[s][b] = fma(w3, tv[3], fma(w2, tv[2], fma(w1, tv[1], w0 * tv[0])));
    }
    scl[s] = scal[s];
  }
}


// ---- cherry tables (round 3) ------------------------------------------------------------------------------------
// A cherry (y, z) under branch c reaches the rest of the tree only through P_c (P_y[:, s_y] o P_z[:, s_z]): 16
// vectors per (sample, rate), 25 when tips can be N.  The prologue tabulates them (ctab[table][s_y * SY + s_z][4],
// global memory: 13 KB per (sample, rate) do not fit beside the tip tables in LDS at three workgroups per CU), and the
// walk executes the schedule K0c rewrote (lh_device.h, W_*): a quarter fewer ops, a fifth fewer vector instructions.

// One table entry: two 16-byte loads per lane (the prologue of THIS workgroup wrote the table; ordered by the barrier)
template <bool kN>
__device__ __forceinline__ void table_entry(const double* ctab, int table, int sy, int sz, double (&c)[4]) {
  constexpr int SY = kN ? 5 : 4;
  // 32-bit byte offset from a wave-uniform base: scalar base + vector offset addressing, no 64-bit vector arithmetic
  const unsigned off = ((unsigned)table * (SY * SY) + (unsigned)(sy * SY + sz)) * 32u;
  const double2* q = reinterpret_cast<const double2*>(reinterpret_cast<const char*>(ctab) + off);
  const double2 q0 = q[0], q1 = q[1];
  c[0] = q0.x, c[1] = q0.y, c[2] = q1.x, c[3] = q1.y;
}

// Walk descriptors: 8 bytes per op, written by K0c, copied into LDS by the workgroup's prologue and read from there
// two ops ahead (a descriptor fetched from global memory sat in the same scalar-memory wait as the op's P-matrix and
// cost a trip to HBM per op: r03 stamps):
//   x  kind [2:0] | the op uses a P-matrix [3] | push slot + 1 [8:4] (0: no push) | pop slot [12:9] |
//      tip B / tip C states are read [13] / [14] | tip A [31:16] (1 when the op has none: its state is loaded anyway)
//   y  tip B [15:0] | tip C [31:16]
// P-matrices and cherry tables are consumed strictly in sequence (K0c lists them in walk order; a pushed subtree's
// branch matrix is applied AT THE PUSH, st = P_first a, so that every op uses at most one matrix), so the walk keeps
// two running offsets instead of per-op addresses, and "the next matrix" / "the next table" are known without
// looking ahead.
typedef int2 WalkOp;
enum : int { WOP_MATRIX = 8, WOP_PUSH_SHIFT = 4, WOP_POP_SHIFT = 9, WOP_HAS_B = 1 << 13, WOP_HAS_C = 1 << 14 };

// timing experiments (results wrong by construction): every table look-up goes to table 0 / every walk matrix comes
// from one of two hot lines
#ifdef LH_EXP_CT_TABHIT
#define LH_CT_TOFF(x) ((x) & 0)
#else
#define LH_CT_TOFF(x) (x)
#endif
#ifdef LH_EXP_CT_PHIT
#define LH_CT_POFF(x) ((x) & 128)
#else
#define LH_CT_POFF(x) (x)
#endif

template <bool kN>
__device__ __forceinline__ void tip_column_at(const char* tiptab_bytes, int entry_off, int st, double (&c)[4]) {
  const double* t = reinterpret_cast<const double*>(tiptab_bytes + entry_off);
  const double2* q = reinterpret_cast<const double2*>(t + (kN ? (st & 3) : st) * 4);
  const double2 q0 = q[0], q1 = q[1];
  c[0] = q0.x, c[1] = q0.y, c[2] = q1.x, c[3] = q1.y;
  if constexpr (kN) {
    if (st == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = ((t[i] + t[4 + i]) + t[8 + i]) + t[12 + i];
    }
  }
}

template <bool kN>
__device__ __forceinline__ void table_entry_at(const char* ctab_bytes, unsigned table_off, int sy, int sz, double (&c)[4]) {
  constexpr int SY = kN ? 5 : 4;
  // 32-bit byte offset from a wave-uniform base: scalar base + vector offset addressing, no 64-bit vector arithmetic
  const unsigned off = table_off + (unsigned)(sy * SY + sz) * 32u;
  const double2* q = reinterpret_cast<const double2*>(ctab_bytes + off);
  const double2 q0 = q[0], q1 = q[1];
  c[0] = q0.x, c[1] = q0.y, c[2] = q1.x, c[3] = q1.y;
}

// The naive branch closed for each possible naive state b (A,C,G,T,N): L_b = sum_i pi_i clv_root[i] P_naive[i][b]
// (N: row sums of P_naive); naive_tab = the naive tip's entry of the tip table.
template <int S>
__device__ __forceinline__ void close_naive_branch(const double (&a)[S][4], const int (&scal)[S], const double* naive_tab,
                                                   const double* __restrict__ p4, double (&lik)[S][5], int (&scl)[S]) {
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const double w0 = p4[0] * a[s][0], w1 = p4[1] * a[s][1], w2 = p4[2] * a[s][2], w3 = p4[3] * a[s][3];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      double tv[4];
      if (b < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = naive_tab[b * 4 + i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = ((naive_tab[i] + naive_tab[4 + i]) + naive_tab[8 + i]) + naive_tab[12 + i];
      }
      lik[s][b] = fma(w3, tv[3], fma(w2, tv[2], fma(w1, tv[1], w0 * tv[0])));
    }
    scl[s] = scal[s];
  }
}

typedef __attribute__((address_space(5))) char* private_ptr;
typedef __attribute__((address_space(3))) const char* lds_ptr;

// The walk in assembly (two or four sites per lane, alignments without N; text and register map: tools/gen_walk_asm.py
// -> lh_prune_walk_asm_s<S>.inc).  Same operations in the same order as the C++ walk below, which stays as the form
// for everything else and as its reference (LH_K1_CXX_WALK=1 selects it; the results are bit-identical).
// wops: the sample's descriptors in global memory; site_base: the wave's first site (lane l carries site_base + l +
// 64 s); ctoff: byte offset of the cherry tables in the scratch region pm points to.
#define LH_WALK_ASM_OPERANDS                                                                                             \
  [nw] "s"(n_w), [wops] "s"(wops), [pm] "s"(pm), [ctoff] "s"(ctoff), [msa] "s"(msa_m), [L] "s"(L), [tip] "s"(tip_lds),  \
      [site0] "s"(site_base), [last] "s"(last_site), [out] "v"((private_ptr)out_mem), [deep] "v"((private_ptr)deep_mem)
// kTipsG: the tip table sits in the scratch region (tip_off bytes behind pm) instead of LDS (large trees).
template <int kDepth, int S, bool kTipsG = false>
__device__ __forceinline__ void prune_wave_asm(int site_base, int site_end, const uint8_t* __restrict__ msa, int L, int n_w,
                                               const WalkOp* __restrict__ wops, pmat_ptr pm, unsigned ctoff, unsigned tip_off,
                                               const double* tiptab, const double* naive_tab,
                                               const double* __restrict__ p4, double (&lik)[S][5], int (&scl)[S]) {
  static_assert(S == 2 || S == 4, "the assembly walk exists for two and four sites per lane");
  __attribute__((aligned(16))) double out_mem[4 * S + 2];                                   // a[S][4], then the packed scaler counts
  __attribute__((aligned(16))) double deep_mem[(kDepth > 1 ? kDepth - 1 : 1) * 4 * S];     // stack slots 1.. : [slot][site][4]
  const uint8_t* msa_m = msa - L;
  const unsigned tip_lds = kTipsG ? tip_off : (unsigned)(size_t)(lds_ptr)tiptab;
  const int last_site = site_end - 1;
  static_assert(!kTipsG || S == 2, "tip columns from the scratch region: two sites per lane only");
  if constexpr (S == 2 && kTipsG) {
    asm volatile(
#include "lh_prune_walk_asm_s2g.inc"
        :
        : LH_WALK_ASM_OPERANDS
        : "memory", "vcc", "scc",
#include "lh_prune_walk_clobbers_s2.inc"
    );
  } else if constexpr (S == 2) {
    asm volatile(
#include "lh_prune_walk_asm_s2.inc"
        :
        : LH_WALK_ASM_OPERANDS
        : "memory", "vcc", "scc",
#include "lh_prune_walk_clobbers_s2.inc"
    );
  } else {
    asm volatile(
#include "lh_prune_walk_asm_s4.inc"
        :
        : LH_WALK_ASM_OPERANDS
        : "memory", "vcc", "scc",
#include "lh_prune_walk_clobbers_s4.inc"
    );
  }
  double a[S][4];
  int scal[S];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[s][i] = out_mem[s * 4 + i];
  const unsigned* packed = reinterpret_cast<const unsigned*>(out_mem + 4 * S);
#pragma unroll
  for (int s = 0; s < S; ++s) scal[s] = (int)((packed[s / 2] >> (16 * (s & 1))) & 0xffffu);
  close_naive_branch<S>(a, scal, naive_tab, p4, lik, scl);
}
#undef LH_WALK_ASM_OPERANDS

template <int kDepth, int S, bool kN>
__device__ __forceinline__ void prune_wave_ct(int site0, int site_end, const uint8_t* __restrict__ msa, int L, int n_w,
                                              const WalkOp* desc /* LDS */, pmat_ptr pm, const double* tiptab,
                                              const double* ctab, const double* naive_tab,
                                              const double* __restrict__ p4, double (&lik)[S][5], int (&scl)[S]) {
  constexpr unsigned kTabBytes = (kN ? 25 : 16) * 32;
  unsigned usite[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int site = site0 + 64 * s;
    usite[s] = (unsigned)(site < site_end ? site : site_end - 1);
  }
  double a[S][4];
  int scal[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    a[s][0] = a[s][1] = a[s][2] = a[s][3] = 1.0;
    scal[s] = 0;
  }
  // Pending siblings (each already multiplied by its branch matrix): slot 0 in registers, deeper slots in a private
  // array (scratch memory).  After K0c's rewrite a directly popped cherry never reaches the stack and K0c numbers the
  // slots by depth: the configs[2] trees push slot 0 ten times and slot 1 twice per tree, nothing deeper.  One
  // register slot instead of three is what lets the walk run at six waves per SIMD without spilling its working set.
  double st0[S][4];
  double deep[kDepth > 1 ? kDepth - 1 : 1][S][4];
#pragma unroll
  for (int s = 0; s < S; ++s) st0[s][0] = st0[s][1] = st0[s][2] = st0[s][3] = 0.0;
  const char* tipb = reinterpret_cast<const char*>(tiptab);
  const char* ctabb = reinterpret_cast<const char*>(ctab);
  const uint8_t* msa_m = msa - L;  // row of tip t (MSA row t - 1) at msa_m + t * L
  const int last = n_w > 0 ? n_w - 1 : 0;
  // descriptor pipeline: d0 = this op, d1 = the next one (both in scalar registers), dv = the one after (LDS read in flight)
  WalkOp d0, d1, dv;
  {
    const WalkOp t0 = desc[0], t1 = desc[last < 1 ? last : 1];
    d0.x = __builtin_amdgcn_readfirstlane(t0.x), d0.y = __builtin_amdgcn_readfirstlane(t0.y);
    d1.x = __builtin_amdgcn_readfirstlane(t1.x), d1.y = __builtin_amdgcn_readfirstlane(t1.y);
    dv = desc[last < 2 ? last : 2];
  }
  int sa[S], sb[S], sc[S];
#pragma unroll
  for (int s = 0; s < S; ++s) sa[s] = sb[s] = sc[s] = 0;
  if (n_w > 0) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      sa[s] = msa_m[(unsigned)(((unsigned)d0.x >> 16) * L) + usite[s]];
      sb[s] = msa_m[(unsigned)((d0.y & 0xffff) * L) + usite[s]];
      sc[s] = msa_m[(unsigned)(((unsigned)d0.y >> 16) * L) + usite[s]];
    }
  }
  unsigned pm_off = 0, tab_off = 0;  // byte offsets of the next unused P-matrix / cherry table
#ifdef LH_EXP_CT_STAMPS
  int ct_prev_kind = 0;
  unsigned long long kt[5] = {0, 0, 0, 0, 0};
  unsigned kn[5] = {0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  for (int k = 0; k < n_w; ++k) {
    const int kind = d0.x & 7;
#ifdef LH_EXP_CT_STAMPS
    {
      const unsigned long long t_now = __builtin_readcyclecounter();
      if (k > 0) {
        const int pk = __builtin_amdgcn_readfirstlane(ct_prev_kind);
#pragma unroll
        for (int q = 0; q < 5; ++q)
          if (pk == q) kt[q] += t_now - t_prev, kn[q] += 1;
      }
      t_prev = t_now;
      ct_prev_kind = kind;
    }
#endif
    const int tip_a = (unsigned)d0.x >> 16, tip_b = d0.y & 0xffff, tip_c = (unsigned)d0.y >> 16;
    // the op's matrix, if it has one, is the next in sequence: x = P a -- its own mat-vec (tip-into-accumulator,
    // table-into-accumulator, pop) or the one a push applies to the subtree it sets aside
    double x[S][4];
    if (d0.x & WOP_MATRIX) {
      const pmat_ptr pb = reinterpret_cast<pmat_ptr>(reinterpret_cast<const char __attribute__((address_space(4)))*>(pm) + LH_CT_POFF(pm_off));
#pragma unroll
      for (int s = 0; s < S; ++s) matvec(pb, a[s], x[s]);
      pm_off += 128;
    }
    const int push = (d0.x >> WOP_PUSH_SHIFT) & 31;
    if (push != 0) {
      if (push == 1) {
#pragma unroll
        for (int s = 0; s < S; ++s) st0[s][0] = x[s][0], st0[s][1] = x[s][1], st0[s][2] = x[s][2], st0[s][3] = x[s][3];
      } else {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          deep[push - 2][s][0] = x[s][0], deep[push - 2][s][1] = x[s][1];
          deep[push - 2][s][2] = x[s][2], deep[push - 2][s][3] = x[s][3];
        }
      }
    }
    // every op ends in an element-wise product: (tip column | table entry | pending sibling) x (x | tip column)
    double u[S][4];
    if (kind == W_TIP_ACC || kind == W_CTAB_ACC || kind == W_POP) {
      if (kind == W_TIP_ACC) {
#pragma unroll
        for (int s = 0; s < S; ++s) tip_column_at<kN>(tipb, tip_a * 128, sa[s], u[s]);
      } else if (kind == W_CTAB_ACC) {
#pragma unroll
        for (int s = 0; s < S; ++s) table_entry_at<kN>(ctabb, LH_CT_TOFF(tab_off), sa[s], sb[s], u[s]);
        tab_off += kTabBytes;
      } else {
        const int slot = (d0.x >> WOP_POP_SHIFT) & 15;
        if (slot == 0) {
#pragma unroll
          for (int s = 0; s < S; ++s) u[s][0] = st0[s][0], u[s][1] = st0[s][1], u[s][2] = st0[s][2], u[s][3] = st0[s][3];
        } else {
#pragma unroll
          for (int s = 0; s < S; ++s) {
            u[s][0] = deep[slot - 1][s][0], u[s][1] = deep[slot - 1][s][1];
            u[s][2] = deep[slot - 1][s][2], u[s][3] = deep[slot - 1][s][3];
          }
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        a[s][0] = u[s][0] * x[s][0];
        a[s][1] = u[s][1] * x[s][1];
        a[s][2] = u[s][2] * x[s][2];
        a[s][3] = u[s][3] * x[s][3];
      }
    } else {
      double v[S][4];
      if (kind == W_CTIP) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          table_entry_at<kN>(ctabb, LH_CT_TOFF(tab_off), sa[s], sb[s], u[s]);
          tip_column_at<kN>(tipb, tip_c * 128, sc[s], v[s]);
        }
        tab_off += kTabBytes;
      } else {  // W_CHERRY
#pragma unroll
        for (int s = 0; s < S; ++s) {
          tip_column_at<kN>(tipb, tip_a * 128, sa[s], u[s]);
          tip_column_at<kN>(tipb, tip_b * 128, sb[s], v[s]);
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        a[s][0] = u[s][0] * v[s][0];
        a[s][1] = u[s][1] * v[s][1];
        a[s][2] = u[s][2] * v[s][2];
        a[s][3] = u[s][3] * v[s][3];
      }
    }
    // rotate the descriptors and request the next op's tip states (tip A always; B and C on their flags)
    d0 = d1;
    d1.x = __builtin_amdgcn_readfirstlane(dv.x), d1.y = __builtin_amdgcn_readfirstlane(dv.y);
    dv = desc[k + 3 < n_w ? k + 3 : last];
#ifdef LH_EXP_CT_NOSTATE  // timing experiment: tip states from arithmetic, no memory (results wrong)
#pragma unroll
    for (int s = 0; s < S; ++s) {
      sa[s] = (usite[s] + ((unsigned)d0.x >> 16)) & 3;
      sb[s] = (usite[s] + (d0.y & 0xffff)) & 3;
      sc[s] = (usite[s] + ((unsigned)d0.y >> 16)) & 3;
    }
#else
#pragma unroll
    for (int s = 0; s < S; ++s) sa[s] = msa_m[(unsigned)(((unsigned)d0.x >> 16) * L) + usite[s]];
    if (d0.x & WOP_HAS_B) {
#pragma unroll
      for (int s = 0; s < S; ++s) sb[s] = msa_m[(unsigned)((d0.y & 0xffff) * L) + usite[s]];
    }
    if (d0.x & WOP_HAS_C) {
#pragma unroll
      for (int s = 0; s < S; ++s) sc[s] = msa_m[(unsigned)(((unsigned)d0.y >> 16) * L) + usite[s]];
    }
#endif
    // per-site, per-rate 2^256 rescaling, as in prune_wave
    unsigned hw[S];
#pragma unroll
    for (int s = 0; s < S; ++s)
      hw[s] = max(max((unsigned)__double2hiint(a[s][0]), (unsigned)__double2hiint(a[s][1])),
                  max((unsigned)__double2hiint(a[s][2]), (unsigned)__double2hiint(a[s][3])));
    unsigned hmin = hw[0];
#pragma unroll
    for (int s = 1; s < S; ++s) hmin = min(hmin, hw[s]);
    if (__builtin_expect(__ballot(hmin < 0x2FF00000u) != 0, 0)) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (hw[s] < 0x2FF00000u) {
          a[s][0] *= kScaleFactor;
          a[s][1] *= kScaleFactor;
          a[s][2] *= kScaleFactor;
          a[s][3] *= kScaleFactor;
          ++scal[s];
        }
      }
    }
  }
#ifdef LH_EXP_CT_STAMPS
  if (S == 2 && blockIdx.z >= 20000 && blockIdx.z < 20128 && (threadIdx.x & 63) == 0) {
    const int w = (blockIdx.z - 20000) * 8 + (threadIdx.x >> 6);
    for (int q = 0; q < 5; ++q) {
      ct_stamps[w][q] = kt[q];
      ct_stamps[w][5 + q] = kn[q];
    }
  }
#endif
  close_naive_branch<S>(a, scal, naive_tab, p4, lik, scl);
}

#undef LH_SLOT_COPY
#undef LH_PUSH_IF
#undef LH_POP_CASE
#undef LH_POP_SLOT
#undef LH_DECL

}  // namespace

// The register-stack form of the workgroup (round 2's kernel): it walks the schedule as lh_schedule_tree wrote it.
// It is what configs[2]-like shapes run (fused: all rates in one workgroup) -- measured against the cherry-table
// form below it has the shorter prologue (1.2 against 2.5 ms per 49 152 with every op skipped) and loses less there
// than the tables save in the walk -- and what large trees run (kSeg: tip table built a schedule
// segment at a time).  Block = n2 two-site waves followed by n1 one-site waves per rate; the tile's sites are
// blockIdx.x * tile .. +tile-1 (clipped to L).
//
// Device-resident schedules are not trusted.  hdr != nullptr: K0c has checked this launch's schedules (hdr[sample].w);
// hdr == nullptr (fused form): the prologue checks them itself -- every op's fields while it builds the matrix list,
// every entry of the list against the op it names -- and a malformed schedule leaves NaN and raises *err_flag.
//
// kFused: the workgroup carries ALL rate categories of its sample (waves [r * wpr, (r + 1) * wpr) walk
// rate r with their own LDS tip table) and, when the walks are done, mixes them itself:
// site_lik[n][1][5][L] then holds the rate mixture (equal weights, scalers aligned to the smallest, the
// arithmetic K2a would do) and K2a runs with a single "rate".  A quarter of the output traffic, and K2a's
// bandwidth-bound assembly shrinks to a quarter.  Used when R * wpr <= 8 waves and the R tip tables fit.
template <int kDepth, bool kTwo, bool kN, bool kFused, bool kSeg = false>
__device__ __forceinline__ void prune_body(int sample, int slot, int n2, int tile, int R, int wpr, const uint8_t* __restrict__ msa, int L,
                                           int T, int n_ops, const int32_t* __restrict__ ops,
                                           const int4* __restrict__ hdr, int32_t* err_flag,
                                           const double* __restrict__ brlen, const double* __restrict__ rates,
                                           const double* __restrict__ eig, double* pmat_w, size_t rate_stride,
                                           const double* __restrict__ pi,
                                           double* __restrict__ site_lik, int32_t* __restrict__ site_scal) {
  extern __shared__ double2 smem2[];
  LH_K1_PHASE(0)
  const int tid = threadIdx.x;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rate = kFused ? wave_all / wpr : (int)blockIdx.y;
  const int wave = kFused ? wave_all - rate * wpr : wave_all;  // within the rate
  const int nthr = kFused ? wpr * 64 : (int)blockDim.x;       // threads working on this rate
  const int rtid = kFused ? tid - rate * nthr : tid;
  if (hdr != nullptr && hdr[sample].w != 0) {  // rejected by the check kernel in front (uniform per workgroup): no number may look like a result
    const int planes = kFused ? 1 : R, plane = kFused ? 0 : rate;
    const int t0 = blockIdx.x * tile, t1 = min(t0 + tile, L);
    for (int site = t0 + tid; site < t1; site += blockDim.x) {
      for (int b = 0; b < 5; ++b) site_lik[(((size_t)sample * planes + plane) * 5 + b) * (size_t)L + site] = __builtin_nan("");
      site_scal[((size_t)sample * planes + plane) * (size_t)L + site] = 0;
    }
    return;
  }
  // scratch area of one (workgroup slot, rate): the schedule's P-matrices
  const size_t pm_off = ((size_t)slot * R + rate) * rate_stride;
  // LDS tip table [T][4][4] (per rate when fused); large trees (kSeg): the segment slots [2 kSegOps][4][4]
  // followed by the naive tip's entry
  double* tiptab = reinterpret_cast<double*>(smem2) + (kFused ? (size_t)rate * T * 16 : 0);
  const double* naive_tab = kSeg ? tiptab + 2 * kSegOps * 16 : tiptab;
  const int4* __restrict__ op_ptr = reinterpret_cast<const int4*>(ops) + (size_t)sample * n_ops;

  // Prologue (formerly a kernel of its own): the P-matrices of this (sample, rate).
  //   P = I + U expm1(lambda t r) U^-1   (pll_update_prob_matrices [3P])
  // One thread per matrix.  The first half of the rate's threads takes the schedule's ops: op k's
  // accumulator-child matrix goes to pmat[k][0], its popped-child matrix to pmat[k][1] -- global memory,
  // because the walk below wants them as SCALAR operands and scalar loads only read memory; the lines
  // are written and, a barrier later, read back on the same CU, so they are served by its L2.  The
  // second half takes the tip branches: a tip child needs no mat-vec, P * onehot(state) is a column of
  // P, and those columns go straight into the LDS table tiptab[tip][state][4] (states A,C,G,T).
  {
    const double* __restrict__ e = eig + (size_t)sample * 36;
    const double rt = rates[(size_t)sample * R + rate];
    const double* __restrict__ bl = brlen + (size_t)sample * (2 * (size_t)T - 2);
    double* pw = pmat_w + pm_off;
    double P[4][4];
    if constexpr (kFused) {
      // Packed form: the T - 3 inner-branch matrices are numbered by the schedule (lh_schedule_tree leaves each
      // op's running count in its descriptor); every op thread notes where its one or two matrices go in a
      // list in LDS (the same for every rate: one copy per workgroup, behind the tip tables), and after a
      // barrier thread t of a rate takes items t, t + nthr, ... of [inner matrices | tips]: no lane idles on a
      // cherry, none computes two matrices while its neighbours compute one.
      uint16_t* mat_list = reinterpret_cast<uint16_t*>(reinterpret_cast<double*>(smem2) + (size_t)R * T * 16);
      const int nodes = 2 * T - 2;
      // (the schedule has been checked by schedule_ranks_kernel: fields in range, ranks = the running matrix count, so the
      // list is written completely and within its T - 3 entries; a rejected sample never gets here)
      if (rate == 0) {
        for (int k = rtid; k < n_ops; k += nthr) {
          const int4 op = op_ptr[k];
          const int kind = op.x & 15, rank = op.x >> OP_RANK_SHIFT;
          if (kind == OP_CHERRY) continue;
          mat_list[rank] = (uint16_t)(2 * k);          // the accumulator child's matrix: node op.z, slot [k][0]
          if (kind == OP_POP_ACC) mat_list[rank + 1] = (uint16_t)(2 * k + 1);  // the popped child's: node op.y, slot [k][1]
        }
      }
      __syncthreads();
      LH_K1_PHASE(1)
      const int n_inner = T - 3;
      for (int it = rtid; it < n_inner + T; it += nthr) {
        if (it < n_inner) {
          const int code = mat_list[it], k = min(code >> 1, n_ops - 1);
          const int4 op = op_ptr[k];
          const int node = min(max((code & 1) ? op.y : op.z, 0), nodes - 1);
          compute_pmatrix(e, bl[node] * rt, P);
          double* o = pw + (size_t)k * 32 + (code & 1) * 16;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) o[i * 4 + q] = P[i][q];
        } else {
          const int j = it - n_inner;
          compute_pmatrix(e, bl[j] * rt, P);
          double* o = tiptab + j * 16;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int st = 0; st < 4; ++st) o[st * 4 + i] = P[i][st];
        }
      }
    } else {
    const int half = nthr >= 128 ? (nthr / 128) * 64 : 0;  // whole waves on either side
    const bool do_ops = half == 0 || rtid < half;
    const bool do_tips = half == 0 || rtid >= half;
    if (do_ops) {
      const int stride = half ? half : nthr;
      for (int k = rtid; k < n_ops; k += stride) {
        const int4 op = op_ptr[k];
        const int kind = op.x & 15;
        if (kind == OP_CHERRY) continue;
        double* o = pw + (size_t)k * 32;
        compute_pmatrix(e, bl[op.z] * rt, P);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) o[i * 4 + q] = P[i][q];
        if (kind == OP_POP_ACC) {
          compute_pmatrix(e, bl[op.y] * rt, P);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) o[16 + i * 4 + q] = P[i][q];
        }
      }
    }
    if (do_tips) {
      const int stride = half ? nthr - half : nthr;
      // (kSeg: only the naive tip here; the other tips' matrices are made a segment at a time, below)
      for (int j = half ? rtid - half : rtid; j < (kSeg ? 1 : T); j += stride) {
        compute_pmatrix(e, bl[j] * rt, P);
        double* o = kSeg ? tiptab + 2 * kSegOps * 16 : tiptab + j * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int st = 0; st < 4; ++st) o[st * 4 + i] = P[i][st];
      }
    }
    }
  }
  SegCtx seg{eig + (size_t)sample * 36, brlen + (size_t)sample * (2 * (size_t)T - 2),
             rates[(size_t)sample * R + rate], tiptab, rtid, nthr};
  if constexpr (kSeg) seg_fill(seg, op_ptr, n_ops, 0);
  // every storing wave's stores have reached L2 (which is where the scalar cache fills from) before any
  // wave passes the barrier
  LH_K1_PHASE(2)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __threadfence_block();
  __syncthreads();
  LH_K1_PHASE(3)

  // P-matrices in schedule order (addresses depend on the op number only), readable from here on; a workgroup that
  // works through several samples has read these addresses before (the scalar cache may still hold the last sample's lines)
  if (slot != sample) asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  const pmat_ptr pm = pmat_after_barrier(pmat_w + pm_off);
  const int lane = tid & 63;
  const int tile0 = blockIdx.x * tile;
  const int site_end = min(tile0 + tile, L);
  const double* __restrict__ p4 = pi + (size_t)sample * 4;
  // results of this wave's walk: five naive-state likelihoods and a scaler count per site
  double lik[2][5];
  int scl[2] = {0, 0};
  int site0, n_own;
  bool two_sites = false;
  if constexpr (kTwo) two_sites = wave < n2;
  if (two_sites) {
    site0 = tile0 + wave * 128 + lane;
    n_own = 2;
    if constexpr (kTwo)
      prune_wave<kDepth, 2, kN, kSeg>(site0, site_end, msa, L, n_ops, op_ptr, pm, tiptab, naive_tab, p4, lik, scl, seg);
  } else {
    site0 = tile0 + n2 * 128 + (wave - n2) * 64 + lane;
    n_own = 1;
    double lik1[1][5];
    int scl1[1];
    prune_wave<kDepth, 1, kN, kSeg>(site0, site_end, msa, L, n_ops, op_ptr, pm, tiptab, naive_tab, p4, lik1, scl1, seg);
#pragma unroll
    for (int b = 0; b < 5; ++b) lik[0][b] = lik1[0][b];
    scl[0] = scl1[0];
  }

  if constexpr (!kFused) {
    double* lik_out = site_lik + (((size_t)sample * R + rate) * 5) * (size_t)L;
    int32_t* scal_out = site_scal + ((size_t)sample * R + rate) * (size_t)L;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int site = site0 + 64 * s;
      if (s < n_own && site < site_end) {
#pragma unroll
        for (int b = 0; b < 5; ++b) lik_out[(size_t)b * L + site] = lik[s][b];
        scal_out[site] = scl[s];
      }
    }
  } else {
    // exchange through LDS (over the tip tables, which no wave needs any more), then mix the rates
    const int pad = n2 * 128 + (wpr - n2) * 64;  // sites a rate's waves cover
    LH_K1_PHASE(4)
    __syncthreads();
    LH_K1_PHASE(5)
    double* X = reinterpret_cast<double*>(smem2);                 // [R][5][pad]
    int* SC = reinterpret_cast<int*>(X + (size_t)R * 5 * pad);    // [R][pad]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int idx = site0 + 64 * s - tile0;
      if (s < n_own && idx < pad) {
#pragma unroll
        for (int b = 0; b < 5; ++b) X[((size_t)rate * 5 + b) * pad + idx] = lik[s][b];
        SC[rate * pad + idx] = scl[s];
      }
    }
    __syncthreads();
    // PhyloHMM::FillXmsaEmission's rate mixture (src/PhyloHMM.cpp:226-237): equal weights, scalers aligned
    // to the smallest one -- the same operations in the same order as K2a performs on unmixed input
    const int n_tile = site_end - tile0;
    const double w = 1.0 / R;
    double* lik_out = site_lik + ((size_t)sample * 5) * (size_t)L;
    int32_t* scal_out = site_scal + (size_t)sample * (size_t)L;
    for (int j = tid; j < 5 * n_tile; j += blockDim.x) {
      const int b = j / n_tile, p = j - b * n_tile;
      int smin = 0x7fffffff;
      for (int r = 0; r < R; ++r) smin = min(smin, SC[r * pad + p]);
      double acc = 0.0;
      for (int r = 0; r < R; ++r) {
        double v = X[((size_t)r * 5 + b) * pad + p];
        const int d = SC[r * pad + p] - smin;
        for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
        acc += w * v;
      }
      lik_out[(size_t)b * L + tile0 + p] = acc;
      if (b == 0) scal_out[tile0 + p] = smin;
    }
    LH_K1_PHASE(6)
  }
}

// x = P * a with P row-major in vector registers (the prologue's table building)
__device__ __forceinline__ void matvec_v(const double (&p)[16], const double (&a)[4], double (&x)[4]) {
  x[0] = fma(p[3], a[3], fma(p[2], a[2], fma(p[1], a[1], p[0] * a[0])));
  x[1] = fma(p[7], a[3], fma(p[6], a[2], fma(p[5], a[1], p[4] * a[0])));
  x[2] = fma(p[11], a[3], fma(p[10], a[2], fma(p[9], a[1], p[8] * a[0])));
  x[3] = fma(p[15], a[3], fma(p[14], a[2], fma(p[13], a[1], p[12] * a[0])));
}

// K0c: checks a sample's schedule (kinds, node ranges, stack discipline: everything K1 indexes with) and
// -- rewrite != 0 -- writes the walk descriptors (WalkOp above; lh_device.h), the branch length of every P-matrix the
// walk consumes (in that order) and the list of cherry tables.  A malformed schedule gets an empty walk, hdr.w = 1 (K1
// then leaves NaN) and sets *err_flag, which lh_family_status reports: device-resident schedules are not trusted.
// The check is a serial walk over the T - 2 ops; two organisations, chosen by the batch size (launch_prune):
//  * kWave (small batches: latency): one WAVE per sample.  It copies the sample's ops into LDS (coalesced; ops_in_lds ==
//    0: trees too large for that, read from global memory), walks them on the scalar unit and notes the NODE of every
//    matrix in LDS, then all lanes fetch those nodes' branch lengths side by side.  60 us per 2048 samples of a 101-tip
//    tree (the ancestral-sequence step's batch); but a CU has ONE scalar unit: 0.73 ms per 49 152.
//    Dynamic LDS: [ops_in_lds ? n_ops : 0] int4 | [n_ops] int matrix nodes | [tabs_stride] int table nodes | [4 + 16] int.
//  * !kWave (large batches: throughput): one THREAD per sample, every lane its own walk, each branch length loaded where
//    it is needed: 0.19 ms per 49 152 samples of a 101-tip tree, but the same 0.2-0.9 ms for a few thousand samples
//    (n_ops dependent round trips to memory).
template <bool kWave>
__global__ void __launch_bounds__(64) schedule_check_kernel(int n, int T, int max_depth, int tabs_stride, int use_tables,
                                                            int rewrite, int ops_in_lds,
                                                            const int32_t* __restrict__ ops,
                                                            const double* __restrict__ brlen, int2* __restrict__ wops,
                                                            double* __restrict__ wlen, int4* __restrict__ tabs,
                                                            int4* __restrict__ hdr, int32_t* err_flag) {
  extern __shared__ int4 k0c_lds[];
  const int lane = threadIdx.x;
  const int smp = kWave ? (int)blockIdx.x : (int)blockIdx.x * 64 + lane;
  if (!kWave && smp >= n) return;
  const int n_ops = T - 2, nodes = 2 * T - 2;
  const int4* __restrict__ og = reinterpret_cast<const int4*>(ops) + (size_t)smp * n_ops;
  int* mnode = reinterpret_cast<int*>(k0c_lds + (ops_in_lds ? n_ops : 0));  // (kWave only, as the three below)
  int* tnode = mnode + n_ops;
  int* verdict = tnode + tabs_stride;
  int* pend_lds = verdict + 4;  // [16] per stack slot of the rewritten walk: where the pushed subtree's matrix sits in the walk's list
  int pend_own[16];             // (the same, !kWave)
  if (kWave && ops_in_lds) {
    for (int k = lane; k < n_ops; k += 64) k0c_lds[k] = og[k];
    __syncthreads();
  }
  int2* wo = wops + (size_t)smp * n_ops;
  int4* tl = tabs + (size_t)smp * tabs_stride;
  double* ml = wlen + (size_t)smp * n_ops;
  const double* __restrict__ bl = brlen + (size_t)smp * nodes;
  // kWave: the walk is the same in every lane and written so that the compiler keeps it on the scalar unit: every op
  // field passes through readfirstlane, all branches are uniform; lane 0 alone stores.
  const bool first = !kWave || lane == 0;
  auto load_op = [&](int k) {
    if constexpr (kWave) {
      const int4 v = ops_in_lds ? k0c_lds[k] : og[k];
      return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y),
                       __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w));
    } else {
      return og[k];
    }
  };
  // where matrix i of the walk's list comes from: noted (kWave) or fetched on the spot
  auto set_mnode = [&](int i, int node) {
    if constexpr (kWave) {
      if (first) mnode[i] = node;
    } else {
      if (rewrite) ml[i] = bl[node];
    }
  };
  auto set_pend = [&](int d, int i) {
    if constexpr (kWave) {
      if (first) pend_lds[d] = i;
    } else {
      pend_own[d] = i;
    }
  };
  auto get_pend = [&](int d) { return kWave ? pend_lds[d] : pend_own[d]; };
  int depth = 0, n_w = 0, n_mat = 0, n_tab = 0;
  int bdepth = 0;  // stack depth of the schedule as written (depth: of the rewritten walk, which pushes less)
  bool bad = false;
  auto tip_ok = [&](int v) { return v >= 1 && v < T; };
  auto inner_ok = [&](int v) { return v >= T && v < nodes; };
  int k = 0;
  int4 op = load_op(0);
  while (k < n_ops && !bad) {
    const bool has_next = k + 1 < n_ops;
    const int4 nx = has_next ? load_op(k + 1) : make_int4(15, 0, 0, 0);
    const int kind = op.x & 15;
    const bool push = (op.x & OP_PUSH_FLAG) != 0;
    if (op.x < 0 || (op.x & 0xe0)) bad = true;
    int2 w = make_int2(0, 0);
    int step = 1;
    if (kind == OP_CHERRY) {
      if (!tip_ok(op.y) || !tip_ok(op.z)) bad = true;
      if (push && (op.w != bdepth || bdepth >= max_depth || depth >= 16 || n_mat >= n_ops)) bad = true;
      if (push != (k != 0)) bad = true;  // a later cherry that does not push would overwrite a live accumulator; the first has none to push
      const int nk = nx.x & 15;
      const bool nx_plain = nx.x >= 0 && (nx.x & 0xf0) == 0;
      if (bad) {
      } else if (use_tables && has_next && nk == OP_POP_ACC && push && nx_plain && nx.w == op.w && inner_ok(nx.y) &&
                 inner_ok(nx.z) && n_tab < tabs_stride) {
        // the cherry is the whole second subtree: the first one stays in the accumulator, nothing is pushed
        w = make_int2(W_CTAB_ACC | WOP_MATRIX | WOP_HAS_B | (op.y << 16), op.z | (1 << 16));
        set_mnode(n_mat, nx.y);
        if (first) {
          if (kWave) tnode[n_tab] = nx.z;
          if (rewrite) tl[n_tab] = make_int4(op.y, op.z, nx.z, 0);
        }
        ++n_mat, ++n_tab;
        step = 2;
      } else {
        int pushbits = 0;
        if (push) {  // the accumulator is set aside as P_first a: the matrix's node is named by the matching pop
          pushbits = WOP_MATRIX | ((depth + 1) << WOP_PUSH_SHIFT);
          set_pend(depth, n_mat);
          ++n_mat, ++depth, ++bdepth;
        }
        if (use_tables && has_next && nk == OP_TIP_ACC && nx_plain && tip_ok(nx.y) && inner_ok(nx.z) && n_tab < tabs_stride) {
          w = make_int2(W_CTIP | pushbits | WOP_HAS_B | WOP_HAS_C | (op.y << 16), op.z | (nx.y << 16));
          if (first) {
            if (kWave) tnode[n_tab] = nx.z;
            if (rewrite) tl[n_tab] = make_int4(op.y, op.z, nx.z, 0);
          }
          ++n_tab;
          step = 2;
        } else {
          w = make_int2(W_CHERRY | pushbits | WOP_HAS_B | (op.y << 16), op.z | (1 << 16));
        }
      }
    } else if (kind == OP_TIP_ACC) {
      if (push || k == 0 || !tip_ok(op.y) || !inner_ok(op.z) || n_mat >= n_ops) {
        bad = true;
      } else {
        w = make_int2(W_TIP_ACC | WOP_MATRIX | (op.y << 16), 1 | (1 << 16));
        set_mnode(n_mat, op.z);
        ++n_mat;
      }
    } else if (kind == OP_POP_ACC) {
      if (push || bdepth < 1 || depth < 1 || op.w != bdepth - 1 || !inner_ok(op.y) || !inner_ok(op.z) || n_mat >= n_ops) {
        bad = true;
      } else {
        --depth, --bdepth;
        w = make_int2(W_POP | WOP_MATRIX | (depth << WOP_POP_SHIFT) | (1 << 16), 1 | (1 << 16));
        set_mnode(get_pend(depth), op.y);  // the popped child: its matrix was applied at the push
        set_mnode(n_mat, op.z);            // the child whose CLV is in the accumulator
        ++n_mat;
      }
    } else {
      bad = true;
    }
    if (!bad) {
      if (rewrite && first) wo[n_w] = w;
      ++n_w;
    }
    k += step;
    if (k < n_ops) op = step == 2 ? load_op(k) : nx;
  }
  if (depth != 0 || bdepth != 0 || n_mat + n_tab > T - 3) bad = true;  // (the scratch area holds T - 3 matrices per rate)
  if (first) {
    if (bad) {
      hdr[smp] = make_int4(0, 0, 0, 1);
      atomicOr(err_flag, 1);
    } else {
      hdr[smp] = make_int4(n_w, n_mat, n_tab, 0);
    }
  }
  if (!rewrite || bad) return;
  // branch length of every inner-branch matrix K1's prologue computes, in the order it stores them: the walk's matrices
  // in walk order, then the cherry branches' (table c at n_mat + c)
  if constexpr (kWave) {
    __syncthreads();
    for (int i = lane; i < n_mat; i += 64) ml[i] = bl[mnode[i]];
    for (int c = lane; c < n_tab; c += 64) ml[n_mat + c] = bl[tnode[c]];
  } else {
    for (int c = 0; c < n_tab; ++c) ml[n_mat + c] = bl[tl[c].z];  // (the thread's own stores)
  }
}

// The register-stack kernels for large trees (segmented tip table) walk the schedule as lh_schedule_tree wrote it; what
// they need from a device-resident schedule is that nothing in it indexes out of bounds: one thread per op checks the
// fields the kernel indexes with (tips: MSA rows and branch lengths; nodes: branch lengths; slots: registers).  hdr is
// cleared before the launch; any bad op marks its sample (hdr.w: K1 leaves NaN there) and sets *err_flag.  (A schedule
// that breaks stack discipline within those bounds computes a wrong number without touching foreign memory; K0c, which
// also checks the discipline, runs where the walk is rewritten anyway.)
__global__ void __launch_bounds__(256) schedule_fields_kernel(int n, int T, int slots, const int32_t* __restrict__ ops,
                                                              int4* __restrict__ hdr, int32_t* err_flag) {
  const int n_ops = T - 2, nodes = 2 * T - 2;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * n_ops) return;
  const int smp = (int)(i / n_ops), k = (int)(i - (size_t)smp * n_ops);
  const int4 op = reinterpret_cast<const int4*>(ops)[i];
  const int kind = op.x & 15;
  const bool push = (op.x & OP_PUSH_FLAG) != 0;
  bool ok = op.x >= 0 && (op.x & 0xe0) == 0 && kind <= OP_POP_ACC;
  if (kind == OP_CHERRY) ok = ok && op.y >= 1 && op.y < T && op.z >= 1 && op.z < T && push == (k != 0);
  if (kind == OP_TIP_ACC) ok = ok && !push && k > 0 && op.y >= 1 && op.y < T && op.z >= T && op.z < nodes;
  if (kind == OP_POP_ACC) ok = ok && !push && k > 0 && op.y >= T && op.y < nodes && op.z >= T && op.z < nodes;
  if (push || kind == OP_POP_ACC) ok = ok && op.w >= 0 && op.w < slots;
  if (!ok) {
    hdr[smp].w = 1;
    atomicOr(err_flag, 1);
  }
}

// The register-stack kernels with all rates in one workgroup place their P-matrices by the RANK each op carries (the
// running count of inner-branch matrices lh_schedule_tree leaves in the descriptor).  One wave per sample checks, 64 ops
// at a time, every field the kernel indexes with and -- a prefix sum across the wave -- that each op's rank is that
// running count and the total T - 3; hdr[smp] = (0, 0, 0, verdict) for every sample.  (Round 3 first made these checks
// inside K1's prologue: their scalars cost the walk registers -- K1 5.63 -> 5.93 ms on one box -- so they moved here:
// 49 152 samples of a 101-tip tree in ~20 us.)
__global__ void __launch_bounds__(64) schedule_ranks_kernel(int T, int slots, const int32_t* __restrict__ ops,
                                                            int4* __restrict__ hdr, int32_t* err_flag) {
  const int smp = blockIdx.x, lane = threadIdx.x;
  const int n_ops = T - 2, nodes = 2 * T - 2;
  const int4* __restrict__ o = reinterpret_cast<const int4*>(ops) + (size_t)smp * n_ops;
  int carry = 0;
  bool bad = false;
  for (int k0 = 0; k0 < n_ops; k0 += 64) {
    const int k = k0 + lane;
    const bool in = k < n_ops;
    const int4 op = in ? o[k] : make_int4(OP_CHERRY, 1, 1, 0);
    const int kind = op.x & 15, rank = op.x >> OP_RANK_SHIFT;
    const bool push = (op.x & OP_PUSH_FLAG) != 0;
    bool ok = op.x >= 0 && (op.x & 0xe0) == 0 && kind <= OP_POP_ACC;
    if (kind == OP_CHERRY) ok = ok && op.y >= 1 && op.y < T && op.z >= 1 && op.z < T && push == (k != 0);
    if (kind == OP_TIP_ACC) ok = ok && !push && k > 0 && op.y >= 1 && op.y < T && op.z >= T && op.z < nodes;
    if (kind == OP_POP_ACC) ok = ok && !push && k > 0 && op.y >= T && op.y < nodes && op.z >= T && op.z < nodes;
    if (push || kind == OP_POP_ACC) ok = ok && op.w >= 0 && op.w < slots;
    const int m = !in || !ok ? 0 : kind == OP_TIP_ACC ? 1 : kind == OP_POP_ACC ? 2 : 0;
    int incl = m;  // inclusive prefix sum over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (in && kind != OP_CHERRY && rank != carry + incl - m) ok = false;
    if (in && !ok) bad = true;
    carry += __shfl(incl, 63);
  }
  if (carry != T - 3) bad = true;
  const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
  if (lane == 0) {
    hdr[smp] = make_int4(0, 0, 0, any_bad ? 1 : 0);
    if (any_bad) atomicOr(err_flag, 1);
  }
}

// The workgroup of the cherry-table form.  Block layout as prune_body (n2 two-site waves + n1 one-site waves per rate;
// kFused: all R rates of the sample in one workgroup, mixed at the end).  scratch: this (sample, rate)'s region of
// rate_stride doubles: [n_mat + n_tab][16] P-matrices | [n_tab][E][4] tables.
template <int kDepth, bool kN, bool kFused, int kS = 2, bool kAsm = false, bool kTipsG = false>
__device__ __forceinline__ void prune_body_ct(int n2, int tile, int R, int wpr, const uint8_t* __restrict__ msa, int L,
                                              int T, const int2* __restrict__ wops, const double* __restrict__ wlen,
                                              const int4* __restrict__ tabs, int tabs_stride,
                                              const int4* __restrict__ hdr, const double* __restrict__ brlen,
                                              const double* __restrict__ rates, const double* __restrict__ eig,
                                              double* pmat_w, size_t rate_stride, const double* __restrict__ pi,
                                              double* __restrict__ site_lik, int32_t* __restrict__ site_scal) {
  extern __shared__ double2 smem2[];
  constexpr int SY = kN ? 5 : 4, E = SY * SY;
  const int tid = threadIdx.x;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rate = kFused ? wave_all / wpr : (int)blockIdx.y;
  const int wave = kFused ? wave_all - rate * wpr : wave_all;
  const int nthr = kFused ? wpr * 64 : (int)blockDim.x;
  const int rtid = kFused ? tid - rate * nthr : tid;
  const int sample = blockIdx.z;
  const int n_ops = T - 2;
  const int4 h = hdr[sample];
#ifdef LH_DEBUG_WALK  // debugging aid: stop every walk after lh_dbg_max_ops ops (LH_DBG_MAXOPS), to bisect a walk against another
  const int n_w = min(__builtin_amdgcn_readfirstlane(h.x), lh_dbg_max_ops);
#else
  const int n_w = __builtin_amdgcn_readfirstlane(h.x);
#endif
  const int n_mat = __builtin_amdgcn_readfirstlane(h.y), n_tab = __builtin_amdgcn_readfirstlane(h.z);
  const bool malformed = h.w != 0;
  double* pw = pmat_w + ((size_t)sample * R + rate) * rate_stride;
  double* ctab = pw + (size_t)(T - 3 > 0 ? T - 3 : 0) * 16;
  // kTipsG (large trees: a 64 KB tip table in LDS would leave two workgroups per CU): the tip table [T][4][4] sits in the
  // scratch region behind the cherry tables; the walk gathers its columns with vector loads, as it does table entries
  static_assert(!kTipsG || !kFused, "tip tables in the scratch region: unfused kernels only");
  const size_t tip_off = (size_t)(T - 3 > 0 ? T - 3 : 0) * 16 + (size_t)tabs_stride * E * 4;  // doubles
  double* tiptab = kTipsG ? pw + tip_off : reinterpret_cast<double*>(smem2) + (kFused ? (size_t)rate * T * 16 : 0);
  const double* naive_tab = tiptab;
  const int4* __restrict__ tl = tabs + (size_t)sample * tabs_stride;
  // the walk descriptors go to LDS behind the tip tables (one copy per workgroup) for the waves that run the C++ walk
  WalkOp* desc = reinterpret_cast<WalkOp*>(reinterpret_cast<double*>(smem2) + (kTipsG ? 0 : (size_t)(kFused ? R : 1) * T * 16));
  // (the assembly walk fetches its descriptors from global memory with scalar loads: this copy also brings their
  // lines into L2 before the walk asks for them -- without it every eighth op waited for HBM)
  for (int i = tid; i < n_w; i += blockDim.x) desc[i] = wops[(size_t)sample * n_ops + i];
  LH_CT_PHASE(0)

  // Prologue, first half: the P-matrices of this (sample, rate), one thread per matrix (K0c left every matrix's
  // branch length in the order they are stored): the walk's inner-branch matrices to the scratch area, the tip
  // branches' into the LDS tip table (column by column, as the walk gathers them), the cherry branches' for the
  // tables.  Without N a table's four state rows are built by the four lanes of a QUAD, and the quad's first lane
  // computes the cherry branch's matrix here and keeps it in registers: it reaches the other three through DPP in the
  // second half, not through memory (tables beyond nthr / 4, and all tables of alignments with N, take the path
  // through the scratch area).
  constexpr bool kQuad = !kN;
  const int n_q = kQuad ? min(n_tab, nthr >> 2) : 0;
  const bool pc_lane = kQuad && (rtid & 3) == 0 && (rtid >> 2) < n_q;
  int4 tcell = make_int4(1, 1, 0, 0);
  if (kQuad && (rtid >> 2) < n_q) tcell = tl[rtid >> 2];  // the quad's table: requested now, used after the barrier
  double pcq[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) pcq[i] = 0.0;
  {
    const double* __restrict__ e = eig + (size_t)sample * 36;
    const double rt = rates[(size_t)sample * R + rate];
    const double* __restrict__ bl = brlen + (size_t)sample * (2 * (size_t)T - 2);
    const double* __restrict__ wl = wlen + (size_t)sample * n_ops;
    double P[4][4];
    // One matrix per thread and round.  Round 0: a quad's first lane takes its cherry matrix, every other thread the
    // item of its rank in the common list (walk matrices | cherry matrices n_q.. (scratch path) | tips); later rounds:
    // the rest of the list, nthr items at a time.  A single compute_pmatrix per round, whatever the item's kind.
    const int n_rest = n_tab - n_q;
    const int n_list = n_mat + n_rest + T;
    const int round1 = nthr - n_q;  // list items taken in round 0
    const int n_rounds = 1 + (n_list > round1 ? (n_list - round1 + nthr - 1) / nthr : 0);
    for (int round = 0; round < n_rounds; ++round) {
      const bool is_pc = round == 0 && pc_lane;
      const int it = round == 0 ? rtid - min(n_q, (rtid + 3) >> 2) : round1 + rtid + (round - 1) * nthr;
      const bool in_list = !is_pc && it < n_list;
      const bool inner = in_list && it < n_mat + n_rest;
      const int slot = it < n_mat ? it : it + n_q;  // matrix slot in the scratch area ([n_mat + c] for table c)
      const int j = it - (n_mat + n_rest);          // tip
      double t = 0.0;
      if (is_pc) t = wl[n_mat + (rtid >> 2)];
      else if (inner) t = wl[slot];
      else if (in_list) t = bl[j];
      compute_pmatrix(e, t * rt, P);
      if (is_pc) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) pcq[i * 4 + q] = P[i][q];
      } else if (inner) {
        double* o = pw + (size_t)slot * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) o[i * 4 + q] = P[i][q];
      } else if (in_list) {
        double* o = tiptab + j * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int st = 0; st < 4; ++st) o[st * 4 + i] = P[i][st];
      }
    }
  }
  LH_CT_PHASE(1)
  // the tip tables are complete (LDS); the scratch-area stores need to have landed only if a table goes that way
  if (kTipsG || n_tab > n_q) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  LH_CT_PHASE(2)
  // Second half: the cherry tables, one thread per (table, state of the first tip): P_c (tipcol_y o tipcol_z) for
  // every state of the second tip -- the very operations the unfused walk performs per lane, done once per state pair.
  auto build_rows = [&](const double (&pc)[16], int c, int sy, int ty, int tz) {
    double py[4];
    tip_column<kN>(tiptab, ty, sy, py);
    double2* o = reinterpret_cast<double2*>(ctab) + ((size_t)c * E + (size_t)sy * SY) * 2;
#pragma unroll
    for (int sz = 0; sz < SY; ++sz) {
      double pz[4], pr[4], x[4];
      tip_column<kN>(tiptab, tz, sz, pz);
#pragma unroll
      for (int i = 0; i < 4; ++i) pr[i] = py[i] * pz[i];
      matvec_v(pc, pr, x);
      o[2 * sz] = make_double2(x[0], x[1]);
      o[2 * sz + 1] = make_double2(x[2], x[3]);
    }
  };
#ifndef LH_EXP_CT_NOPHASEC
  if constexpr (kQuad) {
    // every lane of the wave takes part in the DPP moves (quad_perm [0,0,0,0]: the quad's first lane to all four)
    double pc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lo = __builtin_amdgcn_mov_dpp(__double2loint(pcq[i]), 0, 0xf, 0xf, true);
      const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(pcq[i]), 0, 0xf, 0xf, true);
      pc[i] = __hiloint2double(hi, lo);
    }
    if ((rtid >> 2) < n_q) build_rows(pc, rtid >> 2, rtid & 3, tcell.x, tcell.y);
  }
  for (int it = rtid; it < (n_tab - n_q) * SY; it += nthr) {
    const int c = n_q + it / SY, sy = it - (it / SY) * SY;
    const int4 t = tl[c];
    double pc[16];
    const double2* q = reinterpret_cast<const double2*>(pw + (size_t)(n_mat + c) * 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double2 v = q[j];
      pc[2 * j] = v.x;
      pc[2 * j + 1] = v.y;
    }
    build_rows(pc, c, sy, t.x, t.y);
  }
#endif
  LH_CT_PHASE(3)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __threadfence_block();
  __syncthreads();
  LH_CT_PHASE(4)

  const pmat_ptr pm = pmat_after_barrier(pw);
  const int lane = tid & 63;
  const int tile0 = blockIdx.x * tile;
  const int site_end = min(tile0 + tile, L);
  const double* __restrict__ p4 = pi + (size_t)sample * 4;
  // n2 waves carry kS sites per lane (64 apart), the others one site per lane
  double lik[kS][5];
  int scl[kS];
#pragma unroll
  for (int s = 0; s < kS; ++s) scl[s] = 0;
  int site0, n_own;
  const bool two_sites = wave < n2;
  if (two_sites) {
    site0 = tile0 + wave * (64 * kS) + lane;
    n_own = kS;
    if constexpr (kAsm && !kN)
      prune_wave_asm<kDepth, kS, kTipsG>(tile0 + wave * (64 * kS), site_end, msa, L, n_w, wops + (size_t)sample * n_ops, pm,
                                         (unsigned)((T - 3 > 0 ? T - 3 : 0) * 128), (unsigned)(tip_off * 8), tiptab, naive_tab,
                                         p4, lik, scl);
    else
      prune_wave_ct<kDepth, kS, kN>(site0, site_end, msa, L, n_w, desc, pm, tiptab, ctab, naive_tab, p4, lik, scl);
  } else {
    site0 = tile0 + n2 * (64 * kS) + (wave - n2) * 64 + lane;
    n_own = 1;
    double lik1[1][5];
    int scl1[1];
    prune_wave_ct<kDepth, 1, kN>(site0, site_end, msa, L, n_w, desc, pm, tiptab, ctab, naive_tab, p4, lik1, scl1);
#pragma unroll
    for (int b = 0; b < 5; ++b) lik[0][b] = lik1[0][b];
    scl[0] = scl1[0];
  }
  if (malformed) {  // K0c rejected the schedule: no number may look like a result
#pragma unroll
    for (int s = 0; s < kS; ++s)
#pragma unroll
      for (int b = 0; b < 5; ++b) lik[s][b] = __builtin_nan("");
  }

  if constexpr (!kFused) {
    double* lik_out = site_lik + (((size_t)sample * R + rate) * 5) * (size_t)L;
    int32_t* scal_out = site_scal + ((size_t)sample * R + rate) * (size_t)L;
#pragma unroll
    for (int s = 0; s < kS; ++s) {
      const int site = site0 + 64 * s;
      if (s < n_own && site < site_end) {
#pragma unroll
        for (int b = 0; b < 5; ++b) lik_out[(size_t)b * L + site] = lik[s][b];
        scal_out[site] = scl[s];
      }
    }
  } else {
    // exchange through LDS (over the tip tables, which no wave needs any more), then mix the rates (as prune_body)
    const int pad = n2 * (64 * kS) + (wpr - n2) * 64;
    LH_CT_PHASE(5)
    __syncthreads();
    LH_CT_PHASE(6)
    double* X = reinterpret_cast<double*>(smem2);
    int* SC = reinterpret_cast<int*>(X + (size_t)R * 5 * pad);
#pragma unroll
    for (int s = 0; s < kS; ++s) {
      const int idx = site0 + 64 * s - tile0;
      if (s < n_own && idx < pad) {
#pragma unroll
        for (int b = 0; b < 5; ++b) X[((size_t)rate * 5 + b) * pad + idx] = lik[s][b];
        SC[rate * pad + idx] = scl[s];
      }
    }
    __syncthreads();
    const int n_tile = site_end - tile0;
    const double w = 1.0 / R;
    double* lik_out = site_lik + ((size_t)sample * 5) * (size_t)L;
    int32_t* scal_out = site_scal + (size_t)sample * (size_t)L;
    for (int j = tid; j < 5 * n_tile; j += blockDim.x) {
      const int b = j / n_tile, p = j - b * n_tile;
      int smin = 0x7fffffff;
      for (int r = 0; r < R; ++r) smin = min(smin, SC[r * pad + p]);
      double acc = 0.0;
      for (int r = 0; r < R; ++r) {
        double v = X[((size_t)r * 5 + b) * pad + p];
        const int d = SC[r * pad + p] - smin;
        for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
        acc += w * v;
      }
      lik_out[(size_t)b * L + tile0 + p] = acc;
      if (b == 0) scal_out[tile0 + p] = smin;
    }
    LH_CT_PHASE(7)
  }
}

#define LH_PRUNE_CT_PARAMS                                                                                          \
  int n2, int tile, int R, int wpr, const uint8_t *__restrict__ msa, int L, int T, const int2 *__restrict__ wops,   \
      const double *__restrict__ wlen, const int4 *__restrict__ tabs, int tabs_stride,                              \
      const int4 *__restrict__ hdr, const double *__restrict__ brlen, const double *__restrict__ rates,            \
      const double *__restrict__ eig, double *pmat_w, size_t rate_stride, const double *__restrict__ pi,           \
      double *__restrict__ site_lik, int32_t *__restrict__ site_scal
#define LH_PRUNE_CT_ARGS \
  n2, tile, R, wpr, msa, L, T, wops, wlen, tabs, tabs_stride, hdr, brlen, rates, eig, pmat_w, rate_stride, pi, site_lik, site_scal
#define LH_PRUNE_CT_KERNEL(NAME, WAVES)                                                              \
  template <int kDepth, bool kN, bool kFused, bool kAsm>                                             \
  __global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) NAME(    \
      LH_PRUNE_CT_PARAMS) {                                                                          \
    prune_body_ct<kDepth, kN, kFused, 2, kAsm>(LH_PRUNE_CT_ARGS);                                    \
  }
// Large trees: the tip table in the scratch region, no LDS but the descriptors (six waves per SIMD; assembly walk)
template <int kDepth>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6))) prune_kernel_ctg6(LH_PRUNE_CT_PARAMS) {
  prune_body_ct<kDepth, false, false, 2, true, true>(LH_PRUNE_CT_ARGS);
}
LH_PRUNE_CT_KERNEL(prune_kernel_ct6, 6)
LH_PRUNE_CT_KERNEL(prune_kernel_ct5, 5)
LH_PRUNE_CT_KERNEL(prune_kernel_ct4, 4)
#undef LH_PRUNE_CT_KERNEL
// Four sites per lane (assembly walk only): the per-op instructions that do not depend on the site count -- loads of
// the matrix, descriptor, branches, waits -- are paid once for twice the sites; 156 VGPRs, three waves per SIMD.
template <int kDepth, bool kFused>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3))) prune_kernel_ct_s4(LH_PRUNE_CT_PARAMS) {
  prune_body_ct<kDepth, false, kFused, 4, true>(LH_PRUNE_CT_ARGS);
}
// pmat_w: the scratch area (see prune_body); written in the prologue, read back after the barrier.
#define LH_PRUNE_PARAMS                                                                                     \
  int n_samples, int n2, int tile, int R, int wpr, const uint8_t *__restrict__ msa, int L, int T, int n_ops, \
      const int32_t *__restrict__ ops, const int4 *__restrict__ hdr, int32_t *err_flag,                     \
      const double *__restrict__ brlen, const double *__restrict__ rates, const double *__restrict__ eig,  \
      double *pmat_w, size_t rate_stride, const double *__restrict__ pi,                                   \
      double *__restrict__ site_lik, int32_t *__restrict__ site_scal
#define LH_PRUNE_ARGS \
  n2, tile, R, wpr, msa, L, T, n_ops, ops, hdr, err_flag, brlen, rates, eig, pmat_w, rate_stride, pi, site_lik, site_scal
// -DLH_EXP_K1_PERSIST (experiment, profiles/r03_k1_persistent_slots.txt): a workgroup takes samples blockIdx.z,
// blockIdx.z + gridDim.z, ... and keeps ONE scratch slot for all of them; with LH_K1_PERSIST=<workgroups> in the
// environment the grid has that many workgroups and the scratch area in use is <workgroups> x R x rate_stride instead of
// n x R x rate_stride.  The barrier between two samples: the mixing loop of one reads the LDS the prologue of the next
// writes.  Measured 11 % SLOWER on configs[2] (every sample of a workgroup invalidates its CU's scalar cache, which the
// other resident workgroups are reading their matrices through), so the product build has a workgroup per sample.
#ifdef LH_EXP_K1_PERSIST
#define LH_PRUNE_SAMPLES(BODY)                                                           \
  for (int sample = blockIdx.z; sample < n_samples; sample += gridDim.z) {               \
    if (sample != (int)blockIdx.z) __syncthreads();                                      \
    BODY(sample, (int)blockIdx.z, LH_PRUNE_ARGS);                                        \
  }
#else
#define LH_PRUNE_SAMPLES(BODY) BODY((int)blockIdx.z, (int)blockIdx.z, LH_PRUNE_ARGS);
#endif

// Shallow stacks (depth <= 4, any tree up to a few hundred tips), all rates in one workgroup: two sites per lane.  The
// walk needs ~100 VGPRs; resident waves matter more to it than a few spilled registers, as long as the LDS tip
// tables of that many workgroups fit a CU.  Three register budgets are therefore built -- 6 waves per
// SIMD (80 VGPRs), 5 (96) and 4 (128, no spills) -- and the launcher takes the tightest one whose
// occupancy the tip tables allow: configs[2] runs 6 waves per SIMD, 5 % faster than 5 and 19 % faster than 4.
#define LH_PRUNE_KERNEL(NAME, WAVES)                                                                 \
  template <int kDepth, bool kN>                                                                     \
  __global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) NAME(    \
      LH_PRUNE_PARAMS) {                                                                             \
    LH_PRUNE_SAMPLES((prune_body<kDepth, true, kN, true>))                                           \
  }
LH_PRUNE_KERNEL(prune_kernel_w6, 6)
LH_PRUNE_KERNEL(prune_kernel_w5, 5)
LH_PRUNE_KERNEL(prune_kernel_w4, 4)
#undef LH_PRUNE_KERNEL

// Large trees: tip table built a schedule segment at a time (see SegCtx); register budgets for five and four
// waves per SIMD.
template <int kDepth, bool kN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(5, 5))) prune_kernel_seg(LH_PRUNE_PARAMS) {
  LH_PRUNE_SAMPLES((prune_body<kDepth, true, kN, false, true>))
}
template <int kDepth, bool kN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) prune_kernel_seg4(LH_PRUNE_PARAMS) {
  LH_PRUNE_SAMPLES((prune_body<kDepth, true, kN, false, true>))
}

// Trees from this many tips on (a 24 KB tip table) may run with the tip table in the scratch region
constexpr int kTipsInScratchFrom = 192;

PruneWsSizes prune_ws_sizes(int T, bool mixed_n) {
  PruneWsSizes z;
  z.tabs_per_sample = (size_t)std::max((T - 1) / 2, 1);
  const size_t e = mixed_n ? 25 : 16;
  // the cherry-table form (large trees: + the tip table [T][16], which those kernels keep here instead of in LDS), and
  // never less than the older kernels' [T-2][2][16]
  const size_t tips = debug_options().k1_tips_scratch && T >= kTipsInScratchFrom ? (size_t)T * 16 : 0;
  z.scratch_doubles_per_rate = std::max((size_t)std::max(T - 3, 0) * 16 + z.tabs_per_sample * e * 4 + tips, (size_t)std::max(T - 2, 1) * 32);
  return z;
}

// Returns the number of rate planes it left in site_lik / site_scal: R, or 1 when the workgroups mixed
// the rate categories themselves (the count K2a must then be run with).
int launch_prune(const DevFamily& fam, int n, int R, int T, int max_depth, const int32_t* ops,
                 const double* brlen, const double* rates, const double* eig, const PruneWs& ws, const double* pi,
                 double* site_lik, int32_t* site_scal, hipStream_t stream, bool allow_fused) {
  const int L = fam.n_prune;  // distinct alignment columns; identical ones are pruned once
  const PruneWsSizes sizes = prune_ws_sizes(T, fam.msa_mixed_n != 0);
#ifdef LH_DEBUG_WALK
  {
    const int m = debug_options().dbg_maxops;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(lh_dbg_max_ops), &m, sizeof(m));
  }
#endif
  // K0c: a wave per sample (its ops staged in LDS while they fit, 16 bytes per op) for batches the scalar units get through
  // quickly, a thread per sample from 12 288 samples on (see the kernel)
  auto launch_k0c = [&](bool use_tables, bool rewrite) {
    const size_t n_ops_k = (size_t)std::max(T - 2, 1);
    if (n >= 12288) {
      hipLaunchKernelGGL(schedule_check_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, stream, n, T, max_depth,
                         (int)sizes.tabs_per_sample, use_tables ? 1 : 0, rewrite ? 1 : 0, 0, ops, brlen, ws.wops, ws.wlen, ws.tabs,
                         ws.hdr, ws.err_flag);
      return;
    }
    const size_t tail = (n_ops_k + sizes.tabs_per_sample + 20) * sizeof(int);
    const bool in_lds = n_ops_k * 16 + tail <= 48 * 1024;
    hipLaunchKernelGGL(schedule_check_kernel<true>, dim3(n), dim3(64), (in_lds ? n_ops_k * 16 : 0) + tail, stream, n, T, max_depth,
                       (int)sizes.tabs_per_sample, use_tables ? 1 : 0, rewrite ? 1 : 0, in_lds ? 1 : 0, ops, brlen, ws.wops,
                       ws.wlen, ws.tabs, ws.hdr, ws.err_flag);
  };
  if (L == 0) {               // nothing but all-N padding (K2a reads no plane at all): the schedules still get checked
    launch_k0c(false, false);
    return R;
  }
  double* pmat = ws.scratch;
  const size_t rate_stride = sizes.scratch_doubles_per_rate;
  // tile: up to 1024 sites as two-site waves plus at most one one-site wave for a remainder below 64; tiles
  // rebalanced so that they are equally full.  Large tiles matter for large trees: every workgroup of a (sample,
  // rate) repeats the P-matrix prologue and holds its own T x 128-byte tip table in LDS.
  // (LH_K1_TILE_CAP: test hook that forces small tiles so that the multi-tile path runs on small families)
  const DebugOptions& dbg = debug_options();
  const int cap_env = dbg.k1_tile_cap;
  const int cap = cap_env >= 64 ? std::min(cap_env, 1024) : 1024;
  const int tiles = (L + cap - 1) / cap;
  const int tile = (L + tiles - 1) / tiles;
  // Test hooks / experiments (read once per process): LH_K1_CXX_WALK keeps the cherry-table form's C++ walk;
  // LH_K1_S4 runs its assembly walk with FOUR sites per lane (three waves per SIMD) where a tile is (nearly) whole
  // multiples of 256 sites -- measured on configs[2]: the same time per op as two sites per lane at six waves;
  // LH_K1_TABLES makes the cherry-table form the choice for the fused shapes too (default: the register-stack form,
  // which is faster there: DESIGN.md section 6); LH_K1_NO_TABLES: the cherry-table form without tables.
  const bool cxx_walk = dbg.k1_cxx_walk, s4_env = dbg.k1_s4, tables_env = dbg.k1_tables, no_tables = dbg.k1_no_tables;
  const bool seg_env = dbg.k1_segments;   // test hook: segments on small trees too
  const int seg_waves = dbg.k1_seg_waves; // measured: 4 (128 VGPRs, few spills) beats 5 by 9 %
  const bool no_fuse = dbg.k1_no_fuse;    // test hook: one workgroup per (sample, rate)
  const bool use_asm = !cxx_walk && !fam.msa_mixed_n;
  // (four sites per lane not for large trees: 160 KB / tip table < 3 workgroups means the segmented form, two-site waves)
  const bool s4 = use_asm && s4_env && (tile % 256 == 0 || tile % 256 > 192) && (size_t)T * 128 * 3 <= 160 * 1024 && !seg_env;
  const int spl = s4 ? 256 : 128;  // sites per multi-site wave
  int n2 = tile / spl, n1;
  {
    const int rem = tile - spl * n2;
    if (rem > 64) {
      ++n2;
      n1 = 0;
    } else {
      n1 = rem > 0 ? 1 : 0;
    }
  }
  const int wpr = n2 + n1;  // waves per rate
  const size_t tip_bytes = (size_t)T * 16 * sizeof(double);
  // all rates of a sample in one workgroup, mixed there: at most 8 waves, and R tip tables (later reused
  // as the exchange area [R][5][pad] doubles + [R][pad] ints) within a third of a CU's LDS; behind the tip tables the
  // register-stack form keeps its matrix list and verdict word, the cherry-table form its copy of the walk descriptors
  const size_t pad = (size_t)n2 * spl + (size_t)n1 * 64;
  const size_t tail_bytes = (((size_t)std::max(T - 2, 1) * sizeof(int2)) + 31) & ~(size_t)15;
  const size_t fused_lds = std::max((size_t)R * tip_bytes + tail_bytes, (size_t)R * pad * (5 * sizeof(double) + sizeof(int)));
  const bool fused = allow_fused && !no_fuse && R * wpr <= 8 && fused_lds <= 53 * 1024 && !seg_env;
  // large trees: with the whole tip table in LDS fewer than five waves per SIMD would be resident
  const bool big = !fused && (160 * 1024 / tip_bytes) * wpr / 4 < 5;
  // LH_K1_TIPS_SCRATCH (experiment, profiles/r03_config4.txt): large trees without N through the cherry-table form with
  // the tip table in the scratch region (six waves per SIMD) instead of the segmented register-stack form (four).
  // Measured on the configs[4] shape: 9.4 ms against 7.5 -- a workgroup's matrices, tables and tip table are 256 KB there,
  // eight workgroups per CU put 64 MB in flight per XCD, and every tip and table gather comes from beyond L2.
  const bool tips_g = dbg.k1_tips_scratch && big && !seg_env && !fam.msa_mixed_n && !cxx_walk && T >= kTipsInScratchFrom;
  const bool seg = max_depth <= 4 && !fused && (big || seg_env) && !dbg.k1_no_segments && !tips_g;
  // the register-stack form with all rates in one workgroup runs behind schedule_ranks_kernel (fields and ranks); everything else
  // runs behind K0c
  const bool stack_fused = fused && max_depth <= 4 && !tables_env && !no_tables && !s4;
  // (the register-stack kernels for large trees walk the schedule as written: a field check is all they need)
  if (stack_fused) {
    hipLaunchKernelGGL(schedule_ranks_kernel, dim3(n), dim3(64), 0, stream, T, max_depth <= 3 ? 3 : 4, ops, ws.hdr, ws.err_flag);
  } else if (seg && allow_fused) {
    (void)hipMemsetAsync(ws.hdr, 0, sizeof(int4) * (size_t)n, stream);
    const size_t total = (size_t)n * (size_t)(T - 2);
    hipLaunchKernelGGL(schedule_fields_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, n, T, 4, ops, ws.hdr,
                       ws.err_flag);
  } else if (!stack_fused) {
    // (seg here: the ancestral-sequence step, whose own kernels follow the stack through the schedule -- full check, no rewrite)
    launch_k0c(!no_tables && !seg, !seg);
  }
  const size_t lds = fused ? fused_lds : seg ? (size_t)(2 * kSegOps + 1) * 16 * sizeof(double) : tips_g ? tail_bytes : tip_bytes + tail_bytes;
  const int wg_waves = fused ? R * wpr : wpr;
  dim3 grid(tiles, fused ? 1 : R, n), block(64 * wg_waves);
#ifdef LH_EXP_K1_PERSIST
  // LH_K1_PERSIST=<workgroups>: the register-stack kernels with that many workgroups, each working through samples
  // z, z + grid.z, ... with one scratch slot
  const int persist = dbg.k1_persist;
  if (persist > 0 && (stack_fused || seg)) grid.z = std::min(n, persist);
#endif
  const int n_ops = T - 2;
#define LH_LAUNCH_K(K, HDR)                                                                                   \
  {                                                                                                           \
    if (lds > 64 * 1024)                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                    \
    hipLaunchKernelGGL(K, grid, block, lds, stream, n, n2, tile, R, wpr, fam.msa, L, T, n_ops, ops, HDR, ws.err_flag, brlen, \
                       rates, eig, pmat, rate_stride, pi, site_lik, site_scal);                              \
  }
#define LH_LAUNCH_CT(K)                                                                                       \
  {                                                                                                           \
    if (lds > 64 * 1024)                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                    \
    hipLaunchKernelGGL(K, grid, block, lds, stream, n2, tile, R, wpr, fam.msa, L, T, ws.wops, ws.wlen, ws.tabs, \
                       (int)sizes.tabs_per_sample, ws.hdr, brlen, rates, eig, pmat, rate_stride, pi, site_lik,  \
                       site_scal);                                                                            \
  }
  // waves per SIMD that the LDS of the resident workgroups allows (160 KB per CU, 4 SIMDs)
  const int lds_waves = lds == 0 ? 8 : (int)((160 * 1024 / lds) * wg_waves / 4);
#ifndef LH_EXP_CT_BUDGET
#define LH_EXP_CT_BUDGET 6
#endif
  if (stack_fused) {
    // register-stack form, all rates in one workgroup; three register budgets, the tightest one whose occupancy the
    // tip tables allow (a fourth stack slot spills too much at 80 VGPRs)
    const int null_hdr = 0;
    (void)null_hdr;
#define LH_LAUNCH_STACK(D, N)                                          \
  {                                                                    \
    if (lds_waves >= 6 && D == 3)                                      \
      LH_LAUNCH_K((prune_kernel_w6<D, N>), ws.hdr)                     \
    else if (lds_waves >= 5)                                           \
      LH_LAUNCH_K((prune_kernel_w5<D, N>), ws.hdr)                     \
    else                                                               \
      LH_LAUNCH_K((prune_kernel_w4<D, N>), ws.hdr)                     \
  }
    if (max_depth <= 3 && !fam.msa_mixed_n)
      LH_LAUNCH_STACK(3, false)
    else if (max_depth <= 3)
      LH_LAUNCH_STACK(3, true)
    else if (!fam.msa_mixed_n)
      LH_LAUNCH_STACK(4, false)
    else
      LH_LAUNCH_STACK(4, true)
#undef LH_LAUNCH_STACK
  } else if (seg) {
    // large trees: register-stack form with the tip table built a schedule segment at a time
    if (seg_waves == 4) {
      if (fam.msa_mixed_n)
        LH_LAUNCH_K((prune_kernel_seg4<4, true>), ws.hdr)
      else
        LH_LAUNCH_K((prune_kernel_seg4<4, false>), ws.hdr)
    } else {
      if (fam.msa_mixed_n)
        LH_LAUNCH_K((prune_kernel_seg<4, true>), ws.hdr)
      else
        LH_LAUNCH_K((prune_kernel_seg<4, false>), ws.hdr)
    }
  } else {
    // The cherry-table form (two sites per lane, whole tip table in LDS; one stack slot in registers, deeper ones in
    // scratch memory, so any depth up to 16 runs it): shapes whose rates do not fit one workgroup, deep stacks, the
    // ancestral-sequence step's unmixed planes.  Three register budgets, and for alignments without N the walk in
    // assembly.
#define LH_LAUNCH_BUDGET(D, N, F, A)                          \
  {                                                           \
    if (s4 && A)                                              \
      LH_LAUNCH_CT((prune_kernel_ct_s4<D, F>))                \
    else if (lds_waves >= 6 && LH_EXP_CT_BUDGET >= 6)         \
      LH_LAUNCH_CT((prune_kernel_ct6<D, N, F, A>))            \
    else if (lds_waves >= 5 && LH_EXP_CT_BUDGET >= 5)         \
      LH_LAUNCH_CT((prune_kernel_ct5<D, N, F, A>))            \
    else                                                      \
      LH_LAUNCH_CT((prune_kernel_ct4<D, N, F, A>))            \
  }
#define LH_LAUNCH_FORM(D, N, A)        \
  {                                    \
    if (fused)                         \
      LH_LAUNCH_BUDGET(D, N, true, A)  \
    else                               \
      LH_LAUNCH_BUDGET(D, N, false, A) \
  }
#define LH_LAUNCH_SHALLOW(D)           \
  {                                    \
    if (fam.msa_mixed_n)               \
      LH_LAUNCH_FORM(D, true, false)   \
    else if (use_asm)                  \
      LH_LAUNCH_FORM(D, false, true)   \
    else                               \
      LH_LAUNCH_FORM(D, false, false)  \
  }
    if (tips_g && max_depth <= 4)
      LH_LAUNCH_CT((prune_kernel_ctg6<4>))
    else if (tips_g)
      LH_LAUNCH_CT((prune_kernel_ctg6<16>))
    else if (max_depth <= 4)
      LH_LAUNCH_SHALLOW(4)
    else
      LH_LAUNCH_SHALLOW(16)
#undef LH_LAUNCH_SHALLOW
#undef LH_LAUNCH_FORM
#undef LH_LAUNCH_BUDGET
  }
#undef LH_LAUNCH_K
#undef LH_LAUNCH_CT
#ifdef LH_EXP_CT_STAMPS
  {
    static int calls = 0;
    if (++calls == 3) {
      (void)hipDeviceSynchronize();
      static unsigned long long h[1024][10];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(ct_stamps), sizeof(h));
      const char* names[5] = {"cherry", "tip_acc", "pop", "ctip", "ctab_acc"};
      for (int q = 0; q < 5; ++q) {
        double t = 0, c = 0;
        for (int w = 0; w < 1024; ++w) t += (double)h[w][q], c += (double)h[w][5 + q];
        fprintf(stderr, "[CT stamps] %-9s %.1f ops per wave, %.0f cycles per op\n", names[q], c / 1024, t / (c > 0 ? c : 1));
      }
      static unsigned long long ph[128][8];
      (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(ct_phase), sizeof(ph));
      double acc[8] = {0};
      for (int b = 0; b < 128; ++b)
        for (int i = 1; i < 8; ++i) acc[i] += (double)(ph[b][i] - ph[b][0]);
      fprintf(stderr, "[CT phases, wave 0 of 128 workgroups, cycles since its start] matrices computed %.0f; drained + barrier %.0f; tables "
              "computed %.0f; drained + barrier %.0f; walk done %.0f; all waves done %.0f; mixed and written %.0f\n", acc[1] / 128,
              acc[2] / 128, acc[3] / 128, acc[4] / 128, acc[5] / 128, acc[6] / 128, acc[7] / 128);
    }
  }
#endif
#ifdef LH_EXP_K1_STAMPS
  {
    static int calls = 0;
    if (++calls == 3) {
      (void)hipDeviceSynchronize();
      static unsigned long long h[1024][4];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(k1_stamps), sizeof(h));
      double wait = 0, cnt = 0, walk = 0;
      int m = 0;
      for (int w = 0; w < 1024; ++w)
        if (h[w][1]) {
          wait += (double)h[w][0];
          cnt += (double)h[w][1];
          walk += (double)h[w][2];
          ++m;
        }
      static unsigned long long ph[128][8];
      (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(k1_phase), sizeof(ph));
      double acc[8] = {0};
      for (int b = 0; b < 128; ++b)
        for (int i = 1; i < 7; ++i) acc[i] += (double)(ph[b][i] - ph[b][0]);
      fprintf(stderr, "[K1 phases, wave 0 of 128 workgroups, cycles since its start] matrix list %.0f; matrices computed %.0f; stores drained + barrier %.0f; "
              "walk done %.0f; all waves done %.0f; rates mixed and written %.0f\n", acc[1] / 128, acc[2] / 128, acc[3] / 128, acc[4] / 128, acc[5] / 128, acc[6] / 128);
      fprintf(stderr, "[K1 stamps] %d waves: walk %.0f cycles, %.1f stamped ops per wave, first P load %.0f cycles per op = %.1f %% of the walk "
              "(two counter reads per stamp included)\n", m, walk / std::max(m, 1), cnt / std::max(m, 1), wait / std::max(cnt, 1.0),
              100.0 * wait / std::max(walk, 1.0));
    }
  }
#endif
  return fused ? 1 : R;
}

}  // namespace lh
