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
//
// What is in this file (round 4):
//  * K0c schedule_rewrite_kernel + prune_wave_ct / prune_wave_asm / prune_body_ct / prune_kernel_ct6,ct5,ct4 -- the
//    cherry-table form: schedules rewritten into walk ops (lh_device.h), cherries as tables in the scratch region, the
//    walk in generated assembly (lh_prune_walk_asm_s2.inc, _s2n.inc for alignments with N; tools/gen_walk_asm.py), tip
//    states as bit planes through the scalar path.  What configs[2] and every other family whose tip tables fit a
//    workgroup's share of LDS run, and any stack depth up to 16 (profiles/r04_k1_programme.txt).
//  * prune_wave / prune_body / prune_kernel_w6,w5,w4 -- the register-stack form with all R rates in one workgroup and
//    the rate mixture formed in the kernel: round 3's default, now behind LH_K1_STACK=1 (a second implementation the
//    parity tests hold against the oracle).  Its schedules are checked by schedule_stack_check_kernel in front of it.
//  * prune_kernel_seg / seg4 -- the register-stack walk for large trees: the tip table built a schedule segment at a
//    time (SegCtx); same check kernel.
//  * launch_prune -- the choice between them, and the debug switches (lh::DebugOptions).
#include <cstdlib>

#include <cstdio>

#include "lh_device.h"

namespace lh {

namespace {

// the K1 form the last launch_prune of this thread chose (tests assert it) and its failure message, if any
thread_local char g_prune_form[64] = "";
thread_local char g_prune_error[256] = "";

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

// Column `st` of a tip branch's P (= P * onehot(st)) from the LDS table tiptab[tip][4][4].  A tip whose state is N (4;
// libpll's 1111, src/PhyloHMM.cpp:368-370) contributes P (1,1,1,1)^T, the row sums of a transition matrix: 1 -- the tip
// drops out of the product.  (libpll adds the four entries up and gets 1 to within 2 ulp; the kernel takes the 1: a
// difference of 1e-16 per N tip, far inside the 1e-10 the log-likelihood is held to.)  Only the kN instantiation, which
// alignments mixing N with bases run, knows the state: it reads a vector of ones kept `ones_off` doubles behind the
// table instead of a column -- one compare and one select on the address, no divergence, no fifth column (128 instead of
// 160 bytes per tip keeps the CU at three workgroups).  Round 3 formed the row sums on the spot, four more LDS reads and
// twelve adds behind a divergent branch, which cost the N-aware walk half its speed (profiles/r04_mixed_n.txt).
template <bool kN>
__device__ __forceinline__ void tip_column(const double* tiptab, int tip, int st, double (&c)[4], int ones_off) {
  const int off = tip * 16 + st * 4;
  const double2* q = reinterpret_cast<const double2*>(tiptab + ((kN && st == 4) ? ones_off : off));
  const double2 q0 = q[0], q1 = q[1];
  c[0] = q0.x, c[1] = q0.y, c[2] = q1.x, c[3] = q1.y;
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
                                           pmat_ptr pm, const double* tiptab, const double* naive_tab, int ones_off,
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
        tip_column<kN>(tiptab, row_y, sa[s], u[s], ones_off);
        tip_column<kN>(tiptab, row_z, sb[s], v[s], ones_off);
      }
    } else {
      const pmat_ptr pb = pm + (size_t)k * 32;
#pragma unroll
      for (int s = 0; s < S; ++s) matvec(pb, a[s], v[s]);
      if (kind == OP_TIP_ACC) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          tip_column<kN>(tiptab, row_y, sa[s], u[s], ones_off);
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


template <bool kN>
__device__ __forceinline__ void tip_column_at(const char* tiptab_bytes, int entry_off, int st, double (&c)[4], int ones_off) {
  const int off = entry_off + st * 32;   // (bytes; ones_off in doubles, as for tip_column)
  const double2* q = reinterpret_cast<const double2*>(tiptab_bytes + ((kN && st == 4) ? ones_off * 8 : off));
  const double2 q0 = q[0], q1 = q[1];
  c[0] = q0.x, c[1] = q0.y, c[2] = q1.x, c[3] = q1.y;
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

// The walk in assembly (S = 2 sites per lane, or 1 for the wave that carries a remainder of up to 64 patterns; text and
// register map: tools/gen_walk_asm.py -> lh_prune_walk_asm_s2.inc, _s1.inc and their N-aware twins _s2n.inc, _s1n.inc).
// Same operations in the same order as the C++ walk below, which stays as its reference (LH_K1_CXX_WALK=1 selects it; the
// results are bit-identical).
// wops: the sample's descriptors in global memory; ctoff: byte offset of the cherry tables in the scratch region pm points
// to; planes: the family's tip states as bit planes (DevFamily::msa_planes: [tip][block of 128 patterns][site set][bit]
// 64-bit masks; patterns past the last one repeat it), of which the wave -- lane l carries patterns 128 block + l + 64 s --
// fetches 32 bytes per tip and op with a scalar load.
// kN: alignments that mix N with bases -- a third plane per site set flags the lanes whose state is N (6 masks per tip and
// block instead of 4), and such a tip reads the four ones at `ones` (LDS) instead of a column; lh_prune_walk_asm_s2n.inc.
template <int kDepth, bool kN, int S>
__device__ __forceinline__ void prune_wave_asm(int block128, const uint64_t* __restrict__ planes, int n_blocks, int n_w,
                                               const WalkOp* __restrict__ wops, pmat_ptr pm, unsigned ctoff,
                                               const double* tiptab, const double* naive_tab, const double* ones,
                                               const double* __restrict__ p4, double (&lik)[S][5], int (&scl)[S]) {
  static_assert(S == 1 || S == 2, "one or two sites per lane");
  __attribute__((aligned(16))) double out_mem[4 * S + 2];                                   // a[S][4], then the packed scaler counts
  __attribute__((aligned(16))) double deep_mem[(kDepth > 2 ? kDepth - 2 : 1) * 4 * S];     // stack slots 2.. : [slot][site][4] (0 and 1: registers)
  // tip t (MSA row t - 1) of this wave's block at planes_w + t * pstride bytes; a wave wholly past the last pattern (its
  // results are not stored) reads the last block.  Every wave starts a block of 128 patterns: the one-site wave (S = 1: a
  // remainder of up to 64 patterns behind the tile's two-site waves) reads the block's first site set and ignores the second.
  constexpr int kMasks = kN ? 6 : 4;  // 64-bit masks per (tip, block)
  const int blk = min(block128, n_blocks - 1);
  const uint64_t* planes_w = planes + ((ptrdiff_t)blk - (ptrdiff_t)n_blocks) * kMasks;
  const unsigned pstride = (unsigned)n_blocks * (unsigned)(kMasks * 8);
  const unsigned tip_lds = (unsigned)(size_t)(lds_ptr)tiptab;
  if constexpr (kN && S == 2) {
    const unsigned ones_lds = (unsigned)(size_t)(lds_ptr)ones;
    asm volatile(
#include "lh_prune_walk_asm_s2n.inc"
        :
        : [nw] "s"(n_w), [wops] "s"(wops), [pm] "s"(pm), [ctoff] "s"(ctoff), [planes] "s"(planes_w), [pstride] "s"(pstride),
          [tip] "s"(tip_lds), [ones] "v"(ones_lds), [out] "v"((private_ptr)out_mem), [deep] "v"((private_ptr)deep_mem)
        : "memory", "vcc", "scc",
#include "lh_prune_walk_clobbers_s2.inc"
    );
  } else if constexpr (S == 2) {
    asm volatile(
#include "lh_prune_walk_asm_s2.inc"
        :
        : [nw] "s"(n_w), [wops] "s"(wops), [pm] "s"(pm), [ctoff] "s"(ctoff), [planes] "s"(planes_w), [pstride] "s"(pstride),
          [tip] "s"(tip_lds), [out] "v"((private_ptr)out_mem), [deep] "v"((private_ptr)deep_mem)
        : "memory", "vcc", "scc",
#include "lh_prune_walk_clobbers_s2.inc"
    );
  } else if constexpr (kN) {
    const unsigned ones_lds = (unsigned)(size_t)(lds_ptr)ones;
    asm volatile(
#include "lh_prune_walk_asm_s1n.inc"
        :
        : [nw] "s"(n_w), [wops] "s"(wops), [pm] "s"(pm), [ctoff] "s"(ctoff), [planes] "s"(planes_w), [pstride] "s"(pstride),
          [tip] "s"(tip_lds), [ones] "v"(ones_lds), [out] "v"((private_ptr)out_mem), [deep] "v"((private_ptr)deep_mem)
        : "memory", "vcc", "scc",
#include "lh_prune_walk_clobbers_s1.inc"
    );
  } else {
    asm volatile(
#include "lh_prune_walk_asm_s1.inc"
        :
        : [nw] "s"(n_w), [wops] "s"(wops), [pm] "s"(pm), [ctoff] "s"(ctoff), [planes] "s"(planes_w), [pstride] "s"(pstride),
          [tip] "s"(tip_lds), [out] "v"((private_ptr)out_mem), [deep] "v"((private_ptr)deep_mem)
        : "memory", "vcc", "scc",
#include "lh_prune_walk_clobbers_s1.inc"
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

template <int kDepth, int S, bool kN>
__device__ __forceinline__ void prune_wave_ct(int site0, int site_end, const uint8_t* __restrict__ msa, int L, int n_w,
                                              const WalkOp* desc /* LDS */, pmat_ptr pm, const double* tiptab,
                                              const double* ctab, const double* naive_tab, int ones_off,
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
  for (int k = 0; k < n_w; ++k) {
    const int kind = d0.x & 7;
    const int tip_a = (unsigned)d0.x >> 16, tip_b = d0.y & 0xffff, tip_c = (unsigned)d0.y >> 16;
    // the op's matrix, if it has one, is the next in sequence: x = P a -- its own mat-vec (tip-into-accumulator,
    // table-into-accumulator, pop) or the one a push applies to the subtree it sets aside
    double x[S][4];
    if (d0.x & WOP_MATRIX) {
      const pmat_ptr pb = reinterpret_cast<pmat_ptr>(reinterpret_cast<const char __attribute__((address_space(4)))*>(pm) + pm_off);
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
        for (int s = 0; s < S; ++s) tip_column_at<kN>(tipb, tip_a * 128, sa[s], u[s], ones_off);
      } else if (kind == W_CTAB_ACC) {
#pragma unroll
        for (int s = 0; s < S; ++s) table_entry_at<kN>(ctabb, tab_off, sa[s], sb[s], u[s]);
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
          table_entry_at<kN>(ctabb, tab_off, sa[s], sb[s], u[s]);
          tip_column_at<kN>(tipb, tip_c * 128, sc[s], v[s], ones_off);
        }
        tab_off += kTabBytes;
      } else {  // W_CHERRY
#pragma unroll
        for (int s = 0; s < S; ++s) {
          tip_column_at<kN>(tipb, tip_a * 128, sa[s], u[s], ones_off);
          tip_column_at<kN>(tipb, tip_b * 128, sb[s], v[s], ones_off);
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
__device__ __forceinline__ void prune_body(int sample, int n2, int tile, int R, int wpr, const uint8_t* __restrict__ msa, int L,
                                           int T, int n_ops, const int32_t* __restrict__ ops,
                                           const int4* __restrict__ hdr, int32_t* err_flag,
                                           const double* __restrict__ brlen, const double* __restrict__ rates,
                                           const double* __restrict__ eig, double* pmat_w, size_t rate_stride,
                                           const double* __restrict__ pi,
                                           double* __restrict__ site_lik, int32_t* __restrict__ site_scal) {
  extern __shared__ double2 smem2[];
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
  const size_t pm_off = ((size_t)sample * R + rate) * rate_stride;
  // LDS tip table [T][4][4] (per rate when fused); large trees (kSeg): the segment slots [2 kSegOps][4][4]
  // followed by the naive tip's entry
  double* tiptab = reinterpret_cast<double*>(smem2) + (kFused ? (size_t)rate * T * 16 : 0);
  const double* naive_tab = kSeg ? tiptab + 2 * kSegOps * 16 : tiptab;
  // kN: four ones behind the table(s) -- what a tip whose state is N contributes (tip_column)
  double* ones = reinterpret_cast<double*>(smem2) + (kSeg ? (size_t)(2 * kSegOps + 1) * 16 : (size_t)(kFused ? R : 1) * T * 16);
  const int ones_off = (int)(ones - tiptab);
  if (kN && tid < 4) ones[tid] = 1.0;
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
      uint16_t* mat_list = reinterpret_cast<uint16_t*>(reinterpret_cast<double*>(smem2) + (size_t)R * T * 16 + (kN ? 4 : 0));
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __threadfence_block();
  __syncthreads();

  // P-matrices in schedule order (addresses depend on the op number only), readable from here on
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
      prune_wave<kDepth, 2, kN, kSeg>(site0, site_end, msa, L, n_ops, op_ptr, pm, tiptab, naive_tab, ones_off, p4, lik, scl, seg);
  } else {
    site0 = tile0 + n2 * 128 + (wave - n2) * 64 + lane;
    n_own = 1;
    double lik1[1][5];
    int scl1[1];
    prune_wave<kDepth, 1, kN, kSeg>(site0, site_end, msa, L, n_ops, op_ptr, pm, tiptab, naive_tab, ones_off, p4, lik1, scl1, seg);
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
    __syncthreads();
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
  }
}

// x = P * a with P row-major in vector registers (the prologue's table building)
__device__ __forceinline__ void matvec_v(const double (&p)[16], const double (&a)[4], double (&x)[4]) {
  x[0] = fma(p[3], a[3], fma(p[2], a[2], fma(p[1], a[1], p[0] * a[0])));
  x[1] = fma(p[7], a[3], fma(p[6], a[2], fma(p[5], a[1], p[4] * a[0])));
  x[2] = fma(p[11], a[3], fma(p[10], a[2], fma(p[9], a[1], p[8] * a[0])));
  x[3] = fma(p[15], a[3], fma(p[14], a[2], fma(p[13], a[1], p[12] * a[0])));
}

// K0c: checks a sample's schedule (kinds, node ranges, stack discipline: everything K1 indexes with) and rewrites it into
// the walk the cherry-table kernels execute: the walk descriptors (WalkOp above; lh_device.h), the branch length of every
// P-matrix the walk consumes (in that order) and the list of cherry tables.  A malformed schedule gets hdr.w = 1 (K1 then
// leaves NaN) and sets *err_flag, which lh_family_status reports: device-resident schedules are not trusted.
//
// Round 4: one WAVE per sample, a LANE per op (64 ops a round).  Rounds 2-3 walked the ops serially (a thread, or the scalar
// unit of a wave, per sample: 0.19-0.33 ms per 49 152 samples of a 101-tip tree, 0.7-0.9 ms per 6144 of a 501-tip one); but
// nothing in the rewrite is sequential beyond running counts:
//  * whether a cherry folds into its consumer is decided by the op and its successor alone: cherry (pushing) + pop of the
//    same slot -> W_CTAB_ACC (the push and the pop cancel), cherry + tip-into-accumulator -> W_CTIP; the successor is then
//    consumed;
//  * an op's place in the walk, its matrix's and its table's place in their lists, and the stack depths of the schedule
//    and of the walk before it are exclusive prefix sums over the ops (wave scans, carried from round to round);
//  * a pushed subtree's matrix is applied AT THE PUSH but its branch is named by the matching pop: the pop finds the
//    matrix slot of the last earlier push of its level by a running maximum per level (at most max_depth <= 16 levels).
// Dynamic LDS: [tabs_stride] int table nodes | [16] int last push per level.
__global__ void __launch_bounds__(64) schedule_rewrite_kernel(int T, int max_depth, int tabs_stride, int use_tables,
                                                              const int32_t* __restrict__ ops, const double* __restrict__ brlen,
                                                              int2* __restrict__ wops, double* __restrict__ wlen,
                                                              int4* __restrict__ tabs, int4* __restrict__ hdr, int32_t* err_flag) {
  extern __shared__ int k0c_lds[];
  int* tnode = k0c_lds;                 // [tabs_stride] node of table c's cherry (its branch: wlen[n_mat + c])
  int* lastpush = k0c_lds + tabs_stride;  // [16] matrix slot of the last push at each walk level
  const int lane = threadIdx.x, smp = blockIdx.x;
  const int n_ops = T - 2, nodes = 2 * T - 2;
  const int4* __restrict__ og = reinterpret_cast<const int4*>(ops) + (size_t)smp * n_ops;
  int2* wo = wops + (size_t)smp * n_ops;
  int4* tl = tabs + (size_t)smp * tabs_stride;
  double* ml = wlen + (size_t)smp * n_ops;
  const double* __restrict__ bl = brlen + (size_t)smp * nodes;
  if (lane < 16) lastpush[lane] = -1;
  __syncthreads();
  auto tip_ok = [&](int v) { return v >= 1 && v < T; };
  auto inner_ok = [&](int v) { return v >= T && v < nodes; };
  auto plain = [](int x) { return x >= 0 && (x & 0xf0) == 0; };  // no push flag, no unknown bits
  // does cherry `c` (op j) fold into its successor `nx`?  1: W_CTAB_ACC, 2: W_CTIP, 0: no
  auto fold = [&](int j, const int4& c, const int4& nx) -> int {
    if (!use_tables || j + 1 >= n_ops || (c.x & 15) != OP_CHERRY || c.x < 0 || (c.x & 0xe0)) return 0;
    const bool push = (c.x & OP_PUSH_FLAG) != 0;
    const int nk = nx.x & 15;
    if (push && nk == OP_POP_ACC && plain(nx.x) && nx.w == c.w && inner_ok(nx.y) && inner_ok(nx.z)) return 1;
    if (nk == OP_TIP_ACC && plain(nx.x) && tip_ok(nx.y) && inner_ok(nx.z)) return 2;
    return 0;
  };
  // inclusive wave scan (sum)
  auto scan = [&](int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(v, d);
      if (lane >= d) v += up;
    }
    return v;
  };
  int c_bdepth = 0, c_wi = 0, c_wdepth = 0, c_mi = 0, c_ti = 0;  // running counts before this round (wave-uniform)
  bool bad = false;
  const int4 none = make_int4(15, 0, 0, 0);
  for (int k0 = 0; k0 < n_ops; k0 += 64) {
    const int k = k0 + lane;
    const bool in = k < n_ops;
    const int4 op = in ? og[k] : none;
    const int4 nx = (in && k + 1 < n_ops) ? og[k + 1] : none;
    const int4 pv = (in && k > 0) ? og[k - 1] : none;
    const int kind = op.x & 15;
    const bool push = (op.x & OP_PUSH_FLAG) != 0;
    const int f = in ? fold(k, op, nx) : 0;                         // this cherry folds into op k + 1
    const bool consumed = in && k > 0 && fold(k - 1, pv, op) != 0;  // this op was folded into the cherry before it
    // every field the kernels index with, and the stack discipline of the schedule as written
    bool ok = op.x >= 0 && (op.x & 0xe0) == 0 && kind <= OP_POP_ACC;
    if (kind == OP_CHERRY) ok = ok && tip_ok(op.y) && tip_ok(op.z) && push == (k != 0);
    if (kind == OP_TIP_ACC) ok = ok && !push && k > 0 && tip_ok(op.y) && inner_ok(op.z);
    if (kind == OP_POP_ACC) ok = ok && !push && k > 0 && inner_ok(op.y) && inner_ok(op.z);
    const int db = !in || !ok ? 0 : push ? 1 : kind == OP_POP_ACC ? -1 : 0;
    const int s_db = scan(db);
    const int bdepth = c_bdepth + s_db - db;                        // pending siblings of the schedule before this op
    if (in && push && (op.w != bdepth || bdepth >= max_depth)) ok = false;
    if (in && kind == OP_POP_ACC && op.w != bdepth - 1) ok = false;
    // the walk: ops that are not consumed, their pushes and pops (a W_CTAB_ACC fold cancels a push with its pop)
    const bool emit = in && ok && !consumed;
    const bool wpush = emit && push && f != 1;
    const bool wpop = emit && kind == OP_POP_ACC;
    const bool hasmat = wpush || (emit && (f == 1 || kind == OP_TIP_ACC || kind == OP_POP_ACC));
    const bool hastab = emit && f != 0;
    const int e1 = emit ? 1 : 0, dw = wpush ? 1 : wpop ? -1 : 0, m1 = hasmat ? 1 : 0, t1 = hastab ? 1 : 0;
    const int s_e1 = scan(e1), s_dw = scan(dw), s_m1 = scan(m1), s_t1 = scan(t1);
    const int wi = c_wi + s_e1 - e1, wdepth = c_wdepth + s_dw - dw, mi = c_mi + s_m1 - m1, ti = c_ti + s_t1 - t1;
    if (wpush && wdepth >= 16) ok = false;
    if (wpop && wdepth < 1) ok = false;
    if (hastab && ti >= tabs_stride) ok = false;
    if (hasmat && mi >= n_ops) ok = false;
    if (in && !ok) bad = true;
    // the matrix slot of the last push of every walk level, for the pops of this round (pushes of the round included)
    int mypush = -1;  // (for a pop) where its sibling's matrix sits in the walk's list
    for (int d = 0; d < max_depth; ++d) {
      int v = (wpush && ok && wdepth == d) ? mi : -1;
#pragma unroll
      for (int q = 1; q < 64; q <<= 1) {
        const int up = __shfl_up(v, q);
        if (lane >= q) v = max(v, up);
      }
      v = max(v, lastpush[d]);
      if (wpop && wdepth - 1 == d) mypush = v;
      const int last = __shfl(v, 63);
      __syncthreads();  // (every lane has read lastpush[d])
      if (lane == 0) lastpush[d] = last;
      __syncthreads();
    }
    if (wpop && mypush < 0) {
      ok = false;
      bad = true;
    }
    if (emit && ok) {
      int2 w;
      if (f == 1) {
        w = make_int2(W_CTAB_ACC | WOP_MATRIX | WOP_HAS_B | (op.y << 16), op.z | (1 << 16));
        ml[mi] = bl[nx.y];  // the first subtree stays in the accumulator: its branch matrix
      } else if (kind == OP_CHERRY) {
        const int pushbits = wpush ? WOP_MATRIX | ((wdepth + 1) << WOP_PUSH_SHIFT) : 0;  // (the push's matrix: named by its pop)
        w = f == 2 ? make_int2(W_CTIP | pushbits | WOP_HAS_B | WOP_HAS_C | (op.y << 16), op.z | (nx.y << 16))
                   : make_int2(W_CHERRY | pushbits | WOP_HAS_B | (op.y << 16), op.z | (1 << 16));
      } else if (kind == OP_TIP_ACC) {
        w = make_int2(W_TIP_ACC | WOP_MATRIX | (op.y << 16), 1 | (1 << 16));
        ml[mi] = bl[op.z];
      } else {
        w = make_int2(W_POP | WOP_MATRIX | ((wdepth - 1) << WOP_POP_SHIFT) | (1 << 16), 1 | (1 << 16));
        ml[mypush] = bl[op.y];  // the popped child: its matrix was applied at the push
        ml[mi] = bl[op.z];      // the child whose CLV is in the accumulator
      }
      wo[wi] = w;
      if (hastab) {
        tl[ti] = make_int4(op.y, op.z, nx.z, 0);
        tnode[ti] = nx.z;
      }
    }
    c_bdepth += __shfl(s_db, 63);
    c_wi += __shfl(s_e1, 63);
    c_wdepth += __shfl(s_dw, 63);
    c_mi += __shfl(s_m1, 63);
    c_ti += __shfl(s_t1, 63);
  }
  if (c_bdepth != 0 || c_wdepth != 0 || c_mi + c_ti > T - 3) bad = true;  // (the scratch area holds T - 3 matrices per rate)
  const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
  if (lane == 0) {
    if (any_bad) {
      hdr[smp] = make_int4(0, 0, 0, 1);
      atomicOr(err_flag, 1);
    } else {
      hdr[smp] = make_int4(c_wi, c_mi, c_ti, 0);
    }
  }
  if (any_bad) return;
  // the cherry branches' lengths behind the walk's (table c's matrix is slot n_mat + c of the prologue's list)
  __syncthreads();
  for (int c = lane; c < c_ti; c += 64) ml[c_mi + c] = bl[tnode[c]];
}

// The register-stack kernels walk the schedule as lh_schedule_tree wrote it.  Device-resident schedules are not trusted:
// one WAVE per sample checks, 64 ops at a time,
//  * every field the kernel indexes with (tips: MSA rows and branch lengths; nodes: branch lengths; slots: registers);
//  * the stack discipline, by a prefix sum over +1 per push and -1 per pop: a push goes to slot = the running depth, a pop
//    takes slot depth - 1, and the stack is empty at the end -- a schedule that pops the wrong slot, pops an empty one or
//    pushes over a live one would otherwise compute a finite, wrong likelihood without touching foreign memory;
//  * check_ranks (the kernels with all rates in one workgroup place their P-matrices by the RANK each op carries, the
//    running count of inner-branch matrices lh_schedule_tree leaves in the descriptor): by a second prefix sum, that each
//    op's rank is that running count; the total is T - 3 for any schedule of a binary tree.
// hdr[smp] = (0, 0, 0, verdict) for every sample; a rejected sample's results are NaN and *err_flag is raised.
// (Round 3 first made the field and rank checks inside K1's prologue: their scalars cost the walk registers -- K1 5.63 ->
// 5.93 ms on one box -- so they are a kernel of their own: 49 152 samples of a 101-tip tree in ~20 us.)
__global__ void __launch_bounds__(64) schedule_stack_check_kernel(int T, int slots, int check_ranks,
                                                                  const int32_t* __restrict__ ops, int4* __restrict__ hdr,
                                                                  int32_t* err_flag) {
  const int smp = blockIdx.x, lane = threadIdx.x;
  const int n_ops = T - 2, nodes = 2 * T - 2;
  const int4* __restrict__ o = reinterpret_cast<const int4*>(ops) + (size_t)smp * n_ops;
  int carry = 0, depth0 = 0;  // matrices / pending siblings before this round of 64 ops
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
    const int dl = !in || !ok ? 0 : push ? 1 : kind == OP_POP_ACC ? -1 : 0;
    int incl = m, dincl = dl;  // inclusive prefix sums over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d), dup = __shfl_up(dincl, d);
      if (lane >= d) incl += up, dincl += dup;
    }
    const int depth = depth0 + dincl - dl;  // pending siblings before this op
    if (in && push && op.w != depth) ok = false;
    if (in && kind == OP_POP_ACC && op.w != depth - 1) ok = false;
    if (check_ranks && in && kind != OP_CHERRY && rank != carry + incl - m) ok = false;
    if (in && !ok) bad = true;
    carry += __shfl(incl, 63);
    depth0 += __shfl(dincl, 63);
  }
  if (carry != T - 3 || depth0 != 0) bad = true;
  const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
  if (lane == 0) {
    hdr[smp] = make_int4(0, 0, 0, any_bad ? 1 : 0);
    if (any_bad) atomicOr(err_flag, 1);
  }
}

// The workgroup of the cherry-table form.  Block layout as prune_body (n2 two-site waves + n1 one-site waves per rate;
// kFused: all R rates of the sample in one workgroup, mixed at the end).  scratch: this (sample, rate)'s region of
// rate_stride doubles: [n_mat + n_tab][16] P-matrices | [n_tab][E][4] tables.
template <int kDepth, bool kN, bool kFused, bool kAsm = false>
__device__ __forceinline__ void prune_body_ct(int n2, int tile, int R, int wpr, const uint8_t* __restrict__ msa,
                                              const uint64_t* __restrict__ planes, int L,
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
  const int n_w = __builtin_amdgcn_readfirstlane(h.x);
  const int n_mat = __builtin_amdgcn_readfirstlane(h.y), n_tab = __builtin_amdgcn_readfirstlane(h.z);
  const bool malformed = h.w != 0;
  double* pw = pmat_w + ((size_t)sample * R + rate) * rate_stride;
  double* ctab = pw + (size_t)(T - 3 > 0 ? T - 3 : 0) * 16;
  constexpr int kS = 2;  // sites per lane of the multi-site waves
  double* tiptab = reinterpret_cast<double*>(smem2) + (kFused ? (size_t)rate * T * 16 : 0);  // LDS tip table [T][4][4]
  const double* naive_tab = tiptab;
  // kN: four ones behind the table(s) -- what a tip whose state is N contributes (tip_column)
  double* ones = reinterpret_cast<double*>(smem2) + (size_t)(kFused ? R : 1) * T * 16;
  const int ones_off = (int)(ones - tiptab);
  if (kN && tid < 4) ones[tid] = 1.0;
  const int4* __restrict__ tl = tabs + (size_t)sample * tabs_stride;
  // the walk descriptors go to LDS behind the tip tables (one copy per workgroup) for the waves that run the C++ walk
  WalkOp* desc = reinterpret_cast<WalkOp*>(reinterpret_cast<double*>(smem2) + (size_t)(kFused ? R : 1) * T * 16 + (kN ? 4 : 0));
  // (the assembly walk fetches its descriptors from global memory with scalar loads: this copy also brings their
  // lines into L2 before the walk asks for them -- without it every eighth op waited for HBM)
  for (int i = tid; i < n_w; i += blockDim.x) desc[i] = wops[(size_t)sample * n_ops + i];

  // Prologue, first half: the P-matrices of this (sample, rate), one thread per matrix (K0c left every matrix's
  // branch length in the order they are stored): the walk's inner-branch matrices to the scratch area, the tip
  // branches' into the LDS tip table (column by column, as the walk gathers them), the cherry branches' for the
  // tables.  A table's four base rows are built by the four lanes of a QUAD, and the quad's first lane computes the
  // cherry branch's matrix here and keeps it in registers: it reaches the other three through DPP in the second half,
  // not through memory (tables beyond nthr / 4 take the path through the scratch area).  With N in the alignment a
  // table has a fifth row and column: the quad's lanes share the fifth row's entries (round 4; such tables all took
  // the scratch path before).
  const int n_q = min(n_tab, nthr >> 2);
  const bool pc_lane = (rtid & 3) == 0 && (rtid >> 2) < n_q;
  int4 tcell = make_int4(1, 1, 0, 0);
  if ((rtid >> 2) < n_q) tcell = tl[rtid >> 2];  // the quad's table: requested now, used after the barrier
  double pcq[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) pcq[i] = 0.0;
  {
    const double* __restrict__ e = eig + (size_t)sample * 36;
    const double* __restrict__ bl = brlen + (size_t)sample * (2 * (size_t)T - 2);
    const double* __restrict__ wl = wlen + (size_t)sample * n_ops;
    // One matrix per thread and round.  Round 0: a quad's first lane takes its cherry matrix, every other thread the
    // item of its rank in the common list of its rate (walk matrices | cherry matrices n_q.. (scratch path) | tips).
    // The rest of the lists -- with all rates in one workgroup -- is POOLED over the workgroup's threads (round 4: 71
    // items per rate on configs[2], which kept both waves of every rate busy for a second round; pooled, 284 items keep
    // five of the eight waves busy and three skip the round).  The pooled part runs FIRST: the quads' matrices (sixteen
    // registers each) are then not live across another compute_pmatrix.
    const int n_rest = n_tab - n_q;
    const int n_list = n_mat + n_rest + T;
    const int round1 = nthr - n_q;  // list items of a rate taken in round 0
    auto item = [&](bool is_pc, int it, int rr, int quad) {
      double P[4][4];
      const bool in_list = !is_pc && it < n_list;
      const bool inner = in_list && it < n_mat + n_rest;
      const int slot = it < n_mat ? it : it + n_q;  // matrix slot in the scratch area ([n_mat + c] for table c)
      const int j = it - (n_mat + n_rest);          // tip
      double t = 0.0;
      if (is_pc) t = wl[n_mat + quad];
      else if (inner) t = wl[slot];
      else if (in_list) t = bl[j];
      compute_pmatrix(e, t * rates[(size_t)sample * R + rr], P);
      if (is_pc) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) pcq[i * 4 + q] = P[i][q];
      } else if (inner) {
        double* o = pmat_w + ((size_t)sample * R + rr) * rate_stride + (size_t)slot * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) o[i * 4 + q] = P[i][q];
      } else if (in_list) {
        double* o = reinterpret_cast<double*>(smem2) + (kFused ? (size_t)rr * T * 16 : 0) + j * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int st = 0; st < 4; ++st) o[st * 4 + i] = P[i][st];
      }
    };
    const int rem = n_list - round1;  // items of a rate beyond round 0
    if (rem > 0) {
      if constexpr (kFused) {
        const float inv_rem = 1.0f / (float)rem;
        for (int c = tid; c < R * rem; c += blockDim.x) {
          int rr = min((int)(((float)c + 0.5f) * inv_rem), R - 1);  // c / rem
          rr -= rr * rem > c ? 1 : 0;
          rr += (rr + 1) * rem <= c ? 1 : 0;
          item(false, round1 + c - rr * rem, rr, 0);
        }
      } else {
        for (int it = round1 + rtid; it < n_list; it += nthr) item(false, it, rate, 0);
      }
    }
    item(pc_lane, rtid - min(n_q, (rtid + 3) >> 2), rate, rtid >> 2);
  }
  // the tip tables are complete (LDS); the scratch-area stores need to have landed only if a table goes that way
  if (n_tab > n_q) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // Second half: the cherry tables, one thread per (table, state of the first tip): P_c (tipcol_y o tipcol_z) for
  // every state of the second tip -- the very operations the unfused walk performs per lane, done once per state pair.
  auto build_rows = [&](const double (&pc)[16], int c, int sy, int ty, int tz) {
    double py[4];
    tip_column<kN>(tiptab, ty, sy, py, ones_off);
    double2* o = reinterpret_cast<double2*>(ctab) + ((size_t)c * E + (size_t)sy * SY) * 2;
#pragma unroll
    for (int sz = 0; sz < SY; ++sz) {
      double pz[4], pr[4], x[4];
      tip_column<kN>(tiptab, tz, sz, pz, ones_off);
#pragma unroll
      for (int i = 0; i < 4; ++i) pr[i] = py[i] * pz[i];
      matvec_v(pc, pr, x);
      o[2 * sz] = make_double2(x[0], x[1]);
      o[2 * sz + 1] = make_double2(x[2], x[3]);
    }
  };
  {
    // every lane of the wave takes part in the DPP moves (quad_perm [0,0,0,0]: the quad's first lane to all four)
    double pc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lo = __builtin_amdgcn_mov_dpp(__double2loint(pcq[i]), 0, 0xf, 0xf, true);
      const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(pcq[i]), 0, 0xf, 0xf, true);
      pc[i] = __hiloint2double(hi, lo);
    }
    if ((rtid >> 2) < n_q) {
      build_rows(pc, rtid >> 2, rtid & 3, tcell.x, tcell.y);
      if constexpr (kN) {
        // the fifth row (first tip N: a vector of ones): its entries go to the quad's four lanes one each, the corner to
        // the last lane -- the very operations build_rows performs for sy = 4
        const int sz = rtid & 3;
        double2* o = reinterpret_cast<double2*>(ctab) + ((size_t)(rtid >> 2) * E + 4 * SY) * 2;
        for (int k = sz; k < SY; k += 4) {
          double pz[4], pr[4], x[4];
          tip_column<kN>(tiptab, tcell.y, k, pz, ones_off);
#pragma unroll
          for (int i = 0; i < 4; ++i) pr[i] = 1.0 * pz[i];
          matvec_v(pc, pr, x);
          o[2 * k] = make_double2(x[0], x[1]);
          o[2 * k + 1] = make_double2(x[2], x[3]);
        }
      }
    }
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __threadfence_block();
  __syncthreads();

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
    if constexpr (kAsm)
      prune_wave_asm<kDepth, kN, kS>((tile0 + wave * (64 * kS)) >> 7, planes, (L + 127) >> 7, n_w, wops + (size_t)sample * n_ops, pm,
                                     (unsigned)((T - 3 > 0 ? T - 3 : 0) * 128), tiptab, naive_tab, ones, p4, lik, scl);
    else
      prune_wave_ct<kDepth, kS, kN>(site0, site_end, msa, L, n_w, desc, pm, tiptab, ctab, naive_tab, ones_off, p4, lik, scl);
  } else {
    site0 = tile0 + n2 * (64 * kS) + (wave - n2) * 64 + lane;
    n_own = 1;
    double lik1[1][5];
    int scl1[1];
    if constexpr (kAsm)
      prune_wave_asm<kDepth, kN, 1>((tile0 + n2 * (64 * kS)) >> 7, planes, (L + 127) >> 7, n_w,   // (at most one such wave)
                                    wops + (size_t)sample * n_ops, pm, (unsigned)((T - 3 > 0 ? T - 3 : 0) * 128), tiptab, naive_tab,
                                    ones, p4, lik1, scl1);
    else
      prune_wave_ct<kDepth, 1, kN>(site0, site_end, msa, L, n_w, desc, pm, tiptab, ctab, naive_tab, ones_off, p4, lik1, scl1);
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
    __syncthreads();
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
    // a thread per pattern, its five naive states together (round 4: a thread per (state, pattern) cost an integer division
    // by the tile size per item and read the R scaler counts five times over -- ~250 vector instructions per wave against ~40)
    for (int p = tid; p < n_tile; p += blockDim.x) {
      int smin = 0x7fffffff;
      for (int r = 0; r < R; ++r) smin = min(smin, SC[r * pad + p]);
      double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
      for (int r = 0; r < R; ++r) {
        const int d = SC[r * pad + p] - smin;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          double v = X[((size_t)r * 5 + b) * pad + p];
          for (int q = 0; q < d && v != 0.0; ++q) v *= kScaleThreshold;
          acc[b] += w * v;
        }
      }
#pragma unroll
      for (int b = 0; b < 5; ++b) lik_out[(size_t)b * L + tile0 + p] = acc[b];
      scal_out[tile0 + p] = smin;
    }
  }
}

#define LH_PRUNE_CT_PARAMS                                                                                          \
  int n2, int tile, int R, int wpr, const uint8_t *__restrict__ msa, const uint64_t *__restrict__ planes, int L,    \
      int T, const int2 *__restrict__ wops,                                                                         \
      const double *__restrict__ wlen, const int4 *__restrict__ tabs, int tabs_stride,                              \
      const int4 *__restrict__ hdr, const double *__restrict__ brlen, const double *__restrict__ rates,            \
      const double *__restrict__ eig, double *pmat_w, size_t rate_stride, const double *__restrict__ pi,           \
      double *__restrict__ site_lik, int32_t *__restrict__ site_scal
#define LH_PRUNE_CT_ARGS \
  n2, tile, R, wpr, msa, planes, L, T, wops, wlen, tabs, tabs_stride, hdr, brlen, rates, eig, pmat_w, rate_stride, pi, site_lik, \
      site_scal
#define LH_PRUNE_CT_KERNEL(NAME, WAVES)                                                              \
  template <int kDepth, bool kN, bool kFused, bool kAsm>                                             \
  __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) NAME(   \
      LH_PRUNE_CT_PARAMS) {                                                                          \
    prune_body_ct<kDepth, kN, kFused, kAsm>(LH_PRUNE_CT_ARGS);                                       \
  }
LH_PRUNE_CT_KERNEL(prune_kernel_ct6, 6)
LH_PRUNE_CT_KERNEL(prune_kernel_ct5, 5)
LH_PRUNE_CT_KERNEL(prune_kernel_ct4, 4)
#undef LH_PRUNE_CT_KERNEL
// pmat_w: the scratch area (see prune_body); written in the prologue, read back after the barrier.
#define LH_PRUNE_PARAMS                                                                                     \
  int n_samples, int n2, int tile, int R, int wpr, const uint8_t *__restrict__ msa, int L, int T, int n_ops, \
      const int32_t *__restrict__ ops, const int4 *__restrict__ hdr, int32_t *err_flag,                     \
      const double *__restrict__ brlen, const double *__restrict__ rates, const double *__restrict__ eig,  \
      double *pmat_w, size_t rate_stride, const double *__restrict__ pi,                                   \
      double *__restrict__ site_lik, int32_t *__restrict__ site_scal
#define LH_PRUNE_ARGS \
  n2, tile, R, wpr, msa, L, T, n_ops, ops, hdr, err_flag, brlen, rates, eig, pmat_w, rate_stride, pi, site_lik, site_scal
// One workgroup per sample.  (A scratch slot per RESIDENT workgroup, each working through several samples, was measured
// 11 % slower in round 3 -- profiles/r03_k1_persistent_slots.txt -- and left the source with round 4.)
#define LH_PRUNE_SAMPLES(BODY) BODY((int)blockIdx.z, LH_PRUNE_ARGS);

// Shallow stacks (depth <= 4, any tree up to a few hundred tips), all rates in one workgroup: two sites per lane.  The
// walk needs ~100 VGPRs; resident waves matter more to it than a few spilled registers, as long as the LDS tip
// tables of that many workgroups fit a CU.  Three register budgets are therefore built -- 6 waves per
// SIMD (80 VGPRs), 5 (96) and 4 (128, no spills) -- and the launcher takes the tightest one whose
// occupancy the tip tables allow: configs[2] runs 6 waves per SIMD, 5 % faster than 5 and 19 % faster than 4.
#define LH_PRUNE_KERNEL(NAME, WAVES)                                                                 \
  template <int kDepth, bool kN>                                                                     \
  __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) NAME(   \
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

PruneWsSizes prune_ws_sizes(int T, bool mixed_n) {
  PruneWsSizes z;
  z.tabs_per_sample = (size_t)std::max((T - 1) / 2, 1);
  const size_t e = mixed_n ? 25 : 16;
  // the cherry-table form's matrices and tables, and never less than the register-stack kernels' [T-2][2][16]
  z.scratch_doubles_per_rate = std::max((size_t)std::max(T - 3, 0) * 16 + z.tabs_per_sample * e * 4, (size_t)std::max(T - 2, 1) * 32);
  return z;
}

// Returns the number of rate planes it left in site_lik / site_scal: R, or 1 when the workgroups mixed
// the rate categories themselves (the count K2a must then be run with); -1 when the launch failed (prune_last_error).
int launch_prune(const DevFamily& fam, int n, int R, int T, int max_depth, const int32_t* ops,
                 const double* brlen, const double* rates, const double* eig, const PruneWs& ws, const double* pi,
                 double* site_lik, int32_t* site_scal, hipStream_t stream, bool allow_fused) {
  const int L = fam.n_prune;  // distinct alignment columns; identical ones are pruned once
  const PruneWsSizes sizes = prune_ws_sizes(T, fam.msa_mixed_n != 0);
  if (L == 0) {  // nothing but all-N padding (K2a reads no plane at all): the schedules still get checked
    hipLaunchKernelGGL(schedule_stack_check_kernel, dim3(n), dim3(64), 0, stream, T, std::max(max_depth, 1), 0, ops, ws.hdr, ws.err_flag);
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
  // (several tiles: whole blocks of 128 patterns each, the unit of the assembly walk's state planes)
  const int tile = tiles == 1 ? L : (((L + tiles - 1) / tiles + 127) & ~127);
  int n2 = tile / 128, n1;  // two-site waves, one-site waves per rate
  {
    const int rem = tile - 128 * n2;
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
  // register-stack form keeps its matrix list, the cherry-table form its copy of the walk descriptors
  const size_t pad = (size_t)n2 * 128 + (size_t)n1 * 64;
  const size_t tail_bytes = (((size_t)std::max(T - 2, 1) * sizeof(int2)) + 31) & ~(size_t)15;
  const bool mixed = fam.msa_mixed_n != 0;
  const size_t ones_bytes = mixed ? 32 : 0;  // the N-aware kernels keep four ones behind their tip tables
  const size_t fused_lds = std::max((size_t)R * tip_bytes + ones_bytes + tail_bytes, (size_t)R * pad * (5 * sizeof(double) + sizeof(int)));
  // The form: THREE decisions (the test hooks -- read once per process -- only push a shape onto a form another shape takes
  // by itself: LH_K1_NO_FUSE a workgroup per (sample, rate); LH_K1_SEGMENTS the segmented tip table on small trees too;
  // LH_K1_TABLES / LH_K1_NO_TABLES the cherry-table form, with / without tables, where the register-stack form would run;
  // LH_K1_CXX_WALK that form's C++ walk where the assembly walk would run).
  //  1. fused: all rates in one workgroup -- at most 16 waves, its LDS (R tip tables, later the exchange area) within a
  //     third (up to 8 waves) or half (up to 16) of a CU's; the ancestral-sequence step (allow_fused = false) wants the
  //     rates' planes unmixed.  (Round 4: 16 instead of 8 waves -- families with 257 .. 512 site patterns at R = 4, two
  //     workgroups of 12 or 16 waves per CU, used to fall to a workgroup per (sample, rate) at twice the time.)
  //  2. big: with a whole tip table in LDS fewer than five waves per SIMD would be resident -> segmented tip table.
  //  3. the register-stack form (stack depth <= 4: slots in registers) for big shapes; the cherry-table form (two slots in
  //     registers, deeper ones in scratch memory: any depth up to 16; the walk in assembly) for everything else.
  const bool fused = allow_fused && !dbg.k1_no_fuse && !dbg.k1_segments &&
                     ((R * wpr <= 8 && fused_lds <= 53 * 1024) ||    // three workgroups of up to eight waves per CU
                      (R * wpr > 8 && R * wpr <= 16 && fused_lds <= 80 * 1024));     // two of nine to sixteen
  const bool big = !fused && ((160 * 1024 / tip_bytes) * wpr / 4 < 5 || dbg.k1_segments);
  const bool tables_hook = dbg.k1_tables || dbg.k1_no_tables;
  const bool use_asm = !dbg.k1_cxx_walk;  // (both kinds of alignment: 2-bit state planes, or 2 bits + an N flag)
  // Fused shapes take the cherry-table form (round 4, after the assembly walk got its tip states through the scalar path,
  // a rescaling test every fourth op and a lane-parallel K0c): against the compiler's register-stack walk it is 17-33 %
  // ahead on families of at most 128 patterns (one wave per rate, three waves per SIMD: latency-bound -- 115 patterns 5.22
  // -> 4.28 ms per 49 152, 57 patterns 5.88 -> 4.20), +0.8 % at 167 patterns, +4.7 % on configs[2] (253: K1 5.74 -> 5.37 ms),
  // +6.8 % at 300 (twelve waves per workgroup), +1.6 % on the ragged-read family with N (profiles/r04_k1_small_families.txt,
  // r04_k1_programme.txt).  The register-stack kernels stay what large trees run (segmented tip table) and what the
  // LH_K1_STACK test hook selects for fused shapes.
  const bool stack_fused = fused && max_depth <= 4 && dbg.k1_stack && !tables_hook;
  const bool seg = big && max_depth <= 4 && !tables_hook;
  const size_t seg_bytes = (size_t)(2 * kSegOps + 1) * 16 * sizeof(double) + ones_bytes;
  size_t lds = fused ? fused_lds : seg ? seg_bytes : tip_bytes + ones_bytes + tail_bytes;
  // (A remainder of up to 64 patterns rides in a ONE-site wave: the assembly walk has a one-site variant -- 45 vector
  // registers, half the arithmetic per op; lh_prune_walk_asm_s1.inc.)
  if (lds > 160 * 1024) {
    snprintf(g_prune_error, sizeof(g_prune_error),
             "K1: a tree of %d tips with stack depth %d needs %zu bytes of LDS in the cherry-table form (160 KB per CU)", T, max_depth, lds);
    return -1;
  }
  // the register-stack kernels run behind the stack check (fields, discipline, ranks when fused), the cherry-table kernels
  // behind K0c (the same checks, plus the rewrite)
  if (stack_fused || seg) {
    hipLaunchKernelGGL(schedule_stack_check_kernel, dim3(n), dim3(64), 0, stream, T, stack_fused && max_depth <= 3 ? 3 : 4,
                       stack_fused ? 1 : 0, ops, ws.hdr, ws.err_flag);
  } else {
    hipLaunchKernelGGL(schedule_rewrite_kernel, dim3(n), dim3(64), (sizes.tabs_per_sample + 16) * sizeof(int), stream, T, max_depth,
                       (int)sizes.tabs_per_sample, dbg.k1_no_tables ? 0 : 1, ops, brlen, ws.wops, ws.wlen, ws.tabs, ws.hdr, ws.err_flag);
  }
  const int wg_waves = fused ? R * wpr : wpr;
  dim3 grid(tiles, fused ? 1 : R, n), block(64 * wg_waves);
  const int n_ops = T - 2;
  hipError_t attr_rc = hipSuccess;
#define LH_LAUNCH_K(K, NAME)                                                                                     \
  {                                                                                                              \
    if (lds > 64 * 1024)                                                                                         \
      attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    snprintf(g_prune_form, sizeof(g_prune_form), "%s", NAME);                                                    \
    if (attr_rc == hipSuccess)                                                                                   \
      hipLaunchKernelGGL(K, grid, block, lds, stream, n, n2, tile, R, wpr, fam.msa, L, T, n_ops, ops, ws.hdr, ws.err_flag, \
                         brlen, rates, eig, pmat, rate_stride, pi, site_lik, site_scal);                        \
  }
#define LH_LAUNCH_CT(K, NAME)                                                                                    \
  {                                                                                                              \
    if (lds > 64 * 1024)                                                                                         \
      attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    snprintf(g_prune_form, sizeof(g_prune_form), "%s", NAME);                                                    \
    if (attr_rc == hipSuccess)                                                                                   \
      hipLaunchKernelGGL(K, grid, block, lds, stream, n2, tile, R, wpr, fam.msa, fam.msa_planes, L, T, ws.wops, ws.wlen, ws.tabs, \
                         (int)sizes.tabs_per_sample, ws.hdr, brlen, rates, eig, pmat, rate_stride, pi, site_lik, \
                         site_scal);                                                                             \
  }
  // waves per SIMD that the LDS of the resident workgroups allows (160 KB per CU, 4 SIMDs)
  const int lds_waves = lds == 0 ? 8 : (int)((160 * 1024 / lds) * wg_waves / 4);
  if (stack_fused) {
    // register-stack form, all rates in one workgroup; three register budgets, the tightest one whose occupancy the
    // tip tables allow (a fourth stack slot spills too much at 80 VGPRs)
#define LH_LAUNCH_STACK(D, N)                                                          \
  {                                                                                    \
    if (lds_waves >= 6 && D == 3)                                                      \
      LH_LAUNCH_K((prune_kernel_w6<D, N>), "w6<" #D "," #N ">")                        \
    else if (lds_waves >= 5)                                                           \
      LH_LAUNCH_K((prune_kernel_w5<D, N>), "w5<" #D "," #N ">")                        \
    else                                                                               \
      LH_LAUNCH_K((prune_kernel_w4<D, N>), "w4<" #D "," #N ">")                        \
  }
    if (max_depth <= 3 && !mixed)
      LH_LAUNCH_STACK(3, false)
    else if (max_depth <= 3)
      LH_LAUNCH_STACK(3, true)
    else if (!mixed)
      LH_LAUNCH_STACK(4, false)
    else
      LH_LAUNCH_STACK(4, true)
#undef LH_LAUNCH_STACK
  } else if (seg) {
    // large trees: register-stack form with the tip table built a schedule segment at a time; four waves per SIMD
    // (128 VGPRs, few spills) beat five by 9 % (LH_K1_SEG_WAVES=5: test hook for the other budget)
    if (dbg.k1_seg_waves == 4) {
      if (mixed)
        LH_LAUNCH_K((prune_kernel_seg4<4, true>), "seg4<4,true>")
      else
        LH_LAUNCH_K((prune_kernel_seg4<4, false>), "seg4<4,false>")
    } else {
      if (mixed)
        LH_LAUNCH_K((prune_kernel_seg<4, true>), "seg5<4,true>")
      else
        LH_LAUNCH_K((prune_kernel_seg<4, false>), "seg5<4,false>")
    }
  } else {
    // The cherry-table form (two sites per lane, whole tip table in LDS): shapes whose rates do not fit one workgroup,
    // deep stacks, the ancestral-sequence step's unmixed planes.  Three register budgets; for alignments without N the
    // walk in assembly.
#define LH_LAUNCH_BUDGET(D, N, F, A)                                                          \
  {                                                                                           \
    if (lds_waves >= 6)                                                                       \
      LH_LAUNCH_CT((prune_kernel_ct6<D, N, F, A>), "ct6<" #D "," #N "," #F "," #A ">")        \
    else if (lds_waves >= 5)                                                                  \
      LH_LAUNCH_CT((prune_kernel_ct5<D, N, F, A>), "ct5<" #D "," #N "," #F "," #A ">")        \
    else                                                                                      \
      LH_LAUNCH_CT((prune_kernel_ct4<D, N, F, A>), "ct4<" #D "," #N "," #F "," #A ">")        \
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
    if (mixed && use_asm)              \
      LH_LAUNCH_FORM(D, true, true)    \
    else if (mixed)                    \
      LH_LAUNCH_FORM(D, true, false)   \
    else if (use_asm)                  \
      LH_LAUNCH_FORM(D, false, true)   \
    else                               \
      LH_LAUNCH_FORM(D, false, false)  \
  }
    if (max_depth <= 4)
      LH_LAUNCH_SHALLOW(4)
    else
      LH_LAUNCH_SHALLOW(16)
#undef LH_LAUNCH_SHALLOW
#undef LH_LAUNCH_FORM
#undef LH_LAUNCH_BUDGET
  }
#undef LH_LAUNCH_K
#undef LH_LAUNCH_CT
  const hipError_t launch_rc = attr_rc == hipSuccess ? hipGetLastError() : attr_rc;
  if (launch_rc != hipSuccess) {
    snprintf(g_prune_error, sizeof(g_prune_error), "K1 launch (%s, %zu bytes of LDS): %s", g_prune_form, lds, hipGetErrorString(launch_rc));
    return -1;
  }
  return fused ? 1 : R;
}

const char* prune_last_form() { return g_prune_form; }
const char* prune_last_error() { return g_prune_error; }

}  // namespace lh
