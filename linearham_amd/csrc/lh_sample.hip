// K4: naive-sequence sampling on the device (gfx950).
//
// Replaces the draws of HMM::SampleNaiveSequence (src/HMM.cpp:358-431): SampleInitialState (:323-341),
// SampleJunctionStates (:1222-1278) and SampleGermlineState (:1316-1353) for a whole batch of tree samples, on the
// forward arrays K2b has just written -- they stay in device memory (88 KB per sample for configs[2]; moving them
// to the host was the largest item of `linearham --pipeline`).
//
// The reference draws every state from std::discrete_distribution<int> over a dense weight vector
// (transition column x forward row) with one std::mt19937 stream.  Three facts make that reproducible here:
//  * a sample consumes a fixed number of engine outputs (one generate_canonical<double, 53> = two outputs per
//    draw, one draw per junction row and per germline region with more than one allele), so the host hands every
//    sample its own slice of the stream;
//  * libstdc++'s discrete_distribution is a handful of IEEE operations in a fixed order (accumulate the weights,
//    divide each by the sum, partial sums, last one forced to 1, lower_bound of the uniform), and zeros change
//    neither a sum nor a partial sum -- only the non-zero weights are visited, in dense state order;
//  * the non-zero weights of a backward step are few and known from the structure FillTransition
//    (src/HMM.cpp:964-1089) writes: the predecessors of a state are the left genes' states of the previous row,
//    the NTI states and the previous germline position of its own gene.  Transition values are rebuilt with the
//    association FillTransition uses ((landing_out * gene_prob) * landing_in ...), so every weight has the bits
//    the dense matrices hold.
// SIXTEEN LANES per sample, four samples per wave (round 4; rounds 2-3: one wave per sample).  The sums of a draw are
// sequential by definition (a + b + c in IEEE arithmetic is an order), but its weights are not, and neither are different
// samples: lane l of a sample's group forms the weight of left gene 16 c + l (coalesced 128-byte reads of the tables and of
// the forward row), the group's sixteen values go through LDS and every lane of the group adds them up in lane order on
// reads that are broadcast within the group -- one vector add per element and FOUR samples, where the wave-per-sample form
// spent the same instruction on one.  That form was bound by exactly these chains (about 40 000 instructions per sample,
// 2.8 ms per 49 152 samples: profiles/r04_pipeline.txt), not by the 88 KB of forward arrays it read.  Everything that was
// wave-uniform per sample -- the successor state, its table entries, the running sums, the uniform -- is now a per-lane value
// equal across a group, and nothing branches on it: the single-weight draw of a left gene's predecessor is the general draw
// with one non-zero weight.  Zero weights need no skipping: x + 0 = x, and a zero weight never satisfies the lower_bound
// test its predecessor failed.  The quotients weight / sum of the second pass are one division per lane.
// (First version: one thread per sample, 5 ms for 2048 samples.)
// No fused multiply-adds in here: the host rounds after every operation.
#include <algorithm>

#include "lh_device.h"

namespace lh {

namespace {

#pragma clang fp contract(off)

constexpr int kG = 16;  // lanes per sample (measured: 8 -> 4.88 M pipeline rows/s, 16 -> 5.07-5.10 M, 32 -> 4.77 M; profiles/r04_pipeline.txt)

struct Draw {  // this sample's slice of the engine's output stream, and its group's two 16-slot areas of the wave's LDS
  const uint32_t* words;
  int next;
  double* wbuf;  // [16] weights or quotients of the group's current chunk
  double* cbuf;  // [16] partial sums
  int gl;        // lane within the group
  double* wsave; // [cap] every weight of the group's current draw (nullptr: the family's draws do not fit; recompute them)
};

// std::generate_canonical<double, 53>(std::mt19937&) (bits/random.tcc): two 32-bit outputs
__device__ inline double canonical(Draw& d) {
  const double u0 = (double)d.words[d.next], u1 = (double)d.words[d.next + 1];
  d.next += 2;
  double sum = 0.0, tmp = 1.0;
  sum += u0 * tmp;
  tmp *= 4294967296.0;
  sum += u1 * tmp;
  tmp *= 4294967296.0;
  double r = sum / tmp;
  if (r >= 1.0) r = 0x1.fffffffffffffp-1;  // nextafter(1, 0)
  return r;
}

// std::discrete_distribution<int> over the weights  pre[0..n_pre) | lane_weight(0..n_mid) | post[0..n_post)
// (this is the order of the states in the dense vector; everything the vector holds besides is zero), for the sample of
// the calling lane's group; n_dense, n_pre, n_mid, n_post are the same for every group of the wave, the weights are not.
// Returns the position drawn; kPastEnd when the uniform lies beyond the last partial sum -- the caller then takes
// the vector's last element, whose partial sum libstdc++ sets to 1; kFirst when the answer is the vector's element
// 0 whatever it holds (a uniform of exactly 0, or a vector of fewer than two weights, which is not drawn from and
// takes nothing from the engine).  `n_dense`: size of the dense vector.
constexpr int kPastEnd = -1, kFirst = -2, kNone = -3;

// The 16 values the lanes of a group hold, added to `acc` in lane order: every lane performs the same chain of adds on
// values it reads back from LDS (the same addresses across the group: broadcast reads), so the result is uniform across the
// group.  Lanes beyond the chunk hold 0.
__device__ inline double add_in_group_order(double acc, double mine, const Draw& d) {
  d.wbuf[d.gl] = mine;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const double2* b2 = reinterpret_cast<const double2*>(d.wbuf);
#pragma unroll
  for (int j = 0; j < kG / 2; ++j) {
    const double2 v = b2[j];
    acc += v.x;
    acc += v.y;
  }
  __builtin_amdgcn_wave_barrier();  // (the buffer is rewritten by the next chunk)
  return acc;
}

// (the successor's own five weights come by value and their place -- before or behind the left genes' block -- as a template
// argument: with a pointer and run-time counts the array was indexed dynamically, and the build at -O3 faulted on it)
// Three phases per draw.  A: every lane forms the weights of its genes (one per 16-gene chunk) and parks them in the group's
// LDS row -- no arithmetic depends on a load of another chunk, so all of a row's table and forward reads are in flight
// together (the chunk-by-chunk form of the first 16-lane version paid a memory round trip per chunk and pass: 26 per V-D
// row).  B: the sum, in dense-vector order, from LDS.  C: quotients and partial sums, from LDS.  `nz` remembers which chunks
// hold a non-zero weight in any group of the wave; the others change no sum and cannot be drawn.
template <int kPre, int kPost, typename F>
__device__ int group_draw(int n_dense, const double (&own)[5], int n_mid, F&& lane_weight, Draw& d) {
  constexpr int n_pre = kPre, n_post = kPost;
  const double (&pre)[5] = own;
  const double (&post)[5] = own;
  if (n_dense < 2) return kFirst;
  const int gl = d.gl;
  const int shift = (threadIdx.x & 63) & ~(kG - 1);  // first lane of this group within the wave
  const bool saved = d.wsave != nullptr;
  unsigned long long nz = 0;  // bit c / 16: chunk c has a non-zero weight somewhere in the wave (first 64 chunks)
  if (saved) {
    for (int c = 0; c < n_mid; c += kG) {
      const double w = (c + gl < n_mid) ? lane_weight(c + gl) : 0.0;
      d.wsave[c + gl] = w;
      if (__builtin_amdgcn_ballot_w64(w != 0.0) != 0) nz |= 1ull << ((c / kG) & 63);
    }
    if (n_mid > 64 * kG) nz = ~0ull;  // (more chunks than bits: none is skipped)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  double sum = 0.0;
#pragma unroll
  for (int a = 0; a < n_pre; ++a) sum += pre[a];
  for (int c = 0; c < n_mid; c += kG) {
    if (saved) {
      if (!((nz >> ((c / kG) & 63)) & 1)) continue;
      const double2* b2 = reinterpret_cast<const double2*>(d.wsave + c);
#pragma unroll
      for (int j = 0; j < kG / 2; ++j) {
        const double2 v = b2[j];
        sum += v.x;
        sum += v.y;
      }
    } else {
      const double w = (c + gl < n_mid) ? lane_weight(c + gl) : 0.0;
      if (__builtin_amdgcn_ballot_w64(w != 0.0) == 0) continue;  // (no group of the wave has a weight here)
      sum = add_in_group_order(sum, w, d);
    }
  }
#pragma unroll
  for (int a = 0; a < n_post; ++a) sum += post[a];
  const double p = canonical(d);
  int result = kNone;
  if (!(p > 0.0)) result = kFirst;
  // no positive weight at all: every quotient is 0 / 0, no partial sum compares below the uniform, and lower_bound
  // -- which only ever moves left then -- ends on element 0
  if (!(sum > 0.0)) result = kFirst;
  double cum = 0.0;
  int pos = 0;
#pragma unroll
  for (int a = 0; a < n_pre; ++a, ++pos) {
    cum += pre[a] / sum;
    if (result == kNone && !(cum < p)) result = pos;
  }
  if (saved) {
    // quotients in place (a lane its own elements), then every lane of the group runs the partial sums over them on
    // broadcast reads.  Weights are >= 0 (or the sum is NaN, which was settled above), so the partial sums never fall --
    // or, from an inf weight on (a row the reference's 2^(256 d) equalisation has overflowed), are NaN to the end -- and
    // the element lower_bound picks is the number of leading elements whose partial sum is BELOW the uniform (no NaN is):
    // one compare and one add-with-carry per element, no round trip through LDS between chunks.
    for (int c = 0; c < n_mid; c += kG)
      if ((nz >> ((c / kG) & 63)) & 1) d.wsave[c + gl] = d.wsave[c + gl] / sum;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int below = 0;  // leading elements of the block whose partial sum is below the uniform
    for (int c = 0; c < n_mid; c += kG) {
      if (__builtin_amdgcn_ballot_w64(result == kNone) == 0) break;  // every group of the wave has its answer
      if (!((nz >> ((c / kG) & 63)) & 1)) {  // zeros: the partial sums stand still
        if (result == kNone) below = min(c + kG, n_mid);
        continue;
      }
      const double2* b2 = reinterpret_cast<const double2*>(d.wsave + c);
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < kG / 2; ++j) {
        const double2 v = b2[j];
        cum += v.x;
        cnt += cum < p ? 1 : 0;
        cum += v.y;
        cnt += cum < p ? 1 : 0;
      }
      if (result == kNone) {
        below = c + cnt;  // (every element before this chunk was below: the chunk is reached with result == kNone only then)
        if (cnt < kG && below < n_mid) result = n_pre + below;
      }
    }
  } else {
  for (int c = 0; c < n_mid; c += kG) {
    if (__builtin_amdgcn_ballot_w64(result == kNone) == 0) break;  // every group of the wave has its answer
    const double w = (c + gl < n_mid) ? lane_weight(c + gl) : 0.0;
    if (__builtin_amdgcn_ballot_w64(w != 0.0) == 0) continue;
    // the chunk's partial sums, in lane order: the group's first lane runs the chain (quotients read back from LDS) and
    // leaves partial sum j at cbuf[j]; then every lane compares its own with the uniform and the first hit is the answer
    // (a group whose weights are all zero here leaves its running sum as it is: sixteen times + 0)
    d.wbuf[gl] = w / sum;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (gl == 0) {
      const double2* b2 = reinterpret_cast<const double2*>(d.wbuf);
      double2* c2 = reinterpret_cast<double2*>(d.cbuf);
      double run = cum;
#pragma unroll
      for (int j = 0; j < kG / 2; ++j) {
        const double2 v = b2[j];
        double2 o;
        run += v.x;
        o.x = run;
        run += v.y;
        o.y = run;
        c2[j] = o;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double mine = d.cbuf[gl];
    cum = d.cbuf[kG - 1];  // (lanes beyond the chunk added zeros)
    const int m = min(kG, n_mid - c);
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(result == kNone && gl < m && !(mine < p));
    const unsigned mine_hits = (unsigned)(hit >> shift) & (kG >= 32 ? 0xffffffffu : ((1u << (kG & 31)) - 1u));
    __builtin_amdgcn_wave_barrier();
    if (result == kNone && mine_hits != 0) result = n_pre + c + (int)__builtin_ctz(mine_hits);
  }
  }
  pos = n_pre + n_mid;
#pragma unroll
  for (int a = 0; a < n_post; ++a, ++pos) {
    cum += post[a] / sum;
    if (result == kNone && !(cum < p)) result = pos;
  }
  return result == kNone ? kPastEnd : result;
}

// What a dense junction state is: kind (0 left-gene state, 1 NTI, 2 right-gene germline state), its gene (left
// index for kind 0, right index otherwise) and the NTI base -- one word per state, made by lh_family_set_sampler.
struct Succ {
  int kind;  // 3: gene `gene` of the germline region right of the junction
  int gene;
  int base;
  int row;   // the junction row a germline state belongs to (kinds 0 and 2); -1: any row (NTI states)
};

// The sampled state of row i + 1 normally belongs to row i + 1.  It need not: when the uniform exceeds the last
// partial sum, discrete_distribution returns the LAST element of the state vector whatever it is, and the reference
// carries on from there with the dense transition column of that state.  `row` lets the steps below reproduce
// that: a transition into a germline state exists only from the row before its own.
__device__ inline Succ classify(const DevSampleJunction& J, int dense) {
  const int c = J.state_class[dense];
  Succ s{c & 3, c >> 4, (c >> 2) & 3, -1};
  if (s.kind == 0) s.row = dense - J.left_dense[s.gene];
  if (s.kind == 2) s.row = J.right_first[s.gene] + (dense - J.right_dense[s.gene] - 4);
  return s;
}

// One backward step: draws the state of junction row i given its successor (row i + 1, or the gene of the region
// right of the junction for i = W - 1).  fwd_row: the row's compact forward entries [left nL | nti nR x 4 |
// right nR].  Returns the dense index drawn.  The successor -- and with it every table entry read here -- is a per-lane
// value, the same across a sample's group; nothing branches on it.
__device__ int draw_row(const DevSampleJunction& J, int i, const Succ& sc, const double* __restrict__ fwd_row, Draw& d) {
  const int nL = J.n_left, nR = J.n_right;
  const double* fL = fwd_row;
  const double* fN = fwd_row + nL;
  const double* fR = fwd_row + nL + 4 * (size_t)nR;
  // kind 0 -- the successor is a left gene's state: its only predecessor is the gene's state on the row before, the
  // distribution has one non-zero weight (the host still draws: the uniform may be 0, or the weight's quotient below it),
  // and a state that does not belong to row i + 1 has no predecessor with a forward entry on row i.  r1 indexes the LEFT
  // genes then, the RIGHT genes otherwise.
  const bool single = sc.kind == 0;
  const int r1 = sc.gene;
  const int rr = single ? 0 : r1;  // a valid right-gene index for the table reads the single-weight case does not use
  const double w_single = (single && sc.row == i + 1) ? J.left_trans[(size_t)(i + 1) * nL + r1] * fL[r1] : 0.0;
  // coefficient of the left genes' block: T(state of left gene l on row i -> succ) = (landing_out * gene_prob) * x
  const double gp = J.gp[rr];
  double x, exitp = 1.0;
  if (sc.kind == 1)
    x = J.nli[(size_t)rr * 4 + sc.base];
  else if (sc.kind == 2)
    x = sc.row == i + 1 ? J.li[(size_t)sc.row * nR + rr] : 0.0;  // left states reach it from the row before its own only
  else {
    x = J.exit_li[rr];
    exitp = J.prod[rr];
  }
  const bool exiting = sc.kind == 3;
  auto left_weight = [&](int l) -> double {
    if (single) return l == r1 ? w_single : 0.0;
    if (i >= J.left_rows[l]) return 0.0;  // the gene has no state on this row
    double t = (J.left_lo[(size_t)i * nL + l] * gp) * x;
    if (exiting) t *= exitp;
    return t * fL[l];
  };
  // the successor's own gene: its four NTI states, then its germline state of this row
  double own[5];
  const int srow = sc.kind == 2 ? sc.row : 0;  // (a valid row for the reads below)
  // NOT to be unrolled.  Unrolled (hipcc 7.2 does so from -O2 on) the three-way choice of the table folds into a choice
  // between ADDRESSES followed by one wide load, and the kernel that comes out forms a wild address on some lanes: memory
  // faults at 0xfffff000 / 0x100000000 on the first launch of a 9-tip family.  The same source is correct at -O1, with
  // -fno-unroll-loops (which leaves exactly this loop alone: -Rpass=loop-unroll) and with every read address-checked -- a
  // code-generation fault, found by bisecting the optimisation flags (round 4); tests/test_host_gpu.py's device = host
  // sampler tests are the guard.
#pragma unroll 1
  for (int a = 0; a < 4; ++a) {
    double t;
    if (sc.kind == 1)
      t = J.ntt[(size_t)rr * 16 + a * 4 + sc.base];
    else if (sc.kind == 2)
      t = J.nlo[((size_t)srow * nR + rr) * 4 + a];  // NTI states live on every row: into the state's own position
    else
      t = J.exit_nlo[(size_t)rr * 4 + a];
    own[a] = single ? 0.0 : t * fN[(size_t)rr * 4 + a];
  }
  own[4] = 0.0;
  const int first = J.right_first[rr];
  if (i >= first && sc.kind == 2 && sc.row == i + 1) own[4] = J.rtrans[(size_t)sc.row * nR + rr] * fR[rr];
  if (i >= first && sc.kind == 3) own[4] = J.exit_trans[rr] * fR[rr];
  const bool rf = J.right_first_block != 0;
  const int pos = rf ? group_draw<5, 0>(J.n_states, own, nL, left_weight, d) : group_draw<0, 5>(J.n_states, own, nL, left_weight, d);
  if (pos == kFirst) return 0;
  if (pos == kPastEnd) return J.n_states - 1;
  const int a = rf ? pos : pos - nL;  // position within the own-gene entries
  if (a >= 0 && a < 5) return J.right_dense[rr] + (a < 4 ? a : 4 + (i - first));
  return J.left_dense[rf ? pos - 5 : pos] + i;
}

// SampleGermlineState: the gene of the germline region LEFT of junction J, given the junction's row-0 state
__device__ int draw_left_region(const DevSampleJunction& J, int dense0, const double* __restrict__ germ_fwd, Draw& d) {
  const Succ sc = classify(J, dense0);
  const int rg = sc.kind == 0 ? 0 : sc.gene;  // (a valid right-gene index)
  const double x = sc.kind == 1 ? J.nli[(size_t)rg * 4 + sc.base] : J.li[rg];  // row 0 of li
  const double gpx = J.gp[rg];
  const double none[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int pos = group_draw<0, 0>(
      J.n_left, none, J.n_left,
      [&](int g) -> double {
        double t;
        if (sc.kind == 0) {
          if (g != sc.gene || sc.row != 0) return 0.0;
          t = J.left_trans[g];  // row 0 of left_trans: the transition out of the germline region
        } else {
          if (sc.kind == 2 && sc.row != 0) return 0.0;
          t = (J.enter_lo[g] * gpx) * x;
        }
        return t * germ_fwd[g];
      },
      d);
  return pos == kFirst ? 0 : pos == kPastEnd ? J.n_left - 1 : pos;
}

// Samples junction J backwards: states[i] for i = W-1 .. 0, given gene `right_gene` of the region right of it.
// Returns the row-0 state.
__device__ int sample_junction(const DevSampleJunction& J, int right_gene, const double* __restrict__ fwd_rows,
                                Draw& d, int32_t* __restrict__ states, bool writer) {
  const int W = J.n_rows;
  const size_t stride = (size_t)J.n_left + 5 * (size_t)J.n_right;
  Succ sc{3, right_gene, 0, -1};
  int row0 = 0;
  for (int i = W - 1; i >= 0; --i) {
    const int s = draw_row(J, i, sc, fwd_rows + (size_t)i * stride, d);
    if (writer) states[i] = s;
    sc = classify(J, s);
    row0 = s;
  }
  return row0;
}

}  // namespace

// states[n][1 + W_dj + 1 + W_vd + 1] (igh) / [1 + W_vd + 1] (light chains):
//   J gene | D-J junction rows 0..W-1 | D gene | V-D junction rows | V gene      (dense indices, as the host keeps them)
constexpr int kSampleWaves = 4;            // waves per workgroup
constexpr int kPerWave = 64 / kG;          // samples per wave
__global__ void __launch_bounds__(64 * kSampleWaves)
    sample_kernel(const DevSampler* __restrict__ smp_dev, int n, const double* __restrict__ fwd_all, size_t forward_size,
                  const uint32_t* __restrict__ words_all, int words_per_sample, int32_t* __restrict__ states_all, int save_cap) {
  const DevSampler& smp = *smp_dev;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane / kG;
  const int s_raw = (blockIdx.x * kSampleWaves + wave) * kPerWave + grp;
  // (a group past the end of the batch walks the last sample along with the others and writes nothing: every lane of the
  // wave takes part in every ballot and barrier)
  const int s = min(s_raw, n - 1);
  const bool writer = (lane % kG) == 0 && s_raw < n;
  const double* fwd = fwd_all + (size_t)s * forward_size;
  __shared__ double wave_lds[kSampleWaves][kPerWave][2 * kG];
  extern __shared__ double2 sample_dyn[];  // [kSampleWaves][kPerWave][save_cap] doubles: a draw's weights (save_cap == 0: none)
  double* wsave = save_cap > 0 ? reinterpret_cast<double*>(sample_dyn) + (size_t)(wave * kPerWave + grp) * save_cap : nullptr;
  Draw d{words_all + (size_t)s * words_per_sample, 0, wave_lds[wave][grp], wave_lds[wave][grp] + kG, lane % kG, wsave};
  const DevSampleJunction& VD = smp.vd;
  const DevSampleJunction& DJ = smp.dj;
  const int nV = smp.n_v, nD = smp.n_d, nJ = smp.n_j;
  const size_t vd_size = (size_t)VD.n_rows * (VD.n_left + 5 * (size_t)VD.n_right);
  const size_t dj_size = smp.has_d ? (size_t)DJ.n_rows * (DJ.n_left + 5 * (size_t)DJ.n_right) : 0;
  const double* f_v = fwd;
  const double* f_vd = f_v + nV;
  const double* f_d = f_vd + vd_size;
  const double* f_dj = f_d + (smp.has_d ? nD : 0);
  const double* f_j = f_dj + dj_size;
  int32_t* out = states_all + (size_t)s * smp.states_per_sample;
  // SampleInitialState
  const double none[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  int jg = group_draw<0, 0>(nJ, none, nJ, [&](int g) -> double { return f_j[g]; }, d);
  jg = jg == kFirst ? 0 : jg == kPastEnd ? nJ - 1 : jg;
  int o = 0;
  if (writer) out[o] = jg;
  ++o;
  int row0;
  if (smp.has_d) {
    row0 = sample_junction(DJ, jg, f_dj, d, out + o, writer);
    const int dg = draw_left_region(DJ, row0, f_d, d);
    o += DJ.n_rows;
    if (writer) out[o] = dg;
    ++o;
    row0 = sample_junction(VD, dg, f_vd, d, out + o, writer);
  } else {
    row0 = sample_junction(VD, jg, f_vd, d, out + o, writer);
  }
  const int vg = draw_left_region(VD, row0, f_v, d);
  o += VD.n_rows;
  if (writer) out[o] = vg;
}

void launch_sample(const DevSampler& smp, const DevSampler* smp_dev, int n, const double* fwd, size_t forward_size,
                   const uint32_t* words, int words_per_sample, int32_t* states, hipStream_t stream) {
  const int per_block = kSampleWaves * kPerWave;
  // a draw's weights are kept in LDS between its passes when the widest draw of the family fits 2 KB per sample (32 KB
  // per workgroup): 256 genes; wider families form them again in each pass
  int widest = std::max(std::max(smp.n_v, smp.n_j), std::max(smp.vd.n_left, smp.has_d ? std::max(smp.n_d, smp.dj.n_left) : 0));
  widest = (widest + kG - 1) / kG * kG;
  const int save_cap = widest <= 256 ? widest : 0;
  const size_t dyn = (size_t)per_block * save_cap * sizeof(double);
  hipLaunchKernelGGL(sample_kernel, dim3((n + per_block - 1) / per_block), dim3(64 * kSampleWaves), dyn, stream, smp_dev, n,
                     fwd, forward_size, words, words_per_sample, states, save_cap);
}

}  // namespace lh
