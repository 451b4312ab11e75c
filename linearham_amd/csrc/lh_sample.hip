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
// One WAVE per sample.  The sums of a draw are sequential by definition (a + b + c in IEEE arithmetic is an
// order), but its weights are not: lane l forms the weight of left gene l (64 genes at a time, coalesced reads of
// the tables and of the forward row), the wave then adds the 64 values in lane order: they go through 512 bytes of
// LDS and every lane runs the same chain of adds on broadcast reads (round 3; the first form fetched each value with
// two v_readlane: three vector instructions per element instead of one, 3.4 -> 2.8 ms per 49 152 samples).  Zero weights need no skipping: x + 0 = x, and a zero
// weight never satisfies the lower_bound test its predecessor failed.  The quotients weight / sum of the second
// pass are again one division per lane.  (First version: one thread per sample, 5 ms for 2048 samples; this one:
// see DESIGN.md section 6.)
// No fused multiply-adds in here: the host rounds after every operation.
#include "lh_device.h"

namespace lh {

namespace {

#pragma clang fp contract(off)

struct Draw {  // this sample's slice of the engine's output stream, and the wave's 1 KB of LDS (wave_draw)
  const uint32_t* words;
  int next;
  double* lds;  // [128]: 64 weights or quotients, 64 partial sums
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
// (this is the order of the states in the dense vector; everything the vector holds besides is zero).
// Returns the position drawn; kPastEnd when the uniform lies beyond the last partial sum -- the caller then takes
// the vector's last element, whose partial sum libstdc++ sets to 1; kFirst when the answer is the vector's element
// 0 whatever it holds (a uniform of exactly 0, or a vector of fewer than two weights, which is not drawn from and
// takes nothing from the engine).  `n_dense`: size of the dense vector.
constexpr int kPastEnd = -1, kFirst = -2;

// The 64 values the lanes hold, added to `acc` in lane order: every lane performs the same chain of adds on values it
// reads back from LDS (uniform addresses: broadcast reads), so the result is uniform and no lane-to-scalar traffic is
// needed -- 64 vector adds where the v_readlane form took 64 x (2 v_readlane + 1 add).  Lanes beyond the chunk hold 0.
__device__ inline double add_in_lane_order(double acc, double mine, double* buf) {
  buf[threadIdx.x & 63] = mine;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const double2* b2 = reinterpret_cast<const double2*>(buf);
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const double2 v = b2[j];
    acc += v.x;
    acc += v.y;
  }
  __builtin_amdgcn_wave_barrier();  // (the buffer is rewritten by the next chunk)
  return acc;
}

template <typename F>
__device__ int wave_draw(int n_dense, const double* pre, int n_pre, int n_mid, F&& lane_weight, const double* post,
                         int n_post, Draw& d) {
  if (n_dense < 2) return kFirst;
  const int lane = threadIdx.x & 63;
  double* wbuf = d.lds;
  double* cbuf = d.lds + 64;
  double sum = 0.0;
  for (int a = 0; a < n_pre; ++a) sum += pre[a];
  for (int c = 0; c < n_mid; c += 64) {
    const double w = (c + lane < n_mid) ? lane_weight(c + lane) : 0.0;
    if (__builtin_amdgcn_ballot_w64(w != 0.0) == 0) continue;
    sum = add_in_lane_order(sum, w, wbuf);
  }
  for (int a = 0; a < n_post; ++a) sum += post[a];
  const double p = canonical(d);
  if (!(p > 0.0)) return kFirst;
  // no positive weight at all: every quotient is 0 / 0, no partial sum compares below the uniform, and lower_bound
  // -- which only ever moves left then -- ends on element 0
  if (!(sum > 0.0)) return kFirst;
  double cum = 0.0;
  int pos = 0;
  for (int a = 0; a < n_pre; ++a, ++pos) {
    cum += pre[a] / sum;
    if (cum >= p) return pos;
  }
  for (int c = 0; c < n_mid; c += 64) {
    const double w = (c + lane < n_mid) ? lane_weight(c + lane) : 0.0;
    if (__builtin_amdgcn_ballot_w64(w != 0.0) == 0) continue;
    // the chunk's partial sums, in lane order: lane 0 runs the chain (quotients read back from LDS) and leaves partial
    // sum j at cbuf[j]; then every lane compares its own with the uniform and the first hit is the answer
    wbuf[lane] = w / sum;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const double2* b2 = reinterpret_cast<const double2*>(wbuf);
      double2* c2 = reinterpret_cast<double2*>(cbuf);
      double run = cum;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
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
    const double mine = cbuf[lane];
    cum = cbuf[63];  // (lanes beyond the chunk added zeros)
    const int m = min(64, n_mid - c);
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(lane < m && mine >= p);
    __builtin_amdgcn_wave_barrier();
    if (hit != 0) return n_pre + c + (int)__builtin_ctzll(hit);
  }
  pos = n_pre + n_mid;
  for (int a = 0; a < n_post; ++a, ++pos) {
    cum += post[a] / sum;
    if (cum >= p) return pos;
  }
  return kPastEnd;
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
// right nR].  Returns the dense index drawn.  Everything but the lane-parallel weights is wave-uniform.
__device__ int draw_row(const DevSampleJunction& J, int i, const Succ& sc, const double* __restrict__ fwd_row, Draw& d) {
  const int nL = J.n_left, nR = J.n_right;
  const double* fL = fwd_row;
  const double* fN = fwd_row + nL;
  const double* fR = fwd_row + nL + 4 * (size_t)nR;
  const int r1 = sc.gene;
  if (sc.kind == 0) {
    // the only predecessor of a left gene's state is the gene's state on the row before: the distribution has one
    // non-zero weight, but the host still draws (and the uniform may be 0, or the weight's quotient below it)
    // (a state that does not belong to row i + 1 has no predecessor with a forward entry on row i)
    const double w = sc.row == i + 1 ? J.left_trans[(size_t)(i + 1) * nL + r1] * fL[r1] : 0.0;
    const int pos = wave_draw(J.n_states, &w, 1, 0, [](int) { return 0.0; }, nullptr, 0, d);
    return pos == kFirst ? 0 : pos == kPastEnd ? J.n_states - 1 : J.left_dense[r1] + i;
  }
  // coefficient of the left genes' block: T(state of left gene l on row i -> succ) = (landing_out * gene_prob) * x
  const double gp = J.gp[r1];
  double x, exitp = 1.0;
  if (sc.kind == 1)
    x = J.nli[(size_t)r1 * 4 + sc.base];
  else if (sc.kind == 2)
    x = sc.row == i + 1 ? J.li[(size_t)sc.row * nR + r1] : 0.0;  // left states reach it from the row before its own only
  else {
    x = J.exit_li[r1];
    exitp = J.prod[r1];
  }
  const bool exiting = sc.kind == 3;
  auto left_weight = [&](int l) -> double {
    if (i >= J.left_rows[l]) return 0.0;  // the gene has no state on this row
    double t = (J.left_lo[(size_t)i * nL + l] * gp) * x;
    if (exiting) t *= exitp;
    return t * fL[l];
  };
  // the successor's own gene: its four NTI states, then its germline state of this row
  double own[5];
  for (int a = 0; a < 4; ++a) {
    double t;
    if (sc.kind == 1)
      t = J.ntt[(size_t)r1 * 16 + a * 4 + sc.base];
    else if (sc.kind == 2)
      t = J.nlo[((size_t)sc.row * nR + r1) * 4 + a];  // NTI states live on every row: into the state's own position
    else
      t = J.exit_nlo[(size_t)r1 * 4 + a];
    own[a] = t * fN[(size_t)r1 * 4 + a];
  }
  own[4] = 0.0;
  const int first = J.right_first[r1];
  if (i >= first && sc.kind == 2 && sc.row == i + 1) own[4] = J.rtrans[(size_t)sc.row * nR + r1] * fR[r1];
  if (i >= first && sc.kind == 3) own[4] = J.exit_trans[r1] * fR[r1];
  const bool rf = J.right_first_block != 0;
  const int pos = wave_draw(J.n_states, own, rf ? 5 : 0, nL, left_weight, own, rf ? 0 : 5, d);
  if (pos == kFirst) return 0;
  if (pos == kPastEnd) return J.n_states - 1;
  const int a = rf ? pos : pos - nL;  // position within the own-gene entries
  if (a >= 0 && a < 5) return J.right_dense[r1] + (a < 4 ? a : 4 + (i - first));
  return J.left_dense[rf ? pos - 5 : pos] + i;
}

// SampleGermlineState: the gene of the germline region LEFT of junction J, given the junction's row-0 state
__device__ int draw_left_region(const DevSampleJunction& J, int dense0, const double* __restrict__ germ_fwd, Draw& d) {
  const Succ sc = classify(J, dense0);
  const int pos = wave_draw(
      J.n_left, nullptr, 0, J.n_left,
      [&](int g) -> double {
        double t;
        if (sc.kind == 0) {
          if (g != sc.gene || sc.row != 0) return 0.0;
          t = J.left_trans[g];  // row 0 of left_trans: the transition out of the germline region
        } else {
          if (sc.kind == 2 && sc.row != 0) return 0.0;
          const double x = sc.kind == 1 ? J.nli[(size_t)sc.gene * 4 + sc.base] : J.li[sc.gene];  // row 0 of li
          t = (J.enter_lo[g] * J.gp[sc.gene]) * x;
        }
        return t * germ_fwd[g];
      },
      nullptr, 0, d);
  return pos == kFirst ? 0 : pos == kPastEnd ? J.n_left - 1 : pos;
}

// Samples junction J backwards: states[i] for i = W-1 .. 0, given gene `right_gene` of the region right of it.
// Returns the row-0 state.
__device__ int sample_junction(const DevSampleJunction& J, int right_gene, const double* __restrict__ fwd_rows,
                                Draw& d, int32_t* __restrict__ states) {
  const int W = J.n_rows;
  const size_t stride = (size_t)J.n_left + 5 * (size_t)J.n_right;
  Succ sc{3, right_gene, 0, -1};
  int row0 = 0;
  for (int i = W - 1; i >= 0; --i) {
    const int s = draw_row(J, i, sc, fwd_rows + (size_t)i * stride, d);
    if ((threadIdx.x & 63) == 0) states[i] = s;
    sc = classify(J, s);
    row0 = s;
  }
  return row0;
}

}  // namespace

// states[n][1 + W_dj + 1 + W_vd + 1] (igh) / [1 + W_vd + 1] (light chains):
//   J gene | D-J junction rows 0..W-1 | D gene | V-D junction rows | V gene      (dense indices, as the host keeps them)
constexpr int kSampleWaves = 4;  // samples per workgroup
__global__ void __launch_bounds__(64 * kSampleWaves)
    sample_kernel(const DevSampler smp, int n, const double* __restrict__ fwd_all, size_t forward_size,
                  const uint32_t* __restrict__ words_all, int words_per_sample, int32_t* __restrict__ states_all) {
  const int s = __builtin_amdgcn_readfirstlane(blockIdx.x * kSampleWaves + (int)(threadIdx.x >> 6));
  if (s >= n) return;
  const bool writer = (threadIdx.x & 63) == 0;
  const double* fwd = fwd_all + (size_t)s * forward_size;
  __shared__ double wave_lds[kSampleWaves][128];
  Draw d{words_all + (size_t)s * words_per_sample, 0, wave_lds[threadIdx.x >> 6]};
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
  int jg = wave_draw(nJ, nullptr, 0, nJ, [&](int g) -> double { return f_j[g]; }, nullptr, 0, d);
  jg = jg == kFirst ? 0 : jg == kPastEnd ? nJ - 1 : jg;
  int o = 0;
  if (writer) out[o] = jg;
  ++o;
  int row0;
  if (smp.has_d) {
    row0 = sample_junction(DJ, jg, f_dj, d, out + o);
    const int dg = draw_left_region(DJ, row0, f_d, d);
    o += DJ.n_rows;
    if (writer) out[o] = dg;
    ++o;
    row0 = sample_junction(VD, dg, f_vd, d, out + o);
  } else {
    row0 = sample_junction(VD, jg, f_vd, d, out + o);
  }
  const int vg = draw_left_region(VD, row0, f_v, d);
  o += VD.n_rows;
  if (writer) out[o] = vg;
}

void launch_sample(const DevSampler& smp, int n, const double* fwd, size_t forward_size, const uint32_t* words,
                   int words_per_sample, int32_t* states, hipStream_t stream) {
  hipLaunchKernelGGL(sample_kernel, dim3((n + kSampleWaves - 1) / kSampleWaves), dim3(64 * kSampleWaves), 0, stream, smp, n,
                     fwd, forward_size, words, words_per_sample, states);
}

}  // namespace lh
