// Internal declarations shared by the HIP translation units of liblinearham_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "linearham_amd.h"

namespace lh {

constexpr double kScaleFactor = 0x1p256;      // SCALE_FACTOR, src/utils.hpp:22
constexpr double kScaleThreshold = 0x1p-256;  // SCALE_THRESHOLD, src/utils.hpp:24
constexpr double kLogScaleFactor = 177.445678223345993274;  // log(2^256)

// schedule op kinds (ops[4k] & 15); bit 4 = push the accumulator to stack slot ops[4k+3] first
// bits 8..: for a tip-into-accumulator / pop op, the number of inner-branch P-matrices the earlier ops need (its own
// one or two follow): K1's prologue packs its matrix work by it
enum : int { OP_CHERRY = 0, OP_TIP_ACC = 1, OP_POP_ACC = 2, OP_PUSH_FLAG = 16, OP_RANK_SHIFT = 8 };

// Walk ops: what K1 executes.  K0c (schedule_check_kernel, lh_prune.hip) validates a sample's schedule ON THE
// DEVICE and rewrites it with every cherry that can be folded into its consumer replaced by a table look-up: for
// a cherry (y, z) under branch c the vector P_c (P_y[:, s_y] o P_z[:, s_z]) takes 16 values per (sample, rate)
// (25 with N tips), which K1's prologue tabulates; then
//   cherry + tip-into-accumulator      ->  W_CTIP      a = table[s_y][s_z] o tipcol_w         (no mat-vec)
//   pushed cherry + pop                ->  W_CTAB_ACC  a = table[s_y][s_z] o (P_first a)      (one mat-vec, no push / pop)
// A walk op is an 8-byte descriptor (WalkOp in lh_prune.hip).
enum : int { W_CHERRY = 0, W_TIP_ACC = 1, W_POP = 2, W_CTIP = 3, W_CTAB_ACC = 4 };

// K0c's output and K1's scratch area (all device memory, sized by prune_ws_sizes)
struct PruneWs {
  double* scratch;    // per (sample, rate): [n_mat + n_tab][16] P-matrices (walk order, then the cherry branches'),
                      // then [n_tab][E][4] cherry tables, E = 16 or 25
  int2* wops;         // [n][T-2] walk-op descriptors
  double* wlen;       // [n][T-2] branch length of inner-branch matrix i of the prologue's list (walk order, then the tables')
  int4* tabs;         // [n][(T-1)/2] cherry tables: tip y, tip z, the cherry's node
  int4* hdr;          // [n] walk ops, matrices, tables, error (malformed schedule: the sample's results are NaN)
  int32_t* err_flag;  // set when any sample of any launch had a malformed schedule (lh_family_status reads and clears it)
};
struct PruneWsSizes {
  size_t scratch_doubles_per_rate;  // per (sample, rate)
  size_t tabs_per_sample;
};
PruneWsSizes prune_ws_sizes(int T, bool mixed_n);

// Device copy of lh_segments / lh_junction / family constants (all pointers are device pointers).
struct DevSegments {
  int32_t n_genes;
  int32_t n_chunks;      // ceil(longest segment / 8)
  // Consensus form (lh_family_create builds it when the genes of the set are site-aligned and alike, as the
  // Smith-Waterman candidates of one rearrangement are): per covered alignment site the u-column most genes
  // take there; a gene's product is then (prefix product of the consensus up to its last site) / (prefix up to
  // its first site) x the few factors where it departs from the consensus.  cons_sites == 0: not available.
  int32_t cons_sites;        // covered sites (<= 510)
  int32_t cons_diffs;        // diff entries per gene (padded)
  const uint16_t* cons_col;  // [cons_sites] consensus u-column (index or byte offset like inds_c)
  const uint32_t* cons_rng;  // [n_genes] first | (last + 1) << 16 | rounds of 8 departures << 25, in consensus positions
  const uint32_t* cons_dif;  // [cons_diffs][n_genes] position | own u-column << 16; padding = position
                             // cons_sites (reciprocal 1.0) | the sentinel column (emission 1.0)
  const uint4* inds_c;   // [n_chunks][n_genes] eight 16-bit u-column indices (or byte offsets, see
                         // DevFamily::idx_byte_offsets) per entry (lane g's next
                         // eight factors in one coalesced 16-byte load), padded with the sentinel
                         // column C whose emission is 1.0
};

// Junction tables as K2b reads them: gene dimensions padded to a multiple of 64 with entries that
// contribute nothing (zero transitions, emission index = the zero sentinel), so that every lane of a
// wave loads and computes unconditionally.  Emission indices point into the compact junction-column
// vector (DevFamily::jcols); position n_jcols is the sentinel holding 0.0.
struct DevJunction {
  int32_t n_rows, n_left, n_right;
  int32_t left_pad, right_pad;  // padded gene counts (row strides)
  const double *enter_lo;       // [left_pad]
  const double *left_trans;     // [n_rows][left_pad], row 0 = enter_trans
  const double *left_lo;        // [n_rows][left_pad]
  const int32_t* left_xmsa;     // [n_rows][left_pad]
  const double *right_gp_nli;   // [right_pad][4]
  const double *right_ntt;      // [right_pad][4][4] stored transposed: [r][b][a] = transition a -> b
  const double *right_nlo;      // [n_rows][right_pad][4]
  const double *right_trans, *right_gp_li;  // [n_rows][right_pad]
  const int32_t *right_xmsa;    // [n_rows][right_pad]
  const int32_t *nti_xmsa;      // [n_rows][right_pad][4]
  const double *exit_nlo;       // [right_pad][4]
  const double *exit_trans, *exit_gp_li;  // [right_pad]
  const int32_t* row_pat;       // [n_rows] K1 pattern of the row's alignment site (>= n_prune: the all-N pattern
                                // or a family without an alignment); used by the extended-range mode only
};

// Two reductions happen once per family, on the host (lh_family_create):
//  * identical alignment columns (site patterns) are pruned once: K1 runs over the n_pat distinct
//    columns of the MSA, msa is stored pattern-major;
//  * xMSA columns that pair the same naive base with the same pattern have the same emission: K2 works
//    on (naive base, pattern) pairs ("u-columns").  With an alignment there are always n_ucol = 5 n_prune + 5 slots,
//    numbered by their place in K1's output planes: u = base * n_prune + pattern, the all-N pattern's five pairs last;
//    a pair no xMSA column uses keeps its slot (u_base = 0xff), so K2a fills its emission vector from the planes
//    without a look-up.  Every index table below is in u-column space.  Limits that follow: K2a's LDS emission vector
//    takes (n_ucol + 1) * 8 = (5 n_prune + 6) * 8 bytes per sample, and the segment index chunks can hold 16-bit BYTE
//    offsets (idx_byte_offsets) while that is < 65 536, i.e. n_prune <= 1637.  Families without an alignment
//    (n_seqs == 0) keep their columns one to one.
struct DevFamily {
  int32_t has_d, n_seqs, n_sites, n_xmsa;  // as described by the caller
  int32_t n_pat;                           // distinct alignment columns
  int32_t n_prune;                         // K1's site dimension: n_pat minus the all-N pattern, which
                                           // comes last and whose emission is 1 whatever the tree
  int32_t msa_mixed_n;                     // 1 if some pattern mixes N with bases (K1 then handles N tips)
  int32_t n_ucol;                          // (naive base, pattern) pairs (K2's column dimension): with an alignment
                                           // 5 n_prune + 5, u = base * n_prune + pattern (the all-N pattern's five last)
  int32_t idx_byte_offsets;                // 1: the segment index chunks hold byte offsets (index * 8),
                                           // possible when (n_ucol + 1) * 8 fits 16 bits
  const uint8_t* msa;                      // [n_seqs][n_prune]
  // the same states as BIT PLANES for K1's assembly walk -- [n_seqs][ceil(n_prune / 128)][2 site sets][2 or 3] 64-bit masks:
  // bit l of plane (row i, block b, set s, bit q) is bit q of the state of row i at pattern 128 b + 64 s + l (patterns past
  // the last one repeat it); alignments that mix N with bases (msa_mixed_n) carry a third plane per set flagging N (state
  // bits 0 there).  A wave fetches the 32 / 48 bytes of (row, its block) with one / two scalar loads.
  const uint64_t* msa_planes;
  const int32_t* site_pat;                 // [n_sites] pattern of alignment site j (n_prune = the all-N pattern)
  const int32_t* u_pat;                    // [n_ucol] pattern of u-column u
  const uint8_t* u_base;                   // [n_ucol] its naive base (4 = N; 0xff: no xMSA column is this pair)
  const int32_t* ucol_of_col;              // [n_xmsa] u-column of the caller's column c
  const int32_t* col_of_ucol;              // [n_ucol] one caller column per u-column (-1: none)
  DevSegments vpadding, vgerm, dgerm, jgerm, jpadding;
  const double *vgerm_gene_prob, *vpadding_transition, *vgerm_trans_prod, *jpadding_transition;
  DevJunction vd, dj;
  int32_t max_genes;      // max over regions of the gene count
  int32_t n_jcols;        // distinct u-columns referenced by the junction tables
  const int32_t* jcols;   // [n_jcols] their u-column indices, ascending
  int64_t gem_size;       // doubles per sample of germline/padding emission products (2nV + nD + 2nJ)
  int64_t forward_size;   // doubles per sample in the compact forward output
  int64_t scaler_size;    // ints per sample in the scaler-count output
};

// 1/v to ~1 ulp from the hardware estimate and two Newton steps (normal-range v; a quarter of the IEEE division
// sequence).
__device__ static inline double fast_rcp(double v) {
  double r = __builtin_amdgcn_rcp(v);
  r = fma(fma(-v, r, 1.0), r, r);
  r = fma(fma(-v, r, 1.0), r, r);
  return r;
}

// K4 (lh_sample.hip): what a backward sampling step needs of one junction, in the unfused form FillTransition
// multiplies it together (src/HMM.cpp:964-1089), so that every weight has the bits of the dense matrices.
// Unpadded [W][nL] / [W][nR] tables; dense = index in the reference's junction state vector.
struct DevSampleJunction {
  int32_t n_rows, n_left, n_right, n_states;
  int32_t right_first_block;     // 1: the right genes' states precede the left genes' in the dense vector
  const int32_t* state_class;    // [S] kind (0 left, 1 NTI, 2 right germline) | NTI base << 2 | gene << 4
  const int32_t* left_rows;      // [nL] the gene has states on rows 0 .. left_rows - 1
  const int32_t* left_dense;     // [nL] dense index of its row-0 state
  const double* left_lo;         // [W][nL] landing_out of the row-i state
  const double* left_trans;      // [W][nL] transition into the row-i state (row 0: out of the germline region)
  const double* enter_lo;        // [nL] landing_out of the last germline-region position
  const int32_t* right_dense;    // [nR] dense index of the gene's NTI state A
  const int32_t* right_first;    // [nR] first row with a germline state of the gene (W: none)
  const double* gp;              // [nR] gene_prob
  const double* nli;             // [nR][4] nti_landing_in
  const double* ntt;             // [nR][4][4] nti_transition a -> b
  const double* nlo;             // [W][nR][4] nti_landing_out into the row-i germline state
  const double* li;              // [W][nR] landing_in of the row-i germline state
  const double* rtrans;          // [W][nR] transition into the row-i germline state from the row before
  const double* exit_nlo;        // [nR][4] nti_landing_out into the germline region x prod
  const double* exit_trans;      // [nR] transition into the germline region x prod (0: no last-row state)
  const double* exit_li;         // [nR] landing_in of the germline region's first position
  const double* prod;            // [nR] product of the in-region transitions (src/HMM.cpp:872-876)
};

struct DevSampler {
  int32_t has_d, n_v, n_d, n_j, states_per_sample, words_per_sample;
  DevSampleJunction vd, dj;
};

// K4: states[n][states_per_sample] from the compact forward arrays fwd[n][forward_size] and each sample's
// std::mt19937 outputs words[n][words_per_sample].
// smp_dev: the device copy of smp (the kernel reads the table addresses from it instead of holding forty of them in
// scalar registers).
void launch_sample(const DevSampler& smp, const DevSampler* smp_dev, int n, const double* fwd, size_t forward_size, const uint32_t* words,
                   int words_per_sample, int32_t* states, hipStream_t stream);

// P = I + U expm1(lambda * t*r) Uinv, clamped at 0 (K1's prologue).
// e: lambda[4] | U[4][4] | Uinv[4][4]
// (mode 0 is the stationary one, eigenvalue 0 -- K0a orders them so -- and contributes nothing: three modes are summed)
__device__ static inline void compute_pmatrix(const double* __restrict__ e, double tr, double P[4][4]) {
  double ex[4];
#pragma unroll
  for (int k = 1; k < 4; ++k) ex[k] = expm1(e[k] * tr);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double v = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 1; k < 4; ++k) v = fma(e[4 + i * 4 + k] * ex[k], e[20 + k * 4 + j], v);
      P[i][j] = fmax(v, 0.0);
    }
}

// ---- kernel launchers (each enqueues on `stream`, no synchronisation) -------------------------

// K0a: per sample: discrete-Gamma mean rates from alpha, GTR eigendecomposition.
// eig layout per sample: lambda[4] | U[4][4] | Uinv[4][4]  (36 doubles)
void launch_model_setup(int n, int R, const double* er, const double* pi, const double* alpha,
                        double* rates, double* eig, hipStream_t stream);

// K1: Felsenstein pruning over the MSA sites with the naive tip factored out; each workgroup first
// computes the P-matrices of its (sample, rate): P = I + U expm1(lambda t r) U^-1 for every branch,
// inner-branch matrices in schedule order into the scratch area pmat[n][R][T-2][2][16] (op k: [0] =
// matrix of the child whose CLV is in the accumulator, [1] = matrix of the popped child), tip-branch
// matrices into its LDS tip table.
// site_lik[n][R][5][n_prune], site_scal[n][R][n_prune]
// Returns the number of rate planes left in site_lik / site_scal: R, or 1 if the rates were mixed in K1
// (site_lik[n][1][5][n_prune], site_scal[n][1][n_prune]); K2a is to be run with that count.
int launch_prune(const DevFamily& fam, int n, int R, int T, int max_depth, const int32_t* ops,
                 const double* brlen, const double* rates, const double* eig, const PruneWs& ws, const double* pi,
                 double* site_lik, int32_t* site_scal, hipStream_t stream, bool allow_fused = true);
// the kernel form the calling thread's last launch_prune chose ("w6<3,false>", "seg4<4,true>", "ct6<16,false,false,true>":
// kernel<depth, N-aware, all rates in one workgroup, assembly walk>) and, after a -1, why it failed
const char* prune_last_form();
const char* prune_last_error();

// GTR eigendecomposition only (K0a's last role): eig[n][36]
void launch_gtr_setup(int n, const double* er, const double* pi, double* eig, hipStream_t stream);

// K3: ancestral-sequence sampling (lh_asr.hip).  site_lik / site_scal are K1's UNMIXED per-rate planes
// (launch_prune with allow_fused = false); clv[n][T-2][2][asr_slots(L, R)][2] is scratch; anc[n][T-2][n_sites] receives the
// sampled state of inner node T + i at every site, rate_choice[n][n_sites] the drawn category (K3a -> K3b).
// Returns nonzero if the tree is too large for the kernel's LDS tables.
int launch_asr(const DevFamily& fam, int n, int R, int T, const int32_t* ops, const double* brlen, const double* rates,
               const double* eig, const double* pi, const double* site_lik, const int32_t* site_scal,
               const uint8_t* naive, uint64_t seed, uint64_t sample0, double* clv, void* desc, uint8_t* anc,
               uint8_t* rate_choice, const int4* hdr /* K0c's verdicts, PruneWs::hdr */, hipStream_t stream);
size_t asr_desc_bytes(int T);  // per sample, of the schedule descriptors `desc` (scratch, K3s -> K3b)
size_t asr_lds_bytes(int T, int L, int R, int n_prune);
size_t asr_slots(int L, int R);  // slots per sample in K3's CLV area: clv[n][T-2][2][asr_slots] double2

// K2a + K2b.  site_lik != null: emissions are assembled from K1's output (rate mix and naive
// correction; optionally written to em_out[n][C]); site_lik == null: emissions are taken from
// em_in[n][C].  K2a leaves the germline/padding emission products in gem[n][gem_size], their scaler
// counts in gcnt[n][3] and the junction columns' emissions in jem[n][n_jcols]; K2b runs the scaled
// forward sweep over them -> loglik[n] (+ optional forward rows and scaler counts).
// dxf[n][32], dxc[n]: scratch between the two K2b kernels of the pair form (lh_forward.hip).
// extended: the opt-in extended-range mode (include/linearham_amd.h, lh_family_set_extended_range); jrs[n][rows
// of both junctions] is then the K2a -> K2b hand-off of the junction rows' emission scaler counts.
// fam_dev: the device copy of fam (K2a reads the descriptor from memory instead of taking it by value).
void launch_forward(const DevFamily& fam, const DevFamily* fam_dev, int n, int R, const double* site_lik,
                    const int32_t* site_scal, const double* pi, const double* em_in, double* em_out, double* gem,
                    int32_t* gcnt, double* jem,
                    int32_t* jrs, double* dxf, int32_t* dxc, double* loglik, double* forward_out, int32_t* scaler_out,
                    bool extended, hipStream_t stream);
size_t forward_lds_bytes(const DevFamily& fam);

// Every environment switch of the device library in one place: hooks that tests/ use to push a family onto a kernel
// form another shape takes by itself (timing experiments are built from a copy of csrc/, tools/build_asm_variant.sh);
// none of them part of the C ABI.  Read once per process (the first call of debug_options(), lh_capi.hip); a switch that is not set leaves the
// product behaviour.
struct DebugOptions {
  int chunk = 49152;           // LH_CHUNK=<n>: tree samples per launch group (tests: several groups inside one small call)
  int host_sub = 12288;        // LH_HOST_SUB=<n>: tree samples per staging sub-chunk of lh_eval_batch (host pointers)
  // (LH_K2A_DIRECT -- K2a walks every gene factor by factor, no consensus form -- is a property of a family and is read
  // when one is created: upload_consensus, lh_capi.hip)
  bool k2b_no_pair = false;    // LH_K2B_NO_PAIR: K2b with one sample per wave
  bool k2b_vd_single = false;  // LH_K2B_VD_SINGLE: one sample per V-D wave
  bool sample_timing = false;  // LH_SAMPLE_TIMING: stage times of every lh_eval_sample_batch call on stderr
  int k1_tile_cap = 0;         // LH_K1_TILE_CAP=<sites>: small K1 tiles, so that small families run the multi-tile path
  bool k1_cxx_walk = false;    // LH_K1_CXX_WALK: the cherry-table form with its C++ walk instead of the assembly one
  bool k1_tables = false;      // LH_K1_TABLES: the cherry-table form for large trees too (which take the segmented register-stack form)
  bool k1_stack = false;       // LH_K1_STACK: the register-stack form for fused shapes (which take the cherry-table form by themselves)
  bool k1_no_tables = false;   // LH_K1_NO_TABLES: the cherry-table form's kernels without tables
  bool k1_segments = false;    // LH_K1_SEGMENTS: the segmented tip table (large trees) on small trees too
  int k1_seg_waves = 4;        // LH_K1_SEG_WAVES=<4|5>: register budget of the segmented kernels
  bool k1_no_fuse = false;     // LH_K1_NO_FUSE: one workgroup per (sample, rate)
};
const DebugOptions& debug_options();

}  // namespace lh
