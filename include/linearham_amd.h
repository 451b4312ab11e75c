/* linearham_amd.h -- C ABI of the MI355X phylo-HMM log-likelihood hot path.
 *
 * This is the drop-in boundary for linearham's per-tree-sample evaluation.  The reference has no
 * FFI layer; the seam these entry points replace is the libptpll `pt::pll::Partition` interface
 * plus the forward-pass free functions, as called from PhyloHMM/HMM member functions
 * (citations are file:line into matsengrp/linearham):
 *
 *   lh_family_create      <- PhyloHMM::InitializeXmsaStructs (src/PhyloHMM.cpp:45-89) +
 *                            HMM::InitializeTransition (src/HMM.cpp:190-246): everything that is
 *                            constant for one clonal family, uploaded once.
 *   lh_schedule_tree      <- pt::pll::GetVirtualRoot + the traversal order built inside
 *                            Partition::TraversalUpdate(root, FULL) (src/PhyloHMM.cpp:224-225).
 *   lh_eval_batch[_device]<- pll_compute_gamma_cats (src/PhyloHMM.cpp:425-426) +
 *                            PhyloHMM::InitializePhyloEmission (src/PhyloHMM.cpp:366-383: Partition
 *                            ctor, TraversalUpdate, LogLikelihood, naive correction, exp, the five
 *                            FillGermlinePaddingEmission and two FillJunctionEmission calls) +
 *                            HMM::LogLikelihood / RunForwardAlgorithm (src/HMM.cpp:254-287,345-354),
 *                            for a whole batch of RevBayes tree samples at once.
 *   lh_forward_batch      <- HMM::LogLikelihood on caller-supplied emissions (SimpleHMM,
 *                            src/SimpleHMM.cpp:26-39 + src/HMM.cpp:345-354).
 *   lh_asr_batch[_device] <- the per-tree body of scripts/run_bootstrap_asr_ess.R:48-104
 *                            (phylomd::phylo.likelihood per rate, rate draw, phylomd::asr.sim).
 *
 * All functions return 0 on success and a nonzero status otherwise; lh_last_error() gives the
 * message (the C++ host wrapper turns it into std::runtime_error, mirroring
 * src/linearham.cpp:447-454).  Plain pointers and sizes only.  A handle is bound to the HIP device
 * that was current at lh_family_create and must be used by one host thread at a time.
 */
#ifndef LINEARHAM_AMD_H_
#define LINEARHAM_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LH_ABI_VERSION 1

typedef struct lh_family lh_family; /* opaque: device-resident family constants + workspaces */

/* One running product per gene over its xMSA columns
 * (PhyloHMM::FillGermlinePaddingEmission, src/PhyloHMM.cpp:158-193).  Genes are in std::map
 * (sorted gene name) order, as `*_ggene_ranges_` is iterated (src/PhyloHMM.cpp:169). */
typedef struct {
  int32_t n_genes;
  const int32_t* offsets;   /* [n_genes + 1] into xmsa_inds */
  const int32_t* xmsa_inds; /* [offsets[n_genes]] */
} lh_segments;

/* One junction region (V-D, D-J or V-J) in structured form: exactly the nonzero pattern that
 * FillTransition (src/HMM.cpp:964-1089) writes into the dense germline->junction, junction->junction
 * and junction->germline matrices, indexed by (row = junction site, gene).  "left" genes are the
 * genes whose 3' end lies in the junction (V in V-D), "right" genes own the four NTI states and
 * their 5' germline positions (D in V-D).  A zero/-1 entry means "no such state at this row". */
typedef struct {
  int32_t n_rows;            /* W = junction sites */
  int32_t n_left;            /* nL */
  int32_t n_right;           /* nR */
  const double* enter_trans; /* [nL] transition[last germline-region idx] if a row-0 state exists */
  const double* enter_lo;    /* [nL] landing_out[last germline-region idx] */
  const double* left_trans;  /* [W][nL] transition[p-1] into the row-i state (row 0: 0) */
  const double* left_lo;     /* [W][nL] landing_out[p] of the row-i state */
  const int32_t* left_xmsa;  /* [W][nL] xMSA column of the row-i state, or -1 */
  const double* right_gp_nli;/* [nR][4] gene_prob * nti_landing_in[b] */
  const double* right_ntt;   /* [nR][4][4] nti_transition[b_from][b_to] */
  const double* right_nlo;   /* [W][nR][4] nti_landing_out[b_from][q] of the row-i germline state */
  const double* right_trans; /* [W][nR] transition[q-1] when rows i-1 and i both hold a state */
  const double* right_gp_li; /* [W][nR] gene_prob * landing_in[q] of the row-i germline state */
  const int32_t* right_xmsa; /* [W][nR] xMSA column of the row-i germline state, or -1 */
  const int32_t* nti_xmsa;   /* [W][nR][4] emission column of NTI base b of right gene r at row i
                              * (PhyloHMM: the same for every r; SimpleHMM: per-gene nti_emission) */
  const double* exit_nlo;    /* [nR][4] nti_landing_out[b][q0] * prod(in-region transitions) */
  const double* exit_trans;  /* [nR]    transition[q0-1] * prod (0 if no last-row state) */
  const double* exit_gp_li;  /* [nR]    gene_prob * landing_in[q0] * prod */
} lh_junction;

/* Everything that is constant for one clonal family (host pointers; copied to the device). */
typedef struct {
  int32_t abi_version;    /* LH_ABI_VERSION */
  int32_t has_d;          /* 1: igh (V-D and D-J junctions); 0: igk/igl (single V-J junction in `vd`) */
  int32_t n_seqs;         /* n: MSA rows = tips other than `naive` (0 => forward-only family) */
  int32_t n_sites;        /* L: MSA columns */
  const uint8_t* msa;     /* [n_seqs][L], A,C,G,T,N = 0..4 (HMM::msa_, src/HMM.cpp:71-83) */
  int32_t n_xmsa;         /* C: xMSA columns */
  const int32_t* xmsa_site;       /* [C] MSA site of each xMSA column */
  const uint8_t* xmsa_naive_base; /* [C] naive base 0..4 of each xMSA column (xmsa_ row 0) */
  lh_segments vpadding, vgerm, dgerm, jgerm, jpadding;
  const double* vgerm_gene_prob;     /* [nV] */
  const double* vpadding_transition; /* [nV] (HMM::vpadding_transition_) */
  const double* vgerm_trans_prod;    /* [nV] prod transition[germ_ind_start ... ) (src/HMM.cpp:310-313) */
  const double* jpadding_transition; /* [nJ] */
  lh_junction vd, dj;
} lh_family_desc;

/* Optional per-sample outputs of an evaluation (any pointer may be NULL). Host pointers for
 * lh_eval_batch / lh_forward_batch, device pointers for lh_eval_batch_device. */
typedef struct {
  double* rates;          /* [n][R]   discrete-Gamma category rates (PhyloHMM::sr_) */
  double* xmsa_emission;  /* [n][C]   PhyloHMM::xmsa_emission_ */
  double* forward;        /* [n][lh_forward_size()] compact forward arrays, see lh_forward_layout */
  int32_t* scaler_counts; /* [n][lh_scaler_size()]  vgerm, vd rows..., dgerm, dj rows..., jgerm */
} lh_eval_outputs;

const char* lh_last_error(void);
int lh_device_count(void);

/* Optional helpers for a host that wants its start-up and its copies off the critical path (no reference
 * counterpart): lh_warmup() initialises the HIP runtime and the current device's context (callable from a side
 * thread while the caller parses its inputs); lh_host_alloc / lh_host_free hand out page-locked host memory,
 * which the host-pointer entry points copy to and from at full PCIe rate (any host pointer is accepted). */
int lh_warmup(void);
/* Makes `device` (0 .. lh_device_count() - 1) the calling thread's current device (hipSetDevice, for hosts that do
 * not link the HIP runtime themselves).  A handle belongs to the device that is current when lh_family_create
 * runs, and every entry point that takes a handle switches to that device for its duration: a host with several
 * GPUs creates one handle per device and drives each from its own thread (tree samples dealt
 * i mod N; reference loop src/PhyloHMM.cpp:414-442). */
int lh_set_device(int32_t device);
void* lh_host_alloc(size_t bytes);
void lh_host_free(void* p);

int lh_family_create(const lh_family_desc* desc, lh_family** out);
void lh_family_destroy(lh_family* fam);

/* Number of doubles / ints per sample in lh_eval_outputs.forward / .scaler_counts.
 * forward layout: vgerm[nV] | vd rows i<W: left[nL], nti[nR][4], right[nR] | dgerm[nD] |
 *                 dj rows likewise | jgerm[nJ]   (dgerm/dj absent when has_d == 0). */
int64_t lh_forward_size(const lh_family* fam);
int64_t lh_scaler_size(const lh_family* fam);

/* What lh_family_create reduced the family to: the number of distinct alignment columns (site patterns;
 * the pruning kernel evaluates each once) and of distinct (naive base, pattern) pairs among the xMSA
 * columns (the emission kernels evaluate each once).  Either pointer may be NULL.  Results are per
 * xMSA column / per site all the same. */
int lh_family_info(const lh_family* fam, int32_t* n_patterns, int32_t* n_unique_columns);

/* Which germline / padding sets lh_family_create put into consensus form (bit 0 vpadding, 1 vgerm, 2 dgerm,
 * 3 jgerm, 4 jpadding): when the alleles of a set are site-aligned and alike, a gene's emission product
 * (FillGermlinePaddingEmission, src/PhyloHMM.cpp:158-193) is formed from the prefix products of the set's
 * consensus columns and the few factors where the gene departs from it, instead of factor by factor; values
 * agree to rounding, ScaleMatrix counts exactly.  Environment variable LH_K2A_DIRECT (read at create time)
 * turns the form off. */
int lh_family_consensus_sets(const lh_family* fam);

/* Diagnostic: the pruning-kernel form the handle's last evaluation ran, as "<kernel><stack depth, N-aware[, all rates
 * in one workgroup, assembly walk]>" -- e.g. "w6<3,false>", "seg4<4,true>", "ct6<16,false,false,true>" -- or "" before
 * the first one.  Which form a family takes is a function of its shape (tips, site patterns, rates, stack depth, N
 * inside alignment columns); the parity tests assert that every form is reached by a family that is compared with the
 * oracle.  The string lives as long as the handle and changes with the next evaluation. */
const char* lh_family_prune_form(const lh_family* fam);

/* Opt-in extended-range mode (default off = the reference's arithmetic, overflows included).  The reference
 * loses a tree sample in two places: exp(lnL - log pi) underflows to 0 when a column's likelihood is below
 * 1e-308 (src/PhyloHMM.cpp:237), and the 2^(256 d) equalisation of a region's emission products to the LARGEST
 * ScaleMatrix count overflows to inf when two alleles' counts differ by 4 or more (src/PhyloHMM.cpp:190-192;
 * acknowledged at scripts/run_bootstrap_asr_ess.R:37-39).  With the mode on, emissions are carried as
 * (value, 2^-256 count) pairs into the products and the junction rows, a region is equalised to its SMALLEST
 * count (negligible alleles underflow to 0 instead of likely ones overflowing), and a forward row is rescaled
 * by its largest entry instead of its smallest positive one.  Every such step is an exact power-of-two
 * rescaling, so the log-likelihood equals the default mode's wherever that is finite (tests: 1e-10) and stays
 * finite where the reference returns inf / NaN; the forward arrays and scaler counts of lh_eval_outputs are
 * then in this mode's scaling (value x 2^(-256 count) is what agrees), which is the documented divergence. */
int lh_family_set_extended_range(lh_family* fam, int enable);

/* ---- naive-sequence sampling on the device (HMM::SampleNaiveSequence's draws, src/HMM.cpp:323-341,358-431,
 * 1222-1353) ----
 * One junction in the unfused form FillTransition (src/HMM.cpp:964-1089) multiplies together, so that the
 * sampler's weights carry the bits of the reference's dense transition matrices.  [W][nL] / [W][nR] tables,
 * unpadded; "dense" = index in the junction's state vector (HMM::*_junction_state_strs_). */
typedef struct {
  int32_t n_rows, n_left, n_right, n_states;
  const int32_t* left_rows;    /* [nL] junction rows the gene has states on (rows 0 .. left_rows-1) */
  const int32_t* left_dense;   /* [nL] dense index of its row-0 state */
  const double* left_lo;       /* [W][nL] landing_out[p] of the row-i state */
  const double* left_trans;    /* [W][nL] transition[p-1] into the row-i state; row 0: out of the germline region */
  const double* enter_lo;      /* [nL] landing_out of the last germline-region position */
  const int32_t* right_dense;  /* [nR] dense index of the gene's NTI state A */
  const int32_t* right_first;  /* [nR] first row with a germline state of the gene (W: none) */
  const double* gene_prob;     /* [nR] */
  const double* nti_landing_in;  /* [nR][4] */
  const double* nti_transition;  /* [nR][4][4] from a to b */
  const double* nti_landing_out; /* [W][nR][4] into the row-i germline state */
  const double* landing_in;    /* [W][nR] of the row-i germline state */
  const double* right_trans;   /* [W][nR] transition[q-1] into the row-i germline state from the row before */
  const double* exit_nlo;      /* [nR][4] nti_landing_out[b][q0] * prod */
  const double* exit_trans;    /* [nR] transition[q0-1] * prod (0 if the gene has no last-row state) */
  const double* exit_li;       /* [nR] landing_in[q0] */
  const double* prod;          /* [nR] product of the in-region transitions (src/HMM.cpp:872-876) */
} lh_sampler_junction;

typedef struct {
  lh_sampler_junction vd, dj; /* dj unused when has_d == 0 */
} lh_sampler_desc;

/* Registers the sampler tables of a family (copied to the device).  Fails if the genes of a junction are not laid
 * out as two blocks in its state vector (all left genes before all right genes or the reverse: true of every IG
 * locus, whose gene names sort by segment). */
int lh_family_set_sampler(lh_family* fam, const lh_sampler_desc* desc);

/* std::mt19937 outputs one sample consumes (two per draw; a draw per junction row and per germline region with
 * more than one allele) and ints per sample in `states`. */
int32_t lh_sample_words(const lh_family* fam);
int32_t lh_sample_states(const lh_family* fam);

/* lh_eval_batch followed by the draws of SampleNaiveSequence for every sample, the forward arrays staying on the
 * device.  words [n][lh_sample_words()]: each sample's slice of the engine's output stream, in the order the
 * reference's loop would consume it.  states [n][lh_sample_states()]: J gene | D-J junction rows 0..W-1 | D gene |
 * V-D junction rows | V gene (light chains: J gene | V-J rows | V gene), as indices into the reference's state
 * vectors -- the values HMM::*_state_ind_samp(s)_ take.  rates [n][R] may be NULL. */
int lh_eval_sample_batch(lh_family* fam, int32_t n, int32_t n_tips, int32_t max_depth, const int32_t* ops,
                         const double* brlen, const double* er, const double* pi, const double* alpha,
                         int32_t num_rates, const uint32_t* words, double* loglik, double* rates, int32_t* states);

/* The same with every array resident on the handle's device (words, loglik, rates [may be NULL], states too);
 * enqueued on `hip_stream` without synchronising.  What a host calls that keeps its tree samples on the GPU. */
int lh_eval_sample_batch_device(lh_family* fam, int32_t n, int32_t n_tips, int32_t max_depth, const int32_t* ops,
                                const double* brlen, const double* er, const double* pi, const double* alpha,
                                int32_t num_rates, const uint32_t* words, double* loglik, double* rates,
                                int32_t* states, void* hip_stream);

/* Tree in rooted-at-naive form: tips are nodes 0..T-1 (0 = `naive`, i = MSA row i-1), inner nodes
 * T..2T-3.  children[2*(v-T)+{0,1}] are the two children of inner node v when the tree is rooted at
 * `root`, the inner node adjacent to `naive`.  Writes the kernel's post-order schedule:
 * ops[4*k+{0..3}] for k < T-2 (the last op computes the root; word 0 = kind | flags | running matrix count,
 * words 1-2 = children, word 3 = stack slot: an internal format, validated by lh_eval_batch).  *max_depth receives the number of
 * stack slots the schedule needs.  Pure host integer work. */
int lh_schedule_tree(int32_t n_tips, const int32_t* children, int32_t root, int32_t* ops,
                     int32_t* max_depth);

/* Evaluate n tree samples (host pointers).
 *   ops    [n][T-2][4]  schedules from lh_schedule_tree
 *   brlen  [n][2T-2]    branch length above each node (root entry ignored)
 *   er [n][6] (AC,AG,AT,CG,CT,GT), pi [n][4], alpha [n]; num_rates = R
 *   loglik [n]          HMM::LogLikelihood() per sample */
int lh_eval_batch(lh_family* fam, int32_t n, int32_t n_tips, int32_t max_depth, const int32_t* ops,
                  const double* brlen, const double* er, const double* pi, const double* alpha,
                  int32_t num_rates, double* loglik, const lh_eval_outputs* outs);

/* Same with every array already resident on the handle's device; enqueued on `hip_stream`
 * (a hipStream_t, NULL = default stream) without synchronising.  Device-resident schedules are not trusted: a
 * kernel (K0c) checks every op on the device before anything is indexed with it; a malformed schedule leaves NaN
 * in that sample's results and raises the handle's error word, which lh_family_status reports.  (The reference
 * checks nothing here: src/PhyloHMM.cpp:421 uses the parsed tree unchecked.) */
int lh_eval_batch_device(lh_family* fam, int32_t n, int32_t n_tips, int32_t max_depth,
                         const int32_t* ops, const double* brlen, const double* er, const double* pi,
                         const double* alpha, int32_t num_rates, double* loglik,
                         const lh_eval_outputs* outs, void* hip_stream);

/* Synchronises the handle's device, then reports and clears its asynchronous error state: nonzero (message in
 * lh_last_error) if a launch since the previous call met a malformed schedule.  The host-pointer entry points
 * call it themselves before they return. */
int lh_family_status(lh_family* fam);

/* Forward pass only, on caller-supplied per-column emissions em[n][C] (host pointers). */
int lh_forward_batch(lh_family* fam, int32_t n, const double* em, double* loglik,
                     const lh_eval_outputs* outs);

/* Ancestral-sequence sampling (scripts/run_bootstrap_asr_ess.R:48-104) for n tree samples of the family:
 * per alignment site, draw a rate category with the likelihoods of the column (naive base on the `naive`
 * tip) on the rate-scaled trees, then draw the states of all inner nodes jointly given the tips on the
 * chosen tree.
 *   ops, brlen, er, pi   as for lh_eval_batch
 *   rates  [n][R]        the site rates of the sample (the sr[] columns of the pipeline output)
 *   naive  [n][L]        the sample's NaiveSequence, A,C,G,T,N = 0..4
 *   seed, first_sample   random numbers are Philox4x32-10 with key = seed and counter =
 *                        (site, draw, first_sample + i): draw 0 = rate category, 1 = root (naive's
 *                        neighbour), 2 + (v - T) = inner node v; a draw picks the first category whose
 *                        running weight sum exceeds u * total
 *   anc    [n][T-2][L]   state 0..3 of inner node T + i (lh_schedule_tree numbering) at every site
 *   rate_choice [n][L]   drawn category per site (may be NULL)
 * A sample whose (device-resident) schedule is rejected gets 0xff in every byte of its anc and rate_choice rows --
 * bytes have no NaN -- and raises the handle's error word (lh_family_status).
 * The extra root node that ape::root(..., resolve.root = TRUE) puts on the naive branch (:53) lies at
 * distance 0 from naive's neighbour and has that node's state.  Tips keep their observed characters. */
int lh_asr_batch(lh_family* fam, int32_t n, int32_t n_tips, int32_t max_depth, const int32_t* ops,
                 const double* brlen, const double* er, const double* pi, const double* rates,
                 int32_t num_rates, const uint8_t* naive, uint64_t seed, uint64_t first_sample,
                 uint8_t* anc, uint8_t* rate_choice);

/* Same with every array resident on the handle's device; enqueued on `hip_stream` without synchronising. */
int lh_asr_batch_device(lh_family* fam, int32_t n, int32_t n_tips, int32_t max_depth, const int32_t* ops,
                        const double* brlen, const double* er, const double* pi, const double* rates,
                        int32_t num_rates, const uint8_t* naive, uint64_t seed, uint64_t first_sample,
                        uint8_t* anc, uint8_t* rate_choice, void* hip_stream);

/* Timing of the kernels of the last lh_eval_batch_device call sequence, measured with HIP events
 * on the launch stream when enabled (ms per kernel family: model, prune, forward). */
int lh_profile_enable(lh_family* fam, int enable);
int lh_profile_read(lh_family* fam, double* ms_model, double* ms_prune, double* ms_forward,
                    int64_t* n_launches);

/* Time of the sampling kernel (K3) over the lh_asr_batch_device launches made while profiling was
 * enabled (HIP events on the launch stream); resets the counters. */
int lh_asr_profile_read(lh_family* fam, double* ms_sampling, int64_t* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* LINEARHAM_AMD_H_ */
