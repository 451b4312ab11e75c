// HMM base class of the MI355X-native linearham host (mirrors the class surface of the reference's
// src/HMM.hpp:23-412): cluster YAML parsing, state space, dense transition matrices (kept for the
// accessors and for sampling), and the forward-pass *results*.  The forward pass itself runs on the
// GPU through the C ABI (include/linearham_amd.h); this class has no CPU forward implementation.
#ifndef LINEARHAM_HMM_
#define LINEARHAM_HMM_

#include <map>
#include <memory>
#include <random>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "VDJGermline.hpp"
#include "linearham_amd.h"
#include "utils.hpp"

namespace linearham {

typedef std::map<std::string, std::pair<int, int>> GeneRanges;

/// State-space vectors of one HMM region (the reference keeps them as separate members
/// `<region>_state_strs_`, `<region>_naive_bases_`, ...: src/HMM.hpp:56-110).
struct RegionStates {
  std::vector<std::string> state_strs;
  std::vector<int> left_del, right_del;   // "germline" regions
  std::vector<int> del;                    // "junction" regions
  std::vector<GermlineType> ggene_types;   // "junction" regions
  GeneRanges ggene_ranges;
  std::vector<int> naive_bases, germ_inds, site_inds;
};

/// Host copies of the arrays behind lh_junction (structured form of FillTransition's output).
struct JunctionTables {
  int n_rows = 0, n_left = 0, n_right = 0;
  std::vector<double> enter_trans, enter_lo, left_trans, left_lo, right_gp_nli, right_ntt, right_nlo,
      right_trans, right_gp_li, exit_nlo, exit_trans, exit_gp_li;
  std::vector<int32_t> left_xmsa, right_xmsa, nti_xmsa;
  lh_junction c() const;
};

struct SegmentTables {
  std::vector<int32_t> offsets, xmsa_inds;
  lh_segments c() const;
};

class HMM {
 protected:
  std::string locus_;
  yaml_lite::Node cluster_data_;
  std::map<std::string, std::pair<int, int>> flexbounds_;
  std::map<std::string, int> relpos_;
  std::unordered_map<std::string, GermlineGene> ggenes_;
  std::string alphabet_;
  MatrixXi msa_;
  std::mt19937 rng_;
  std::discrete_distribution<int> distr_;

  RegionStates vpadding_, vgerm_, vd_junction_, dgerm_, dj_junction_, jgerm_, jpadding_;

  VectorXd vpadding_transition_;
  MatrixXd vgerm_vd_junction_transition_, vd_junction_transition_, vd_junction_dgerm_transition_;
  MatrixXd dgerm_dj_junction_transition_, dj_junction_transition_, dj_junction_jgerm_transition_;
  VectorXd jpadding_transition_;

  bool cache_forward_ = false;
  double loglikelihood_ = 0.0;

  VectorXd vgerm_forward_, dgerm_forward_, jgerm_forward_;
  MatrixXd vd_junction_forward_, dj_junction_forward_;
  int vgerm_scaler_count_ = 0, dgerm_scaler_count_ = 0, jgerm_scaler_count_ = 0;
  std::vector<int> vd_junction_scaler_counts_, dj_junction_scaler_counts_;

  // naive sequence sample
  std::string naive_seq_samp_;
  std::string vgerm_state_str_samp_;
  int vgerm_state_ind_samp_ = 0, vgerm_left_del_samp_ = 0, vgerm_right_del_samp_ = 0;
  std::string vgerm_left_insertion_samp_;
  std::vector<std::string> vd_junction_state_str_samps_;
  std::vector<int> vd_junction_state_ind_samps_;
  std::string vd_junction_insertion_samp_;
  std::string dgerm_state_str_samp_;
  int dgerm_state_ind_samp_ = 0, dgerm_left_del_samp_ = 0, dgerm_right_del_samp_ = 0;
  std::vector<std::string> dj_junction_state_str_samps_;
  std::vector<int> dj_junction_state_ind_samps_;
  std::string dj_junction_insertion_samp_;
  std::string jgerm_state_str_samp_;
  int jgerm_state_ind_samp_ = 0, jgerm_left_del_samp_ = 0, jgerm_right_del_samp_ = 0;
  std::string jgerm_right_insertion_samp_;

  // host-side accelerators of the sampling pass, built on first use
  std::vector<int> vd_scatter_, dj_scatter_;  // compact forward slot -> position in the dense matrix
  struct SamplingLists;
  std::unique_ptr<SamplingLists> sampling_lists_;

  // GPU side
  lh_family* family_ = nullptr;
  bool device_sampler_ = false;  // lh_family_set_sampler accepted this family's junctions
  // Several GPUs in one process (the loop that shards is src/PhyloHMM.cpp:414-442): devices_ lists the
  // HIP devices RunPipeline deals its rows to (row i -> devices_[i mod N]); family_ belongs to devices_[0] (or to the
  // calling thread's current device when the list is empty), more_families_[k] to devices_[k + 1].
  std::vector<int> devices_;
  std::vector<lh_family*> more_families_;

  void InitializeMsa();
  void InitializeStateSpace();
  void InitializeTransition();

  void EnsureSamplingLists();

 public:
  /// Everything one naive-sequence sample touches, outside the HMM object: RunPipeline's worker threads each own
  /// one and sample different rows of the table at the same time (the HMM itself is only read).
  struct RowSampler {
    VectorXd vgerm_forward, dgerm_forward, jgerm_forward;
    MatrixXd vd_junction_forward, dj_junction_forward;
    std::vector<int> vd_scatter, dj_scatter;
    std::discrete_distribution<int> distr;
    std::string naive_seq;
    std::string vgerm_state_str, dgerm_state_str, jgerm_state_str;
    int vgerm_state_ind = 0, vgerm_left_del = 0, vgerm_right_del = 0;
    int dgerm_state_ind = 0, dgerm_left_del = 0, dgerm_right_del = 0;
    int jgerm_state_ind = 0, jgerm_left_del = 0, jgerm_right_del = 0;
    std::vector<std::string> vd_junction_state_strs, dj_junction_state_strs;
    std::vector<int> vd_junction_state_inds, dj_junction_state_inds;
    std::string vgerm_left_insertion, vd_junction_insertion, dj_junction_insertion, jgerm_right_insertion;
  };
  /// HMM::SampleNaiveSequence (src/HMM.cpp:358-431) on the compact forward arrays `fwd` of one evaluation
  /// (lh_eval_outputs.forward), drawing from `rng`; const, safe to call from several threads with distinct
  /// samplers.  EnsureSamplingLists() must have run.
  void SampleRow(RowSampler& s, const double* fwd, std::mt19937& rng) const;
  /// std::mt19937 outputs one SampleNaiveSequence consumes: every draw of a discrete_distribution over two or
  /// more weights takes one generate_canonical<double, 53> = two outputs, and the number of draws per sample is
  /// fixed by the family (one per germline region with more than one allele, one per junction site).
  int RawDrawsPerSample() const;
  /// The rest of a sample once its states are known (drawn on the device, lh_eval_sample_batch): naive sequence,
  /// gene names, deletions and insertions exactly as SampleRow derives them from the same state indices.
  /// states: J gene | D-J junction rows | D gene | V-D junction rows | V gene (light chains: J | V-J rows | V).
  void ApplySampledStates(RowSampler& s, const int32_t* states) const;
  /// Sampler tables of the device path (lh_family_set_sampler) with their storage.
  struct SamplerJunction {
    int n_rows = 0, n_left = 0, n_right = 0, n_states = 0;
    std::vector<int32_t> left_rows, left_dense, right_dense, right_first;
    std::vector<double> left_lo, left_trans, enter_lo, gene_prob, nti_landing_in, nti_transition, nti_landing_out,
        landing_in, right_trans, exit_nlo, exit_trans, exit_li, prod;
    lh_sampler_junction c() const;
  };
  SamplerJunction BuildSamplerJunction(const RegionStates& J, const RegionStates& G_left, const RegionStates& G_right,
                                       std::pair<int, int> left_fb, std::pair<int, int> right_fb) const;

 protected:
  /// Runs the device forward pass if needed (pure virtual: the derived class knows how the
  /// emissions are produced) and unpacks the compact forward arrays into the dense members.
  virtual void RunForwardAlgorithm() = 0;
  void UnpackForward(const double* fwd, const int32_t* sco);
  void SampleInitialState();

  /// Structured junction tables for the C ABI. `xmsa_inds` is the W x S index matrix of the junction
  /// (row-major), or empty when the caller fills the *_xmsa arrays itself.
  JunctionTables BuildJunctionTables(const RegionStates& J, const RegionStates& G_left,
                                     const RegionStates& G_right, std::pair<int, int> left_fb,
                                     std::pair<int, int> right_fb, const MatrixXi& xmsa_inds) const;

 public:
  HMM(const std::string& yaml_path, int cluster_ind, const std::string& hmm_param_dir, int seed);
  virtual ~HMM();
  HMM(const HMM&) = delete;
  HMM& operator=(const HMM&) = delete;

  const std::string& locus() const { return locus_; }
  const std::map<std::string, std::pair<int, int>>& flexbounds() const { return flexbounds_; }
  const std::map<std::string, int>& relpos() const { return relpos_; }
  const std::unordered_map<std::string, GermlineGene>& ggenes() const { return ggenes_; }
  const std::string& alphabet() const { return alphabet_; }
  const MatrixXi& msa() const { return msa_; }

#define LH_REGION_ACCESSORS(R)                                                               \
  const GeneRanges& R##_ggene_ranges() const { return R##_.ggene_ranges; }                   \
  const std::vector<int>& R##_naive_bases() const { return R##_.naive_bases; }               \
  const std::vector<int>& R##_site_inds() const { return R##_.site_inds; }
#define LH_GERM_ACCESSORS(R)                                                                 \
  LH_REGION_ACCESSORS(R)                                                                     \
  const std::vector<std::string>& R##_state_strs() const { return R##_.state_strs; }         \
  const std::vector<int>& R##_left_del() const { return R##_.left_del; }                     \
  const std::vector<int>& R##_right_del() const { return R##_.right_del; }                   \
  const std::vector<int>& R##_germ_inds() const { return R##_.germ_inds; }
#define LH_JUNCTION_ACCESSORS(R)                                                             \
  LH_REGION_ACCESSORS(R)                                                                     \
  const std::vector<std::string>& R##_state_strs() const { return R##_.state_strs; }         \
  const std::vector<int>& R##_del() const { return R##_.del; }                               \
  const std::vector<GermlineType>& R##_ggene_types() const { return R##_.ggene_types; }      \
  const std::vector<int>& R##_germ_inds() const { return R##_.germ_inds; }
  LH_REGION_ACCESSORS(vpadding)
  LH_GERM_ACCESSORS(vgerm)
  LH_JUNCTION_ACCESSORS(vd_junction)
  LH_GERM_ACCESSORS(dgerm)
  LH_JUNCTION_ACCESSORS(dj_junction)
  LH_GERM_ACCESSORS(jgerm)
  LH_REGION_ACCESSORS(jpadding)
#undef LH_REGION_ACCESSORS
#undef LH_GERM_ACCESSORS
#undef LH_JUNCTION_ACCESSORS

  const VectorXd& vpadding_transition() const { return vpadding_transition_; }
  const MatrixXd& vgerm_vd_junction_transition() const { return vgerm_vd_junction_transition_; }
  const MatrixXd& vd_junction_transition() const { return vd_junction_transition_; }
  const MatrixXd& vd_junction_dgerm_transition() const { return vd_junction_dgerm_transition_; }
  const MatrixXd& dgerm_dj_junction_transition() const { return dgerm_dj_junction_transition_; }
  const MatrixXd& dj_junction_transition() const { return dj_junction_transition_; }
  const MatrixXd& dj_junction_jgerm_transition() const { return dj_junction_jgerm_transition_; }
  const VectorXd& jpadding_transition() const { return jpadding_transition_; }
  bool cache_forward() const { return cache_forward_; }
  const VectorXd& vgerm_forward() const { return vgerm_forward_; }
  const MatrixXd& vd_junction_forward() const { return vd_junction_forward_; }
  const VectorXd& dgerm_forward() const { return dgerm_forward_; }
  const MatrixXd& dj_junction_forward() const { return dj_junction_forward_; }
  const VectorXd& jgerm_forward() const { return jgerm_forward_; }
  int vgerm_scaler_count() const { return vgerm_scaler_count_; }
  const std::vector<int>& vd_junction_scaler_counts() const { return vd_junction_scaler_counts_; }
  int dgerm_scaler_count() const { return dgerm_scaler_count_; }
  const std::vector<int>& dj_junction_scaler_counts() const { return dj_junction_scaler_counts_; }
  int jgerm_scaler_count() const { return jgerm_scaler_count_; }
  const std::string& naive_seq_samp() const { return naive_seq_samp_; }
  const std::string& vgerm_state_str_samp() const { return vgerm_state_str_samp_; }
  int vgerm_state_ind_samp() const { return vgerm_state_ind_samp_; }
  int vgerm_left_del_samp() const { return vgerm_left_del_samp_; }
  int vgerm_right_del_samp() const { return vgerm_right_del_samp_; }
  const std::string& vgerm_left_insertion_samp() const { return vgerm_left_insertion_samp_; }
  const std::vector<std::string>& vd_junction_state_str_samps() const { return vd_junction_state_str_samps_; }
  const std::vector<int>& vd_junction_state_ind_samps() const { return vd_junction_state_ind_samps_; }
  const std::string& vd_junction_insertion_samp() const { return vd_junction_insertion_samp_; }
  const std::string& dgerm_state_str_samp() const { return dgerm_state_str_samp_; }
  int dgerm_state_ind_samp() const { return dgerm_state_ind_samp_; }
  int dgerm_left_del_samp() const { return dgerm_left_del_samp_; }
  int dgerm_right_del_samp() const { return dgerm_right_del_samp_; }
  const std::vector<std::string>& dj_junction_state_str_samps() const { return dj_junction_state_str_samps_; }
  const std::vector<int>& dj_junction_state_ind_samps() const { return dj_junction_state_ind_samps_; }
  const std::string& dj_junction_insertion_samp() const { return dj_junction_insertion_samp_; }
  const std::string& jgerm_state_str_samp() const { return jgerm_state_str_samp_; }
  int jgerm_state_ind_samp() const { return jgerm_state_ind_samp_; }
  int jgerm_left_del_samp() const { return jgerm_left_del_samp_; }
  int jgerm_right_del_samp() const { return jgerm_right_del_samp_; }
  const std::string& jgerm_right_insertion_samp() const { return jgerm_right_insertion_samp_; }

  double LogLikelihood();
  std::string SampleNaiveSequence();
};

// Free functions mirroring src/HMM.hpp:421-540.
void CacheGermlineStates(GermlinePtr germ_ptr, std::pair<int, int> left_flexbounds,
                         std::pair<int, int> right_flexbounds, int relpos, bool left_end, bool right_end,
                         RegionStates& R);
void CacheJunctionStates(const GermlineGene& ggene, std::pair<int, int> left_flexbounds,
                         std::pair<int, int> right_flexbounds, int relpos, bool left_end, RegionStates& R);
void CachePaddingStates(GermlinePtr germ_ptr, std::pair<int, int> leftright_flexbounds, int relpos,
                        bool left_end, RegionStates& R);
void ComputeGermlineJunctionTransition(const RegionStates& G, const RegionStates& J, GermlineType left_gtype,
                                       GermlineType right_gtype,
                                       const std::unordered_map<std::string, GermlineGene>& ggenes, MatrixXd& T);
void ComputeJunctionTransition(const RegionStates& J, GermlineType left_gtype, GermlineType right_gtype,
                               const std::unordered_map<std::string, GermlineGene>& ggenes, MatrixXd& T);
void ComputeJunctionGermlineTransition(const RegionStates& J, const RegionStates& G, GermlineType left_gtype,
                                       GermlineType right_gtype,
                                       const std::unordered_map<std::string, GermlineGene>& ggenes, MatrixXd& T);
void ComputePaddingTransition(const GeneRanges& ranges, const std::unordered_map<std::string, GermlineGene>& ggenes,
                              VectorXd& transition);
void FillTransition(const GermlineGene& from_ggene, const GermlineGene& to_ggene, GermlineType left_gtype,
                    GermlineType right_gtype, int germ_ind_row_start, int germ_ind_col_start,
                    int site_ind_row_start, int site_ind_col_start, int nti_row_start, int nti_col_start,
                    int nti_row_length, int nti_col_length, int germ_row_start, int germ_col_start,
                    int germ_row_length, int germ_col_length, MatrixXd& T, int row_off, int col_off);
/// The non-zero rows of every column of a dense transition matrix (the junction matrices are >99 %
/// zeros; sampling reads one column per step).
struct ColumnLists {
  std::vector<int> start;  // [cols + 1]
  std::vector<int> rows;   // ascending within a column
  static ColumnLists Build(const MatrixXd& M);
};

/// One draw of std::discrete_distribution<int> over a weight vector of length `size` of which only the
/// entries idx[0..k) (ascending) are non-zero -- the same std::mt19937 consumption and, operation for
/// operation, the same arithmetic as libstdc++ performs on the full vector (adding the zeros changes
/// neither the sum nor a partial sum), so the sampled index is the one the dense call returns.
int DrawDiscreteSparse(std::mt19937& rng, const int* idx, const double* weights, int k, int size);

/// germ_cols / junction_cols (optional): column lists of the two transition matrices; with them a step
/// costs the handful of states that can precede the sampled one instead of all S.
void SampleJunctionStates(int germ_state_ind_samp, const MatrixXd& junction_germ_transition, const RegionStates& J,
                          const MatrixXd& junction_transition, const MatrixXd& junction_forward,
                          GermlineType left_gtype, GermlineType right_gtype, std::pair<int, int> left_flexbounds,
                          const std::string& alphabet, std::mt19937& rng, std::discrete_distribution<int>& distr,
                          std::string& naive_seq_samp, int& germ_left_del_samp,
                          std::vector<std::string>& junction_state_str_samps,
                          std::vector<int>& junction_state_ind_samps, std::string& junction_insertion_samp,
                          int& germ_right_del_samp, const ColumnLists* germ_cols = nullptr,
                          const ColumnLists* junction_cols = nullptr);
void SampleGermlineState(const std::vector<int>& junction_state_ind_samps, const MatrixXd& germ_junction_transition,
                         const RegionStates& G, const VectorXd& germ_forward, const std::string& alphabet,
                         std::mt19937& rng, std::discrete_distribution<int>& distr, std::string& naive_seq_samp,
                         std::string& germ_state_str_samp, int& germ_state_ind_samp, int& germ_left_del_samp,
                         int& germ_right_del_samp);

/// Throws std::runtime_error(lh_last_error()) when a C-ABI call fails.
void CheckHip(int rc, const char* what);

}  // namespace linearham

#endif  // LINEARHAM_HMM_
