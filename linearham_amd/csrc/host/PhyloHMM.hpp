// PhyloHMM of the MI355X-native linearham host (class surface of src/PhyloHMM.hpp:23-126).
// Per tree sample the reference builds a libpll partition, prunes every xMSA column, fills the
// emission matrices and runs the forward algorithm on one CPU core; here all of that is ONE batched
// call into the HIP library (lh_eval_batch) and this class only prepares inputs / unpacks outputs.
#ifndef LINEARHAM_PHYLOHMM_
#define LINEARHAM_PHYLOHMM_

#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "HMM.hpp"
#include "newick.hpp"

namespace linearham {

class PhyloHMM : public HMM {
 private:
  MatrixXi xmsa_;
  std::vector<std::string> xmsa_labels_, xmsa_seqs_;
  int xmsa_naive_ind_ = 0;
  VectorXd xmsa_emission_;
  VectorXi vpadding_xmsa_inds_, vgerm_xmsa_inds_, dgerm_xmsa_inds_, jgerm_xmsa_inds_, jpadding_xmsa_inds_;
  MatrixXi vd_junction_xmsa_inds_, dj_junction_xmsa_inds_;
  std::vector<int32_t> xmsa_site_;       // MSA site of each xMSA column
  std::vector<uint8_t> xmsa_base_;       // naive base of each xMSA column

  int iteration_ = 0;
  double rb_loglikelihood_ = 0, prior_ = 0, alpha_ = 1.0;
  std::vector<double> er_, pi_, sr_;
  TreeArrays tree_;
  bool have_tree_ = false;
  int num_rates_ = 1;
  double lh_loglikelihood_ = 0, logweight_ = 0;
  std::string naive_sequence_;
  const std::string* pending_newick_ = nullptr;  // RunPipeline: this row's tree, already exported

  // raw device outputs of the pending evaluation (unpacked by RunForwardAlgorithm)
  std::vector<double> pending_forward_;
  std::vector<int32_t> pending_scalers_;
  double pending_loglik_ = 0;

  void InitializeXmsaStructs();
  void CreateFamily();
  void RunForwardAlgorithm() override;
  void WriteOutputHeaders(std::ofstream& outfile) const;
  void WriteOutputLine(std::ofstream& outfile) const;
  void FormatOutputLine(std::string& line, int iteration, double rb_loglikelihood, double prior, double alpha,
                        const double* er, const double* pi, const std::string& tree, const double* sr, int num_rates,
                        double lh_loglikelihood, const RowSampler& sample) const;
  struct TsvTable;
  struct TableBatch;

 public:
  PhyloHMM(const std::string& yaml_path, int cluster_ind, const std::string& hmm_param_dir, int seed);

  const MatrixXi& xmsa() const { return xmsa_; }
  const std::vector<std::string>& xmsa_labels() const { return xmsa_labels_; }
  const std::vector<std::string>& xmsa_seqs() const { return xmsa_seqs_; }
  int xmsa_naive_ind() const { return xmsa_naive_ind_; }
  const VectorXd& xmsa_emission() const { return xmsa_emission_; }
  const VectorXi& vpadding_xmsa_inds() const { return vpadding_xmsa_inds_; }
  const VectorXi& vgerm_xmsa_inds() const { return vgerm_xmsa_inds_; }
  const MatrixXi& vd_junction_xmsa_inds() const { return vd_junction_xmsa_inds_; }
  const VectorXi& dgerm_xmsa_inds() const { return dgerm_xmsa_inds_; }
  const MatrixXi& dj_junction_xmsa_inds() const { return dj_junction_xmsa_inds_; }
  const VectorXi& jgerm_xmsa_inds() const { return jgerm_xmsa_inds_; }
  const VectorXi& jpadding_xmsa_inds() const { return jpadding_xmsa_inds_; }
  int iteration() const { return iteration_; }
  double rb_loglikelihood() const { return rb_loglikelihood_; }
  double prior() const { return prior_; }
  double alpha() const { return alpha_; }
  const std::vector<double>& er() const { return er_; }
  const std::vector<double>& pi() const { return pi_; }
  const TreeArrays& tree() const { return tree_; }
  /// Discrete-Gamma category rates; computed on the device, valid after InitializePhyloEmission().
  const std::vector<double>& sr() const { return sr_; }
  double lh_loglikelihood() const { return lh_loglikelihood_; }
  double logweight() const { return logweight_; }
  const std::string& naive_sequence() const { return naive_sequence_; }

  void InitializePhyloParameters(const std::string& newick_path, const std::vector<double>& er,
                                 const std::vector<double>& pi, double alpha, int num_rates);
  /// Same with the Newick text given directly (RunPipeline rows).
  void InitializePhyloParametersFromString(const std::string& newick, const std::vector<double>& er,
                                           const std::vector<double>& pi, double alpha, int num_rates);
  void InitializePhyloEmission();
  void RunPipeline(const std::string& input_path, const std::string& output_path, int num_rates);
  /// Test entry: the state draws of SampleNaiveSequence for the current tree and parameters with the engine's
  /// outputs GIVEN (n_words >= RawDrawsPerSample(), at most 624), once by the device sampler (lh_eval_sample_batch)
  /// and once by the host sampler (HMM::SampleRow on a std::mt19937 whose state is set so that it returns exactly these
  /// words).  States as lh_eval_sample_batch lays them out.
  void SampleStatesWithWords(const uint32_t* words, int n_words, std::vector<int32_t>& device_states,
                             std::vector<int32_t>& host_states);

  /// The per-tree body of scripts/run_bootstrap_asr_ess.R:48-104 for every row of a RunPipeline output table
  /// (columns er[1..6], pi[1..4], tree, sr[1..R], NaiveSequence): per alignment site a rate category is drawn
  /// with the column likelihoods on the rate-scaled trees, then the inner-node states are drawn jointly given
  /// the tips (K3, lh_asr_batch).  Writes one Newick string per row, rooted on the naive branch as
  /// ape::root(tree, "naive", resolve.root = TRUE) does, every node annotated [&ancestral="<L bases>"] (tips:
  /// their observed sequence).  Random numbers: Philox stream `seed`, sample number = row number.
  void RunAsr(const std::string& input_path, const std::string& output_path, uint64_t seed);
  /// One annotated tree (RunAsr's output line) from the sampled states anc[(T-2)][L] of a row.
  std::string AnnotatedNewick(const TreeArrays& tree, const std::string& naive_sequence, const uint8_t* anc) const;

  /// Batched log-likelihoods of many tree samples (the GPU-native entry point RunPipeline uses).
  /// Rows are (newick, er[6], pi[4], alpha).  Returns HMM::LogLikelihood() per row.
  struct TreeSample {
    std::string newick;
    std::vector<double> er, pi;
    double alpha;
  };
  std::vector<double> LogLikelihoodBatch(const std::vector<TreeSample>& samples, int num_rates);

  /// Flattened device inputs of a batch (used by LogLikelihoodBatch and by the benchmark harness).
  struct DeviceBatch {
    int n = 0, n_tips = 0, max_depth = 0;
    std::vector<int32_t> ops;
    std::vector<double> brlen, er, pi, alpha;
  };
  /// `trees` / `exported` (optional): the parsed trees and their re-exported Newick strings, produced by the
  /// same worker threads (RunPipeline needs both per row and would otherwise redo them one by one).
  DeviceBatch FlattenBatch(const std::vector<TreeSample>& samples, std::vector<TreeArrays>* trees = nullptr,
                           std::vector<std::string>* exported = nullptr) const;
  /// The whole RevBayes table `path` as device inputs (rows parsed and scheduled by worker threads).
  DeviceBatch FlattenTsv(const std::string& path, int* n_rows) const;
  /// Only the table rows `row_ids[0..n)` (any order, repeats allowed), in that order.
  DeviceBatch FlattenTsvRows(const std::string& path, const int64_t* row_ids, int n, int* n_rows) const;
  lh_family* family() {
    CreateFamily();
    return family_;
  }
  /// Opt-in extended-range arithmetic of the device path (lh_family_set_extended_range): finite log-likelihoods
  /// where the reference's equalisation overflows or its exp underflows; off by default.
  void SetExtendedRange(bool on);
  /// The HIP devices RunPipeline uses (before the first evaluation): one family handle and one host thread per
  /// entry, table row i evaluated and sampled on devices[i mod N], output in file order -- the split of
  /// src/PhyloHMM.cpp:414-442's loop over one node's GPUs.  The same device may be listed more than once (two
  /// handles on one GPU).  Single-row members (LogLikelihood, SampleNaiveSequence ...) use devices[0].
  void SetDevices(const std::vector<int>& devices);
  int n_xmsa() const { return xmsa_.cols(); }

 private:
  TableBatch FlattenTable(const TsvTable& table, std::size_t r0, std::size_t r1, bool with_export, bool with_scalars,
                          const std::string& path) const;
};

typedef std::shared_ptr<PhyloHMM> PhyloHMMPtr;

void StoreGermlinePaddingXmsaIndices(const std::vector<int>& naive_bases, const std::vector<int>& site_inds,
                                     std::map<std::pair<int, int>, int>& xmsa_ids, VectorXi& xmsa_inds);
void StoreJunctionXmsaIndices(const std::vector<int>& naive_bases, const std::vector<int>& site_inds,
                              std::pair<int, int> left_flexbounds, std::pair<int, int> right_flexbounds,
                              std::map<std::pair<int, int>, int>& xmsa_ids, MatrixXi& xmsa_inds);
void StoreXmsaIndex(std::pair<int, int> id, std::map<std::pair<int, int>, int>& xmsa_ids, int& xmsa_ind);

}  // namespace linearham

#endif  // LINEARHAM_PHYLOHMM_
