// SimpleHMM (class surface of src/SimpleHMM.hpp): star-tree emissions taken straight from the partis
// HMM emission tables (no phylogeny).  The emission tables are sample-invariant family constants and
// are assembled on the host; the forward pass runs on the GPU through lh_forward_batch.
#ifndef LINEARHAM_SIMPLEHMM_
#define LINEARHAM_SIMPLEHMM_

#include <memory>
#include <string>
#include <vector>

#include "HMM.hpp"

namespace linearham {

class SimpleHMM : public HMM {
 private:
  std::vector<double> em_;  // per-"column" emission factors consumed by the device forward pass
  SegmentTables vpad_t_, vger_t_, dger_t_, jger_t_, jpad_t_;
  JunctionTables vd_t_, dj_t_;
  std::vector<double> gene_prob_t_, trans_prod_t_;
  void CreateFamily();
  void InitializeEmission();
  void RunForwardAlgorithm() override;

 public:
  SimpleHMM(const std::string& yaml_path, int cluster_ind, const std::string& hmm_param_dir, int seed);
};

typedef std::shared_ptr<SimpleHMM> SimpleHMMPtr;

}  // namespace linearham

#endif  // LINEARHAM_SIMPLEHMM_
