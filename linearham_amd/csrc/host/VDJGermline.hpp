// Germline parameter model: Germline, NTInsertion, NPadding, V/D/JGermline, GermlineGene and
// CreateGermlineGeneMap (src/Germline.hpp:19-54, src/NTInsertion.hpp:18-43, src/NPadding.hpp:15-25,
// src/VDJGermline.hpp:20-66 of the reference), parsed from partis per-allele HMM YAML files.
#ifndef LINEARHAM_VDJGERMLINE_
#define LINEARHAM_VDJGERMLINE_

#include <memory>
#include <string>
#include <unordered_map>

#include "utils.hpp"

namespace linearham {

class Germline {
 protected:
  VectorXd landing_in_, landing_out_, transition_;
  double gene_prob_ = 0;
  std::string alphabet_, name_;
  MatrixXd emission_;
  VectorXi bases_;

 public:
  Germline() {}
  explicit Germline(const yaml_lite::Node& root);
  const VectorXd& landing_in() const { return landing_in_; }
  const VectorXd& landing_out() const { return landing_out_; }
  const VectorXd& transition() const { return transition_; }
  double gene_prob() const { return gene_prob_; }
  const std::string& alphabet() const { return alphabet_; }
  const std::string& name() const { return name_; }
  const MatrixXd& emission() const { return emission_; }
  const VectorXi& bases() const { return bases_; }
  int length() const { return (int)bases_.size(); }
};

class NTInsertion {
 protected:
  VectorXd nti_landing_in_;
  MatrixXd nti_landing_out_, nti_transition_, nti_emission_;

 public:
  NTInsertion() {}
  explicit NTInsertion(const yaml_lite::Node& root);
  const VectorXd& nti_landing_in() const { return nti_landing_in_; }
  const MatrixXd& nti_landing_out() const { return nti_landing_out_; }
  const MatrixXd& nti_transition() const { return nti_transition_; }
  const MatrixXd& nti_emission() const { return nti_emission_; }
};

class NPadding {
 protected:
  double n_transition_ = 0;
  VectorXd n_emission_;

 public:
  NPadding() {}
  explicit NPadding(const yaml_lite::Node& root);
  double n_transition() const { return n_transition_; }
  const VectorXd& n_emission() const { return n_emission_; }
};

class VGermline : public Germline, public NPadding {
 public:
  explicit VGermline(const yaml_lite::Node& root) : Germline(root), NPadding(root) {}
};
class DGermline : public Germline, public NTInsertion {
 public:
  explicit DGermline(const yaml_lite::Node& root) : Germline(root), NTInsertion(root) {}
};
class JGermline : public Germline, public NTInsertion, public NPadding {
 public:
  explicit JGermline(const yaml_lite::Node& root) : Germline(root), NTInsertion(root), NPadding(root) {}
};

typedef std::shared_ptr<Germline> GermlinePtr;
typedef std::shared_ptr<VGermline> VGermlinePtr;
typedef std::shared_ptr<DGermline> DGermlinePtr;
typedef std::shared_ptr<JGermline> JGermlinePtr;

enum class GermlineType { V, D, J };

struct GermlineGene {
  GermlineType type;
  GermlinePtr germ_ptr;
  VGermlinePtr VGermlinePtrCast() const;
  DGermlinePtr DGermlinePtrCast() const;
  JGermlinePtr JGermlinePtrCast() const;
  /// NTInsertion view of a D or J gene (the reference picks it with a type test at every use).
  const NTInsertion& nti() const;
  /// NPadding view of a V or J gene.
  const NPadding& npadding() const;
};

std::unordered_map<std::string, GermlineGene> CreateGermlineGeneMap(std::string hmm_param_dir);

}  // namespace linearham

#endif  // LINEARHAM_VDJGERMLINE_
