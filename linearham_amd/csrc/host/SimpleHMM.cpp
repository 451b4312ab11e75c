#include "SimpleHMM.hpp"

namespace linearham {

// src/SimpleHMM.cpp:26-39
SimpleHMM::SimpleHMM(const std::string& yaml_path, int cluster_ind, const std::string& hmm_param_dir, int seed)
    : HMM(yaml_path, cluster_ind, hmm_param_dir, seed) {
  InitializeEmission();
  cache_forward_ = true;
}

namespace {

// FillGermlineEmission / FillPaddingEmission (src/SimpleHMM.cpp:95-139, 224-271) multiply one table
// entry per (site, sequence) into a running product with ScaleMatrix after every factor.  The same
// chain is handed to the device as a segment of indices into `em`.
struct EmissionBuilder {
  std::vector<double> em;
  int Add(double v) {
    em.push_back(v);
    return (int)em.size() - 1;
  }
};

}  // namespace

// src/SimpleHMM.cpp:47-77
void SimpleHMM::InitializeEmission() {
  const bool igh = locus_ == "igh";
  const int n = msa_.rows();
  const int N = (int)alphabet_.size() - 1;
  EmissionBuilder eb;

  auto germline_segments = [&](const RegionStates& R) {
    SegmentTables s;
    s.offsets.push_back(0);
    for (auto it = R.ggene_ranges.begin(); it != R.ggene_ranges.end(); ++it) {
      const Germline& g = *ggenes_.at(it->first).germ_ptr;
      // one em entry per (germline position, base) of this gene, created on demand
      std::map<std::pair<int, int>, int> idx;
      for (int j = it->second.first; j < it->second.second; ++j)
        for (int k = 0; k < n; ++k) {
          const int b = msa_(k, R.site_inds[j]);
          if (b == N) continue;
          const std::pair<int, int> key(R.germ_inds[j], b);
          auto f = idx.find(key);
          if (f == idx.end()) f = idx.emplace(key, eb.Add(g.emission()(b, R.germ_inds[j]))).first;
          s.xmsa_inds.push_back(f->second);
        }
      s.offsets.push_back((int32_t)s.xmsa_inds.size());
    }
    return s;
  };
  auto padding_segments = [&](const RegionStates& R) {
    SegmentTables s;
    s.offsets.push_back(0);
    for (auto it = R.ggene_ranges.begin(); it != R.ggene_ranges.end(); ++it) {
      const VectorXd& ne = ggenes_.at(it->first).npadding().n_emission();
      int idx[4] = {-1, -1, -1, -1};
      for (int j = it->second.first; j < it->second.second; ++j)
        for (int k = 0; k < n; ++k) {
          const int b = msa_(k, R.site_inds[j]);
          if (b == N) continue;
          if (idx[b] < 0) idx[b] = eb.Add(ne[b]);
          s.xmsa_inds.push_back(idx[b]);
        }
      s.offsets.push_back((int32_t)s.xmsa_inds.size());
    }
    return s;
  };
  // FillJunctionEmission (src/SimpleHMM.cpp:160-211): plain products over the sequences, no scaling.
  auto junction_indices = [&](const RegionStates& R, std::pair<int, int> left_fb, std::pair<int, int> right_fb) {
    const int site_start = left_fb.first, site_end = right_fb.second;
    MatrixXi M;
    M.setConstant(site_end - site_start, (int)R.naive_bases.size(), -1);
    for (auto it = R.ggene_ranges.begin(); it != R.ggene_ranges.end(); ++it) {
      const GermlineGene& gg = ggenes_.at(it->first);
      for (int i = it->second.first; i < it->second.second; ++i) {
        if (R.site_inds[i] == -1) {
          const MatrixXd& ne = gg.nti().nti_emission();
          for (int site = site_start; site < site_end; ++site) {
            double v = 1;
            for (int j = 0; j < n; ++j)
              if (msa_(j, site) != N) v *= ne(msa_(j, site), R.naive_bases[i]);
            M(site - site_start, i) = eb.Add(v);
          }
        } else {
          double v = 1;
          for (int j = 0; j < n; ++j)
            if (msa_(j, R.site_inds[i]) != N) v *= gg.germ_ptr->emission()(msa_(j, R.site_inds[i]), R.germ_inds[i]);
          M(R.site_inds[i] - site_start, i) = eb.Add(v);
        }
      }
    }
    return M;
  };

  vpad_t_ = padding_segments(vpadding_);
  vger_t_ = germline_segments(vgerm_);
  dger_t_.offsets.assign(1, 0);
  if (igh) {
    const MatrixXi x1 = junction_indices(vd_junction_, flexbounds_.at("v_r"), flexbounds_.at("d_l"));
    dger_t_ = germline_segments(dgerm_);
    const MatrixXi x2 = junction_indices(dj_junction_, flexbounds_.at("d_r"), flexbounds_.at("j_l"));
    vd_t_ = BuildJunctionTables(vd_junction_, vgerm_, dgerm_, flexbounds_.at("v_r"), flexbounds_.at("d_l"), x1);
    dj_t_ = BuildJunctionTables(dj_junction_, dgerm_, jgerm_, flexbounds_.at("d_r"), flexbounds_.at("j_l"), x2);
  } else {
    const MatrixXi x1 = junction_indices(vd_junction_, flexbounds_.at("v_r"), flexbounds_.at("j_l"));
    vd_t_ = BuildJunctionTables(vd_junction_, vgerm_, jgerm_, flexbounds_.at("v_r"), flexbounds_.at("j_l"), x1);
  }
  jger_t_ = germline_segments(jgerm_);
  jpad_t_ = padding_segments(jpadding_);
  if (eb.em.empty()) eb.Add(1.0);
  em_ = eb.em;
  for (auto it = vgerm_.ggene_ranges.begin(); it != vgerm_.ggene_ranges.end(); ++it) {
    const Germline& g = *ggenes_.at(it->first).germ_ptr;
    gene_prob_t_.push_back(g.gene_prob());
    const int gis = vgerm_.germ_inds[it->second.first];
    double prod = 1.0;
    for (int k = 0; k < it->second.second - it->second.first - 1; ++k) prod *= g.transition()[gis + k];
    trans_prod_t_.push_back(prod);
  }
}

// Forward-only family (n_seqs = 0): created lazily at the first evaluation (needs a GPU).
void SimpleHMM::CreateFamily() {
  if (family_) return;
  const bool igh = locus_ == "igh";
  lh_family_desc d{};
  d.abi_version = LH_ABI_VERSION;
  d.has_d = igh ? 1 : 0;
  d.n_seqs = 0;
  d.n_sites = 0;
  d.n_xmsa = (int32_t)em_.size();
  d.vpadding = vpad_t_.c();
  d.vgerm = vger_t_.c();
  d.dgerm = dger_t_.c();
  d.jgerm = jger_t_.c();
  d.jpadding = jpad_t_.c();
  d.vgerm_gene_prob = gene_prob_t_.data();
  d.vpadding_transition = vpadding_transition_.data();
  d.vgerm_trans_prod = trans_prod_t_.data();
  d.jpadding_transition = jpadding_transition_.data();
  d.vd = vd_t_.c();
  if (igh) d.dj = dj_t_.c();
  CheckHip(lh_family_create(&d, &family_), "lh_family_create");
}

void SimpleHMM::RunForwardAlgorithm() {
  CreateFamily();
  std::vector<double> fwd(lh_forward_size(family_));
  std::vector<int32_t> sco(lh_scaler_size(family_));
  lh_eval_outputs outs{nullptr, nullptr, fwd.data(), sco.data()};
  CheckHip(lh_forward_batch(family_, 1, em_.data(), &loglikelihood_, &outs), "lh_forward_batch");
  UnpackForward(fwd.data(), sco.data());
}

}  // namespace linearham
