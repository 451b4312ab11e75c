#include "HMM.hpp"

#include <algorithm>
#include <limits>
#include <cmath>
#include <tuple>

namespace linearham {

void CheckHip(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + lh_last_error());
}

lh_segments SegmentTables::c() const {
  lh_segments s;
  s.n_genes = (int32_t)offsets.size() - 1;
  s.offsets = offsets.data();
  s.xmsa_inds = xmsa_inds.data();
  return s;
}

lh_junction JunctionTables::c() const {
  lh_junction j;
  j.n_rows = n_rows;
  j.n_left = n_left;
  j.n_right = n_right;
  j.enter_trans = enter_trans.data();
  j.enter_lo = enter_lo.data();
  j.left_trans = left_trans.data();
  j.left_lo = left_lo.data();
  j.left_xmsa = left_xmsa.data();
  j.right_gp_nli = right_gp_nli.data();
  j.right_ntt = right_ntt.data();
  j.right_nlo = right_nlo.data();
  j.right_trans = right_trans.data();
  j.right_gp_li = right_gp_li.data();
  j.right_xmsa = right_xmsa.data();
  j.nti_xmsa = nti_xmsa.data();
  j.exit_nlo = exit_nlo.data();
  j.exit_trans = exit_trans.data();
  j.exit_gp_li = exit_gp_li.data();
  return j;
}

// src/HMM.cpp:27-63
HMM::HMM(const std::string& yaml_path, int cluster_ind, const std::string& hmm_param_dir, int seed) {
  StageTimer timer;
  yaml_lite::Node root = yaml_lite::LoadFile(yaml_path);
  try {
    locus_ = root["germline-info"]["locus"].as_string();
    cluster_data_ = root["events"][cluster_ind];
  } catch (const std::runtime_error& e) {
    throw std::runtime_error("Can't read one of \"locus\" (under \"germline-info\"), \"events\" from  " +
                             yaml_path + " . Check that yaml file contains these keys.");
  }
  try {
    const yaml_lite::Node& info = cluster_data_["linearham-info"];
    for (const auto& kv : info["flexbounds"].map)
      flexbounds_[kv.first] = {kv.second[0].as_int(), kv.second[1].as_int()};
    for (const auto& kv : info["relpos"].map) relpos_[kv.first] = kv.second.as_int();
  } catch (const std::runtime_error& e) {
    throw std::runtime_error("Can't read one of \"flexbounds\", \"relpos\" from  " + yaml_path +
                             " . Check that \"linearham-info\" was written to the yaml file and is not null.");
  }
  timer.Mark("cluster yaml");
  ggenes_ = CreateGermlineGeneMap(hmm_param_dir);
  timer.Mark("germline parameter files");
  Require(!ggenes_.empty(), "no germline parameter files found in " + hmm_param_dir);
  alphabet_ = ggenes_.begin()->second.germ_ptr->alphabet() + "N";
  Require(locus_ == "igh" || locus_ == "igk" || locus_ == "igl", "locus must be igh, igk or igl");
  for (const auto& kv : relpos_)
    Require(ggenes_.count(kv.first) == 1, "gene \"" + kv.first + "\" of relpos has no parameter file");
  InitializeMsa();
  rng_.seed(seed);
  InitializeStateSpace();
  timer.Mark("state space");
  InitializeTransition();
  timer.Mark("transition matrices");
}

HMM::~HMM() {
  if (family_) lh_family_destroy(family_);
  for (lh_family* f : more_families_)
    if (f) lh_family_destroy(f);
}

// src/HMM.cpp:71-83
void HMM::InitializeMsa() {
  const int n = (int)cluster_data_["unique_ids"].size();
  const int L = (int)cluster_data_["naive_seq"].as_string().size();
  msa_.setConstant(n, L, -1);
  for (int i = 0; i < n; i++) {
    const char* seq_type = cluster_data_["has_shm_indels"][i].as_bool() ? "indel_reversed_seqs" : "input_seqs";
    const std::string& seq = cluster_data_[seq_type][i].as_string();
    Require((int)seq.size() == L, "sequence length differs from naive_seq length");
    const VectorXi ints = ConvertSeqToInts(seq, alphabet_);
    for (int j = 0; j < L; j++) msa_(i, j) = ints[j];
  }
}

// src/HMM.cpp:94-185
void HMM::InitializeStateSpace() {
  const bool igh = locus_ == "igh";
  for (auto it = relpos_.begin(); it != relpos_.end(); ++it) {
    const std::string& gname = it->first;
    const int relpos = it->second;
    const GermlineGene& ggene = ggenes_.at(gname);
    if (ggene.type == GermlineType::V) {
      CachePaddingStates(ggene.germ_ptr, flexbounds_.at("v_l"), relpos, true, vpadding_);
      CacheGermlineStates(ggene.germ_ptr, flexbounds_.at("v_l"), flexbounds_.at("v_r"), relpos, true, false,
                          vgerm_);
      CacheJunctionStates(ggene, flexbounds_.at("v_r"), flexbounds_.at(igh ? "d_l" : "j_l"), relpos, false,
                          vd_junction_);
    } else if (ggene.type == GermlineType::D) {
      CacheJunctionStates(ggene, flexbounds_.at("v_r"), flexbounds_.at("d_l"), relpos, true, vd_junction_);
      CacheGermlineStates(ggene.germ_ptr, flexbounds_.at("d_l"), flexbounds_.at("d_r"), relpos, false, false,
                          dgerm_);
      CacheJunctionStates(ggene, flexbounds_.at("d_r"), flexbounds_.at("j_l"), relpos, false, dj_junction_);
    } else {
      if (igh) {
        CacheJunctionStates(ggene, flexbounds_.at("d_r"), flexbounds_.at("j_l"), relpos, true, dj_junction_);
      } else {
        CacheJunctionStates(ggene, flexbounds_.at("v_r"), flexbounds_.at("j_l"), relpos, true, vd_junction_);
      }
      CacheGermlineStates(ggene.germ_ptr, flexbounds_.at("j_l"), flexbounds_.at("j_r"), relpos, false, true,
                          jgerm_);
      CachePaddingStates(ggene.germ_ptr, flexbounds_.at("j_r"), relpos, false, jpadding_);
    }
  }
}

// src/HMM.cpp:190-246
void HMM::InitializeTransition() {
  ComputePaddingTransition(vpadding_.ggene_ranges, ggenes_, vpadding_transition_);
  if (locus_ == "igh") {
    ComputeGermlineJunctionTransition(vgerm_, vd_junction_, GermlineType::V, GermlineType::D, ggenes_,
                                      vgerm_vd_junction_transition_);
    ComputeJunctionTransition(vd_junction_, GermlineType::V, GermlineType::D, ggenes_, vd_junction_transition_);
    ComputeJunctionGermlineTransition(vd_junction_, dgerm_, GermlineType::V, GermlineType::D, ggenes_,
                                      vd_junction_dgerm_transition_);
    ComputeGermlineJunctionTransition(dgerm_, dj_junction_, GermlineType::D, GermlineType::J, ggenes_,
                                      dgerm_dj_junction_transition_);
    ComputeJunctionTransition(dj_junction_, GermlineType::D, GermlineType::J, ggenes_, dj_junction_transition_);
    ComputeJunctionGermlineTransition(dj_junction_, jgerm_, GermlineType::D, GermlineType::J, ggenes_,
                                      dj_junction_jgerm_transition_);
  } else {
    ComputeGermlineJunctionTransition(vgerm_, vd_junction_, GermlineType::V, GermlineType::J, ggenes_,
                                      vgerm_vd_junction_transition_);
    ComputeJunctionTransition(vd_junction_, GermlineType::V, GermlineType::J, ggenes_, vd_junction_transition_);
    ComputeJunctionGermlineTransition(vd_junction_, jgerm_, GermlineType::V, GermlineType::J, ggenes_,
                                      vd_junction_dgerm_transition_);
  }
  ComputePaddingTransition(jpadding_.ggene_ranges, ggenes_, jpadding_transition_);
}

// src/HMM.cpp:345-354 -- the forward pass runs on the device; this returns its result.
double HMM::LogLikelihood() {
  if (cache_forward_) {
    RunForwardAlgorithm();
    cache_forward_ = false;
  }
  return loglikelihood_;
}

namespace {

// Scatter one compact junction row block (include/linearham_amd.h, lh_forward_size) into the dense
// W x S matrix of the reference (HMM::*_junction_forward_).  Which dense entry a compact slot lands on
// depends on the family only: the positions are worked out once (`scatter`), the matrix is zeroed once,
// and every later call overwrites the same entries.
const double* UnpackJunction(const double* p, const RegionStates& J, const RegionStates& G_left,
                             const RegionStates& G_right, int js, int W, MatrixXd& F, std::vector<int>& scatter) {
  const int S = (int)J.state_strs.size();
  const int nL = (int)G_left.ggene_ranges.size(), nR = (int)G_right.ggene_ranges.size();
  const int stride = nL + 5 * nR;
  if (scatter.empty() || F.rows() != W || F.cols() != S) {
    F.setZero(W, S);
    scatter.assign((std::size_t)W * stride, -1);
    for (int i = 0; i < W; ++i) {
      int* sc = scatter.data() + (std::size_t)i * stride;
      int l = 0;
      for (auto it = G_left.ggene_ranges.begin(); it != G_left.ggene_ranges.end(); ++it, ++l) {
        const auto& rg = J.ggene_ranges.at(it->first);
        if (i < rg.second - rg.first) sc[l] = i * S + rg.first + i;
      }
      int r = 0;
      for (auto it = G_right.ggene_ranges.begin(); it != G_right.ggene_ranges.end(); ++it, ++r) {
        const auto& rg = J.ggene_ranges.at(it->first);
        for (int b = 0; b < 4; ++b) sc[nL + 4 * r + b] = i * S + rg.first + b;
        if (rg.second > rg.first + 4) {
          const int first_site = J.site_inds[rg.first + 4];
          const int k = rg.first + 4 + (js + i - first_site);
          if (js + i >= first_site && k < rg.second) sc[nL + 4 * nR + r] = i * S + k;
        }
      }
    }
  }
  double* f = F.data();
  const std::size_t total = (std::size_t)W * stride;
  for (std::size_t t = 0; t < total; ++t)
    if (scatter[t] >= 0) f[scatter[t]] = p[t];
  return p + total;
}

}  // namespace

void HMM::UnpackForward(const double* fwd, const int32_t* sco) {
  const int nV = (int)vgerm_.state_strs.size(), nJ = (int)jgerm_.state_strs.size();
  vgerm_forward_.assign(fwd, fwd + nV);
  fwd += nV;
  vgerm_scaler_count_ = *sco++;
  const bool igh = locus_ == "igh";
  const int W1 = flexbounds_.at(igh ? "d_l" : "j_l").second - flexbounds_.at("v_r").first;
  fwd = UnpackJunction(fwd, vd_junction_, vgerm_, igh ? dgerm_ : jgerm_, flexbounds_.at("v_r").first, W1,
                       vd_junction_forward_, vd_scatter_);
  vd_junction_scaler_counts_.assign(sco, sco + W1);
  sco += W1;
  if (igh) {
    const int nD = (int)dgerm_.state_strs.size();
    dgerm_forward_.assign(fwd, fwd + nD);
    fwd += nD;
    dgerm_scaler_count_ = *sco++;
    const int W2 = flexbounds_.at("j_l").second - flexbounds_.at("d_r").first;
    fwd = UnpackJunction(fwd, dj_junction_, dgerm_, jgerm_, flexbounds_.at("d_r").first, W2, dj_junction_forward_,
                         dj_scatter_);
    dj_junction_scaler_counts_.assign(sco, sco + W2);
    sco += W2;
  }
  jgerm_forward_.assign(fwd, fwd + nJ);
  jgerm_scaler_count_ = *sco;
}

// src/HMM.cpp:323-341
void HMM::SampleInitialState() {
  distr_.param(std::discrete_distribution<int>::param_type(jgerm_forward_.data(),
                                                           jgerm_forward_.data() + jgerm_forward_.size()));
  jgerm_state_ind_samp_ = distr_(rng_);
  jgerm_state_str_samp_ = jgerm_.state_strs[jgerm_state_ind_samp_];
  jgerm_left_del_samp_ = jgerm_.left_del[jgerm_state_ind_samp_];
  jgerm_right_del_samp_ = jgerm_.right_del[jgerm_state_ind_samp_];
  int range_start, range_end;
  std::tie(range_start, range_end) = jgerm_.ggene_ranges.at(jgerm_state_str_samp_);
  for (int i = range_start; i < range_end; i++)
    naive_seq_samp_[jgerm_.site_inds[i]] = alphabet_[jgerm_.naive_bases[i]];
}

struct HMM::SamplingLists {
  ColumnLists vd_germ, vd_junction, dj_germ, dj_junction;
};

void HMM::EnsureSamplingLists() {
  if (sampling_lists_) return;
  sampling_lists_.reset(new SamplingLists());
  sampling_lists_->vd_germ = ColumnLists::Build(vd_junction_dgerm_transition_);
  sampling_lists_->vd_junction = ColumnLists::Build(vd_junction_transition_);
  if (locus_ == "igh") {
    sampling_lists_->dj_germ = ColumnLists::Build(dj_junction_jgerm_transition_);
    sampling_lists_->dj_junction = ColumnLists::Build(dj_junction_transition_);
  }
}

int HMM::RawDrawsPerSample() const {
  const bool igh = locus_ == "igh";
  int draws = 0;
  draws += jgerm_.state_strs.size() >= 2;
  draws += vgerm_.state_strs.size() >= 2;
  // a junction row draws over all S junction states (S >= 2 whenever a junction exists: four NTI states per right gene)
  draws += flexbounds_.at(igh ? "d_l" : "j_l").second - flexbounds_.at("v_r").first;
  if (igh) {
    draws += dgerm_.state_strs.size() >= 2;
    draws += flexbounds_.at("j_l").second - flexbounds_.at("d_r").first;
  }
  return 2 * draws;  // generate_canonical<double, 53> on a 32-bit engine
}

// The same sequence of operations as SampleNaiveSequence below, on caller-owned state.
void HMM::SampleRow(RowSampler& s, const double* fwd, std::mt19937& rng) const {
  const SamplingLists& sl = *sampling_lists_;
  const bool igh = locus_ == "igh";
  const int nV = (int)vgerm_.state_strs.size(), nJ = (int)jgerm_.state_strs.size();
  s.vgerm_forward.assign(fwd, fwd + nV);
  fwd += nV;
  const int W1 = flexbounds_.at(igh ? "d_l" : "j_l").second - flexbounds_.at("v_r").first;
  fwd = UnpackJunction(fwd, vd_junction_, vgerm_, igh ? dgerm_ : jgerm_, flexbounds_.at("v_r").first, W1,
                       s.vd_junction_forward, s.vd_scatter);
  if (igh) {
    const int nD = (int)dgerm_.state_strs.size();
    s.dgerm_forward.assign(fwd, fwd + nD);
    fwd += nD;
    const int W2 = flexbounds_.at("j_l").second - flexbounds_.at("d_r").first;
    fwd = UnpackJunction(fwd, dj_junction_, dgerm_, jgerm_, flexbounds_.at("d_r").first, W2, s.dj_junction_forward,
                         s.dj_scatter);
  }
  s.jgerm_forward.assign(fwd, fwd + nJ);

  s.naive_seq.assign(msa_.cols(), 'N');
  {  // SampleInitialState
    s.distr.param(std::discrete_distribution<int>::param_type(s.jgerm_forward.data(),
                                                              s.jgerm_forward.data() + s.jgerm_forward.size()));
    s.jgerm_state_ind = s.distr(rng);
    s.jgerm_state_str = jgerm_.state_strs[s.jgerm_state_ind];
    s.jgerm_left_del = jgerm_.left_del[s.jgerm_state_ind];
    s.jgerm_right_del = jgerm_.right_del[s.jgerm_state_ind];
    const auto& rg = jgerm_.ggene_ranges.at(s.jgerm_state_str);
    for (int i = rg.first; i < rg.second; i++) s.naive_seq[jgerm_.site_inds[i]] = alphabet_[jgerm_.naive_bases[i]];
  }
  if (igh) {
    SampleJunctionStates(s.jgerm_state_ind, dj_junction_jgerm_transition_, dj_junction_, dj_junction_transition_,
                         s.dj_junction_forward, GermlineType::D, GermlineType::J, flexbounds_.at("d_r"), alphabet_, rng,
                         s.distr, s.naive_seq, s.jgerm_left_del, s.dj_junction_state_strs, s.dj_junction_state_inds,
                         s.dj_junction_insertion, s.dgerm_right_del, &sl.dj_germ, &sl.dj_junction);
    SampleGermlineState(s.dj_junction_state_inds, dgerm_dj_junction_transition_, dgerm_, s.dgerm_forward, alphabet_,
                        rng, s.distr, s.naive_seq, s.dgerm_state_str, s.dgerm_state_ind, s.dgerm_left_del,
                        s.dgerm_right_del);
    SampleJunctionStates(s.dgerm_state_ind, vd_junction_dgerm_transition_, vd_junction_, vd_junction_transition_,
                         s.vd_junction_forward, GermlineType::V, GermlineType::D, flexbounds_.at("v_r"), alphabet_, rng,
                         s.distr, s.naive_seq, s.dgerm_left_del, s.vd_junction_state_strs, s.vd_junction_state_inds,
                         s.vd_junction_insertion, s.vgerm_right_del, &sl.vd_germ, &sl.vd_junction);
  } else {
    SampleJunctionStates(s.jgerm_state_ind, vd_junction_dgerm_transition_, vd_junction_, vd_junction_transition_,
                         s.vd_junction_forward, GermlineType::V, GermlineType::J, flexbounds_.at("v_r"), alphabet_, rng,
                         s.distr, s.naive_seq, s.jgerm_left_del, s.vd_junction_state_strs, s.vd_junction_state_inds,
                         s.vd_junction_insertion, s.vgerm_right_del, &sl.vd_germ, &sl.vd_junction);
  }
  SampleGermlineState(s.vd_junction_state_inds, vgerm_vd_junction_transition_, vgerm_, s.vgerm_forward, alphabet_, rng,
                      s.distr, s.naive_seq, s.vgerm_state_str, s.vgerm_state_ind, s.vgerm_left_del, s.vgerm_right_del);
  const std::string& q = s.naive_seq;
  std::size_t a = 0, b = q.size();
  while (a < q.size() && q[a] == 'N') ++a;
  while (b > a && q[b - 1] == 'N') --b;
  bool ok = b > a;
  for (std::size_t i = a; i < b && ok; ++i) ok = alphabet_.find(q[i]) != std::string::npos && q[i] != 'N';
  s.vgerm_left_insertion = ok ? q.substr(0, a) : "";
  s.jgerm_right_insertion = ok ? q.substr(b) : "";
}

// src/HMM.cpp:358-431.  Sampling stays on the host: it consumes ONE std::mt19937 stream in file
// order (src/HMM.cpp:56), with libstdc++'s discrete_distribution, exactly like the reference.
std::string HMM::SampleNaiveSequence() {
  if (cache_forward_) {
    RunForwardAlgorithm();
    cache_forward_ = false;
  }
  naive_seq_samp_.assign(msa_.cols(), 'N');
  EnsureSamplingLists();
  const SamplingLists& sl = *sampling_lists_;
  SampleInitialState();
  if (locus_ == "igh") {
    SampleJunctionStates(jgerm_state_ind_samp_, dj_junction_jgerm_transition_, dj_junction_, dj_junction_transition_,
                         dj_junction_forward_, GermlineType::D, GermlineType::J, flexbounds_.at("d_r"), alphabet_,
                         rng_, distr_, naive_seq_samp_, jgerm_left_del_samp_, dj_junction_state_str_samps_,
                         dj_junction_state_ind_samps_, dj_junction_insertion_samp_, dgerm_right_del_samp_, &sl.dj_germ,
                         &sl.dj_junction);
    SampleGermlineState(dj_junction_state_ind_samps_, dgerm_dj_junction_transition_, dgerm_, dgerm_forward_,
                        alphabet_, rng_, distr_, naive_seq_samp_, dgerm_state_str_samp_, dgerm_state_ind_samp_,
                        dgerm_left_del_samp_, dgerm_right_del_samp_);
    SampleJunctionStates(dgerm_state_ind_samp_, vd_junction_dgerm_transition_, vd_junction_, vd_junction_transition_,
                         vd_junction_forward_, GermlineType::V, GermlineType::D, flexbounds_.at("v_r"), alphabet_,
                         rng_, distr_, naive_seq_samp_, dgerm_left_del_samp_, vd_junction_state_str_samps_,
                         vd_junction_state_ind_samps_, vd_junction_insertion_samp_, vgerm_right_del_samp_, &sl.vd_germ,
                         &sl.vd_junction);
  } else {
    SampleJunctionStates(jgerm_state_ind_samp_, vd_junction_dgerm_transition_, vd_junction_, vd_junction_transition_,
                         vd_junction_forward_, GermlineType::V, GermlineType::J, flexbounds_.at("v_r"), alphabet_,
                         rng_, distr_, naive_seq_samp_, jgerm_left_del_samp_, vd_junction_state_str_samps_,
                         vd_junction_state_ind_samps_, vd_junction_insertion_samp_, vgerm_right_del_samp_, &sl.vd_germ,
                         &sl.vd_junction);
  }
  SampleGermlineState(vd_junction_state_ind_samps_, vgerm_vd_junction_transition_, vgerm_, vgerm_forward_, alphabet_,
                      rng_, distr_, naive_seq_samp_, vgerm_state_str_samp_, vgerm_state_ind_samp_,
                      vgerm_left_del_samp_, vgerm_right_del_samp_);

  // GetFrameworkInsertionRegex "^(N*)[ACGT]+(N*)$" (src/utils.cpp:97-99, src/HMM.cpp:422-428)
  const std::string& s = naive_seq_samp_;
  std::size_t a = 0, b = s.size();
  while (a < s.size() && s[a] == 'N') ++a;
  while (b > a && s[b - 1] == 'N') --b;
  bool ok = b > a;
  for (std::size_t i = a; i < b && ok; ++i) ok = alphabet_.find(s[i]) != std::string::npos && s[i] != 'N';
  vgerm_left_insertion_samp_ = ok ? s.substr(0, a) : "";
  jgerm_right_insertion_samp_ = ok ? s.substr(b) : "";
  return naive_seq_samp_;
}

// src/HMM.cpp:466-498
void CacheGermlineStates(GermlinePtr germ_ptr, std::pair<int, int> left_flexbounds,
                         std::pair<int, int> right_flexbounds, int relpos, bool left_end, bool right_end,
                         RegionStates& R) {
  const int site_start = left_end ? std::max(relpos, left_flexbounds.first) : left_flexbounds.second;
  const int site_end =
      right_end ? std::min(relpos + germ_ptr->length(), right_flexbounds.second) : right_flexbounds.first;
  Require(site_end > site_start, "germline region of " + germ_ptr->name() + " must contain at least one site");
  Require(site_start - relpos >= 0 && site_end - relpos <= germ_ptr->length(),
          "allele " + germ_ptr->name() + " does not cover its germline region");
  const int range_start = (int)R.naive_bases.size();
  R.ggene_ranges.emplace(germ_ptr->name(), std::make_pair(range_start, range_start + (site_end - site_start)));
  R.state_strs.push_back(germ_ptr->name());
  R.left_del.push_back(site_start - relpos);
  R.right_del.push_back(relpos + germ_ptr->length() - site_end);
  for (int i = site_start; i < site_end; i++) {
    R.naive_bases.push_back(germ_ptr->bases()[i - relpos]);
    R.germ_inds.push_back(i - relpos);
    R.site_inds.push_back(i);
  }
}

// src/HMM.cpp:528-576
void CacheJunctionStates(const GermlineGene& ggene, std::pair<int, int> left_flexbounds,
                         std::pair<int, int> right_flexbounds, int relpos, bool left_end, RegionStates& R) {
  const Germline& g = *ggene.germ_ptr;
  const int site_start = left_end ? std::max(relpos, left_flexbounds.first) : left_flexbounds.first;
  const int site_end = left_end ? right_flexbounds.second : std::min(relpos + g.length(), right_flexbounds.second);
  Require(site_end >= site_start, "junction range of " + g.name() + " is negative");
  const int range_start = (int)R.naive_bases.size();
  int range_end = range_start + (site_end - site_start);
  if (left_end) range_end += (int)g.alphabet().size();
  R.ggene_ranges.emplace(g.name(), std::make_pair(range_start, range_end));
  if (left_end) {
    for (std::size_t i = 0; i < g.alphabet().size(); i++) {
      R.state_strs.push_back(g.name() + ":N_" + g.alphabet()[i]);
      R.del.push_back(-1);
      R.ggene_types.push_back(ggene.type);
      R.naive_bases.push_back((int)i);
      R.germ_inds.push_back(-1);
      R.site_inds.push_back(-1);
    }
  }
  for (int i = site_start; i < site_end; i++) {
    R.state_strs.push_back(g.name() + ":" + std::to_string(i - relpos));
    R.del.push_back(left_end ? i - relpos : relpos + g.length() - i - 1);
    R.ggene_types.push_back(ggene.type);
    R.naive_bases.push_back(g.bases().at(i - relpos));
    R.germ_inds.push_back(i - relpos);
    R.site_inds.push_back(i);
  }
}

// src/HMM.cpp:595-619
void CachePaddingStates(GermlinePtr germ_ptr, std::pair<int, int> fb, int relpos, bool left_end,
                        RegionStates& R) {
  const int site_start = left_end ? fb.first : std::min(relpos + germ_ptr->length(), fb.second);
  const int site_end = left_end ? std::max(relpos, fb.first) : fb.second;
  const int range_start = (int)R.naive_bases.size();
  R.ggene_ranges.emplace(germ_ptr->name(), std::make_pair(range_start, range_start + (site_end - site_start)));
  for (int i = site_start; i < site_end; i++) {
    R.naive_bases.push_back((int)germ_ptr->alphabet().size());
    R.site_inds.push_back(i);
  }
}

namespace {

struct JInfo {
  int range_start, range_end, nti_length, germ_start, germ_length, germ_ind_start, site_ind_start;
};

JInfo JunctionInfo(const RegionStates& J, const std::string& gname, const GermlineGene& gg,
                   GermlineType right_gtype) {
  JInfo o;
  std::tie(o.range_start, o.range_end) = J.ggene_ranges.at(gname);
  o.nti_length = (gg.type == right_gtype) ? (int)gg.germ_ptr->alphabet().size() : 0;
  o.germ_start = o.range_start + o.nti_length;
  o.germ_length = o.range_end - o.germ_start;
  o.germ_ind_start = (o.germ_length > 0) ? J.germ_inds[o.germ_start] : -1;
  o.site_ind_start = (o.germ_length > 0) ? J.site_inds[o.germ_start] : -1;
  return o;
}

}  // namespace

// src/HMM.cpp:647-706
void ComputeGermlineJunctionTransition(const RegionStates& G, const RegionStates& J, GermlineType left_gtype,
                                       GermlineType right_gtype,
                                       const std::unordered_map<std::string, GermlineGene>& ggenes, MatrixXd& T) {
  T.setZero((int)G.state_strs.size(), (int)J.state_strs.size());
  int from_i = 0;
  for (auto from_it = G.ggene_ranges.begin(); from_it != G.ggene_ranges.end(); ++from_it, from_i++) {
    const GermlineGene& from_ggene = ggenes.at(from_it->first);
    const int from_range_end = from_it->second.second;
    const int from_germ_ind_start = G.germ_inds[from_range_end - 1];
    const int from_site_ind_start = G.site_inds[from_range_end - 1];
    for (auto to_it = J.ggene_ranges.begin(); to_it != J.ggene_ranges.end(); ++to_it) {
      const GermlineGene& to_ggene = ggenes.at(to_it->first);
      const JInfo t = JunctionInfo(J, to_it->first, to_ggene, right_gtype);
      FillTransition(from_ggene, to_ggene, left_gtype, right_gtype, from_germ_ind_start, t.germ_ind_start,
                     from_site_ind_start, t.site_ind_start, 0, t.range_start, 0, t.nti_length, 0, t.germ_start, 1,
                     t.germ_length, T, from_i, 0);
    }
  }
}

// src/HMM.cpp:726-784
void ComputeJunctionTransition(const RegionStates& J, GermlineType left_gtype, GermlineType right_gtype,
                               const std::unordered_map<std::string, GermlineGene>& ggenes, MatrixXd& T) {
  T.setZero((int)J.state_strs.size(), (int)J.state_strs.size());
  for (auto from_it = J.ggene_ranges.begin(); from_it != J.ggene_ranges.end(); ++from_it) {
    const GermlineGene& from_ggene = ggenes.at(from_it->first);
    const JInfo f = JunctionInfo(J, from_it->first, from_ggene, right_gtype);
    for (auto to_it = J.ggene_ranges.begin(); to_it != J.ggene_ranges.end(); ++to_it) {
      const GermlineGene& to_ggene = ggenes.at(to_it->first);
      const JInfo t = JunctionInfo(J, to_it->first, to_ggene, right_gtype);
      FillTransition(from_ggene, to_ggene, left_gtype, right_gtype, f.germ_ind_start, t.germ_ind_start,
                     f.site_ind_start, t.site_ind_start, f.range_start, t.range_start, f.nti_length, t.nti_length,
                     f.germ_start, t.germ_start, f.germ_length, t.germ_length, T, 0, 0);
    }
  }
}

// src/HMM.cpp:812-879
void ComputeJunctionGermlineTransition(const RegionStates& J, const RegionStates& G, GermlineType left_gtype,
                                       GermlineType right_gtype,
                                       const std::unordered_map<std::string, GermlineGene>& ggenes, MatrixXd& T) {
  T.setZero((int)J.state_strs.size(), (int)G.state_strs.size());
  for (auto from_it = J.ggene_ranges.begin(); from_it != J.ggene_ranges.end(); ++from_it) {
    const GermlineGene& from_ggene = ggenes.at(from_it->first);
    const JInfo f = JunctionInfo(J, from_it->first, from_ggene, right_gtype);
    int to_i = 0;
    for (auto to_it = G.ggene_ranges.begin(); to_it != G.ggene_ranges.end(); ++to_it, to_i++) {
      const GermlineGene& to_ggene = ggenes.at(to_it->first);
      const int to_range_start = to_it->second.first, to_range_end = to_it->second.second;
      const int to_germ_ind_start = G.germ_inds[to_range_start];
      const int to_site_ind_start = G.site_inds[to_range_start];
      FillTransition(from_ggene, to_ggene, left_gtype, right_gtype, f.germ_ind_start, to_germ_ind_start,
                     f.site_ind_start, to_site_ind_start, f.range_start, 0, f.nti_length, 0, f.germ_start, 0,
                     f.germ_length, 1, T, 0, to_i);
      double prod = 1.0;
      const VectorXd& tr = to_ggene.germ_ptr->transition();
      for (int k = 0; k < to_range_end - to_range_start - 1; ++k) prod *= tr[to_germ_ind_start + k];
      for (int r = f.range_start; r < f.range_end; ++r) T(r, to_i) *= prod;
    }
  }
}

// src/HMM.cpp:891-915
void ComputePaddingTransition(const GeneRanges& ranges, const std::unordered_map<std::string, GermlineGene>& ggenes,
                              VectorXd& transition) {
  transition.assign(ranges.size(), 0.0);
  int i = 0;
  for (auto it = ranges.begin(); it != ranges.end(); ++it, i++) {
    const double n_transition = ggenes.at(it->first).npadding().n_transition();
    transition[i] = (1.0 - n_transition) * std::pow(n_transition, it->second.second - it->second.first);
  }
}

// src/HMM.cpp:964-1089.  (row_off, col_off) locate the Eigen::Ref block the reference passes in.
void FillTransition(const GermlineGene& from_ggene, const GermlineGene& to_ggene, GermlineType left_gtype,
                    GermlineType right_gtype, int germ_ind_row_start, int germ_ind_col_start,
                    int site_ind_row_start, int site_ind_col_start, int nti_row_start, int nti_col_start,
                    int nti_row_length, int nti_col_length, int germ_row_start, int germ_col_start,
                    int germ_row_length, int germ_col_length, MatrixXd& T, int row_off, int col_off) {
  const Germline& from = *from_ggene.germ_ptr;
  const Germline& to = *to_ggene.germ_ptr;
  if (from.name() == to.name()) {
    if (from_ggene.type == right_gtype) {
      const NTInsertion& nti = from_ggene.nti();
      if (nti_col_length > 0)
        for (int r = 0; r < nti_row_length; ++r)
          for (int c = 0; c < nti_col_length; ++c)
            T(row_off + nti_row_start + r, col_off + nti_col_start + c) = nti.nti_transition()(r, c);
      if (germ_col_length > 0)
        for (int r = 0; r < nti_row_length; ++r)
          for (int c = 0; c < germ_col_length; ++c)
            T(row_off + nti_row_start + r, col_off + germ_col_start + c) =
                nti.nti_landing_out()(r, germ_ind_col_start + c);
    }
    if (germ_row_length > 0 && germ_col_length > 0) {
      if (germ_ind_row_start == germ_ind_col_start) {
        for (int k = 0; k < germ_row_length - 1; ++k)
          T(row_off + germ_row_start + k, col_off + germ_col_start + k + 1) = from.transition()[germ_ind_row_start + k];
      } else {
        // Eigen's block.diagonal(-(rows-1)) = the bottom-left element; vector.diagonal(-k) = element k
        T(row_off + germ_row_start + germ_row_length - 1, col_off + germ_col_start) =
            from.transition().at(germ_ind_row_start + germ_row_length - 1);
      }
    }
  }
  if (from_ggene.type == left_gtype && to_ggene.type == right_gtype) {
    if (germ_row_length > 0 && nti_col_length > 0) {
      const VectorXd& nti_landing_in = to_ggene.nti().nti_landing_in();
      for (int r = 0; r < germ_row_length; ++r)
        for (int c = 0; c < nti_col_length; ++c) {
          double v = 1.0;
          v = from.landing_out()[germ_ind_row_start + r] * v;
          v *= to.gene_prob();
          v = nti_landing_in[c] * v;
          T(row_off + germ_row_start + r, col_off + nti_col_start + c) = v;
        }
    }
    if (germ_row_length > 0 && germ_col_length > 0) {
      // Germline-to-germline entries across two genes exist only between states on consecutive alignment
      // sites: row a sits on site site_ind_row_start + a, column b on site site_ind_col_start + b, so the
      // nonzeros lie on the one diagonal b - a = shift, clipped to the block.
      const int shift = site_ind_row_start + 1 - site_ind_col_start;
      const int a0 = shift < 0 ? -shift : 0, b0 = shift > 0 ? shift : 0;
      const int run = std::min(germ_row_length - a0, germ_col_length - b0);
      const double gp = to.gene_prob();
      for (int k = 0; k < run; ++k)
        T(row_off + germ_row_start + a0 + k, col_off + germ_col_start + b0 + k) =
            from.landing_out()[germ_ind_row_start + a0 + k] * gp * to.landing_in()[germ_ind_col_start + b0 + k];
    }
  }
}

// Structured form of the three matrices above for one junction, as consumed by the HIP forward
// kernel (lh_junction, include/linearham_amd.h).  Every entry equals the corresponding nonzero of the
// dense matrices (checked by tests/test_host_*.py against the dense accessors).
JunctionTables HMM::BuildJunctionTables(const RegionStates& J, const RegionStates& G_left,
                                        const RegionStates& G_right, std::pair<int, int> left_fb,
                                        std::pair<int, int> right_fb, const MatrixXi& xmsa_inds) const {
  JunctionTables t;
  const int js = left_fb.first;
  const int W = right_fb.second - left_fb.first;
  const int nL = (int)G_left.ggene_ranges.size(), nR = (int)G_right.ggene_ranges.size();
  Require(W >= 1, "junction regions must contain at least one site");
  t.n_rows = W;
  t.n_left = nL;
  t.n_right = nR;
  t.enter_trans.assign(nL, 0.0);
  t.enter_lo.assign(nL, 0.0);
  t.left_trans.assign((std::size_t)W * nL, 0.0);
  t.left_lo.assign((std::size_t)W * nL, 0.0);
  t.left_xmsa.assign((std::size_t)W * nL, -1);
  t.right_gp_nli.assign((std::size_t)nR * 4, 0.0);
  t.right_ntt.assign((std::size_t)nR * 16, 0.0);
  t.right_nlo.assign((std::size_t)W * nR * 4, 0.0);
  t.right_trans.assign((std::size_t)W * nR, 0.0);
  t.right_gp_li.assign((std::size_t)W * nR, 0.0);
  t.right_xmsa.assign((std::size_t)W * nR, -1);
  t.nti_xmsa.assign((std::size_t)W * nR * 4, -1);
  t.exit_nlo.assign((std::size_t)nR * 4, 0.0);
  t.exit_trans.assign(nR, 0.0);
  t.exit_gp_li.assign(nR, 0.0);
  const bool have_x = xmsa_inds.size() > 0;
  int l = 0;
  for (auto it = G_left.ggene_ranges.begin(); it != G_left.ggene_ranges.end(); ++it, ++l) {
    const Germline& g = *ggenes_.at(it->first).germ_ptr;
    const int p_last = G_left.germ_inds[it->second.second - 1];
    t.enter_lo[l] = g.landing_out()[p_last];
    const auto& rg = J.ggene_ranges.at(it->first);
    const int cnt = rg.second - rg.first;
    if (cnt > 0) t.enter_trans[l] = g.transition().at(p_last);
    for (int i = 0; i < cnt; ++i) {
      const int p = J.germ_inds[rg.first + i];
      Require(J.site_inds[rg.first + i] == js + i, "left-gene junction states must start at the junction start");
      if (i >= 1) t.left_trans[(std::size_t)i * nL + l] = g.transition()[p - 1];
      t.left_lo[(std::size_t)i * nL + l] = g.landing_out()[p];
      if (have_x) t.left_xmsa[(std::size_t)i * nL + l] = xmsa_inds(i, rg.first + i);
    }
  }
  int r = 0;
  for (auto it = G_right.ggene_ranges.begin(); it != G_right.ggene_ranges.end(); ++it, ++r) {
    const GermlineGene& gg = ggenes_.at(it->first);
    const Germline& g = *gg.germ_ptr;
    const NTInsertion& nti = gg.nti();
    const auto& rg = J.ggene_ranges.at(it->first);
    for (int b = 0; b < 4; ++b) {
      t.right_gp_nli[(std::size_t)r * 4 + b] = g.gene_prob() * nti.nti_landing_in()[b];
      for (int c = 0; c < 4; ++c) t.right_ntt[(std::size_t)r * 16 + b * 4 + c] = nti.nti_transition()(b, c);
      if (have_x)
        for (int i = 0; i < W; ++i) t.nti_xmsa[((std::size_t)i * nR + r) * 4 + b] = xmsa_inds(i, rg.first + b);
    }
    bool first = true;
    int last_row = -1;
    for (int k = rg.first + 4; k < rg.second; ++k) {
      const int q = J.germ_inds[k];
      const int i = J.site_inds[k] - js;
      for (int b = 0; b < 4; ++b) t.right_nlo[((std::size_t)i * nR + r) * 4 + b] = nti.nti_landing_out()(b, q);
      if (!first) t.right_trans[(std::size_t)i * nR + r] = g.transition()[q - 1];
      t.right_gp_li[(std::size_t)i * nR + r] = g.gene_prob() * g.landing_in()[q];
      if (have_x) t.right_xmsa[(std::size_t)i * nR + r] = xmsa_inds(i, k);
      first = false;
      last_row = i;
    }
    const int trs = it->second.first, tre = it->second.second;
    const int q0 = G_right.germ_inds[trs];
    double prod = 1.0;
    for (int k = 0; k < tre - trs - 1; ++k) prod *= g.transition()[q0 + k];
    for (int b = 0; b < 4; ++b) t.exit_nlo[(std::size_t)r * 4 + b] = nti.nti_landing_out()(b, q0) * prod;
    if (last_row == W - 1) t.exit_trans[r] = g.transition().at(q0 - 1) * prod;
    t.exit_gp_li[r] = g.gene_prob() * g.landing_in()[q0] * prod;
  }
  return t;
}

lh_sampler_junction HMM::SamplerJunction::c() const {
  lh_sampler_junction j;
  j.n_rows = n_rows;
  j.n_left = n_left;
  j.n_right = n_right;
  j.n_states = n_states;
  j.left_rows = left_rows.data();
  j.left_dense = left_dense.data();
  j.left_lo = left_lo.data();
  j.left_trans = left_trans.data();
  j.enter_lo = enter_lo.data();
  j.right_dense = right_dense.data();
  j.right_first = right_first.data();
  j.gene_prob = gene_prob.data();
  j.nti_landing_in = nti_landing_in.data();
  j.nti_transition = nti_transition.data();
  j.nti_landing_out = nti_landing_out.data();
  j.landing_in = landing_in.data();
  j.right_trans = right_trans.data();
  j.exit_nlo = exit_nlo.data();
  j.exit_trans = exit_trans.data();
  j.exit_li = exit_li.data();
  j.prod = prod.data();
  return j;
}

// The factors FillTransition (src/HMM.cpp:964-1089) multiplies, kept apart, per (junction row, gene): the device
// sampler forms every transition value with the reference's own association (see lh_sample.hip).
HMM::SamplerJunction HMM::BuildSamplerJunction(const RegionStates& J, const RegionStates& G_left,
                                               const RegionStates& G_right, std::pair<int, int> left_fb,
                                               std::pair<int, int> right_fb) const {
  SamplerJunction t;
  const int js = left_fb.first;
  const int W = right_fb.second - left_fb.first;
  const int nL = (int)G_left.ggene_ranges.size(), nR = (int)G_right.ggene_ranges.size();
  t.n_rows = W;
  t.n_left = nL;
  t.n_right = nR;
  t.n_states = (int)J.state_strs.size();
  t.left_rows.assign(nL, 0);
  t.left_dense.assign(nL, 0);
  t.left_lo.assign((std::size_t)W * nL, 0.0);
  t.left_trans.assign((std::size_t)W * nL, 0.0);
  t.enter_lo.assign(nL, 0.0);
  t.right_dense.assign(nR, 0);
  t.right_first.assign(nR, W);
  t.gene_prob.assign(nR, 0.0);
  t.nti_landing_in.assign((std::size_t)nR * 4, 0.0);
  t.nti_transition.assign((std::size_t)nR * 16, 0.0);
  t.nti_landing_out.assign((std::size_t)W * nR * 4, 0.0);
  t.landing_in.assign((std::size_t)W * nR, 0.0);
  t.right_trans.assign((std::size_t)W * nR, 0.0);
  t.exit_nlo.assign((std::size_t)nR * 4, 0.0);
  t.exit_trans.assign(nR, 0.0);
  t.exit_li.assign(nR, 0.0);
  t.prod.assign(nR, 1.0);
  int l = 0;
  for (auto it = G_left.ggene_ranges.begin(); it != G_left.ggene_ranges.end(); ++it, ++l) {
    const Germline& g = *ggenes_.at(it->first).germ_ptr;
    const int p_last = G_left.germ_inds[it->second.second - 1];
    t.enter_lo[l] = g.landing_out()[p_last];
    const auto& rg = J.ggene_ranges.at(it->first);
    const int cnt = rg.second - rg.first;
    t.left_rows[l] = cnt;
    t.left_dense[l] = rg.first;
    if (cnt > 0) t.left_trans[l] = g.transition().at(p_last);  // row 0: out of the germline region
    for (int i = 0; i < cnt; ++i) {
      const int p = J.germ_inds[rg.first + i];
      if (i >= 1) t.left_trans[(std::size_t)i * nL + l] = g.transition()[p - 1];
      t.left_lo[(std::size_t)i * nL + l] = g.landing_out()[p];
    }
  }
  int r = 0;
  for (auto it = G_right.ggene_ranges.begin(); it != G_right.ggene_ranges.end(); ++it, ++r) {
    const GermlineGene& gg = ggenes_.at(it->first);
    const Germline& g = *gg.germ_ptr;
    const NTInsertion& nti = gg.nti();
    const auto& rg = J.ggene_ranges.at(it->first);
    t.right_dense[r] = rg.first;
    t.gene_prob[r] = g.gene_prob();
    for (int b = 0; b < 4; ++b) {
      t.nti_landing_in[(std::size_t)r * 4 + b] = nti.nti_landing_in()[b];
      for (int c = 0; c < 4; ++c) t.nti_transition[(std::size_t)r * 16 + b * 4 + c] = nti.nti_transition()(b, c);
    }
    bool first = true;
    int last_row = -1;
    for (int k = rg.first + 4; k < rg.second; ++k) {
      const int q = J.germ_inds[k];
      const int i = J.site_inds[k] - js;
      if (first) t.right_first[r] = i;
      for (int b = 0; b < 4; ++b) t.nti_landing_out[((std::size_t)i * nR + r) * 4 + b] = nti.nti_landing_out()(b, q);
      if (!first) t.right_trans[(std::size_t)i * nR + r] = g.transition()[q - 1];
      t.landing_in[(std::size_t)i * nR + r] = g.landing_in()[q];
      first = false;
      last_row = i;
    }
    const int trs = it->second.first, tre = it->second.second;
    const int q0 = G_right.germ_inds[trs];
    double prod = 1.0;
    for (int k = 0; k < tre - trs - 1; ++k) prod *= g.transition()[q0 + k];
    t.prod[r] = prod;
    for (int b = 0; b < 4; ++b) t.exit_nlo[(std::size_t)r * 4 + b] = nti.nti_landing_out()(b, q0) * prod;
    if (last_row == W - 1) t.exit_trans[r] = g.transition().at(q0 - 1) * prod;
    t.exit_li[r] = g.landing_in()[q0];
  }
  return t;
}

namespace {

// What SampleJunctionStates (src/HMM.cpp:1222-1278) does with the drawn states, in its order (last row first).
void ApplyJunctionStates(const int32_t* k_of_row, const RegionStates& J, int W, GermlineType left_gtype,
                         GermlineType right_gtype, int site_start, const std::string& alphabet, std::string& naive_seq,
                         int& germ_left_del, std::vector<std::string>& strs, std::vector<int>& inds,
                         std::string& insertion, int& germ_right_del) {
  strs.assign(W, "");
  inds.assign(W, -1);
  insertion = "";
  germ_right_del = -1;
  for (int i = W - 1; i >= 0; i--) {
    const int k = k_of_row[i];
    inds[i] = k;
    strs[i] = J.state_strs[k];
    naive_seq[site_start + i] = alphabet[J.naive_bases[k]];
    if (J.ggene_types[k] == right_gtype) {
      if (J.del[k] != -1) {
        germ_left_del = J.del[k];
      } else {
        insertion = alphabet[J.naive_bases[k]] + insertion;
      }
    } else if (J.ggene_types[k] == left_gtype && germ_right_del == -1) {
      germ_right_del = J.del[k];
    }
  }
}

// ... and SampleGermlineState (src/HMM.cpp:1316-1353)
void ApplyGermlineState(int ind, const RegionStates& G, const std::string& alphabet, std::string& naive_seq,
                        std::string& str, int& state_ind, int& left_del, int& right_del) {
  state_ind = ind;
  str = G.state_strs[ind];
  left_del = G.left_del[ind];
  if (right_del == -1) right_del = G.right_del[ind];
  const auto& rg = G.ggene_ranges.at(str);
  for (int i = rg.first; i < rg.second; i++) naive_seq[G.site_inds[i]] = alphabet[G.naive_bases[i]];
}

}  // namespace

void HMM::ApplySampledStates(RowSampler& s, const int32_t* st) const {
  const bool igh = locus_ == "igh";
  s.naive_seq.assign(msa_.cols(), 'N');
  int o = 0;
  {  // SampleInitialState
    s.jgerm_state_ind = st[o++];
    s.jgerm_state_str = jgerm_.state_strs[s.jgerm_state_ind];
    s.jgerm_left_del = jgerm_.left_del[s.jgerm_state_ind];
    s.jgerm_right_del = jgerm_.right_del[s.jgerm_state_ind];
    const auto& rg = jgerm_.ggene_ranges.at(s.jgerm_state_str);
    for (int i = rg.first; i < rg.second; i++) s.naive_seq[jgerm_.site_inds[i]] = alphabet_[jgerm_.naive_bases[i]];
  }
  const int W1 = flexbounds_.at(igh ? "d_l" : "j_l").second - flexbounds_.at("v_r").first;
  if (igh) {
    const int W2 = flexbounds_.at("j_l").second - flexbounds_.at("d_r").first;
    ApplyJunctionStates(st + o, dj_junction_, W2, GermlineType::D, GermlineType::J, flexbounds_.at("d_r").first, alphabet_,
                        s.naive_seq, s.jgerm_left_del, s.dj_junction_state_strs, s.dj_junction_state_inds,
                        s.dj_junction_insertion, s.dgerm_right_del);
    o += W2;
    ApplyGermlineState(st[o++], dgerm_, alphabet_, s.naive_seq, s.dgerm_state_str, s.dgerm_state_ind, s.dgerm_left_del,
                       s.dgerm_right_del);
    ApplyJunctionStates(st + o, vd_junction_, W1, GermlineType::V, GermlineType::D, flexbounds_.at("v_r").first, alphabet_,
                        s.naive_seq, s.dgerm_left_del, s.vd_junction_state_strs, s.vd_junction_state_inds,
                        s.vd_junction_insertion, s.vgerm_right_del);
  } else {
    ApplyJunctionStates(st + o, vd_junction_, W1, GermlineType::V, GermlineType::J, flexbounds_.at("v_r").first, alphabet_,
                        s.naive_seq, s.jgerm_left_del, s.vd_junction_state_strs, s.vd_junction_state_inds,
                        s.vd_junction_insertion, s.vgerm_right_del);
  }
  o += W1;
  ApplyGermlineState(st[o], vgerm_, alphabet_, s.naive_seq, s.vgerm_state_str, s.vgerm_state_ind, s.vgerm_left_del,
                     s.vgerm_right_del);
  const std::string& q = s.naive_seq;
  std::size_t a = 0, b = q.size();
  while (a < q.size() && q[a] == 'N') ++a;
  while (b > a && q[b - 1] == 'N') --b;
  bool ok = b > a;
  for (std::size_t i = a; i < b && ok; ++i) ok = alphabet_.find(q[i]) != std::string::npos && q[i] != 'N';
  s.vgerm_left_insertion = ok ? q.substr(0, a) : "";
  s.jgerm_right_insertion = ok ? q.substr(b) : "";
}

ColumnLists ColumnLists::Build(const MatrixXd& M) {
  ColumnLists c;
  const int R = M.rows(), C = M.cols();
  c.start.assign(C + 1, 0);
  for (int r = 0; r < R; ++r) {
    const double* row = M.row(r);
    for (int k = 0; k < C; ++k)
      if (row[k] != 0.0) ++c.start[k + 1];
  }
  for (int k = 0; k < C; ++k) c.start[k + 1] += c.start[k];
  c.rows.resize(c.start[C]);
  std::vector<int> fill(c.start.begin(), c.start.end() - 1);
  for (int r = 0; r < R; ++r) {  // rows visited in order: ascending within every column
    const double* row = M.row(r);
    for (int k = 0; k < C; ++k)
      if (row[k] != 0.0) c.rows[fill[k]++] = r;
  }
  return c;
}

// libstdc++'s discrete_distribution (bits/random.tcc): probabilities = weights / sum, cumulative by
// partial_sum, last cumulative forced to 1.0, one generate_canonical<double, 53> draw p, result =
// lower_bound(cumulative, p).  Zeros add 0.0 to the sum and to every partial sum, so only the non-zero
// entries have to be visited; a zero entry can be returned in two corner cases only, kept here: p == 0
// returns index 0, and p above the last computed partial sum returns the last index (whose cumulative is
// the forced 1.0).
int DrawDiscreteSparse(std::mt19937& rng, const int* idx, const double* weights, int k, int size) {
  if (size < 2) return 0;  // libstdc++ keeps no table for fewer than two weights and draws nothing
  double sum = 0.0;
  for (int j = 0; j < k; ++j) sum += weights[j];
  const double p = std::generate_canonical<double, std::numeric_limits<double>::digits>(rng);
  // (all weights zero, or a NaN among them: libstdc++'s table is all NaN from its first element, index 0 comes back)
  if (!(p > 0.0) || k == 0 || sum != sum) return 0;
  double cum = 0.0;
  for (int j = 0; j < k; ++j) {
    cum += weights[j] / sum;
    const double cp = (j == k - 1 && idx[j] == size - 1) ? 1.0 : cum;
    // lower_bound: the first element that is NOT below p.  Written as !(cp < p), not cp >= p: on a row the reference's
    // 2^(256 d) equalisation has overflowed (src/PhyloHMM.cpp:190-192) weights are inf, their quotients by the sum NaN,
    // every partial sum from the first of them on NaN, and lower_bound -- no NaN is below anything -- stops there.
    if (!(cp < p)) return idx[j];
  }
  return size - 1;
}

// src/HMM.cpp:1222-1278
void SampleJunctionStates(int germ_state_ind_samp, const MatrixXd& junction_germ_transition, const RegionStates& J,
                          const MatrixXd& junction_transition, const MatrixXd& junction_forward,
                          GermlineType left_gtype, GermlineType right_gtype, std::pair<int, int> left_flexbounds,
                          const std::string& alphabet, std::mt19937& rng, std::discrete_distribution<int>& distr,
                          std::string& naive_seq_samp, int& germ_left_del_samp,
                          std::vector<std::string>& junction_state_str_samps,
                          std::vector<int>& junction_state_ind_samps, std::string& junction_insertion_samp,
                          int& germ_right_del_samp, const ColumnLists* germ_cols, const ColumnLists* junction_cols) {
  const int site_start = left_flexbounds.first;
  const int W = junction_forward.rows(), S = junction_forward.cols();
  junction_state_str_samps.assign(W, "");
  junction_state_ind_samps.assign(W, -1);
  junction_insertion_samp = "";
  germ_right_del_samp = -1;
  VectorXd probs(S);
  std::vector<int> nz_idx;
  std::vector<double> nz_w;
  for (int i = W - 1; i >= 0; i--) {
    int k;
    const ColumnLists* cols = (i == W - 1) ? germ_cols : junction_cols;
    if (cols) {
      // only the states with a non-zero transition into the sampled successor can have weight
      const MatrixXd& T = (i == W - 1) ? junction_germ_transition : junction_transition;
      const int col = (i == W - 1) ? germ_state_ind_samp : junction_state_ind_samps[i + 1];
      nz_idx.clear();
      nz_w.clear();
      for (int t = cols->start[col]; t < cols->start[col + 1]; ++t) {
        const int s = cols->rows[t];
        const double w = T(s, col) * junction_forward(i, s);
        if (w != 0.0) {
          nz_idx.push_back(s);
          nz_w.push_back(w);
        }
      }
      k = DrawDiscreteSparse(rng, nz_idx.data(), nz_w.data(), (int)nz_idx.size(), S);
    } else {
      for (int s = 0; s < S; ++s) {
        const double tr = (i == W - 1) ? junction_germ_transition(s, germ_state_ind_samp)
                                       : junction_transition(s, junction_state_ind_samps[i + 1]);
        probs[s] = tr * junction_forward(i, s);
      }
      distr.param(std::discrete_distribution<int>::param_type(probs.data(), probs.data() + probs.size()));
      k = distr(rng);
    }
    junction_state_ind_samps[i] = k;
    junction_state_str_samps[i] = J.state_strs[k];
    naive_seq_samp[site_start + i] = alphabet[J.naive_bases[k]];
    if (J.ggene_types[k] == right_gtype) {
      if (J.del[k] != -1) {
        germ_left_del_samp = J.del[k];
      } else {
        junction_insertion_samp = alphabet[J.naive_bases[k]] + junction_insertion_samp;
      }
    } else if (J.ggene_types[k] == left_gtype && germ_right_del_samp == -1) {
      germ_right_del_samp = J.del[k];
    }
  }
}

// src/HMM.cpp:1316-1353
void SampleGermlineState(const std::vector<int>& junction_state_ind_samps, const MatrixXd& germ_junction_transition,
                         const RegionStates& G, const VectorXd& germ_forward, const std::string& alphabet,
                         std::mt19937& rng, std::discrete_distribution<int>& distr, std::string& naive_seq_samp,
                         std::string& germ_state_str_samp, int& germ_state_ind_samp, int& germ_left_del_samp,
                         int& germ_right_del_samp) {
  const int n = germ_junction_transition.rows();
  VectorXd probs(n);
  for (int g = 0; g < n; ++g) probs[g] = germ_junction_transition(g, junction_state_ind_samps.front()) * germ_forward[g];
  distr.param(std::discrete_distribution<int>::param_type(probs.data(), probs.data() + probs.size()));
  germ_state_ind_samp = distr(rng);
  germ_state_str_samp = G.state_strs[germ_state_ind_samp];
  germ_left_del_samp = G.left_del[germ_state_ind_samp];
  if (germ_right_del_samp == -1) germ_right_del_samp = G.right_del[germ_state_ind_samp];
  int range_start, range_end;
  std::tie(range_start, range_end) = G.ggene_ranges.at(germ_state_str_samp);
  for (int i = range_start; i < range_end; i++) naive_seq_samp[G.site_inds[i]] = alphabet[G.naive_bases[i]];
}

}  // namespace linearham
