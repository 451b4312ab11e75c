#include "VDJGermline.hpp"

#include <exception>
#include <thread>

#include <dirent.h>

#include <algorithm>
#include <map>
#include <tuple>
#include <vector>

namespace linearham {

// src/Germline.cpp:20-115
Germline::Germline(const yaml_lite::Node& root) {
  alphabet_ = GetAlphabet(root);
  name_ = root["name"].as_string();
  const std::string gname = name_;
  int gstart, gend;
  std::tie(gstart, gend) = FindGermlineStartEnd(root, gname);
  const int nstates = (int)root["states"].size();
  Require(gstart == 2 || gstart == (int)alphabet_.size() + 1, "unexpected germline start state");
  Require(gend == nstates - 1 || gend == nstates - 2, "unexpected germline end state");
  const int gcount = gend - gstart + 1;
  name_ = FixGeneName(name_);

  landing_in_.assign(gcount, 0.0);
  landing_out_.assign(gcount, 0.0);
  transition_.assign(gcount - 1, 0.0);
  emission_.setZero((int)alphabet_.size(), gcount);
  bases_.assign(gcount, 0);
  gene_prob_ = root["extras"]["gene_prob"].as_double();

  const yaml_lite::Node& init_state = root["states"][0];
  Require(init_state["name"].as_string() == "init", "first state must be init");
  std::vector<std::string> state_names;
  VectorXd probs;
  std::tie(state_names, probs) = ParseStringProbMap(init_state["transitions"]);
  int idx;
  for (std::size_t i = 0; i < state_names.size(); i++) {
    if (MatchGermlineState(state_names[i], gname, &idx)) {
      landing_in_.at(idx) = probs[i];
    } else {
      Require(state_names[i].find("insert_left_") != std::string::npos, "init must land in insert_left_*");
    }
  }

  for (int i = gstart; i < gend + 1; i++) {
    const yaml_lite::Node& gstate = root["states"][i];
    Require(MatchGermlineState(gstate["name"].as_string(), gname, &idx), "germline state name");
    const int gindex = idx;
    Require(gindex == i - gstart, "germline state numbering");
    std::tie(state_names, probs) = ParseStringProbMap(gstate["transitions"]);
    for (std::size_t j = 0; j < state_names.size(); j++) {
      if (MatchGermlineState(state_names[j], gname, &idx)) {
        Require(idx == gindex + 1, "germline state must transition to its successor");
        transition_.at(gindex) = probs[j];
      } else if (state_names[j] == "end") {
        landing_out_[gindex] = probs[j];
      } else {
        Require(state_names[j] == "insert_right_N", "unexpected transition target " + state_names[j]);
      }
    }
    std::tie(state_names, probs) = ParseStringProbMap(gstate["emissions"]["probs"]);
    Require(gstate["emissions"]["track"].as_string() == "nukes", "emission track must be nukes");
    for (std::size_t j = 0; j < state_names.size(); j++)
      emission_(GetAlphabetIndex(alphabet_, state_names[j][0]), gindex) = probs[j];
    bases_[gindex] = GetAlphabetIndex(alphabet_, gstate["extras"]["germline"].as_char());
  }
}

// src/NTInsertion.cpp:21-104
NTInsertion::NTInsertion(const yaml_lite::Node& root) {
  const std::string alphabet = GetAlphabet(root);
  const std::string gname = root["name"].as_string();
  int gstart, gend;
  std::tie(gstart, gend) = FindGermlineStartEnd(root, gname);
  const int nstates = (int)root["states"].size();
  Require(gstart == (int)alphabet.size() + 1, "NTI states expected before the germline states");
  Require(gend == nstates - 1 || gend == nstates - 2, "unexpected germline end state");
  const int gcount = gend - gstart + 1;
  const int na = (int)alphabet.size();
  nti_landing_in_.assign(na, 0.0);
  nti_landing_out_.setZero(na, gcount);
  nti_transition_.setZero(na, na);
  nti_emission_.setZero(na, na);

  const yaml_lite::Node& init_state = root["states"][0];
  Require(init_state["name"].as_string() == "init", "first state must be init");
  std::vector<std::string> state_names;
  VectorXd probs;
  std::tie(state_names, probs) = ParseStringProbMap(init_state["transitions"]);
  int idx;
  char base;
  for (std::size_t i = 0; i < state_names.size(); i++) {
    if (MatchNTIState(state_names[i], alphabet, &base)) {
      nti_landing_in_[GetAlphabetIndex(alphabet, base)] = probs[i];
    } else {
      Require(MatchGermlineState(state_names[i], gname, &idx), "init must land in NTI or germline state");
    }
  }
  for (int i = 1; i <= na; i++) {
    const yaml_lite::Node& nti_state = root["states"][i];
    Require(MatchNTIState(nti_state["name"].as_string(), alphabet, &base), "NTI state name");
    const int nti_base = GetAlphabetIndex(alphabet, base);
    std::tie(state_names, probs) = ParseStringProbMap(nti_state["transitions"]);
    for (std::size_t j = 0; j < state_names.size(); j++) {
      if (MatchGermlineState(state_names[j], gname, &idx)) {
        Require(idx < gcount, "NTI landing-out position out of range");
        nti_landing_out_(nti_base, idx) = probs[j];
      } else {
        Require(MatchNTIState(state_names[j], alphabet, &base), "NTI transition target");
        nti_transition_(nti_base, GetAlphabetIndex(alphabet, base)) = probs[j];
      }
    }
    std::tie(state_names, probs) = ParseStringProbMap(nti_state["emissions"]["probs"]);
    Require(nti_state["emissions"]["track"].as_string() == "nukes", "emission track must be nukes");
    for (std::size_t j = 0; j < state_names.size(); j++)
      nti_emission_(GetAlphabetIndex(alphabet, state_names[j][0]), nti_base) = probs[j];
  }
}

// src/NPadding.cpp:22-109
NPadding::NPadding(const yaml_lite::Node& root) {
  const std::string alphabet = GetAlphabet(root);
  const std::string gname = root["name"].as_string();
  int gstart, gend;
  std::tie(gstart, gend) = FindGermlineStartEnd(root, gname);
  const int nstates = (int)root["states"].size();
  Require(gstart == 2 || gend == nstates - 2, "expected insert_left_N or insert_right_N");
  n_emission_.assign(alphabet.size(), 0.0);
  int n_index, n_check_index;
  std::string n_name, next_name;
  if (gstart == 2) {
    n_index = gstart - 1;
    n_check_index = gstart - 2;
    n_name = "insert_left_N";
    next_name = gname + "_0";
  } else {
    n_index = gend + 1;
    n_check_index = gend;
    n_name = "insert_right_N";
    next_name = "end";
  }
  const yaml_lite::Node& n_state = root["states"][n_index];
  const yaml_lite::Node& n_check_state = root["states"][n_check_index];
  Require(n_state["name"].as_string() == n_name, "padding state name");
  std::map<std::string, double> a, b;
  for (const auto& kv : n_state["transitions"].map) a[kv.first] = kv.second.as_double();
  for (const auto& kv : n_check_state["transitions"].map) b[kv.first] = kv.second.as_double();
  Require(a.size() == b.size(), "padding state transitions must mirror the neighbouring state");
  for (auto it = a.begin(), cit = b.begin(); it != a.end(); ++it, ++cit) {
    Require(it->first == cit->first, "padding transition keys differ");
    Require(std::fabs(it->second - cit->second) <= EPS, "padding transition probabilities differ");
    if (it->first == n_name) {
      n_transition_ = it->second;
    } else {
      Require(it->first == next_name, "unexpected padding transition target");
    }
  }
  std::vector<std::string> names;
  VectorXd probs;
  std::tie(names, probs) = ParseStringProbMap(n_state["emissions"]["probs"]);
  Require(n_state["emissions"]["track"].as_string() == "nukes", "emission track must be nukes");
  for (std::size_t i = 0; i < names.size(); i++) {
    Require(probs[i] == 0.25, "N emission must be 0.25");
    n_emission_[GetAlphabetIndex(alphabet, names[i][0])] = probs[i];
  }
  Require(n_state["extras"]["germline"].as_string() == "N", "padding germline must be N");
  Require(n_state["extras"]["ambiguous_emission_prob"].as_double() == 0.25, "ambiguous_emission_prob");
}

VGermlinePtr GermlineGene::VGermlinePtrCast() const {
  Require(type == GermlineType::V, "not a V gene");
  return std::static_pointer_cast<VGermline>(germ_ptr);
}
DGermlinePtr GermlineGene::DGermlinePtrCast() const {
  Require(type == GermlineType::D, "not a D gene");
  return std::static_pointer_cast<DGermline>(germ_ptr);
}
JGermlinePtr GermlineGene::JGermlinePtrCast() const {
  Require(type == GermlineType::J, "not a J gene");
  return std::static_pointer_cast<JGermline>(germ_ptr);
}
const NTInsertion& GermlineGene::nti() const {
  if (type == GermlineType::D) return *DGermlinePtrCast();
  return *JGermlinePtrCast();
}
const NPadding& GermlineGene::npadding() const {
  if (type == GermlineType::V) return *VGermlinePtrCast();
  return *JGermlinePtrCast();
}

// src/VDJGermline.cpp:46-108.  File names must match ^(IG([HKL])([VDJ]).*_star_.*)\.yaml$.
std::unordered_map<std::string, GermlineGene> CreateGermlineGeneMap(std::string hmm_param_dir) {
  if (hmm_param_dir.empty() || hmm_param_dir.back() != '/') hmm_param_dir += "/";
  DIR* dir = opendir(hmm_param_dir.c_str());
  if (dir == nullptr) throw std::runtime_error("--hmm-param-dir \"" + hmm_param_dir + "\" does not exist");
  std::vector<std::string> files;
  struct dirent* e;
  while ((e = readdir(dir)) != nullptr) files.push_back(e->d_name);
  closedir(dir);
  std::sort(files.begin(), files.end());
  // which files are per-allele parameter files (src/VDJGermline.cpp:56,81)
  struct Item {
    std::string file, gname;
    char seg;
    GermlineGene ggene;
    std::exception_ptr error;
  };
  std::vector<Item> items;
  for (const std::string& fn : files) {
    if (fn.size() < 10 || fn.compare(0, 2, "IG") != 0 || fn.compare(fn.size() - 5, 5, ".yaml") != 0) continue;
    const char locus = fn[2], seg = fn[3];
    if (locus != 'H' && locus != 'K' && locus != 'L') continue;
    if (seg != 'V' && seg != 'D' && seg != 'J') continue;
    const std::string stem = fn.substr(0, fn.size() - 5);
    if (stem.find("_star_", 4) == std::string::npos) continue;
    if (seg == 'D' && (locus == 'K' || locus == 'L')) continue;
    items.push_back(Item{hmm_param_dir + fn, FixGeneName(stem), seg, GermlineGene(), nullptr});
  }
  // a full germline set is a few hundred files, each parsed independently: spread over the cores
  auto load = [&](std::size_t lo, std::size_t hi) {
    for (std::size_t i = lo; i < hi; ++i) {
      Item& it = items[i];
      try {
        const yaml_lite::Node root = yaml_lite::LoadFile(it.file);
        if (it.seg == 'V') {
          it.ggene.type = GermlineType::V;
          it.ggene.germ_ptr.reset(new VGermline(root));
        } else if (it.seg == 'D') {
          it.ggene.type = GermlineType::D;
          it.ggene.germ_ptr.reset(new DGermline(root));
        } else {
          it.ggene.type = GermlineType::J;
          it.ggene.germ_ptr.reset(new JGermline(root));
        }
      } catch (...) {
        it.error = std::current_exception();
      }
    }
  };
  const std::size_t n_threads =
      std::max<std::size_t>(1, std::min<std::size_t>(std::min(std::thread::hardware_concurrency(), 16u), items.size() / 8));
  if (n_threads == 1) {
    load(0, items.size());
  } else {
    std::vector<std::thread> pool;
    for (std::size_t t = 0; t < n_threads; ++t)
      pool.emplace_back(load, items.size() * t / n_threads, items.size() * (t + 1) / n_threads);
    for (std::thread& th : pool) th.join();
  }
  std::unordered_map<std::string, GermlineGene> ggenes;
  std::string alphabet;
  for (Item& it : items) {  // file order, as the serial loop reported its first error
    if (it.error) std::rethrow_exception(it.error);
    if (alphabet.empty()) alphabet = it.ggene.germ_ptr->alphabet();
    Require(alphabet == it.ggene.germ_ptr->alphabet(), "all germline alphabets must be identical");
    ggenes.emplace(it.gname, it.ggene);
  }
  return ggenes;
}

}  // namespace linearham
