#include "utils.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <numeric>

namespace linearham {

// src/utils.cpp:20-35
std::pair<std::vector<std::string>, VectorXd> ParseStringProbMap(const yaml_lite::Node& node) {
  Require(node.IsMap(), "probability map expected");
  std::vector<std::string> names(node.size());
  VectorXd probs(node.size());
  double sum = 0;
  int i = 0;
  for (const auto& kv : node.map) {
    names[i] = kv.first;
    probs[i] = kv.second.as_double();
    sum += probs[i];
    i++;
  }
  Require(std::fabs(sum - 1) <= EPS, "probability map does not sum to one");
  return {names, probs};
}

// src/utils.cpp:43-51
std::string GetAlphabet(const yaml_lite::Node& root) {
  Require(root.IsMap(), "YAML root must be a map");
  std::string a;
  for (const auto& n : root["tracks"]["nukes"].seq) a.push_back(n.as_char());
  std::sort(a.begin(), a.end());
  return a;
}

// src/utils.cpp:61-66
int GetAlphabetIndex(const std::string& alphabet, char base) {
  const auto it = std::find(alphabet.begin(), alphabet.end(), base);
  Require(it != alphabet.end(), std::string("base not in alphabet: ") + base);
  return it - alphabet.begin();
}

// GetGermlineStateRegex ("^<gname>_([0-9]+)$", src/utils.cpp:74-79) without <regex>
bool MatchGermlineState(const std::string& s, const std::string& gname, int* index) {
  if (s.size() <= gname.size() + 1 || s.compare(0, gname.size(), gname) != 0 || s[gname.size()] != '_')
    return false;
  int v = 0;
  for (std::size_t i = gname.size() + 1; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (s[i] - '0');
  }
  *index = v;
  return true;
}

// GetNTIStateRegex ("^insert_left_([ACGT])$", src/utils.cpp:87-89)
bool MatchNTIState(const std::string& s, const std::string& alphabet, char* base) {
  static const std::string prefix = "insert_left_";
  if (s.size() != prefix.size() + 1 || s.compare(0, prefix.size(), prefix) != 0) return false;
  if (alphabet.find(s.back()) == std::string::npos) return false;
  *base = s.back();
  return true;
}

// src/utils.cpp:110-125
std::pair<int, int> FindGermlineStartEnd(const yaml_lite::Node& root, const std::string& gname) {
  const yaml_lite::Node& states = root["states"];
  int gstart = 0, gend = (int)states.size() - 1;
  while (states[gstart]["name"].as_string().find(gname) == std::string::npos) gstart++;
  while (states[gend]["name"].as_string().find(gname) == std::string::npos) gend--;
  return {gstart, gend};
}

// src/utils.cpp:155-164
VectorXi ConvertSeqToInts(const std::string& seq_str, const std::string& alphabet) {
  VectorXi seq(seq_str.size());
  for (std::size_t i = 0; i < seq_str.size(); i++) seq[i] = GetAlphabetIndex(alphabet, seq_str[i]);
  return seq;
}

// src/utils.cpp:175-184
std::string ConvertIntsToSeq(const VectorXi& seq, const std::string& alphabet) {
  std::string s(seq.size(), ' ');
  for (std::size_t i = 0; i < seq.size(); i++) s[i] = alphabet.at(seq[i]);
  return s;
}

// "_star_" -> "*", "_slash_" -> "/" (src/Germline.cpp:43-44)
std::string FixGeneName(std::string name) {
  for (const auto& rep : {std::make_pair(std::string("_star_"), std::string("*")),
                          std::make_pair(std::string("_slash_"), std::string("/"))}) {
    std::size_t p;
    while ((p = name.find(rep.first)) != std::string::npos) name.replace(p, rep.first.size(), rep.second);
  }
  return name;
}

const HostOptions& host_options() {
  static const HostOptions opts = [] {
    HostOptions o;
    o.pipeline_timing = std::getenv("LH_PIPELINE_TIMING") != nullptr;
    o.host_sampling = std::getenv("LH_HOST_SAMPLING") != nullptr;
    if (const char* e = std::getenv("LH_HOST_THREADS")) o.host_threads = std::max(1, std::atoi(e));
    return o;
  }();
  return opts;
}

double StageTimer::Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
StageTimer::StageTimer() : on(host_options().pipeline_timing), t0(Now()) {}
void StageTimer::Mark(const char* what) {
  if (!on) return;
  const double t = Now();
  std::fprintf(stderr, "[host] %-28s %.3f s\n", what, t - t0);
  t0 = t;
}

}  // namespace linearham
