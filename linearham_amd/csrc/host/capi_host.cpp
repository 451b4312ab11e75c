// Test/bench-facing C entry points of liblinearham_host.so: construct the C++ host classes and dump
// their accessors as JSON so that the pytest suite can compare them with the reference's goldens
// (test/test.cpp) -- the role the Catch binary plays upstream.
#include <cmath>
#include <cstring>
#include <iomanip>
#include <sstream>
#include <random>
#include <string>

#include "PhyloHMM.hpp"
#include "SimpleHMM.hpp"

using namespace linearham;

namespace {

thread_local std::string g_err, g_out;

struct Json {
  std::ostringstream o;
  bool first = true;
  Json() { o << std::setprecision(17); }
  void key(const std::string& k) {
    o << (first ? "{" : ",") << "\"" << k << "\":";
    first = false;
  }
  static std::string esc(const std::string& s) {
    std::string r = "\"";
    for (char c : s) {
      if (c == '"' || c == '\\') r.push_back('\\');
      r.push_back(c);
    }
    return r + "\"";
  }
  void num(double v) {
    if (std::isnan(v)) o << "NaN";
    else if (std::isinf(v)) o << (v > 0 ? "Infinity" : "-Infinity");
    else o << v;
  }
  void put(const std::string& k, const std::string& v) { key(k); o << esc(v); }
  void put(const std::string& k, int v) { key(k); o << v; }
  void put(const std::string& k, bool v) { key(k); o << (v ? "true" : "false"); }
  void put(const std::string& k, double v) { key(k); num(v); }
  void put(const std::string& k, const std::vector<int>& v) {
    key(k); o << "[";
    for (std::size_t i = 0; i < v.size(); ++i) o << (i ? "," : "") << v[i];
    o << "]";
  }
  void put(const std::string& k, const std::vector<double>& v) {
    key(k); o << "[";
    for (std::size_t i = 0; i < v.size(); ++i) { if (i) o << ","; num(v[i]); }
    o << "]";
  }
  void put(const std::string& k, const std::vector<std::string>& v) {
    key(k); o << "[";
    for (std::size_t i = 0; i < v.size(); ++i) o << (i ? "," : "") << esc(v[i]);
    o << "]";
  }
  void put(const std::string& k, const std::vector<GermlineType>& v) {
    key(k); o << "[";
    for (std::size_t i = 0; i < v.size(); ++i)
      o << (i ? "," : "") << (v[i] == GermlineType::V ? "\"V\"" : v[i] == GermlineType::D ? "\"D\"" : "\"J\"");
    o << "]";
  }
  template <typename T>
  void put(const std::string& k, const Matrix<T>& m) {
    key(k); o << "[";
    for (int r = 0; r < m.rows(); ++r) {
      o << (r ? ",[" : "[");
      for (int c = 0; c < m.cols(); ++c) { if (c) o << ","; num((double)m(r, c)); }
      o << "]";
    }
    o << "]";
  }
  void put(const std::string& k, const GeneRanges& g) {
    key(k); o << "{";
    bool f = true;
    for (const auto& kv : g) {
      o << (f ? "" : ",") << esc(kv.first) << ":[" << kv.second.first << "," << kv.second.second << "]";
      f = false;
    }
    o << "}";
  }
  void put(const std::string& k, const std::map<std::string, int>& g) {
    key(k); o << "{";
    bool f = true;
    for (const auto& kv : g) { o << (f ? "" : ",") << esc(kv.first) << ":" << kv.second; f = false; }
    o << "}";
  }
  std::string str() { return first ? "{}" : o.str() + "}"; }
};

void DumpHMM(const HMM& h, Json& j) {
  j.put("locus", h.locus());
  j.put("flexbounds", GeneRanges(h.flexbounds().begin(), h.flexbounds().end()));
  j.put("relpos", h.relpos());
  j.put("alphabet", h.alphabet());
  j.put("msa", h.msa());
#define DUMP(name) j.put(#name, h.name());
  DUMP(vpadding_ggene_ranges) DUMP(vpadding_naive_bases) DUMP(vpadding_site_inds)
  DUMP(vgerm_state_strs) DUMP(vgerm_left_del) DUMP(vgerm_right_del) DUMP(vgerm_ggene_ranges)
  DUMP(vgerm_naive_bases) DUMP(vgerm_germ_inds) DUMP(vgerm_site_inds)
  DUMP(vd_junction_state_strs) DUMP(vd_junction_del) DUMP(vd_junction_ggene_types)
  DUMP(vd_junction_ggene_ranges) DUMP(vd_junction_naive_bases) DUMP(vd_junction_germ_inds)
  DUMP(vd_junction_site_inds)
  DUMP(dgerm_state_strs) DUMP(dgerm_left_del) DUMP(dgerm_right_del) DUMP(dgerm_ggene_ranges)
  DUMP(dgerm_naive_bases) DUMP(dgerm_germ_inds) DUMP(dgerm_site_inds)
  DUMP(dj_junction_state_strs) DUMP(dj_junction_del) DUMP(dj_junction_ggene_types)
  DUMP(dj_junction_ggene_ranges) DUMP(dj_junction_naive_bases) DUMP(dj_junction_germ_inds)
  DUMP(dj_junction_site_inds)
  DUMP(jgerm_state_strs) DUMP(jgerm_left_del) DUMP(jgerm_right_del) DUMP(jgerm_ggene_ranges)
  DUMP(jgerm_naive_bases) DUMP(jgerm_germ_inds) DUMP(jgerm_site_inds)
  DUMP(jpadding_ggene_ranges) DUMP(jpadding_naive_bases) DUMP(jpadding_site_inds)
  DUMP(vpadding_transition) DUMP(vgerm_vd_junction_transition) DUMP(vd_junction_transition)
  DUMP(vd_junction_dgerm_transition) DUMP(dgerm_dj_junction_transition) DUMP(dj_junction_transition)
  DUMP(dj_junction_jgerm_transition) DUMP(jpadding_transition) DUMP(cache_forward)
}

void DumpForward(const HMM& h, Json& j) {
  DUMP(vgerm_forward) DUMP(vd_junction_forward) DUMP(dgerm_forward) DUMP(dj_junction_forward) DUMP(jgerm_forward)
  DUMP(vgerm_scaler_count) DUMP(vd_junction_scaler_counts) DUMP(dgerm_scaler_count)
  DUMP(dj_junction_scaler_counts) DUMP(jgerm_scaler_count)
}

void DumpSample(const HMM& h, Json& j) {
  DUMP(naive_seq_samp) DUMP(vgerm_state_str_samp) DUMP(vgerm_state_ind_samp) DUMP(vgerm_left_del_samp)
  DUMP(vgerm_right_del_samp) DUMP(vgerm_left_insertion_samp) DUMP(vd_junction_state_str_samps)
  DUMP(vd_junction_state_ind_samps) DUMP(vd_junction_insertion_samp) DUMP(dgerm_state_str_samp)
  DUMP(dgerm_state_ind_samp) DUMP(dgerm_left_del_samp) DUMP(dgerm_right_del_samp)
  DUMP(dj_junction_state_str_samps) DUMP(dj_junction_state_ind_samps) DUMP(dj_junction_insertion_samp)
  DUMP(jgerm_state_str_samp) DUMP(jgerm_state_ind_samp) DUMP(jgerm_left_del_samp) DUMP(jgerm_right_del_samp)
  DUMP(jgerm_right_insertion_samp)
}

void DumpPhylo(const PhyloHMM& h, Json& j) {
  DUMP(xmsa) DUMP(xmsa_labels) DUMP(xmsa_seqs) DUMP(xmsa_naive_ind) DUMP(vpadding_xmsa_inds) DUMP(vgerm_xmsa_inds)
  DUMP(vd_junction_xmsa_inds) DUMP(dgerm_xmsa_inds) DUMP(dj_junction_xmsa_inds) DUMP(jgerm_xmsa_inds)
  DUMP(jpadding_xmsa_inds)
}
#undef DUMP

template <typename F>
int Guard(F f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

}  // namespace

extern "C" {

const char* lhh_last_error() { return g_err.c_str(); }

// Germline parameter parsing only (no GPU): JSON of one allele file parsed as V, D or J.
int lhh_germline_json(const char* yaml_path, char type, const char** out) {
  return Guard([&] {
    const yaml_lite::Node root = yaml_lite::LoadFile(yaml_path);
    Json j;
    const Germline g(root);
    j.put("landing_in", g.landing_in());
    j.put("landing_out", g.landing_out());
    j.put("transition", g.transition());
    j.put("gene_prob", g.gene_prob());
    j.put("alphabet", g.alphabet());
    j.put("name", g.name());
    j.put("emission", g.emission());
    j.put("bases", g.bases());
    j.put("length", g.length());
    if (type == 'D' || type == 'J') {
      const NTInsertion n(root);
      j.put("nti_landing_in", n.nti_landing_in());
      j.put("nti_landing_out", n.nti_landing_out());
      j.put("nti_transition", n.nti_transition());
      j.put("nti_emission", n.nti_emission());
    }
    if (type == 'V' || type == 'J') {
      const NPadding p(root);
      j.put("n_transition", p.n_transition());
      j.put("n_emission", p.n_emission());
    }
    g_out = j.str();
    *out = g_out.c_str();
  });
}

int lhh_simple_create(const char* yaml, int cluster_ind, const char* dir, int seed, void** out) {
  return Guard([&] { *out = new SimpleHMM(yaml, cluster_ind, dir, seed); });
}
int lhh_phylo_create(const char* yaml, int cluster_ind, const char* dir, int seed, void** out) {
  return Guard([&] { *out = static_cast<HMM*>(new PhyloHMM(yaml, cluster_ind, dir, seed)); });
}
void lhh_destroy(void* h) { delete static_cast<HMM*>(h); }

int lhh_phylo_init_parameters(void* h, const char* newick_path, const double* er, const double* pi, double alpha,
                              int num_rates) {
  return Guard([&] {
    dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h))
        .InitializePhyloParameters(newick_path, std::vector<double>(er, er + 6), std::vector<double>(pi, pi + 4),
                                   alpha, num_rates);
  });
}
int lhh_phylo_init_parameters_str(void* h, const char* newick, const double* er, const double* pi, double alpha,
                                  int num_rates) {
  return Guard([&] {
    dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h))
        .InitializePhyloParametersFromString(newick, std::vector<double>(er, er + 6),
                                             std::vector<double>(pi, pi + 4), alpha, num_rates);
  });
}
int lhh_phylo_init_emission(void* h) {
  return Guard([&] { dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h)).InitializePhyloEmission(); });
}
int lhh_phylo_set_extended_range(void* h, int on) {
  return Guard([&] { dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h)).SetExtendedRange(on != 0); });
}
int lhh_loglikelihood(void* h, double* out) {
  return Guard([&] { *out = static_cast<HMM*>(h)->LogLikelihood(); });
}
int lhh_sample(void* h, const char** out) {
  return Guard([&] {
    g_out = static_cast<HMM*>(h)->SampleNaiveSequence();
    *out = g_out.c_str();
  });
}
// Test entry (PhyloHMM::SampleStatesWithWords): device_states / host_states [n_states]; returns the count in *n_states.
int lhh_phylo_sample_words(void* h, const uint32_t* words, int n_words, int32_t* device_states, int32_t* host_states,
                           int cap, int* n_states) {
  return Guard([&] {
    std::vector<int32_t> d, s;
    static_cast<linearham::PhyloHMM*>(h)->SampleStatesWithWords(words, n_words, d, s);
    if ((int)d.size() > cap || d.size() != s.size()) throw std::runtime_error("lhh_phylo_sample_words: state count");
    std::copy(d.begin(), d.end(), device_states);
    std::copy(s.begin(), s.end(), host_states);
    *n_states = (int)d.size();
  });
}

int lhh_run_pipeline(void* h, const char* input_path, const char* output_path, int num_rates) {
  return Guard([&] { dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h)).RunPipeline(input_path, output_path, num_rates); });
}
int lhh_run_asr(void* h, const char* input_path, const char* output_path, uint64_t seed) {
  return Guard([&] { dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h)).RunAsr(input_path, output_path, seed); });
}
// what: bit 0 = state space + transitions, bit 1 = forward arrays, bit 2 = sample, bit 3 = xMSA structures

int lhh_dump_json(void* h, int what, const char** out) {
  return Guard([&] {
    HMM* hmm = static_cast<HMM*>(h);
    Json j;
    if (what & 1) DumpHMM(*hmm, j);
    if (what & 2) DumpForward(*hmm, j);
    if (what & 4) DumpSample(*hmm, j);
    if (what & 8) {
      PhyloHMM& p = dynamic_cast<PhyloHMM&>(*hmm);
      DumpPhylo(p, j);
      j.put("xmsa_emission", p.xmsa_emission());
      j.put("sr", p.sr());
      j.put("er", p.er());
      j.put("pi", p.pi());
      j.put("alpha", p.alpha());
    }
    g_out = j.str();
    *out = g_out.c_str();
  });
}

// Flatten a RevBayes table into the device inputs of lh_eval_batch_device (bench.py): fills
// caller-allocated arrays; returns max_depth through *max_depth.  n rows are taken cyclically from
// the table.  ops [n][T-2][4], brlen [n][2T-2], er [n][6], pi [n][4], alpha [n].
int lhh_phylo_flatten_tsv(void* h, const char* tsv_path, int n, int32_t* ops, double* brlen, double* er,
                          double* pi, double* alpha, int* n_tips, int* max_depth, int* n_rows_in_file,
                          int need_family, void** family) {
  return Guard([&] {
    PhyloHMM& p = dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h));
    int n_file = 0;
    const PhyloHMM::DeviceBatch b = p.FlattenTsv(tsv_path, &n_file);
    *n_rows_in_file = n_file;
    struct {
      std::size_t n;
      std::size_t size() const { return n; }
    } rows{(std::size_t)n_file};
    *n_tips = b.n_tips;
    *max_depth = b.max_depth;
    *family = need_family ? p.family() : nullptr;
    if (n > 0 && ops) {
      const std::size_t T = b.n_tips, no = (T - 2) * 4, nb = 2 * T - 2;
      for (int s = 0; s < n; ++s) {
        const std::size_t r = s % rows.size();
        std::memcpy(ops + s * no, b.ops.data() + r * no, no * sizeof(int32_t));
        std::memcpy(brlen + s * nb, b.brlen.data() + r * nb, nb * sizeof(double));
        std::memcpy(er + s * 6, b.er.data() + r * 6, 6 * sizeof(double));
        std::memcpy(pi + s * 4, b.pi.data() + r * 4, 4 * sizeof(double));
        alpha[s] = b.alpha[r];
      }
    }
  });
}

// The rows `row_ids[0..n)` of the table (any order, repeats allowed) as device inputs: only these rows are parsed and
// scheduled (a rank of a multi-GPU run flattens the rows it evaluates, not the whole table).
int lhh_phylo_flatten_tsv_rows(void* h, const char* tsv_path, int n, const int64_t* row_ids, int32_t* ops, double* brlen,
                               double* er, double* pi, double* alpha, int* n_tips, int* max_depth, int* n_rows_in_file,
                               int need_family, void** family) {
  return Guard([&] {
    PhyloHMM& p = dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h));
    int n_file = 0;
    const PhyloHMM::DeviceBatch b = p.FlattenTsvRows(tsv_path, row_ids, n, &n_file);
    *n_rows_in_file = n_file;
    *n_tips = b.n_tips;
    *max_depth = b.max_depth;
    *family = need_family ? p.family() : nullptr;
    std::memcpy(ops, b.ops.data(), b.ops.size() * sizeof(int32_t));
    std::memcpy(brlen, b.brlen.data(), b.brlen.size() * sizeof(double));
    std::memcpy(er, b.er.data(), b.er.size() * sizeof(double));
    std::memcpy(pi, b.pi.data(), b.pi.size() * sizeof(double));
    std::memcpy(alpha, b.alpha.data(), b.alpha.size() * sizeof(double));
  });
}

int lhh_phylo_set_devices(void* h, const int* devices, int n) {
  return Guard([&] { dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h)).SetDevices(std::vector<int>(devices, devices + n)); });
}

int lhh_phylo_sizes(void* h, int* n_tips, int* n_sites, int* n_xmsa, int* s_vd, int* s_dj, int* w_vd, int* w_dj,
                    int* g_total) {
  return Guard([&] {
    PhyloHMM& p = dynamic_cast<PhyloHMM&>(*static_cast<HMM*>(h));
    *n_tips = p.msa().rows() + 1;
    *n_sites = p.msa().cols();
    *n_xmsa = p.n_xmsa();
    *s_vd = (int)p.vd_junction_state_strs().size();
    *s_dj = (int)p.dj_junction_state_strs().size();
    *w_vd = p.vd_junction_xmsa_inds().rows();
    *w_dj = p.dj_junction_xmsa_inds().rows();
    *g_total = (int)(p.vpadding_xmsa_inds().size() + p.vgerm_xmsa_inds().size() + p.dgerm_xmsa_inds().size() +
                     p.jgerm_xmsa_inds().size() + p.jpadding_xmsa_inds().size());
  });
}

// Self-test of DrawDiscreteSparse (HMM.hpp) against std::discrete_distribution on the full weight vector:
// random sizes and sparsity patterns plus the corner cases (first / last weight zero or not, a single
// non-zero weight, all weights zero, fewer than two weights).  Both generators start from the same seed
// and must agree on every sampled index and stay in the same state.  Returns the number of disagreements.
int lhh_selftest_sparse_draw(int seed, int trials) {
  std::mt19937 gen(seed), a(seed + 1), b(seed + 1);
  int bad = 0;
  for (int t = 0; t < trials; ++t) {
    const int size = (t % 17 == 0) ? 1 : 2 + (int)(gen() % 60);
    std::vector<double> w(size, 0.0);
    const int mode = (int)(gen() % 6);
    for (int i = 0; i < size; ++i) {
      const bool on = mode == 0 ? (gen() % 4 == 0) : mode == 1 ? (i == size - 1) : mode == 2 ? (i == 0)
                    : mode == 3 ? false : mode == 4 ? (i != 0 && i != size - 1 && gen() % 3 == 0) : true;
      if (on) w[i] = std::ldexp((double)(gen() % 100000 + 1), -(int)(gen() % 600));
    }
    std::vector<int> idx;
    std::vector<double> nz;
    for (int i = 0; i < size; ++i)
      if (w[i] != 0.0) {
        idx.push_back(i);
        nz.push_back(w[i]);
      }
    std::discrete_distribution<int> dist;
    dist.param(std::discrete_distribution<int>::param_type(w.data(), w.data() + size));
    const int dense = dist(a);
    const int sparse = DrawDiscreteSparse(b, idx.data(), nz.data(), (int)idx.size(), size);
    if (dense != sparse || a != b) ++bad;
  }
  return bad;
}

// Self-test of ParseDouble / AppendFixed6 (newick.hpp) against strtod / printf("%f"): random values in the forms
// a RevBayes table and libpll's exporter produce, plus exact ties of the sixth decimal.  Returns the mismatches.
int lhh_selftest_numbers(int seed, int trials) {
  std::mt19937_64 rng((uint64_t)seed);
  int bad = 0;
  char buf[96];
  auto check_parse = [&](const char* s) {
    const char* e1 = nullptr;
    char* e2 = nullptr;
    const double a = ParseDouble(s, &e1), b = std::strtod(s, &e2);
    if (std::memcmp(&a, &b, sizeof a) != 0 || e1 != e2) ++bad;
  };
  auto check_fixed = [&](double v) {
    std::string o;
    AppendFixed6(o, v);
    std::snprintf(buf, sizeof buf, "%f", v);
    if (o != buf) ++bad;
  };
  const char* fixed_cases[] = {"0", "0.0", "1e-06", "1E-6", "0.0078125", ".5", "5.", "12345678901234567890", "1e400",
                               "0.1e-320", "123.456e+10", "7e", "1e+", "0.30000000000000004", "9007199254740993",
                               "4.9e-324", "x", "", "-0.25", "+3.5e2:"};
  for (const char* c : fixed_cases) check_parse(c);
  for (int i = 0; i < trials; ++i) {
    const double u = (double)(rng() >> 11) * 0x1p-53;
    const int e = (int)(rng() % 14) - 9;
    const double v = u * std::pow(10.0, e);
    const char* fmts[] = {"%.6g", "%.8g", "%.10g", "%.17g", "%f", "%.12f", "%e"};
    std::snprintf(buf, sizeof buf, fmts[rng() % 7], v);
    check_parse(buf);
    check_fixed(v);
    // exact halves of the sixth decimal that are binary fractions: (2m + 1) * 15625 / 2^(7 + j)
    const double tie = (double)(2 * (rng() % 4096) + 1) * 15625.0 * std::ldexp(1.0, -7 - (int)(rng() % 6)) / 15625.0 /
                       15625.0 * 15625.0;
    check_fixed(tie);
    check_fixed((double)(rng() % 2000001) * 0.0078125);
    check_fixed((double)(rng() % 1000) + 0.5e-6 * (double)(rng() % 3));
  }
  check_fixed(0.0078125);
  check_fixed(0.0);
  check_fixed(1e-7);
  check_fixed(4e9);
  check_fixed(123456789.9999995);
  return bad;
}

// Newick ingest + re-export without a device: `labels` is a '\n'-separated list (first = naive).  *out is the
// output table's tree column (ExportNewick); children/brlen/root may be NULL.
int lhh_newick_roundtrip(const char* newick, const char* labels, const char** out, int32_t* children,
                         double* brlen, int32_t* root) {
  return Guard([&] {
    std::vector<std::string> labs;
    std::string cur;
    for (const char* c = labels; *c; ++c) {
      if (*c == '\n') {
        labs.push_back(cur);
        cur.clear();
      } else {
        cur.push_back(*c);
      }
    }
    if (!cur.empty()) labs.push_back(cur);
    const TreeArrays tr = ParseNewick(newick, labs, EPS, true);
    g_out = ExportNewick(tr, labs);
    *out = g_out.c_str();
    if (children) std::copy(tr.children.begin(), tr.children.end(), children);
    if (brlen) std::copy(tr.brlen.begin(), tr.brlen.end(), brlen);
    if (root) *root = tr.root;
  });
}

}  // extern "C"
