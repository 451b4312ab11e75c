#include "PhyloHMM.hpp"

#include <cerrno>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

namespace linearham {

// src/PhyloHMM.cpp:29-34
PhyloHMM::PhyloHMM(const std::string& yaml_path, int cluster_ind, const std::string& hmm_param_dir, int seed)
    : HMM(yaml_path, cluster_ind, hmm_param_dir, seed) {
  InitializeXmsaStructs();
}

// src/PhyloHMM.cpp:45-89 (+ BuildXmsa, :123-144)
void PhyloHMM::InitializeXmsaStructs() {
  xmsa_labels_.push_back("naive");
  for (const auto& n : cluster_data_["unique_ids"].seq) xmsa_labels_.push_back(n.as_string());
  xmsa_naive_ind_ = 0;
  std::map<std::pair<int, int>, int> xmsa_ids;
  StoreGermlinePaddingXmsaIndices(vpadding_.naive_bases, vpadding_.site_inds, xmsa_ids, vpadding_xmsa_inds_);
  StoreGermlinePaddingXmsaIndices(vgerm_.naive_bases, vgerm_.site_inds, xmsa_ids, vgerm_xmsa_inds_);
  if (locus_ == "igh") {
    StoreJunctionXmsaIndices(vd_junction_.naive_bases, vd_junction_.site_inds, flexbounds_.at("v_r"),
                             flexbounds_.at("d_l"), xmsa_ids, vd_junction_xmsa_inds_);
    StoreGermlinePaddingXmsaIndices(dgerm_.naive_bases, dgerm_.site_inds, xmsa_ids, dgerm_xmsa_inds_);
    StoreJunctionXmsaIndices(dj_junction_.naive_bases, dj_junction_.site_inds, flexbounds_.at("d_r"),
                             flexbounds_.at("j_l"), xmsa_ids, dj_junction_xmsa_inds_);
  } else {
    StoreJunctionXmsaIndices(vd_junction_.naive_bases, vd_junction_.site_inds, flexbounds_.at("v_r"),
                             flexbounds_.at("j_l"), xmsa_ids, vd_junction_xmsa_inds_);
  }
  StoreGermlinePaddingXmsaIndices(jgerm_.naive_bases, jgerm_.site_inds, xmsa_ids, jgerm_xmsa_inds_);
  StoreGermlinePaddingXmsaIndices(jpadding_.naive_bases, jpadding_.site_inds, xmsa_ids, jpadding_xmsa_inds_);

  const int n = msa_.rows(), C = (int)xmsa_ids.size();
  xmsa_.setConstant(n + 1, C, -1);
  xmsa_site_.assign(C, 0);
  xmsa_base_.assign(C, 0);
  for (auto it = xmsa_ids.begin(); it != xmsa_ids.end(); ++it) {
    const int naive_base = it->first.first, msa_ind = it->first.second, xmsa_ind = it->second;
    xmsa_(0, xmsa_ind) = naive_base;
    for (int r = 0; r < n; ++r) xmsa_(r + 1, xmsa_ind) = msa_(r, msa_ind);
    xmsa_site_[xmsa_ind] = msa_ind;
    xmsa_base_[xmsa_ind] = (uint8_t)naive_base;
  }
  xmsa_seqs_.assign(n + 1, "");
  for (int r = 0; r < n + 1; ++r) {
    VectorXi row(xmsa_.row(r), xmsa_.row(r) + C);
    xmsa_seqs_[r] = ConvertIntsToSeq(row, alphabet_);
  }
}

namespace {

// Worker threads a host stage may start: the cores this process may run on (its affinity mask; a container's share can
// be smaller than the machine), at most `cap`; LH_HOST_THREADS overrides the count (experiments, small containers).
// The stages of RunPipeline run side by side with up to 16 workers each (measured on a 256-thread host, 262144 rows of
// configs[2]: 8 workers 1.22 s, 16 1.02 s, 24 1.22 s, 32 1.05 s -- profiles/r03_pipeline_e2e.txt).
int HostThreads(int cap) {
  static const int avail = [] {
    if (host_options().host_threads > 0) return host_options().host_threads;
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : 1 << 20, CPU_COUNT(&set));
    return std::max(1, n);
  }();
  return std::max(1, std::min(avail, cap));
}

SegmentTables MakeSegments(const GeneRanges& ranges, const VectorXi& inds) {
  SegmentTables s;
  s.offsets.push_back(0);
  for (auto it = ranges.begin(); it != ranges.end(); ++it) {
    for (int j = it->second.first; j < it->second.second; ++j) s.xmsa_inds.push_back(inds[j]);
    s.offsets.push_back((int32_t)s.xmsa_inds.size());
  }
  return s;
}

}  // namespace

// Upload everything that is constant for this clonal family (lh_family_create).  Done lazily at the
// first evaluation so that the host-only state (state space, transitions, xMSA) can be inspected on a
// machine without a GPU; any evaluation without a GPU fails here (there is no CPU path).
void PhyloHMM::CreateFamily() {
  if (family_) return;
  const bool igh = locus_ == "igh";
  const int n = msa_.rows(), L = msa_.cols();
  std::vector<uint8_t> msa8((std::size_t)n * L);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < L; ++c) msa8[(std::size_t)r * L + c] = (uint8_t)msa_(r, c);
  const SegmentTables vpad = MakeSegments(vpadding_.ggene_ranges, vpadding_xmsa_inds_);
  const SegmentTables vger = MakeSegments(vgerm_.ggene_ranges, vgerm_xmsa_inds_);
  const SegmentTables dger = MakeSegments(dgerm_.ggene_ranges, dgerm_xmsa_inds_);
  const SegmentTables jger = MakeSegments(jgerm_.ggene_ranges, jgerm_xmsa_inds_);
  const SegmentTables jpad = MakeSegments(jpadding_.ggene_ranges, jpadding_xmsa_inds_);
  std::vector<double> gene_prob, trans_prod;
  for (auto it = vgerm_.ggene_ranges.begin(); it != vgerm_.ggene_ranges.end(); ++it) {
    const Germline& g = *ggenes_.at(it->first).germ_ptr;
    gene_prob.push_back(g.gene_prob());
    const int gis = vgerm_.germ_inds[it->second.first];
    double prod = 1.0;  // src/HMM.cpp:310-313
    for (int k = 0; k < it->second.second - it->second.first - 1; ++k) prod *= g.transition()[gis + k];
    trans_prod.push_back(prod);
  }
  JunctionTables vd, dj;
  if (igh) {
    vd = BuildJunctionTables(vd_junction_, vgerm_, dgerm_, flexbounds_.at("v_r"), flexbounds_.at("d_l"),
                             vd_junction_xmsa_inds_);
    dj = BuildJunctionTables(dj_junction_, dgerm_, jgerm_, flexbounds_.at("d_r"), flexbounds_.at("j_l"),
                             dj_junction_xmsa_inds_);
  } else {
    vd = BuildJunctionTables(vd_junction_, vgerm_, jgerm_, flexbounds_.at("v_r"), flexbounds_.at("j_l"),
                             vd_junction_xmsa_inds_);
  }
  lh_family_desc d{};
  d.abi_version = LH_ABI_VERSION;
  d.has_d = igh ? 1 : 0;
  d.n_seqs = n;
  d.n_sites = L;
  d.msa = msa8.data();
  d.n_xmsa = xmsa_.cols();
  d.xmsa_site = xmsa_site_.data();
  d.xmsa_naive_base = xmsa_base_.data();
  d.vpadding = vpad.c();
  d.vgerm = vger.c();
  d.dgerm = dger.c();
  d.jgerm = jger.c();
  d.jpadding = jpad.c();
  d.vgerm_gene_prob = gene_prob.data();
  d.vpadding_transition = vpadding_transition_.data();
  d.vgerm_trans_prod = trans_prod.data();
  d.jpadding_transition = jpadding_transition_.data();
  d.vd = vd.c();
  if (igh) d.dj = dj.c();
  StageTimer timer;
  if (!devices_.empty()) CheckHip(lh_set_device(devices_[0]), "lh_set_device");
  CheckHip(lh_family_create(&d, &family_), "lh_family_create");
  timer.Mark("lh_family_create (+ HIP init)");
  // Device-side naive-sequence sampling (lh_eval_sample_batch).  The one structural condition it has -- the left and
  // the right genes of a junction occupy two blocks of the junction's state vector, true for gene names that start
  // with their locus and segment letters -- is checked by the library; a family that does not meet it keeps the
  // host sampler (the same algorithm, HMM::SampleRow), and so does LH_HOST_SAMPLING=1.
  if (!host_options().host_sampling) {
    SamplerJunction svd, sdj;
    lh_sampler_desc sd{};
    if (igh) {
      svd = BuildSamplerJunction(vd_junction_, vgerm_, dgerm_, flexbounds_.at("v_r"), flexbounds_.at("d_l"));
      sdj = BuildSamplerJunction(dj_junction_, dgerm_, jgerm_, flexbounds_.at("d_r"), flexbounds_.at("j_l"));
      sd.dj = sdj.c();
    } else {
      svd = BuildSamplerJunction(vd_junction_, vgerm_, jgerm_, flexbounds_.at("v_r"), flexbounds_.at("j_l"));
    }
    sd.vd = svd.c();
    device_sampler_ = lh_family_set_sampler(family_, &sd) == 0 && lh_sample_words(family_) == RawDrawsPerSample();
    timer.Mark("lh_family_set_sampler");
    // the other devices' handles: the same descriptors, uploaded with that device current (only when the rows will be
    // sampled on the devices: nothing else shards)
    for (std::size_t k = 1; device_sampler_ && k < devices_.size(); ++k) {
      CheckHip(lh_set_device(devices_[k]), "lh_set_device");
      lh_family* f = nullptr;
      CheckHip(lh_family_create(&d, &f), "lh_family_create");
      more_families_.push_back(f);
      if (device_sampler_ && lh_family_set_sampler(f, &sd) != 0) throw std::runtime_error(lh_last_error());
    }
  }
  // Only RunPipeline's device-sampling branch deals rows to several handles.  A family that keeps the host sampler
  // (LH_HOST_SAMPLING, or genes that do not form two blocks of a junction's state vector) evaluates everything on the
  // first listed device: say so instead of building handles nobody uses.
  if (devices_.size() > 1 && !device_sampler_) {
    std::fprintf(stderr, "linearham: --devices lists %zu devices, but this run samples on the host: all rows are evaluated on "
                 "device %d\n", devices_.size(), devices_[0]);
    for (lh_family* f : more_families_) lh_family_destroy(f);
    more_families_.clear();
  }
  if (devices_.size() > 1) CheckHip(lh_set_device(devices_[0]), "lh_set_device");
}

void PhyloHMM::SetDevices(const std::vector<int>& devices) {
  Require(family_ == nullptr, "SetDevices must be called before the first evaluation");
  for (int dev : devices) Require(dev >= 0 && dev < lh_device_count(), "SetDevices: no such device");
  devices_ = devices;
}

// src/PhyloHMM.cpp:350-361
void PhyloHMM::InitializePhyloParameters(const std::string& newick_path, const std::vector<double>& er,
                                         const std::vector<double>& pi, double alpha, int num_rates) {
  std::ifstream in(newick_path);
  if (!in) throw std::runtime_error("Can't open Newick file " + newick_path);
  std::stringstream ss;
  ss << in.rdbuf();
  InitializePhyloParametersFromString(ss.str(), er, pi, alpha, num_rates);
}

void PhyloHMM::InitializePhyloParametersFromString(const std::string& newick, const std::vector<double>& er,
                                                   const std::vector<double>& pi, double alpha, int num_rates) {
  Require(er.size() == 6 && pi.size() == 4, "er must have 6 and pi 4 entries");
  Require(num_rates >= 1, "num_rates must be positive");
  tree_ = ParseNewick(newick, xmsa_labels_, EPS, true);
  have_tree_ = true;
  er_ = er;
  pi_ = pi;
  alpha_ = alpha;
  num_rates_ = num_rates;
  sr_.assign(num_rates, 0.0);
}

PhyloHMM::DeviceBatch PhyloHMM::FlattenBatch(const std::vector<TreeSample>& samples, std::vector<TreeArrays>* trees,
                                             std::vector<std::string>* exported) const {
  DeviceBatch b;
  const int T = (int)xmsa_labels_.size();
  b.n = (int)samples.size();
  b.n_tips = T;
  b.ops.resize((std::size_t)b.n * (T - 2) * 4);
  b.brlen.resize((std::size_t)b.n * (2 * T - 2));
  b.er.resize((std::size_t)b.n * 6);
  b.pi.resize((std::size_t)b.n * 4);
  b.alpha.resize(b.n);
  if (trees) trees->resize(b.n);
  if (exported) exported->resize(b.n);
  // Rows are independent (parse, unroot at naive's neighbour, schedule): the GPU evaluates a few million
  // trees per second, one host core flattens a few ten thousand, so the rows are spread over the cores.
  auto flatten_rows = [&](int lo, int hi, int* max_depth) {
    for (int s = lo; s < hi; ++s) {
      const TreeSample& ts = samples[s];
      Require(ts.er.size() == 6 && ts.pi.size() == 4, "er must have 6 and pi 4 entries");
      TreeArrays tr = ParseNewick(ts.newick, xmsa_labels_, EPS, exported != nullptr);
      int32_t depth = 0;
      CheckHip(lh_schedule_tree(T, tr.children.data(), tr.root, b.ops.data() + (std::size_t)s * (T - 2) * 4, &depth),
               "lh_schedule_tree");
      *max_depth = std::max(*max_depth, (int)depth);
      std::copy(tr.brlen.begin(), tr.brlen.end(), b.brlen.begin() + (std::size_t)s * (2 * T - 2));
      std::copy(ts.er.begin(), ts.er.end(), b.er.begin() + (std::size_t)s * 6);
      std::copy(ts.pi.begin(), ts.pi.end(), b.pi.begin() + (std::size_t)s * 4);
      b.alpha[s] = ts.alpha;
      if (exported) {
        (*exported)[s] = ExportNewick(tr, xmsa_labels_);
        tr.as_parsed = std::string();
      }
      if (trees) (*trees)[s] = std::move(tr);
    }
  };
  const int hw = HostThreads(16);
  const int n_threads = std::max(1, std::min(hw, b.n / 64));
  if (n_threads == 1) {
    flatten_rows(0, b.n, &b.max_depth);
  } else {
    std::vector<std::thread> pool;
    std::vector<int> depths(n_threads, 0);
    std::vector<std::exception_ptr> errors(n_threads);
    for (int t = 0; t < n_threads; ++t) {
      const int lo = (int)((long long)b.n * t / n_threads), hi = (int)((long long)b.n * (t + 1) / n_threads);
      pool.emplace_back([&, t, lo, hi] {
        try {
          flatten_rows(lo, hi, &depths[t]);
        } catch (...) {
          errors[t] = std::current_exception();
        }
      });
    }
    for (std::thread& th : pool) th.join();
    for (const std::exception_ptr& e : errors)
      if (e) std::rethrow_exception(e);  // the first failing row range, in file order
    for (int d : depths) b.max_depth = std::max(b.max_depth, d);
  }
  return b;
}

std::vector<double> PhyloHMM::LogLikelihoodBatch(const std::vector<TreeSample>& samples, int num_rates) {
  CreateFamily();
  const DeviceBatch b = FlattenBatch(samples);
  std::vector<double> ll(b.n);
  if (b.n == 0) return ll;
  CheckHip(lh_eval_batch(family_, b.n, b.n_tips, b.max_depth, b.ops.data(), b.brlen.data(), b.er.data(),
                         b.pi.data(), b.alpha.data(), num_rates, ll.data(), nullptr),
           "lh_eval_batch");
  return ll;
}

// src/PhyloHMM.cpp:366-383: one evaluation on the device (gamma rates, P-matrices, pruning, emission
// assembly, forward sweep).  The results are unpacked lazily, like the reference's cache_forward_.
void PhyloHMM::InitializePhyloEmission() {
  Require(have_tree_, "InitializePhyloParameters must be called first");
  CreateFamily();
  const int T = tree_.n_tips;
  std::vector<int32_t> ops((std::size_t)(T - 2) * 4);
  int32_t depth = 0;
  CheckHip(lh_schedule_tree(T, tree_.children.data(), tree_.root, ops.data(), &depth), "lh_schedule_tree");
  xmsa_emission_.assign(xmsa_.cols(), 0.0);
  pending_forward_.assign(lh_forward_size(family_), 0.0);
  pending_scalers_.assign(lh_scaler_size(family_), 0);
  lh_eval_outputs outs{sr_.data(), xmsa_emission_.data(), pending_forward_.data(), pending_scalers_.data()};
  CheckHip(lh_eval_batch(family_, 1, T, depth, ops.data(), tree_.brlen.data(), er_.data(), pi_.data(), &alpha_,
                         num_rates_, &pending_loglik_, &outs),
           "lh_eval_batch");
  cache_forward_ = true;
}

namespace {

// the state word whose tempered value is y (inverse of std::mt19937's output transformation)
uint32_t Untemper(uint32_t y) {
  y ^= y >> 18;
  y ^= (y << 15) & 0xEFC60000u;
  uint32_t t = y;
  for (int i = 0; i < 5; ++i) t = y ^ ((t << 7) & 0x9D2C5680u);
  y = t;
  t = y;
  for (int i = 0; i < 3; ++i) t = y ^ (t >> 11);
  return t;
}

}  // namespace

void PhyloHMM::SampleStatesWithWords(const uint32_t* words, int n_words, std::vector<int32_t>& device_states,
                                     std::vector<int32_t>& host_states) {
  Require(have_tree_, "InitializePhyloParameters must be called first");
  CreateFamily();
  Require(device_sampler_, "the family has no device sampler");
  const int raw = RawDrawsPerSample();
  Require(n_words >= raw && n_words <= 624, "SampleStatesWithWords: need RawDrawsPerSample() .. 624 words");
  const int T = tree_.n_tips;
  std::vector<int32_t> ops((std::size_t)(T - 2) * 4);
  int32_t depth = 0;
  CheckHip(lh_schedule_tree(T, tree_.children.data(), tree_.root, ops.data(), &depth), "lh_schedule_tree");
  // device
  device_states.assign(lh_sample_states(family_), -1);
  double ll = 0;
  std::vector<double> rates(num_rates_);
  CheckHip(lh_eval_sample_batch(family_, 1, T, depth, ops.data(), tree_.brlen.data(), er_.data(), pi_.data(), &alpha_,
                                num_rates_, words, &ll, rates.data(), device_states.data()),
           "lh_eval_sample_batch");
  // host: the same forward arrays, an engine that returns the same words
  std::vector<double> fwd(lh_forward_size(family_));
  std::vector<int32_t> sco(lh_scaler_size(family_));
  lh_eval_outputs outs{nullptr, nullptr, fwd.data(), sco.data()};
  CheckHip(lh_eval_batch(family_, 1, T, depth, ops.data(), tree_.brlen.data(), er_.data(), pi_.data(), &alpha_,
                         num_rates_, &ll, &outs),
           "lh_eval_batch");
  std::ostringstream st;
  for (int i = 0; i < 624; ++i) st << Untemper(i < n_words ? words[i] : 0u) << ' ';
  st << 0;  // position: the next output is the first word
  std::istringstream in(st.str());
  std::mt19937 rng;
  in >> rng;
  for (int i = 0; i < std::min(n_words, 4); ++i) {
    std::mt19937 probe = rng;
    probe.discard(i);
    Require((uint32_t)probe() == words[i], "SampleStatesWithWords: engine state construction failed");
  }
  EnsureSamplingLists();
  RowSampler s;
  SampleRow(s, fwd.data(), rng);
  host_states.clear();
  host_states.push_back(s.jgerm_state_ind);
  if (locus_ == "igh") {
    host_states.insert(host_states.end(), s.dj_junction_state_inds.begin(), s.dj_junction_state_inds.end());
    host_states.push_back(s.dgerm_state_ind);
  }
  host_states.insert(host_states.end(), s.vd_junction_state_inds.begin(), s.vd_junction_state_inds.end());
  host_states.push_back(s.vgerm_state_ind);
}

void PhyloHMM::RunForwardAlgorithm() {
  UnpackForward(pending_forward_.data(), pending_scalers_.data());
  loglikelihood_ = pending_loglik_;
}

// src/PhyloHMM.cpp:244-282
void PhyloHMM::WriteOutputHeaders(std::ofstream& outfile) const {
  outfile << "Iteration\tRBLogLikelihood\tPrior\talpha\t";
  for (std::size_t i = 1; i <= er_.size(); i++) outfile << ("er[" + std::to_string(i) + "]\t");
  for (std::size_t i = 1; i <= pi_.size(); i++) outfile << ("pi[" + std::to_string(i) + "]\t");
  outfile << "tree\t";
  for (std::size_t i = 1; i <= sr_.size(); i++) outfile << ("sr[" + std::to_string(i) + "]\t");
  outfile << "LHLogLikelihood\tLogWeight\tNaiveSequence\tVGene\tV5pDel\tV3pDel\tVFwkInsertion\t";
  if (locus_ == "igh") {
    outfile << "VDInsertion\tDGene\tD5pDel\tD3pDel\tDJInsertion\t";
  } else {
    outfile << "VJInsertion\t";
  }
  outfile << "JGene\tJ5pDel\tJ3pDel\tJFwkInsertion\n";
}

namespace {

// operator<<(std::ostream&, double) with the stream's defaults = printf("%g")
void AppendG(std::string& o, double v) {
  char b[40];
  const int n = std::snprintf(b, sizeof b, "%g", v);
  o.append(b, (std::size_t)n);
}
void AppendInt(std::string& o, long long v) {
  char b[24];
  const int n = std::snprintf(b, sizeof b, "%lld", v);
  o.append(b, (std::size_t)n);
}

}  // namespace

// One line of the output table (src/PhyloHMM.cpp:288-327), from explicit pieces so that RunPipeline's worker
// threads can format rows side by side.
void PhyloHMM::FormatOutputLine(std::string& o, int iteration, double rb_loglikelihood, double prior, double alpha,
                                const double* er, const double* pi, const std::string& tree, const double* sr,
                                int num_rates, double lh_loglikelihood, const RowSampler& s) const {
  AppendInt(o, iteration);
  o.push_back('\t');
  AppendG(o, rb_loglikelihood);
  o.push_back('\t');
  AppendG(o, prior);
  o.push_back('\t');
  AppendG(o, alpha);
  o.push_back('\t');
  for (int k = 0; k < 6; ++k) AppendG(o, er[k]), o.push_back('\t');
  for (int k = 0; k < 4; ++k) AppendG(o, pi[k]), o.push_back('\t');
  o += tree;
  o.push_back('\t');
  for (int k = 0; k < num_rates; ++k) AppendG(o, sr[k]), o.push_back('\t');
  AppendG(o, lh_loglikelihood);
  o.push_back('\t');
  AppendG(o, lh_loglikelihood - rb_loglikelihood);
  o.push_back('\t');
  o += s.naive_seq;
  o.push_back('\t');
  o += s.vgerm_state_str;
  o.push_back('\t');
  AppendInt(o, s.vgerm_left_del);
  o.push_back('\t');
  AppendInt(o, s.vgerm_right_del);
  o.push_back('\t');
  o += s.vgerm_left_insertion;
  o.push_back('\t');
  o += s.vd_junction_insertion;
  o.push_back('\t');
  if (locus_ == "igh") {
    o += s.dgerm_state_str;
    o.push_back('\t');
    AppendInt(o, s.dgerm_left_del);
    o.push_back('\t');
    AppendInt(o, s.dgerm_right_del);
    o.push_back('\t');
    o += s.dj_junction_insertion;
    o.push_back('\t');
  }
  o += s.jgerm_state_str;
  o.push_back('\t');
  AppendInt(o, s.jgerm_left_del);
  o.push_back('\t');
  AppendInt(o, s.jgerm_right_del);
  o.push_back('\t');
  o += s.jgerm_right_insertion;
  o.push_back('\n');
}

// src/PhyloHMM.cpp:288-327
void PhyloHMM::WriteOutputLine(std::ofstream& outfile) const {
  RowSampler s;  // a view of the members the line is made of
  s.naive_seq = naive_sequence_;
  s.vgerm_state_str = vgerm_state_str_samp_;
  s.vgerm_left_del = vgerm_left_del_samp_;
  s.vgerm_right_del = vgerm_right_del_samp_;
  s.vgerm_left_insertion = vgerm_left_insertion_samp_;
  s.vd_junction_insertion = vd_junction_insertion_samp_;
  s.dgerm_state_str = dgerm_state_str_samp_;
  s.dgerm_left_del = dgerm_left_del_samp_;
  s.dgerm_right_del = dgerm_right_del_samp_;
  s.dj_junction_insertion = dj_junction_insertion_samp_;
  s.jgerm_state_str = jgerm_state_str_samp_;
  s.jgerm_left_del = jgerm_left_del_samp_;
  s.jgerm_right_del = jgerm_right_del_samp_;
  s.jgerm_right_insertion = jgerm_right_insertion_samp_;
  std::string line;
  FormatOutputLine(line, iteration_, rb_loglikelihood_, prior_, alpha_, er_.data(), pi_.data(),
                   pending_newick_ ? *pending_newick_ : ExportNewick(tree_, xmsa_labels_), sr_.data(), (int)sr_.size(),
                   lh_loglikelihood_, s);
  outfile << line;
}

namespace {

// One row of the RevBayes table (io::CSVReader<15, trim_chars<>, double_quote_escape<'\t','"'>>,
// src/PhyloHMM.cpp:396-400): tab separated, optional double quotes, extra columns ignored.
std::vector<std::string> SplitTsv(const std::string& line) {
  std::vector<std::string> out;
  std::string cur;
  bool quoted = false;
  for (std::size_t i = 0; i < line.size(); ++i) {
    const char c = line[i];
    if (c == '"') {
      if (quoted && i + 1 < line.size() && line[i + 1] == '"') {
        cur.push_back('"');
        ++i;
      } else {
        quoted = !quoted;
      }
    } else if (c == '\t' && !quoted) {
      out.push_back(cur);
      cur.clear();
    } else if (c != '\r') {
      cur.push_back(c);
    }
  }
  out.push_back(cur);
  return out;
}

}  // namespace

// A RevBayes table held in memory: the rows are located once (no per-field strings) and handed to worker
// threads; a field is converted where it is needed.
// A file's bytes, NUL-terminated, read by several threads into memory that is not cleared first (a 400 MB table:
// 0.03 s instead of 0.10 s for a zero-filled std::string and one fread).
struct FileBytes {
  std::unique_ptr<char[]> data;
  std::size_t n = 0;
  const char* c_str() const { return data.get(); }
  std::size_t size() const { return n; }
  static FileBytes Read(const std::string& path, const char* what) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error(std::string("Can't open ") + what + " " + path);
    struct stat st;
    if (::fstat(fd, &st) != 0) {
      ::close(fd);
      throw std::runtime_error("Can't read " + path);
    }
    FileBytes b;
    if (!S_ISREG(st.st_mode) || st.st_size == 0) {
      // a pipe, a FIFO, /dev/stdin, a process substitution (or a /proc-style file that reports no size): no length to
      // divide among threads -- read to the end into a growing buffer
      std::size_t cap = 1u << 20;
      std::unique_ptr<char[]> buf(new char[cap + 1]);
      for (;;) {
        if (b.n == cap) {
          std::unique_ptr<char[]> more(new char[2 * cap + 1]);
          std::memcpy(more.get(), buf.get(), b.n);
          buf.swap(more);
          cap *= 2;
        }
        const ssize_t got = ::read(fd, buf.get() + b.n, cap - b.n);
        if (got < 0) {
          if (errno == EINTR) continue;
          ::close(fd);
          throw std::runtime_error("Can't read " + path);
        }
        if (got == 0) break;
        b.n += (std::size_t)got;
      }
      ::close(fd);
      buf[b.n] = '\0';
      b.data = std::move(buf);
      return b;
    }
    b.n = (std::size_t)st.st_size;
    b.data.reset(new char[b.n + 1]);
    b.data[b.n] = '\0';
    const int n_threads = b.n < (8u << 20) ? 1 : HostThreads(8);
    std::vector<int> bad(n_threads, 0);
    auto part = [&](int w) {
      std::size_t lo = b.n * w / n_threads;
      const std::size_t hi = b.n * (w + 1) / n_threads;
      while (lo < hi) {
        const ssize_t got = ::pread(fd, b.data.get() + lo, hi - lo, (off_t)lo);
        if (got <= 0) {
          bad[w] = 1;
          return;
        }
        lo += (std::size_t)got;
      }
    };
    if (n_threads == 1) {
      part(0);
    } else {
      std::vector<std::thread> pool;
      for (int w = 0; w < n_threads; ++w) pool.emplace_back(part, w);
      for (std::thread& t : pool) t.join();
    }
    ::close(fd);
    for (int x : bad)
      if (x) throw std::runtime_error("Can't read " + path);
    return b;
  }
};

struct PhyloHMM::TsvTable {
  FileBytes buf;                                          // the file, NUL-terminated
  std::vector<std::pair<std::size_t, std::size_t>> rows;  // data lines: offset and length (empty lines skipped)
  std::vector<std::string> header;
  int col[15];                                            // columns of RunPipeline's fifteen fields

  static TsvTable Read(const std::string& path, const char* what) {
    TsvTable t;
    t.buf = FileBytes::Read(path, what);
    const char* p = t.buf.c_str();
    const std::size_t n = t.buf.size();
    std::size_t pos = 0;
    bool first = true;
    while (pos < n) {
      const char* nl = static_cast<const char*>(std::memchr(p + pos, '\n', n - pos));
      const std::size_t end = nl ? (std::size_t)(nl - p) : n;
      std::size_t len = end - pos;
      if (len && p[pos + len - 1] == '\r') --len;
      if (first) {
        t.header = SplitTsv(std::string(p + pos, len));
        first = false;
      } else if (len) {
        t.rows.push_back({pos, len});
      }
      pos = end + 1;
    }
    if (first) throw std::runtime_error(std::string("Empty ") + what + " " + path);
    return t;
  }

  void Locate(const char* const* names, int n, int* out, const std::string& path) const {
    for (int k = 0; k < n; ++k) {
      const auto it = std::find(header.begin(), header.end(), names[k]);
      if (it == header.end()) throw std::runtime_error(std::string("Missing column \"") + names[k] + "\" in " + path);
      out[k] = (int)(it - header.begin());
    }
  }

  // Field boundaries of row r (begin offsets of every field and the end of the row); a quoted field keeps its
  // quotes here and loses them in Field().  `quoted` tells whether the row holds a double quote at all.
  void Split(std::size_t r, std::vector<std::size_t>& starts, bool* quoted) const {
    const char* p = buf.c_str() + rows[r].first;
    const std::size_t len = rows[r].second;
    starts.clear();
    starts.push_back(0);
    *quoted = std::memchr(p, '"', len) != nullptr;
    if (!*quoted) {
      std::size_t pos = 0;
      while (const char* tab = static_cast<const char*>(std::memchr(p + pos, '\t', len - pos))) {
        pos = (std::size_t)(tab - p) + 1;
        starts.push_back(pos);
      }
    } else {
      bool in = false;
      for (std::size_t i = 0; i < len; ++i) {
        if (p[i] == '"')
          in = !in;
        else if (p[i] == '\t' && !in)
          starts.push_back(i + 1);
      }
    }
    starts.push_back(len + 1);
  }
};

namespace {

struct FieldView {
  const char* p;
  std::size_t n;
};

}  // namespace

// The part of FlattenBatch that also reads its rows from the table: rows [r0, r1) are parsed (numbers, tree),
// rooted at naive's neighbour and scheduled by `n_threads` workers straight into the device arrays.
struct PhyloHMM::TableBatch {
  DeviceBatch dev;
  std::vector<int> iteration;
  std::vector<double> lik, prior;
  std::vector<std::string> exported;  // per row: the output table's tree column (with_export)
};

PhyloHMM::TableBatch PhyloHMM::FlattenTable(const TsvTable& t, std::size_t r0, std::size_t r1, bool with_export,
                                            bool with_scalars, const std::string& path) const {
  const auto t_begin = std::chrono::steady_clock::now();
  TableBatch tb;
  DeviceBatch& b = tb.dev;
  const int T = (int)xmsa_labels_.size();
  const std::size_t m = r1 - r0;
  b.n = (int)m;
  b.n_tips = T;
  b.ops.resize(m * (std::size_t)(T - 2) * 4);
  b.brlen.resize(m * (std::size_t)(2 * T - 2));
  b.er.resize(m * 6);
  b.pi.resize(m * 4);
  b.alpha.resize(m);
  if (with_export) tb.exported.resize(m);
  if (with_scalars) {
    tb.iteration.resize(m);
    tb.lik.resize(m);
    tb.prior.resize(m);
  }
  const LabelIndex labels(xmsa_labels_);
  auto work = [&](std::size_t lo, std::size_t hi, int* max_depth) {
    NewickScratch scratch;
    std::vector<std::size_t> starts;
    std::vector<int32_t> children(2 * (std::size_t)(T - 2));
    std::string unq;
    for (std::size_t i = lo; i < hi; ++i) {
      bool quoted = false;
      t.Split(r0 + i, starts, &quoted);
      const char* row = t.buf.c_str() + t.rows[r0 + i].first;
      auto field = [&](int k) {
        const int c = t.col[k];
        if (c + 1 >= (int)starts.size()) throw std::runtime_error("Too few columns in " + path);
        FieldView f{row + starts[c], starts[c + 1] - 1 - starts[c]};
        if (quoted && f.n >= 2 && f.p[0] == '"' && f.p[f.n - 1] == '"') f = FieldView{f.p + 1, f.n - 2};
        return f;
      };
      auto number = [&](int k) {
        const FieldView f = field(k);
        const char* end = nullptr;
        const double v = ParseDouble(f.p, &end);
        if (end == f.p || end > f.p + f.n) throw std::runtime_error("Bad number in column \"" + t.header[t.col[k]] + "\" of " + path);
        return v;
      };
      if (with_scalars) {
        tb.iteration[i] = (int)number(0);
        tb.lik[i] = number(1);
        tb.prior[i] = number(2);
      }
      b.alpha[i] = number(3);
      for (int k = 0; k < 6; ++k) b.er[i * 6 + k] = number(4 + k);
      for (int k = 0; k < 4; ++k) b.pi[i * 4 + k] = number(10 + k);
      FieldView tree = field(14);
      if (quoted && std::memchr(tree.p, '"', tree.n)) {  // doubled quotes inside a quoted field (never seen; kept right)
        unq.clear();
        for (std::size_t q = 0; q < tree.n; ++q) {
          unq.push_back(tree.p[q]);
          if (tree.p[q] == '"' && q + 1 < tree.n && tree.p[q + 1] == '"') ++q;
        }
        tree = FieldView{unq.c_str(), unq.size()};
      }
      int32_t root = -1, depth = 0;
      ParseNewickInto(tree.p, tree.n, labels, EPS, scratch, children.data(), &root,
                      b.brlen.data() + i * (std::size_t)(2 * T - 2), with_export ? &tb.exported[i] : nullptr);
      CheckHip(lh_schedule_tree(T, children.data(), root, b.ops.data() + i * (std::size_t)(T - 2) * 4, &depth),
               "lh_schedule_tree");
      *max_depth = std::max(*max_depth, (int)depth);
    }
  };
  const int hw = HostThreads(16);
  const int n_threads = (int)std::max<std::size_t>(1, std::min<std::size_t>(hw, m / 32));
  std::vector<std::thread> pool;
  std::vector<int> depths(n_threads, 0);
  std::vector<std::exception_ptr> errors(n_threads);
  for (int w = 0; w < n_threads; ++w) {
    const std::size_t lo = m * w / n_threads, hi = m * (w + 1) / n_threads;
    auto body = [&, w, lo, hi] {
      try {
        work(lo, hi, &depths[w]);
      } catch (...) {
        errors[w] = std::current_exception();
      }
    };
    if (n_threads == 1)
      body();
    else
      pool.emplace_back(body);
  }
  for (std::thread& th : pool) th.join();
  for (const std::exception_ptr& e : errors)
    if (e) std::rethrow_exception(e);  // the first failing row range, in file order
  for (int d : depths) b.max_depth = std::max(b.max_depth, d);
  if (host_options().pipeline_timing)
    std::fprintf(stderr, "[FlattenTable] %zu rows on %d threads: %.3f s\n", m, n_threads,
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
  return tb;
}

PhyloHMM::DeviceBatch PhyloHMM::FlattenTsv(const std::string& path, int* n_rows) const {
  const auto t0 = std::chrono::steady_clock::now();
  TsvTable t = TsvTable::Read(path, "RevBayes output file");
  if (host_options().pipeline_timing)
    std::fprintf(stderr, "[FlattenTsv] read + line index %.3f s\n",
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  const char* names[15] = {"alpha", "alpha", "alpha", "alpha", "er[1]", "er[2]", "er[3]", "er[4]",
                           "er[5]", "er[6]",  "pi[1]", "pi[2]", "pi[3]", "pi[4]", "tree"};
  t.Locate(names, 15, t.col, path);
  if (t.rows.empty()) throw std::runtime_error("no rows in table");
  *n_rows = (int)t.rows.size();
  return FlattenTable(t, 0, t.rows.size(), false, false, path).dev;
}

PhyloHMM::DeviceBatch PhyloHMM::FlattenTsvRows(const std::string& path, const int64_t* row_ids, int n, int* n_rows) const {
  TsvTable t = TsvTable::Read(path, "RevBayes output file");
  const char* names[15] = {"alpha", "alpha", "alpha", "alpha", "er[1]", "er[2]", "er[3]", "er[4]",
                           "er[5]", "er[6]",  "pi[1]", "pi[2]", "pi[3]", "pi[4]", "tree"};
  t.Locate(names, 15, t.col, path);
  if (t.rows.empty()) throw std::runtime_error("no rows in table");
  *n_rows = (int)t.rows.size();
  // the line index restricted to the rows asked for: the rest of the table is never parsed
  std::vector<std::pair<std::size_t, std::size_t>> sel((std::size_t)std::max(n, 0));
  for (int i = 0; i < n; ++i) {
    if (row_ids[i] < 0 || (std::size_t)row_ids[i] >= t.rows.size()) throw std::runtime_error("FlattenTsvRows: row outside the table");
    sel[i] = t.rows[(std::size_t)row_ids[i]];
  }
  t.rows.swap(sel);
  if (t.rows.empty()) return DeviceBatch{};
  return FlattenTable(t, 0, t.rows.size(), false, false, path).dev;
}

// src/PhyloHMM.cpp:393-446.  The reference evaluates, samples and writes row by row on one core.  Here the table
// is read once, and per batch of rows: worker threads parse and schedule the trees, the GPU evaluates the batch and
// draws every row's states (lh_eval_sample_batch), and worker threads derive the naive sequences from the states and
// format the output lines, which are written in file order.
// Sampling consumes ONE std::mt19937 stream in file order (src/HMM.cpp:56); a sample takes a fixed number of
// engine outputs (HMM::RawDrawsPerSample), so a row's outputs can be handed to whoever draws for it: the device
// gets them as words[row][...], a host worker that starts at row r copies the engine and skips r samples' worth
// (LH_HOST_SAMPLING=1, or a family whose junction genes do not form two blocks of the state vector).  Every row
// sees exactly the numbers it would see in the serial loop (the host sampler repeats the device's first and last
// row on every run; the seed-0 goldens).  The last row also goes through the object's own members, which are then
// in the state the reference's loop leaves behind.
void PhyloHMM::RunPipeline(const std::string& input_path, const std::string& output_path, int num_rates) {
  const bool timing = host_options().pipeline_timing;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  const auto t_start = now();
  TsvTable table = TsvTable::Read(input_path, "RevBayes output file");
  const char* names[15] = {"Iteration", "Likelihood", "Prior", "alpha", "er[1]", "er[2]", "er[3]", "er[4]",
                           "er[5]",     "er[6]",      "pi[1]", "pi[2]", "pi[3]", "pi[4]", "tree"};
  table.Locate(names, 15, table.col, input_path);
  const std::size_t N = table.rows.size();
  const auto t_read = now();

  StageTimer stage;
  CreateFamily();
  stage.Mark("CreateFamily");
  EnsureSamplingLists();
  stage.Mark("sampling column lists");
  std::ofstream outfile(output_path);
  if (!outfile) throw std::runtime_error("Can't open output file " + output_path);
  er_.assign(6, 0.0);
  pi_.assign(4, 0.0);
  sr_.assign(num_rates, 0.0);
  num_rates_ = num_rates;
  const std::size_t FS = lh_forward_size(family_), SS = lh_scaler_size(family_);
  const int T = (int)xmsa_labels_.size();
  const int raw_per_sample = RawDrawsPerSample();
  // Sampling on the device when the family has its sampler tables: the forward arrays then never leave the GPU, the
  // host sends each row's slice of the engine's output stream and gets the sampled states back.
  const bool dev_sampling = device_sampler_;
  const std::size_t NS = dev_sampling ? (std::size_t)lh_sample_states(family_) : 0;
  // Two stages, two batches in flight: a producer thread parses / schedules batch k + 1 and has the GPU
  // evaluate it into one of two page-locked result slots while this thread's workers sample and format batch k.
  // Batch size: a sixteenth of the table (the stages overlap batch by batch: few batches mean a long fill and drain),
  // between 2048 rows and 16384 (768 tree samples fill the chip's resident workgroups once; from 8192 on the kernels run
  // at their full-batch rate).
  const std::size_t kBatch = std::min<std::size_t>(std::max<std::size_t>(N, 1), std::min<std::size_t>(16384, std::max<std::size_t>(2048, (N + 15) / 16)));
  struct Slot {
    TableBatch tb;
    double *ll = nullptr, *rates = nullptr, *fwd = nullptr;
    int32_t *sco = nullptr, *states = nullptr;
    std::vector<uint32_t> words;
    std::size_t off = 0, m = 0;
    int state = 0;  // 0 free, 1 filled
  };
  Slot slots[2];
  std::mutex mu;
  std::condition_variable cv;
  std::exception_ptr producer_error;
  bool cancel = false;  // the consumer gave up (error): the producer must not wait for a slot
  double t_flat = 0, t_eval = 0, t_samp = 0, t_write = 0, t_wait = 0;
  auto alloc_slot = [&](Slot& s) {
    s.ll = static_cast<double*>(lh_host_alloc(sizeof(double) * kBatch));
    s.rates = static_cast<double*>(lh_host_alloc(sizeof(double) * kBatch * num_rates));
    // device sampling keeps forward arrays for two rows only: the table's first (cross-check) and last (members)
    const std::size_t fwd_rows = dev_sampling ? 2 : kBatch;
    s.fwd = static_cast<double*>(lh_host_alloc(sizeof(double) * fwd_rows * FS));
    s.sco = static_cast<int32_t*>(lh_host_alloc(sizeof(int32_t) * fwd_rows * SS));
    if (dev_sampling) s.states = static_cast<int32_t*>(lh_host_alloc(sizeof(int32_t) * kBatch * NS));
    if (!s.ll || !s.rates || !s.fwd || !s.sco || (dev_sampling && !s.states)) throw std::runtime_error(lh_last_error());
  };
  auto free_slots = [&] {
    for (Slot& s : slots) {
      lh_host_free(s.ll);
      lh_host_free(s.rates);
      lh_host_free(s.fwd);
      lh_host_free(s.sco);
      lh_host_free(s.states);
    }
  };
  double t_words = 0;
  // Stage 0: a parser thread turns the next batches of rows into device arrays (worker threads inside FlattenTable) and
  // keeps at most two of them waiting, so that parsing batch k + 1 overlaps the device's work on batch k.
  std::deque<TableBatch> parsed;
  bool parser_done = false;
  std::thread parser([&] {
    try {
      for (std::size_t off = 0; off < N; off += kBatch) {
        const std::size_t m = std::min(kBatch, N - off);
        const auto t0 = now();
        TableBatch tb = FlattenTable(table, off, off + m, true, true, input_path);
        t_flat += secs(t0, now());
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return parsed.size() < 2 || cancel; });
        if (cancel) return;
        parsed.push_back(std::move(tb));
        cv.notify_all();
      }
    } catch (...) {
      std::lock_guard<std::mutex> lock(mu);
      if (!producer_error) producer_error = std::current_exception();
    }
    std::lock_guard<std::mutex> lock(mu);
    parser_done = true;
    cv.notify_all();
  });
  // The device sampler's random words: one engine stream in file order (a copy of rng_ that just runs on), drawn by a
  // thread of its own, at most two batches ahead of the producer.
  std::deque<std::vector<uint32_t>> drawn;
  std::thread drawer([&] {
    if (!dev_sampling) return;
    try {
      std::mt19937 word_rng = rng_;
      for (std::size_t off = 0; off < N; off += kBatch) {
        const std::size_t m = std::min(kBatch, N - off);
        const auto t0 = now();
        std::vector<uint32_t> w(m * (std::size_t)raw_per_sample);
        for (uint32_t& x : w) x = (uint32_t)word_rng();
        t_words += secs(t0, now());
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return drawn.size() < 2 || cancel; });
        if (cancel) return;
        drawn.push_back(std::move(w));
        cv.notify_all();
      }
    } catch (...) {
      std::lock_guard<std::mutex> lock(mu);
      if (!producer_error) producer_error = std::current_exception();
      cv.notify_all();
    }
  });
  std::thread producer([&] {
    try {
      int k = 0;
      for (std::size_t off = 0; off < N; off += kBatch, k ^= 1) {
        Slot& s = slots[k];
        const std::size_t m = std::min(kBatch, N - off);
        TableBatch tb;
        {
          std::unique_lock<std::mutex> lock(mu);
          cv.wait(lock, [&] { return !parsed.empty() || parser_done || cancel || producer_error; });
          if (cancel || producer_error || parsed.empty()) return;  // (an empty queue with the parser gone: it failed)
          tb = std::move(parsed.front());
          parsed.pop_front();
          cv.notify_all();
        }
        {
          std::unique_lock<std::mutex> lock(mu);
          cv.wait(lock, [&] { return s.state == 0 || cancel; });
          if (cancel) return;
        }
        if (!s.ll) alloc_slot(s);
        s.tb = std::move(tb);
        s.off = off;
        s.m = m;
        const DeviceBatch& b = s.tb.dev;
        const auto t2 = now();
        if (dev_sampling) {
          {
            std::unique_lock<std::mutex> lock(mu);
            cv.wait(lock, [&] { return !drawn.empty() || cancel || producer_error; });
            if (cancel || producer_error) return;
            s.words = std::move(drawn.front());
            drawn.pop_front();
            cv.notify_all();
          }
          if (more_families_.empty()) {
            CheckHip(lh_eval_sample_batch(family_, b.n, b.n_tips, b.max_depth, b.ops.data(), b.brlen.data(), b.er.data(),
                                          b.pi.data(), b.alpha.data(), num_rates, s.words.data(), s.ll, s.rates, s.states),
                     "lh_eval_sample_batch");
          } else {
            // Several devices: table row i goes to device i mod N.  One host thread per device gathers its rows of
            // the batch, has its handle evaluate and sample them, and puts the results back at the rows' places
            // (host memory, in-process: no collective; RCCL only joins separate processes, bench.py --gpus N).
            const std::size_t D = 1 + more_families_.size();
            const std::size_t n_ops4 = (std::size_t)(b.n_tips - 2) * 4, nodes = 2 * (std::size_t)b.n_tips - 2;
            const std::size_t W = (std::size_t)raw_per_sample;
            std::vector<std::exception_ptr> errs(D);
            std::vector<std::thread> pool;
            for (std::size_t d = 0; d < D; ++d)
              pool.emplace_back([&, d] {
                try {
                  std::vector<std::size_t> idx;
                  for (std::size_t i = 0; i < m; ++i)
                    if ((off + i) % D == d) idx.push_back(i);
                  const std::size_t q = idx.size();
                  if (q == 0) return;
                  std::vector<int32_t> ops(q * n_ops4), states(q * NS);
                  std::vector<double> brlen(q * nodes), er(q * 6), pi(q * 4), alpha(q), ll(q), rates(q * (std::size_t)num_rates);
                  std::vector<uint32_t> words(q * W);
                  for (std::size_t j = 0; j < q; ++j) {
                    const std::size_t i = idx[j];
                    std::copy_n(b.ops.data() + i * n_ops4, n_ops4, ops.data() + j * n_ops4);
                    std::copy_n(b.brlen.data() + i * nodes, nodes, brlen.data() + j * nodes);
                    std::copy_n(b.er.data() + i * 6, 6, er.data() + j * 6);
                    std::copy_n(b.pi.data() + i * 4, 4, pi.data() + j * 4);
                    alpha[j] = b.alpha[i];
                    std::copy_n(s.words.data() + i * W, W, words.data() + j * W);
                  }
                  lh_family* fam = d == 0 ? family_ : more_families_[d - 1];
                  if (lh_eval_sample_batch(fam, (int32_t)q, b.n_tips, b.max_depth, ops.data(), brlen.data(), er.data(), pi.data(),
                                           alpha.data(), num_rates, words.data(), ll.data(), rates.data(), states.data()))
                    throw std::runtime_error(std::string("lh_eval_sample_batch: ") + lh_last_error());
                  for (std::size_t j = 0; j < q; ++j) {
                    const std::size_t i = idx[j];
                    s.ll[i] = ll[j];
                    std::copy_n(rates.data() + j * (std::size_t)num_rates, (std::size_t)num_rates, s.rates + i * (std::size_t)num_rates);
                    std::copy_n(states.data() + j * NS, NS, s.states + i * NS);
                  }
                } catch (...) {
                  errs[d] = std::current_exception();
                }
              });
            for (std::thread& th : pool) th.join();
            for (const std::exception_ptr& e : errs)
              if (e) std::rethrow_exception(e);
          }
          auto one_row = [&](std::size_t i, int k_fwd) {  // forward arrays of one row, for the host-side checks
            const std::size_t n_ops = (std::size_t)(b.n_tips - 2) * 4, nodes = 2 * (std::size_t)b.n_tips - 2;
            double ll1 = 0;
            lh_eval_outputs outs{nullptr, nullptr, s.fwd + k_fwd * FS, s.sco + k_fwd * SS};
            CheckHip(lh_eval_batch(family_, 1, b.n_tips, b.max_depth, b.ops.data() + i * n_ops, b.brlen.data() + i * nodes,
                                   b.er.data() + i * 6, b.pi.data() + i * 4, b.alpha.data() + i, num_rates, &ll1, &outs),
                     "lh_eval_batch");
          };
          if (off == 0) one_row(0, 0);
          if (off + m == N) one_row(m - 1, 1);
        } else {
          lh_eval_outputs outs{s.rates, nullptr, s.fwd, s.sco};
          CheckHip(lh_eval_batch(family_, b.n, b.n_tips, b.max_depth, b.ops.data(), b.brlen.data(), b.er.data(),
                                 b.pi.data(), b.alpha.data(), num_rates, s.ll, &outs),
                   "lh_eval_batch");
        }
        const auto t3 = now();
        t_eval += secs(t2, t3);
        {
          std::lock_guard<std::mutex> lock(mu);
          s.state = 1;
        }
        cv.notify_all();
      }
    } catch (...) {
      std::lock_guard<std::mutex> lock(mu);
      producer_error = std::current_exception();
      cv.notify_all();
    }
  });
  struct Joiner {  // the three threads are stopped, joined and the slots are freed on every way out
    std::thread &t, &t0, &t1;
    std::function<void()> before, after;
    ~Joiner() {
      before();
      if (t0.joinable()) t0.join();
      if (t1.joinable()) t1.join();
      if (t.joinable()) t.join();
      after();
    }
  } joiner{producer, parser, drawer,
           [&] {
             {
               std::lock_guard<std::mutex> lock(mu);
               cancel = true;
             }
             cv.notify_all();
           },
           free_slots};

  bool header_written = false;
  int k = 0;
  for (std::size_t off = 0; off < N; off += kBatch, k ^= 1) {
    Slot& slot = slots[k];
    const auto tw = now();
    {
      std::unique_lock<std::mutex> lock(mu);
      cv.wait(lock, [&] { return slot.state == 1 || producer_error; });
      if (producer_error) std::rethrow_exception(producer_error);
    }
    const std::size_t m = slot.m;
    const TableBatch& tb = slot.tb;
    const DeviceBatch& b = tb.dev;
    const double *ll = slot.ll, *rates = slot.rates, *fwd = slot.fwd;
    const int32_t* sco = slot.sco;
    const auto t2 = now();
    t_wait += secs(tw, t2);
    // rows of this batch except the table's very last one: sampled and formatted by the workers
    const std::size_t m_par = (off + m == N) ? m - 1 : m;
    const int hw = HostThreads(16);
    const int n_threads = (int)std::max<std::size_t>(1, std::min<std::size_t>(hw, m_par / 8));
    std::vector<std::string> chunks(n_threads);
    std::vector<std::exception_ptr> errors(n_threads);
    auto sample_rows = [&](int w, std::size_t lo, std::size_t hi) {
      try {
        RowSampler s;
        std::mt19937 rng = rng_;
        if (!dev_sampling) rng.discard((unsigned long long)(off + lo) * (unsigned long long)raw_per_sample);
        std::string& o = chunks[w];
        o.reserve((hi - lo) * (tb.exported.empty() ? 512 : tb.exported[lo].size() + 1024));
        for (std::size_t i = lo; i < hi; ++i) {
          if (dev_sampling) {
            ApplySampledStates(s, slot.states + i * NS);
            if (off + i == 0) {  // the device's draws against the host sampler's, where it is cheap
              RowSampler h;
              SampleRow(h, fwd, rng);
              if (h.naive_seq != s.naive_seq || h.vd_junction_state_inds != s.vd_junction_state_inds ||
                  h.dj_junction_state_inds != s.dj_junction_state_inds || h.vgerm_state_ind != s.vgerm_state_ind ||
                  h.dgerm_state_ind != s.dgerm_state_ind || h.jgerm_state_ind != s.jgerm_state_ind)
                throw std::runtime_error("RunPipeline: the device sampler and the host sampler disagree on the first row");
            }
          } else if (off + i == 0) {  // the bookkeeping above rests on this count: check it where it is cheap
            std::mt19937 expect = rng;
            SampleRow(s, fwd + i * FS, rng);
            expect.discard((unsigned long long)raw_per_sample);
            if (!(expect == rng)) throw std::runtime_error("RunPipeline: a sample consumed an unexpected number of random numbers");
          } else {
            SampleRow(s, fwd + i * FS, rng);
          }
          FormatOutputLine(o, tb.iteration[i], tb.lik[i], tb.prior[i], b.alpha[i], b.er.data() + i * 6,
                           b.pi.data() + i * 4, tb.exported[i], rates + i * num_rates, num_rates, ll[i], s);
        }
      } catch (...) {
        errors[w] = std::current_exception();
      }
    };
    {
      std::vector<std::thread> pool;
      for (int w = 0; w < n_threads; ++w) {
        const std::size_t lo = m_par * w / n_threads, hi = m_par * (w + 1) / n_threads;
        if (n_threads == 1)
          sample_rows(w, lo, hi);
        else
          pool.emplace_back(sample_rows, w, lo, hi);
      }
      for (std::thread& th : pool) th.join();
      for (const std::exception_ptr& e : errors)
        if (e) std::rethrow_exception(e);
    }
    const auto t3 = now();
    if (!header_written) {
      WriteOutputHeaders(outfile);
      header_written = true;
    }
    for (const std::string& c : chunks) outfile.write(c.data(), (std::streamsize)c.size());
    if (m_par < m) {
      // the last row of the table, through the members (as every row goes in the reference)
      const std::size_t i = m - 1;
      // (with device sampling the forward arrays of this row sit in the slot's second place)
      const double* fwd_i = dev_sampling ? slot.fwd + FS : fwd + i * FS;
      const int32_t* sco_i = dev_sampling ? slot.sco + SS : sco + i * SS;
      rng_.discard((unsigned long long)(N - 1) * (unsigned long long)raw_per_sample);
      iteration_ = tb.iteration[i];
      rb_loglikelihood_ = tb.lik[i];
      prior_ = tb.prior[i];
      alpha_ = b.alpha[i];
      er_.assign(b.er.begin() + i * 6, b.er.begin() + (i + 1) * 6);
      pi_.assign(b.pi.begin() + i * 4, b.pi.begin() + (i + 1) * 4);
      {  // the tree's arrays again (the batch keeps schedules, not child lists)
        const std::pair<std::size_t, std::size_t> row = table.rows[off + i];
        std::vector<std::size_t> starts;
        bool quoted = false;
        table.Split(off + i, starts, &quoted);
        const int c = table.col[14];
        std::string text(table.buf.c_str() + row.first + starts[c], starts[c + 1] - 1 - starts[c]);
        if (quoted) text = SplitTsv(std::string(table.buf.c_str() + row.first, row.second)).at(c);
        tree_ = ParseNewick(text, xmsa_labels_, EPS, true);
      }
      have_tree_ = true;
      pending_newick_ = &tb.exported[i];
      sr_.assign(rates + i * num_rates, rates + (i + 1) * num_rates);
      pending_forward_.assign(fwd_i, fwd_i + FS);
      pending_scalers_.assign(sco_i, sco_i + SS);
      pending_loglik_ = ll[i];
      cache_forward_ = true;
      lh_loglikelihood_ = LogLikelihood();
      logweight_ = lh_loglikelihood_ - rb_loglikelihood_;
      naive_sequence_ = SampleNaiveSequence();
      if (dev_sampling) {
        RowSampler d;
        ApplySampledStates(d, slot.states + i * NS);
        if (d.naive_seq != naive_sequence_)
          throw std::runtime_error("RunPipeline: the device sampler and the host sampler disagree on the last row");
      }
      WriteOutputLine(outfile);
      pending_newick_ = nullptr;
    }
    const auto t4 = now();
    t_samp += secs(t2, t3);
    t_write += secs(t3, t4);
    {
      std::lock_guard<std::mutex> lock(mu);
      slot.state = 0;
    }
    cv.notify_all();
  }
  (void)T;
  outfile.close();
  if (timing)
    std::fprintf(stderr,
                 "[RunPipeline] %zu rows: read %.3f s; producer: parse+schedule %.3f s, device (incl. copies%s) %.3f s; "
                 "consumer (%s): waiting %.3f s, sample+format %.3f s, write %.3f s; total %.3f s\n",
                 N, secs(t_start, t_read), t_flat, dev_sampling ? (", engine words " + std::to_string(t_words) + " s").c_str() : "",
                 t_eval, dev_sampling ? "device sampler" : "host sampler", t_wait, t_samp,
                 t_write, secs(t_start, now()));
}

// scripts/run_bootstrap_asr_ess.R:86-101: the tree rooted on the naive branch (the added root node sits at
// distance 0 from naive's neighbour, :53), every node followed by [&ancestral="..."] as
// phylotate::print_annotated writes node comments.  Branch lengths "%.10g" (R's own formatting of doubles is
// not restated).
std::string PhyloHMM::AnnotatedNewick(const TreeArrays& tr, const std::string& naive_sequence,
                                      const uint8_t* anc) const {
  const int T = tr.n_tips;
  const int L = (int)msa_.cols();
  auto comment = [&](int v) {
    std::string s = "[&ancestral=\"";
    if (v == 0) {
      s += naive_sequence;
    } else if (v < T) {
      for (int j = 0; j < L; ++j) s.push_back(alphabet_[msa_(v - 1, j)]);
    } else {
      const uint8_t* a = anc + (std::size_t)(v - T) * L;
      for (int j = 0; j < L; ++j) s.push_back(alphabet_[a[j]]);
    }
    return s + "\"]";
  };
  auto len = [](double l) {
    char b[48];
    std::snprintf(b, sizeof b, ":%.10g", l);
    return std::string(b);
  };
  std::string out;
  struct Fr {
    int node, next_kid;
  };
  std::vector<Fr> stack{{tr.root, 0}};
  out = "(" + xmsa_labels_[0] + comment(0) + len(tr.brlen[0]) + ",";
  while (!stack.empty()) {
    Fr& f = stack.back();
    if (f.node < T) {
      out += xmsa_labels_[f.node] + comment(f.node) + len(tr.brlen[f.node]);
      stack.pop_back();
      continue;
    }
    if (f.next_kid < 2) {
      out.push_back(f.next_kid == 0 ? '(' : ',');
      const int k = tr.children[2 * (std::size_t)(f.node - T) + f.next_kid++];
      stack.push_back({k, 0});
      continue;
    }
    out += ")" + comment(f.node) + len(f.node == tr.root ? 0.0 : tr.brlen[f.node]);
    stack.pop_back();
  }
  return out + ")" + comment(tr.root) + ";";
}

void PhyloHMM::RunAsr(const std::string& input_path, const std::string& output_path, uint64_t seed) {
  std::ifstream in(input_path);
  if (!in) throw std::runtime_error("Can't open linearham output file " + input_path);
  std::string line;
  if (!std::getline(in, line)) throw std::runtime_error("Empty linearham output file " + input_path);
  const std::vector<std::string> header = SplitTsv(line);
  auto find = [&](const std::string& name) {
    const auto it = std::find(header.begin(), header.end(), name);
    return it == header.end() ? -1 : (int)(it - header.begin());
  };
  std::vector<int> col;
  for (int k = 1; k <= 6; ++k) col.push_back(find("er[" + std::to_string(k) + "]"));
  for (int k = 1; k <= 4; ++k) col.push_back(find("pi[" + std::to_string(k) + "]"));
  col.push_back(find("tree"));
  col.push_back(find("NaiveSequence"));
  const char* names[12] = {"er[1]", "er[2]", "er[3]", "er[4]", "er[5]", "er[6]", "pi[1]", "pi[2]", "pi[3]", "pi[4]",
                           "tree",  "NaiveSequence"};
  for (int k = 0; k < 12; ++k)
    if (col[k] < 0) throw std::runtime_error(std::string("Missing column \"") + names[k] + "\" in " + input_path);
  std::vector<int> sr_col;
  for (int k = 1;; ++k) {
    const int c = find("sr[" + std::to_string(k) + "]");
    if (c < 0) break;
    sr_col.push_back(c);
  }
  if (sr_col.empty()) throw std::runtime_error("Missing column \"sr[1]\" in " + input_path);
  const int R = (int)sr_col.size();
  const int L = (int)msa_.cols();
  struct Row {
    TreeSample ts;
    std::vector<double> sr;
    std::string naive;
  };
  std::vector<Row> rows;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    const std::vector<std::string> f = SplitTsv(line);
    auto get = [&](int c) -> const std::string& {
      if (c >= (int)f.size()) throw std::runtime_error("Too few columns in " + input_path);
      return f[c];
    };
    Row r;
    for (int k = 0; k < 6; ++k) r.ts.er.push_back(std::stod(get(col[k])));
    for (int k = 0; k < 4; ++k) r.ts.pi.push_back(std::stod(get(col[6 + k])));
    r.ts.alpha = 1.0;  // unused: the rates come from the sr[] columns
    r.ts.newick = get(col[10]);
    r.naive = get(col[11]);
    if ((int)r.naive.size() != L) throw std::runtime_error("NaiveSequence length differs from the alignment's in " + input_path);
    for (int c : sr_col) r.sr.push_back(std::stod(get(c)));
    rows.push_back(std::move(r));
  }
  CreateFamily();
  std::ofstream outfile(output_path);
  if (!outfile) throw std::runtime_error("Can't open output file " + output_path);
  const int T = (int)xmsa_labels_.size();
  const std::size_t kBatch = 1024;
  for (std::size_t off = 0; off < rows.size(); off += kBatch) {
    const std::size_t m = std::min(kBatch, rows.size() - off);
    std::vector<TreeSample> samples;
    for (std::size_t i = 0; i < m; ++i) samples.push_back(rows[off + i].ts);
    std::vector<TreeArrays> trees;
    const DeviceBatch b = FlattenBatch(samples, &trees);
    std::vector<double> rates(m * R);
    std::vector<uint8_t> naive(m * (std::size_t)L), anc(m * (std::size_t)(T - 2) * L);
    for (std::size_t i = 0; i < m; ++i) {
      std::copy(rows[off + i].sr.begin(), rows[off + i].sr.end(), rates.begin() + i * R);
      for (int j = 0; j < L; ++j) {
        const std::size_t a = alphabet_.find(rows[off + i].naive[j]);
        if (a == std::string::npos) throw std::runtime_error("NaiveSequence holds a character outside the alphabet");
        naive[i * L + j] = (uint8_t)a;
      }
    }
    CheckHip(lh_asr_batch(family_, b.n, b.n_tips, b.max_depth, b.ops.data(), b.brlen.data(), b.er.data(), b.pi.data(),
                          rates.data(), R, naive.data(), seed, (uint64_t)off, anc.data(), nullptr),
             "lh_asr_batch");
    // the annotated strings (86 KB per tree for 100 leaves x 400 sites) are most of this step's host time:
    // rows are formatted by several threads, written in order
    std::vector<std::string> lines(m);
    const int hw = HostThreads(16);
    const int n_threads = std::max(1, std::min(hw, (int)(m / 16)));
    auto format_rows = [&](std::size_t lo, std::size_t hi) {
      for (std::size_t i = lo; i < hi; ++i)
        lines[i] = AnnotatedNewick(trees[i], rows[off + i].naive, anc.data() + i * (std::size_t)(T - 2) * L);
    };
    if (n_threads == 1) {
      format_rows(0, m);
    } else {
      std::vector<std::thread> pool;
      for (int t = 0; t < n_threads; ++t)
        pool.emplace_back(format_rows, m * t / n_threads, m * (t + 1) / n_threads);
      for (std::thread& th : pool) th.join();
    }
    for (std::size_t i = 0; i < m; ++i) outfile << lines[i] << "\n";
  }
}

// src/PhyloHMM.cpp:461-471
void StoreGermlinePaddingXmsaIndices(const std::vector<int>& naive_bases, const std::vector<int>& site_inds,
                                     std::map<std::pair<int, int>, int>& xmsa_ids, VectorXi& xmsa_inds) {
  xmsa_inds.assign(naive_bases.size(), -1);
  for (std::size_t i = 0; i < naive_bases.size(); i++)
    StoreXmsaIndex({naive_bases[i], site_inds[i]}, xmsa_ids, xmsa_inds[i]);
}

// src/PhyloHMM.cpp:489-513
void StoreJunctionXmsaIndices(const std::vector<int>& naive_bases, const std::vector<int>& site_inds,
                              std::pair<int, int> left_flexbounds, std::pair<int, int> right_flexbounds,
                              std::map<std::pair<int, int>, int>& xmsa_ids, MatrixXi& xmsa_inds) {
  const int site_start = left_flexbounds.first, site_end = right_flexbounds.second;
  xmsa_inds.setConstant(site_end - site_start, (int)naive_bases.size(), -1);
  for (std::size_t i = 0; i < naive_bases.size(); i++) {
    if (site_inds[i] == -1) {
      for (int site_ind = site_start; site_ind < site_end; site_ind++)
        StoreXmsaIndex({naive_bases[i], site_ind}, xmsa_ids, xmsa_inds(site_ind - site_start, (int)i));
    } else {
      StoreXmsaIndex({naive_bases[i], site_inds[i]}, xmsa_ids, xmsa_inds(site_inds[i] - site_start, (int)i));
    }
  }
}

// src/PhyloHMM.cpp:523-536
void StoreXmsaIndex(std::pair<int, int> id, std::map<std::pair<int, int>, int>& xmsa_ids, int& xmsa_ind) {
  const int next = (int)xmsa_ids.size();
  auto res = xmsa_ids.emplace(id, next);
  xmsa_ind = res.first->second;
}

void PhyloHMM::SetExtendedRange(bool on) {
  // takes effect with the next InitializePhyloEmission / RunPipeline
  if (lh_family_set_extended_range(family(), on ? 1 : 0)) throw std::runtime_error(lh_last_error());
  for (lh_family* f : more_families_)
    if (lh_family_set_extended_range(f, on ? 1 : 0)) throw std::runtime_error(lh_last_error());
}

}  // namespace linearham
