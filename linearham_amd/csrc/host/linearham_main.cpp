// `linearham` command line (same sub-commands and flag names as src/linearham.cpp:268-455 of the
// reference, without TCLAP): --compute-logl | --sample | --pipeline; plus --asr, the per-tree body of
// scripts/run_bootstrap_asr_ess.R:48-104 on a --pipeline output table.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "PhyloHMM.hpp"

namespace {

struct Args {
  std::map<std::string, std::vector<std::string>> v;
  const std::string& one(const std::string& k) const {
    auto it = v.find(k);
    if (it == v.end() || it->second.empty()) throw std::invalid_argument("Required argument missing: " + k);
    if (it->second.size() > 1) throw std::invalid_argument("Argument already set! for arg --" + k);
    return it->second[0];
  }
  std::string opt(const std::string& k, const std::string& dflt) const {
    auto it = v.find(k);
    return (it == v.end() || it->second.empty()) ? dflt : it->second.back();
  }
  std::vector<double> multi(const std::string& k) const {
    auto it = v.find(k);
    if (it == v.end() || it->second.empty()) throw std::invalid_argument("Required argument missing: " + k);
    std::vector<double> out;
    for (const auto& s : it->second) out.push_back(std::stod(s));
    return out;
  }
};

Args Parse(int argc, char** argv, int first) {
  Args a;
  for (int i = first; i < argc; ++i) {
    std::string k = argv[i];
    if (k.rfind("--", 0) != 0) throw std::invalid_argument("Couldn't find match for argument " + k);
    k = k.substr(2);
    if (i + 1 >= argc) throw std::invalid_argument("Missing a value for this argument! --" + k);
    a.v[k].push_back(argv[++i]);
  }
  return a;
}

}  // namespace

int main(int argc, char** argv) {
  try {
    if (argc < 2 || std::string(argv[1]) == "-h" || std::string(argv[1]) == "--help") {
      std::cout << "A Phylo-HMM implementation for B cell receptor sequence analysis.\n"
                   "USAGE: linearham {--compute-logl|--sample|--pipeline|--asr} --yaml-path <string> --cluster-ind <int> "
                   "--hmm-param-dir <string> [--seed <int>] [--num-rates <int>] [--extended-range <0|1>] [--devices <a,b,...>] ...\n";
      return argc < 2 ? EXIT_FAILURE : EXIT_SUCCESS;
    }
    const auto t_main = std::chrono::steady_clock::now();
    const bool timing = linearham::host_options().pipeline_timing;
    auto since_start = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count(); };
    const std::string subcmd = argv[1];
    const Args a = Parse(argc, argv, 2);
    if (subcmd != "--compute-logl" && subcmd != "--sample" && subcmd != "--pipeline" && subcmd != "--asr")
      throw std::invalid_argument("'" + subcmd + "' is not a valid subcommand.");
    const std::string yaml_path = a.one("yaml-path");
    const int cluster_ind = std::stoi(a.one("cluster-ind"));
    const std::string hmm_param_dir = a.one("hmm-param-dir");
    const int seed = std::stoi(a.opt("seed", "0"));
    const int num_rates = std::stoi(a.opt("num-rates", "1"));
    // not in the reference: --devices a,b,... -- the GPUs --pipeline deals the table's rows to (row i -> device i mod N);
    // every other subcommand evaluates on the first one
    std::vector<int> device_list;
    {
      const std::string devs = a.opt("devices", "");
      std::size_t pos = 0;
      while (!devs.empty() && pos <= devs.size()) {
        const std::size_t comma = std::min(devs.find(',', pos), devs.size());
        device_list.push_back(std::stoi(devs.substr(pos, comma - pos)));
        pos = comma + 1;
      }
      if (device_list.size() > 1 && subcmd != "--pipeline")
        std::fprintf(stderr, "linearham: %s evaluates on one device; of --devices only device %d is used\n", subcmd.c_str(),
                     device_list[0]);
    }
    // the HIP runtime and the context of the device the run evaluates on come up on a side thread while the parameter
    // files are read
    const int first_device = device_list.empty() ? -1 : device_list[0];
    std::thread warmup([first_device] {
      if (first_device >= 0 && first_device < lh_device_count()) (void)lh_set_device(first_device);
      (void)lh_warmup();
    });
    struct Join {
      std::thread& t;
      ~Join() {
        if (t.joinable()) t.join();
      }
    } join_warmup{warmup};
    linearham::PhyloHMMPtr phylo_hmm_ptr =
        std::make_shared<linearham::PhyloHMM>(yaml_path, cluster_ind, hmm_param_dir, seed);
    warmup.join();
    if (timing) std::fprintf(stderr, "[main] family object + HIP context ready at %.3f s\n", since_start());
    if (!device_list.empty()) {
      if (subcmd != "--pipeline") device_list.resize(1);
      phylo_hmm_ptr->SetDevices(device_list);
    }
    // not in the reference: finite log-likelihoods where its scaling over/underflows (include/linearham_amd.h)
    if (std::stoi(a.opt("extended-range", "0")) != 0) phylo_hmm_ptr->SetExtendedRange(true);
    if (subcmd == "--pipeline") {
      phylo_hmm_ptr->RunPipeline(a.one("input-path"), a.one("output-path"), num_rates);
      if (timing) std::fprintf(stderr, "[main] RunPipeline returned at %.3f s\n", since_start());
      phylo_hmm_ptr.reset();
      if (timing) std::fprintf(stderr, "[main] family released at %.3f s\n", since_start());
      return EXIT_SUCCESS;
    }
    if (subcmd == "--asr") {
      phylo_hmm_ptr->RunAsr(a.one("input-path"), a.one("output-path"), (uint64_t)std::stoll(a.opt("seed", "0")));
      return EXIT_SUCCESS;
    }
    phylo_hmm_ptr->InitializePhyloParameters(a.one("newick-path"), a.multi("er"), a.multi("pi"),
                                             std::stod(a.opt("alpha", "1.0")), num_rates);
    phylo_hmm_ptr->InitializePhyloEmission();
    if (subcmd == "--compute-logl") {
      std::cout << phylo_hmm_ptr->LogLikelihood() << std::endl;
    } else {
      const int N = std::stoi(a.opt("N", "1"));
      for (int i = 0; i < N; i++) std::cout << phylo_hmm_ptr->SampleNaiveSequence() << std::endl;
    }
    return EXIT_SUCCESS;
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
  }
  return EXIT_FAILURE;
}
