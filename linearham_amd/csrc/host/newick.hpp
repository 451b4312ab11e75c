// Newick ingest/export for the clonal tree (replaces pll_utree_parse_newick[_string],
// pt::pll::set_missing_branch_length and pll_utree_export_newick [libpll/libptpll, third party];
// call sites src/PhyloHMM.cpp:299-300,354-355,419-422).
#ifndef LINEARHAM_NEWICK_
#define LINEARHAM_NEWICK_

#include <string>
#include <vector>

namespace linearham {

/// Unrooted binary tree in the C ABI's rooted-at-naive form (include/linearham_amd.h):
/// tips 0..T-1 follow `labels` (0 = "naive"), inner nodes T..2T-3, `root` is naive's neighbour.
struct TreeArrays {
  int n_tips = 0;
  int root = -1;
  std::vector<int> children;   // [(T-2)*2]
  std::vector<double> brlen;   // [2T-2], branch above each node (root entry unused = 0)
  std::string as_parsed;       // the tree as libpll would re-export it (input order; see ExportNewick)
};

/// Strips "[&index=N]" (and any other bracket comment), parses the unrooted tree (trifurcating top
/// level, or a bifurcating one whose two root branches are merged), replaces missing/zero branch
/// lengths by `eps`, and maps tip labels onto `labels`.  Throws std::runtime_error on malformed
/// input or label mismatch (the reference does not check the libpll return value, :421).
/// `with_export` also fills TreeArrays::as_parsed (the output table's tree column).
TreeArrays ParseNewick(const std::string& text, const std::vector<std::string>& labels, double eps,
                       bool with_export = false);

/// The tree column of the output table (pll_utree_export_newick at src/PhyloHMM.cpp:299-300): the input's
/// own nesting and order, "%f" branch lengths, missing/zero lengths replaced, comments gone.
std::string ExportNewick(const TreeArrays& tree, const std::vector<std::string>& labels);

}  // namespace linearham

#endif
