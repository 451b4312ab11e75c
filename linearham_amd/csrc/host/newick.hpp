// Newick ingest/export for the clonal tree (replaces pll_utree_parse_newick[_string],
// pt::pll::set_missing_branch_length and pll_utree_export_newick [libpll/libptpll, third party];
// call sites src/PhyloHMM.cpp:299-300,354-355,419-422).
#ifndef LINEARHAM_NEWICK_
#define LINEARHAM_NEWICK_

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace linearham {

/// strtod restricted to what it is called on here (decimal numbers of a RevBayes table), bit for bit the same
/// result; *end = first character not consumed (== s if there is no number).
double ParseDouble(const char* s, const char** end);

/// Appends printf("%f", v) (libpll's branch-length format), bit for bit.
void AppendFixed6(std::string& out, double v);

/// Tip labels of a family -> tip numbers (0 = naive), looked up without building strings.
class LabelIndex {
  std::vector<std::string> labels_;
  std::vector<int> slots_;

 public:
  explicit LabelIndex(const std::vector<std::string>& labels);
  int size() const { return (int)labels_.size(); }
  const std::string& label(int i) const { return labels_[i]; }
  int Find(const char* s, std::size_t n) const;  // -1: unknown
};

/// Reusable working arrays of ParseNewickInto (one per thread).
struct NewickScratch {
  struct Frame {
    int node, from;
  };
  std::vector<int> parent, nk, kid, tip, newid;
  std::vector<double> len;
  std::vector<uint32_t> lab_off, lab_len;
  std::vector<char> seen;
  std::vector<Frame> stack;
};

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

/// The same, single pass over `text[0..n)` (which must be followed by a NUL or any non-numeric character), results
/// written straight into caller arrays: children [(T-2)*2], *root, brlen [2T-2]; `exported` (optional) receives the
/// output table's tree column.  No allocation once the scratch arrays have grown to the family's size.
void ParseNewickInto(const char* text, std::size_t n, const LabelIndex& labels, double eps, NewickScratch& scratch,
                     int32_t* children, int32_t* root, double* brlen, std::string* exported);

/// The tree column of the output table (pll_utree_export_newick at src/PhyloHMM.cpp:299-300): the input's
/// own nesting and order, "%f" branch lengths, missing/zero lengths replaced, comments gone.
std::string ExportNewick(const TreeArrays& tree, const std::vector<std::string>& labels);

}  // namespace linearham

#endif
