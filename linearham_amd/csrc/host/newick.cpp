#include "newick.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>

namespace linearham {

namespace {

struct PNode {
  std::string label;
  double len = -1.0;  // < 0: missing
  std::vector<int> kids;
};

struct NewickParser {
  const std::string& t;
  std::size_t pos = 0;
  std::vector<PNode> nodes;
  explicit NewickParser(const std::string& text) : t(text) {}

  void ws() {
    while (pos < t.size() && (t[pos] == ' ' || t[pos] == '\t' || t[pos] == '\n' || t[pos] == '\r')) ++pos;
  }
  [[noreturn]] void fail(const std::string& m) const {
    throw std::runtime_error("newick: " + m + " at offset " + std::to_string(pos));
  }
  void parse_len(PNode& n) {
    ws();
    if (pos < t.size() && t[pos] == ':') {
      ++pos;
      ws();
      char* end = nullptr;
      const double v = std::strtod(t.c_str() + pos, &end);
      if (end == t.c_str() + pos) fail("bad branch length");
      pos = end - t.c_str();
      n.len = v;
    }
  }
  std::string parse_label() {
    ws();
    std::string s;
    if (pos < t.size() && (t[pos] == '\'' || t[pos] == '"')) {
      const char q = t[pos++];
      while (pos < t.size() && t[pos] != q) s.push_back(t[pos++]);
      if (pos >= t.size()) fail("unterminated quoted label");
      ++pos;
      return s;
    }
    while (pos < t.size() && t[pos] != ',' && t[pos] != '(' && t[pos] != ')' && t[pos] != ':' &&
           t[pos] != ';' && t[pos] != ' ' && t[pos] != '\t' && t[pos] != '\n')
      s.push_back(t[pos++]);
    return s;
  }
  int parse_node(int depth) {
    if (depth > 100000) fail("tree too deep");
    ws();
    if (pos >= t.size()) fail("unexpected end");
    const int id = (int)nodes.size();
    nodes.emplace_back();
    if (t[pos] == '(') {
      ++pos;
      while (true) {
        const int k = parse_node(depth + 1);
        nodes[id].kids.push_back(k);
        ws();
        if (pos >= t.size()) fail("unexpected end");
        if (t[pos] == ',') {
          ++pos;
          continue;
        }
        if (t[pos] != ')') fail("expected ',' or ')'");
        ++pos;
        break;
      }
      nodes[id].label = parse_label();  // inner label: kept for the export only
    } else {
      nodes[id].label = parse_label();
      if (nodes[id].label.empty()) fail("empty tip label");
    }
    parse_len(nodes[id]);
    return id;
  }
};

std::string StripComments(const std::string& s) {
  std::string o;
  int depth = 0;
  for (char c : s) {
    if (c == '[') {
      ++depth;
    } else if (c == ']') {
      if (depth > 0) --depth;
    } else if (depth == 0) {
      o.push_back(c);
    }
  }
  return o;
}

// pll_utree_export_newick(GetVirtualRoot(tree), NULL) [libpll-2, third party; src/PhyloHMM.cpp:299-300] on a
// tree that pll_utree_parse_newick_string built: libpll's parser hangs the top level's three subtrees on
// the virtual root's ring in input order and every inner node's two subtrees likewise, and the exporter
// walks the rings in that order, so the output is the input's own nesting and order with tips printed as
// "label:%f", inner nodes as "(a,b)label:%f" and the top level as "(a,b,c)label;" (its length dropped),
// missing and zero lengths already replaced (pt::pll::set_missing_branch_length).
std::string ExportAsParsed(const std::vector<PNode>& N, int top, double eps) {
  std::string out;
  out.reserve(N.size() * 24);
  char b[64];
  struct Fr {
    int node;
    std::size_t next_kid;
  };
  std::vector<Fr> stack{{top, 0}};
  while (!stack.empty()) {
    Fr& f = stack.back();
    const PNode& n = N[f.node];
    if (f.next_kid < n.kids.size()) {
      out.push_back(f.next_kid == 0 ? '(' : ',');
      const int k = n.kids[f.next_kid++];
      stack.push_back({k, 0});
      continue;
    }
    if (!n.kids.empty()) out.push_back(')');
    out += n.label;
    if (f.node != top) {
      std::snprintf(b, sizeof b, ":%f", (n.len < 0.0 || n.len == 0.0) ? eps : n.len);
      out += b;
    }
    stack.pop_back();
  }
  out.push_back(';');
  return out;
}

}  // namespace

TreeArrays ParseNewick(const std::string& text, const std::vector<std::string>& labels, double eps,
                       bool with_export) {
  const std::string clean = StripComments(text);
  NewickParser p(clean);
  const int top = p.parse_node(0);
  const std::vector<PNode>& N = p.nodes;
  const int T = (int)labels.size();
  if (T < 3) throw std::runtime_error("newick: need at least 3 tips (naive + 2 sequences)");
  std::map<std::string, int> lab2id;
  for (int i = 0; i < T; ++i) lab2id[labels[i]] = i;

  // undirected adjacency over parser node ids (top-level node handled below)
  std::vector<std::vector<std::pair<int, double>>> adj(N.size());
  auto fix = [eps](double l) { return (l < 0.0 || l == 0.0) ? eps : l; };
  auto link = [&](int a, int b, double l) {
    adj[a].push_back({b, l});
    adj[b].push_back({a, l});
  };
  int n_tip = 0;
  for (std::size_t v = 0; v < N.size(); ++v) {
    if (N[v].kids.empty()) {
      ++n_tip;
      if (!lab2id.count(N[v].label)) throw std::runtime_error("newick: unknown tip label \"" + N[v].label + "\"");
    } else if ((int)v != top) {
      if (N[v].kids.size() != 2) throw std::runtime_error("newick: inner nodes must be binary");
    }
    if ((int)v != top)
      for (int k : N[v].kids) link((int)v, k, fix(N[k].len));
  }
  if (n_tip != T) throw std::runtime_error("newick: tree has " + std::to_string(n_tip) + " tips, expected " +
                                           std::to_string(T));
  if (N[top].kids.size() == 3) {
    for (int k : N[top].kids) link(top, k, fix(N[k].len));
  } else if (N[top].kids.size() == 2) {
    const int a = N[top].kids[0], b = N[top].kids[1];
    const double la = N[a].len < 0 ? 0.0 : N[a].len, lb = N[b].len < 0 ? 0.0 : N[b].len;
    link(a, b, fix(la + lb));
  } else {
    throw std::runtime_error("newick: top level must have 2 or 3 children");
  }

  int naive = -1;
  std::vector<char> seen_label(T, 0);
  for (std::size_t v = 0; v < N.size(); ++v)
    if (N[v].kids.empty()) {
      const int id = lab2id[N[v].label];
      if (seen_label[id]) throw std::runtime_error("newick: duplicate tip label \"" + N[v].label + "\"");
      seen_label[id] = 1;
      if (id == 0) naive = (int)v;
    }
  if (naive < 0) throw std::runtime_error("newick: tip \"" + labels[0] + "\" not found");

  TreeArrays out;
  out.n_tips = T;
  out.children.assign(2 * (std::size_t)(T - 2), -1);
  out.brlen.assign(2 * (std::size_t)T - 2, 0.0);
  if (adj[naive].size() != 1) throw std::runtime_error("newick: naive must be a tip");
  const int root_old = adj[naive][0].first;
  if (N[root_old].kids.empty()) throw std::runtime_error("newick: naive's neighbour must be an inner node");
  out.brlen[0] = adj[naive][0].second;
  // iterative DFS assigning inner ids T.. in pre-order
  std::vector<int> newid(N.size(), -1);
  int next = T;
  struct Fr {
    int node, parent;
  };
  std::vector<Fr> stack{{root_old, naive}};
  newid[root_old] = next++;
  out.root = newid[root_old];
  while (!stack.empty()) {
    const Fr f = stack.back();
    stack.pop_back();
    int slot = 0;
    for (const auto& nb : adj[f.node]) {
      if (nb.first == f.parent) continue;
      int cid;
      if (N[nb.first].kids.empty()) {
        cid = lab2id[N[nb.first].label];
      } else {
        if (next >= 2 * T - 2 + 1) throw std::runtime_error("newick: too many inner nodes");
        cid = newid[nb.first] = next++;
        stack.push_back({nb.first, f.node});
      }
      if (slot >= 2) throw std::runtime_error("newick: node of degree > 3");
      out.children[2 * (std::size_t)(newid[f.node] - T) + slot++] = cid;
      out.brlen[cid] = nb.second;
    }
    if (slot != 2) throw std::runtime_error("newick: inner node of degree < 3");
  }
  if (next != 2 * T - 2) throw std::runtime_error("newick: not an unrooted binary tree");
  // a rooted (bifurcating) top level is something libpll's unrooted parser rejects: no libpll order exists
  if (with_export && N[top].kids.size() == 3) out.as_parsed = ExportAsParsed(N, top, eps);
  return out;
}

std::string ExportNewick(const TreeArrays& tr, const std::vector<std::string>& labels) {
  if (!tr.as_parsed.empty()) return tr.as_parsed;
  // rooted input (accepted here, rejected by the reference): the unrooted tree, trifurcating at naive's neighbour
  const int T = tr.n_tips;
  char buf[64];
  struct Rec {
    const TreeArrays& tr;
    const std::vector<std::string>& labels;
    int T;
    std::string go(int v) const {
      char b[64];
      std::string s;
      if (v < T) {
        s = labels[v];
      } else {
        s = "(" + go(tr.children[2 * (v - T)]) + "," + go(tr.children[2 * (v - T) + 1]) + ")";
      }
      std::snprintf(b, sizeof b, ":%f", tr.brlen[v]);
      return s + b;
    }
  } rec{tr, labels, T};
  std::snprintf(buf, sizeof buf, ":%f", tr.brlen[0]);
  return "(" + labels[0] + buf + "," + rec.go(tr.children[2 * (tr.root - T)]) + "," +
         rec.go(tr.children[2 * (tr.root - T) + 1]) + ");";
}

}  // namespace linearham
