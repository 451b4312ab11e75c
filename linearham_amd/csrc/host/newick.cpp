#include "newick.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace linearham {

// ---------------------------------------------------------------------------------------------------
// numbers
// ---------------------------------------------------------------------------------------------------

// strtod for the decimal forms a RevBayes table holds ("0.0123", "1e-06", "12.5E3"): up to 19 significant
// digits and a power of ten within +-22 are converted by one correctly rounded multiplication or division
// (both operands exact), which is what strtod returns too; anything else goes to strtod itself.
double ParseDouble(const char* s, const char** end) {
  const char* p = s;
  bool neg = false;
  if (*p == '-' || *p == '+') neg = *p++ == '-';
  uint64_t mant = 0;
  int digits = 0, exp10 = 0;
  bool any = false, simple = true;
  while (*p >= '0' && *p <= '9') {
    any = true;
    if (mant || *p != '0') {
      if (digits < 19)
        mant = mant * 10 + (uint64_t)(*p - '0'), ++digits;
      else
        simple = false;
    }
    ++p;
  }
  if (*p == '.') {
    ++p;
    while (*p >= '0' && *p <= '9') {
      any = true;
      if (mant || *p != '0') {
        if (digits < 19)
          mant = mant * 10 + (uint64_t)(*p - '0'), ++digits;
        else
          simple = false;
      }
      if (simple) --exp10;
      ++p;
    }
  }
  if (!any) {
    char* e = nullptr;
    const double v = std::strtod(s, &e);
    *end = e;
    return v;
  }
  if (*p == 'e' || *p == 'E') {
    const char* q = p + 1;
    bool eneg = false;
    if (*q == '-' || *q == '+') eneg = *q++ == '-';
    if (*q >= '0' && *q <= '9') {
      int ev = 0;
      while (*q >= '0' && *q <= '9') {
        if (ev < 100000) ev = ev * 10 + (*q - '0');
        ++q;
      }
      exp10 += eneg ? -ev : ev;
      p = q;
    }
  }
  static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  if (simple && mant < ((uint64_t)1 << 53) && exp10 >= -22 && exp10 <= 22) {
    double v = (double)mant;
    v = exp10 < 0 ? v / kPow10[-exp10] : v * kPow10[exp10];
    *end = p;
    return neg ? -v : v;
  }
  char* e = nullptr;
  const double v = std::strtod(s, &e);
  *end = e;
  return v;
}

// printf("%f") for a non-negative length: six decimals, the exact binary value rounded half to even -- in
// integers: v = M 2^E, so v 10^6 = (M 10^6) >> -E with the remainder deciding the rounding (no libm call:
// nearbyint / fma were measured not to scale across threads here).
void AppendFixed6(std::string& out, double v) {
  if (!(v >= 0.0) || v >= 4e9) {
    char b[64];
    std::snprintf(b, sizeof b, "%f", v);
    out += b;
    return;
  }
  uint64_t bits;
  std::memcpy(&bits, &v, sizeof bits);
  const int ef = (int)(bits >> 52) & 0x7ff;
  const uint64_t frac = bits & (((uint64_t)1 << 52) - 1);
  const uint64_t M = ef ? (frac | ((uint64_t)1 << 52)) : frac;
  const int E = ef ? ef - 1075 : -1074;  // v = M * 2^E; v < 4e9 < 2^32, so E < 0 whenever M != 0
  const unsigned __int128 N = (unsigned __int128)M * 1000000u;  // < 2^73
  uint64_t q;
  const int sh = -E;
  if (M == 0) {
    q = 0;
  } else if (sh >= 100) {
    q = 0;  // N < 2^73 << 2^(sh - 1): rounds to zero
  } else {
    const unsigned __int128 one = 1;
    const unsigned __int128 quo = N >> sh, rem = N & ((one << sh) - 1), half = one << (sh - 1);
    q = (uint64_t)quo;
    if (rem > half || (rem == half && (q & 1))) ++q;
  }
  const uint64_t ip = q / 1000000u;
  uint32_t fp = (uint32_t)(q % 1000000u);
  char b[32];
  int n = 0;
  char tmp[24];
  int m = 0;
  uint64_t x = ip;
  do {
    tmp[m++] = (char)('0' + x % 10);
    x /= 10;
  } while (x);
  while (m) b[n++] = tmp[--m];
  b[n++] = '.';
  for (int i = 5; i >= 0; --i) {
    b[n + i] = (char)('0' + fp % 10);
    fp /= 10;
  }
  n += 6;
  out.append(b, (std::size_t)n);
}

// ---------------------------------------------------------------------------------------------------
// labels
// ---------------------------------------------------------------------------------------------------

static inline uint64_t HashBytes(const char* s, std::size_t n) {
  uint64_t h = 1469598103934665603ull;  // FNV-1a
  for (std::size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)s[i]) * 1099511628211ull;
  return h;
}

LabelIndex::LabelIndex(const std::vector<std::string>& labels) : labels_(labels) {
  std::size_t cap = 16;
  while (cap < 2 * labels.size() + 2) cap <<= 1;
  slots_.assign(cap, -1);
  for (std::size_t i = 0; i < labels.size(); ++i) {
    std::size_t h = HashBytes(labels[i].data(), labels[i].size()) & (cap - 1);
    while (slots_[h] >= 0) {
      if (labels_[slots_[h]] == labels[i]) throw std::runtime_error("duplicate sequence label \"" + labels[i] + "\"");
      h = (h + 1) & (cap - 1);
    }
    slots_[h] = (int)i;
  }
}

int LabelIndex::Find(const char* s, std::size_t n) const {
  const std::size_t cap = slots_.size();
  std::size_t h = HashBytes(s, n) & (cap - 1);
  while (slots_[h] >= 0) {
    const std::string& l = labels_[slots_[h]];
    if (l.size() == n && std::memcmp(l.data(), s, n) == 0) return slots_[h];
    h = (h + 1) & (cap - 1);
  }
  return -1;
}

// ---------------------------------------------------------------------------------------------------
// parser
// ---------------------------------------------------------------------------------------------------

namespace {

[[noreturn]] void Fail(const std::string& m, std::size_t pos) {
  throw std::runtime_error("newick: " + m + " at offset " + std::to_string(pos));
}

}  // namespace

// One pass over the text, no per-node allocations (the scratch vectors are reused from tree to tree):
// nodes are numbered as they open (pre-order), an inner node's children are recorded as they close.
void ParseNewickInto(const char* t, std::size_t n, const LabelIndex& labels, double eps, NewickScratch& sc,
                     int32_t* children, int32_t* root_out, double* brlen, std::string* exported) {
  const int T = labels.size();
  if (T < 3) throw std::runtime_error("newick: need at least 3 tips (naive + 2 sequences)");
  const int max_nodes = 2 * T + 2;
  sc.parent.assign(max_nodes, -1);
  sc.nk.assign(max_nodes, 0);
  sc.kid.assign((std::size_t)3 * max_nodes, -1);
  sc.len.assign(max_nodes, -1.0);  // < 0: missing
  sc.tip.assign(max_nodes, -1);
  sc.lab_off.assign(max_nodes, 0);
  sc.lab_len.assign(max_nodes, 0);
  sc.seen.assign(T, 0);
  std::size_t pos = 0;
  auto ws = [&] {  // whitespace and [comments] ("[&index=N]" annotations of RevBayes trees)
    for (;;) {
      while (pos < n && (t[pos] == ' ' || t[pos] == '\t' || t[pos] == '\n' || t[pos] == '\r')) ++pos;
      if (pos < n && t[pos] == '[') {
        int depth = 0;
        while (pos < n) {
          if (t[pos] == '[')
            ++depth;
          else if (t[pos] == ']' && --depth == 0) {
            ++pos;
            break;
          }
          ++pos;
        }
        continue;
      }
      return;
    }
  };
  auto label = [&](int v) {
    ws();
    if (pos < n && (t[pos] == '\'' || t[pos] == '"')) {
      const char q = t[pos++];
      sc.lab_off[v] = (uint32_t)pos;
      while (pos < n && t[pos] != q) ++pos;
      if (pos >= n) Fail("unterminated quoted label", pos);
      sc.lab_len[v] = (uint32_t)(pos - sc.lab_off[v]);
      ++pos;
      return;
    }
    sc.lab_off[v] = (uint32_t)pos;
    while (pos < n && t[pos] != ',' && t[pos] != '(' && t[pos] != ')' && t[pos] != ':' && t[pos] != ';' &&
           t[pos] != ' ' && t[pos] != '\t' && t[pos] != '\n' && t[pos] != '[')
      ++pos;
    sc.lab_len[v] = (uint32_t)(pos - sc.lab_off[v]);
  };
  auto length = [&](int v) {
    ws();
    if (pos < n && t[pos] == ':') {
      ++pos;
      ws();
      const char* end = nullptr;
      const double x = ParseDouble(t + pos, &end);
      if (end == t + pos) Fail("bad branch length", pos);
      pos = (std::size_t)(end - t);
      sc.len[v] = x;
    }
  };
  int n_nodes = 0, n_tip = 0, cur = -1, top = -1, naive = -1;
  auto open_node = [&](int parent) {
    if (n_nodes >= max_nodes) Fail("more nodes than a binary tree of these sequences has", pos);
    const int v = n_nodes++;
    sc.parent[v] = parent;
    if (parent >= 0) {
      if (sc.nk[parent] >= (parent == top ? 3 : 2))
        Fail(parent == top ? "top level must have 2 or 3 children" : "inner nodes must be binary", pos);
      sc.kid[(std::size_t)3 * parent + sc.nk[parent]++] = v;
    }
    return v;
  };
  // the text is a tree: '(' opens an inner node, a label is a tip, ',' separates siblings, ')' closes
  bool expect_node = true;
  while (true) {
    ws();
    if (expect_node) {
      if (pos >= n) Fail("unexpected end", pos);
      if (t[pos] == '(') {
        ++pos;
        const int v = open_node(cur);
        if (cur < 0) top = v;
        cur = v;
        continue;  // its first child follows
      }
      if (cur < 0) Fail("a tree starts with '('", pos);
      const int v = open_node(cur);
      label(v);
      if (sc.lab_len[v] == 0) Fail("empty tip label", pos);
      const int id = labels.Find(t + sc.lab_off[v], sc.lab_len[v]);
      if (id < 0) throw std::runtime_error("newick: unknown tip label \"" + std::string(t + sc.lab_off[v], sc.lab_len[v]) + "\"");
      if (sc.seen[id]) throw std::runtime_error("newick: duplicate tip label \"" + labels.label(id) + "\"");
      sc.seen[id] = 1;
      sc.tip[v] = id;
      if (id == 0) naive = v;
      ++n_tip;
      length(v);
      expect_node = false;
      continue;
    }
    // a node has just been completed: ',' -> a sibling follows, ')' -> the parent closes
    if (pos >= n) Fail("unexpected end", pos);
    if (t[pos] == ',') {
      ++pos;
      expect_node = true;
      continue;
    }
    if (t[pos] != ')') Fail("expected ',' or ')'", pos);
    ++pos;
    label(cur);  // inner label: kept for the export only
    length(cur);
    if (sc.nk[cur] < 2) Fail(cur == top ? "top level must have 2 or 3 children" : "inner nodes must be binary", pos);
    const int closed = cur;
    cur = sc.parent[cur];
    if (closed == top) break;
  }
  if (n_tip != T)
    throw std::runtime_error("newick: tree has " + std::to_string(n_tip) + " tips, expected " + std::to_string(T));
  if (naive < 0) throw std::runtime_error("newick: tip \"" + labels.label(0) + "\" not found");
  if (n_nodes != 2 * T - 2 + (sc.nk[top] == 2 ? 1 : 0)) throw std::runtime_error("newick: not an unrooted binary tree");

  // the unrooted tree, re-rooted at naive's neighbour: neighbours of v = its parser children and its parser
  // parent; a bifurcating top level is a node of degree two whose two branches merge into one.
  const bool rooted = sc.nk[top] == 2;
  auto fix = [eps](double l) { return (l < 0.0 || l == 0.0) ? eps : l; };
  auto up = [&](int v) {  // the neighbour on the parent side and the length of that branch
    const int p = sc.parent[v];
    if (rooted && p == top) {
      const int sib = sc.kid[(std::size_t)3 * top] == v ? sc.kid[(std::size_t)3 * top + 1] : sc.kid[(std::size_t)3 * top];
      const double la = sc.len[v] < 0 ? 0.0 : sc.len[v], lb = sc.len[sib] < 0 ? 0.0 : sc.len[sib];
      return std::pair<int, double>(sib, fix(la + lb));
    }
    return std::pair<int, double>(p, fix(sc.len[v]));
  };
  const std::pair<int, double> nb = up(naive);
  const int root_old = nb.first;
  if (root_old < 0 || sc.tip[root_old] >= 0) throw std::runtime_error("newick: naive's neighbour must be an inner node");
  for (std::size_t i = 0; i < 2 * (std::size_t)T - 2; ++i) brlen[i] = 0.0;
  brlen[0] = nb.second;
  // depth-first from the root, inner ids T.. in the order the nodes are reached
  sc.newid.assign(n_nodes, -1);
  sc.stack.clear();
  int next = T;
  sc.newid[root_old] = next++;
  *root_out = sc.newid[root_old];
  sc.stack.push_back({root_old, naive});
  while (!sc.stack.empty()) {
    const NewickScratch::Frame f = sc.stack.back();
    sc.stack.pop_back();
    int slot = 0;
    auto visit = [&](int u, double l) {
      if (u == f.from) return;
      int cid;
      if (sc.tip[u] >= 0) {
        cid = sc.tip[u];
      } else {
        cid = sc.newid[u] = next++;
        sc.stack.push_back({u, f.node});
      }
      if (slot >= 2) throw std::runtime_error("newick: node of degree > 3");
      children[2 * (std::size_t)(sc.newid[f.node] - T) + slot++] = cid;
      brlen[cid] = l;
    };
    for (int c = 0; c < sc.nk[f.node]; ++c) {
      const int k = sc.kid[(std::size_t)3 * f.node + c];
      if (rooted && f.node == top) continue;  // (never reached: the merged top node is not a node of the tree)
      visit(k, fix(sc.len[k]));
    }
    if (f.node != top) {
      const std::pair<int, double> u = up(f.node);
      visit(u.first, u.second);
    }
    if (slot != 2) throw std::runtime_error("newick: inner node of degree < 3");
  }
  if (next != 2 * T - 2) throw std::runtime_error("newick: not an unrooted binary tree");

  if (!exported) return;
  // pll_utree_export_newick(GetVirtualRoot(tree), NULL) [libpll-2, third party; src/PhyloHMM.cpp:299-300] on a
  // tree that pll_utree_parse_newick_string built: libpll's parser hangs the top level's three subtrees on
  // the virtual root's ring in input order and every inner node's two subtrees likewise, and the exporter
  // walks the rings in that order, so the output is the input's own nesting and order with tips printed as
  // "label:%f", inner nodes as "(a,b)label:%f" and the top level as "(a,b,c)label;" (its length dropped),
  // missing and zero lengths already replaced (pt::pll::set_missing_branch_length).
  std::string& out = *exported;
  out.clear();
  if (rooted) {
    // a rooted (bifurcating) top level is something libpll's unrooted parser rejects: no libpll order exists;
    // written as the unrooted tree, trifurcating at naive's neighbour
    struct Rec {
      const int32_t* ch;
      const double* bl;
      const LabelIndex& labels;
      int T;
      void go(int v, std::string& o) const {
        if (v < T) {
          o += labels.label(v);
        } else {
          o.push_back('(');
          go(ch[2 * (v - T)], o);
          o.push_back(',');
          go(ch[2 * (v - T) + 1], o);
          o.push_back(')');
        }
        o.push_back(':');
        AppendFixed6(o, bl[v]);
      }
    } rec{children, brlen, labels, T};
    out.push_back('(');
    out += labels.label(0);
    out.push_back(':');
    AppendFixed6(out, brlen[0]);
    out.push_back(',');
    rec.go(children[2 * (*root_out - T)], out);
    out.push_back(',');
    rec.go(children[2 * (*root_out - T) + 1], out);
    out += ");";
    return;
  }
  out.reserve((std::size_t)n_nodes * 24);
  sc.stack.clear();
  sc.stack.push_back({top, 0});
  while (!sc.stack.empty()) {
    NewickScratch::Frame& f = sc.stack.back();  // from = next child to write
    if (f.from < sc.nk[f.node]) {
      out.push_back(f.from == 0 ? '(' : ',');
      const int k = sc.kid[(std::size_t)3 * f.node + f.from++];
      sc.stack.push_back({k, 0});
      continue;
    }
    if (sc.nk[f.node]) out.push_back(')');
    out.append(t + sc.lab_off[f.node], sc.lab_len[f.node]);
    if (f.node != top) {
      out.push_back(':');
      AppendFixed6(out, fix(sc.len[f.node]));
    }
    sc.stack.pop_back();
  }
  out.push_back(';');
}

TreeArrays ParseNewick(const std::string& text, const std::vector<std::string>& labels, double eps,
                       bool with_export) {
  const LabelIndex index(labels);
  NewickScratch sc;
  TreeArrays out;
  const int T = (int)labels.size();
  out.n_tips = T;
  if (T < 3) throw std::runtime_error("newick: need at least 3 tips (naive + 2 sequences)");
  out.children.assign(2 * (std::size_t)(T - 2), -1);
  out.brlen.assign(2 * (std::size_t)T - 2, 0.0);
  int32_t root = -1;
  ParseNewickInto(text.c_str(), text.size(), index, eps, sc, out.children.data(), &root, out.brlen.data(),
                  with_export ? &out.as_parsed : nullptr);
  out.root = root;
  return out;
}

std::string ExportNewick(const TreeArrays& tr, const std::vector<std::string>& labels) {
  if (!tr.as_parsed.empty()) return tr.as_parsed;
  // no text at hand (a tree given as arrays): the unrooted tree, trifurcating at naive's neighbour
  const int T = tr.n_tips;
  std::string out;
  struct Rec {
    const TreeArrays& tr;
    const std::vector<std::string>& labels;
    int T;
    void go(int v, std::string& o) const {
      if (v < T) {
        o += labels[v];
      } else {
        o.push_back('(');
        go(tr.children[2 * (v - T)], o);
        o.push_back(',');
        go(tr.children[2 * (v - T) + 1], o);
        o.push_back(')');
      }
      o.push_back(':');
      AppendFixed6(o, tr.brlen[v]);
    }
  } rec{tr, labels, T};
  out.push_back('(');
  out += labels[0];
  out.push_back(':');
  AppendFixed6(out, tr.brlen[0]);
  out.push_back(',');
  rec.go(tr.children[2 * (tr.root - T)], out);
  out.push_back(',');
  rec.go(tr.children[2 * (tr.root - T) + 1], out);
  out += ");";
  return out;
}

}  // namespace linearham
