// Minimal YAML reader for the two dialects on linearham's hot path (replaces yaml-cpp, which the
// reference links as a system library: SConstruct:270):
//   * partis per-allele HMM files: block maps / block sequences (also "key:\n- item" at the same
//     indent), flow maps and sequences (possibly wrapped over several lines), plain scalars,
//     `null`, `#` comments          (data/hmm_params/*.yaml);
//   * partis cluster files written as JSON-style flow YAML (data/phylo_hmm_input.yaml:1-19).
// Maps keep document order (yaml-cpp iterates maps in document order, which
// ParseStringProbMap relies on: src/utils.cpp:20-35).
#ifndef LINEARHAM_YAML_LITE_
#define LINEARHAM_YAML_LITE_

#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace yaml_lite {

class Node {
 public:
  enum Type { Null, Scalar, Sequence, Map };
  Type type = Null;
  std::string scalar;
  std::vector<Node> seq;
  std::vector<std::pair<std::string, Node>> map;

  bool IsNull() const { return type == Null; }
  bool IsMap() const { return type == Map; }
  bool IsSequence() const { return type == Sequence; }
  bool IsScalar() const { return type == Scalar; }
  std::size_t size() const { return type == Sequence ? seq.size() : type == Map ? map.size() : 0; }

  bool has(const std::string& key) const {
    if (type != Map) return false;
    for (const auto& kv : map)
      if (kv.first == key) return true;
    return false;
  }
  const Node& operator[](const std::string& key) const {
    if (type != Map) throw std::runtime_error("yaml: node is not a map (key \"" + key + "\")");
    for (const auto& kv : map)
      if (kv.first == key) return kv.second;
    throw std::runtime_error("yaml: missing key \"" + key + "\"");
  }
  const Node& operator[](const char* key) const { return (*this)[std::string(key)]; }
  const Node& operator[](std::size_t i) const {
    if (type != Sequence || i >= seq.size()) throw std::runtime_error("yaml: bad sequence index");
    return seq[i];
  }
  const Node& operator[](int i) const { return (*this)[static_cast<std::size_t>(i)]; }

  const std::string& as_string() const {
    if (type != Scalar) throw std::runtime_error("yaml: node is not a scalar");
    return scalar;
  }
  double as_double() const {
    const std::string& s = as_string();
    char* end = nullptr;
    const double v = std::strtod(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') throw std::runtime_error("yaml: not a number: \"" + s + "\"");
    return v;
  }
  int as_int() const {
    const std::string& s = as_string();
    char* end = nullptr;
    const long v = std::strtol(s.c_str(), &end, 10);
    if (end == s.c_str() || *end != '\0') throw std::runtime_error("yaml: not an integer: \"" + s + "\"");
    return static_cast<int>(v);
  }
  bool as_bool() const {
    const std::string& s = as_string();
    if (s == "true" || s == "True" || s == "TRUE" || s == "yes") return true;
    if (s == "false" || s == "False" || s == "FALSE" || s == "no") return false;
    throw std::runtime_error("yaml: not a boolean: \"" + s + "\"");
  }
  char as_char() const {
    const std::string& s = as_string();
    if (s.size() != 1) throw std::runtime_error("yaml: not a single character: \"" + s + "\"");
    return s[0];
  }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : t_(text) {}

  Node ParseDocument() {
    SkipBlankLines();
    if (pos_ >= t_.size()) return Node();
    if (t_.compare(pos_, 3, "---") == 0) {
      SkipLine();
      SkipBlankLines();
    }
    const std::size_t ind = Indent();
    const char c = t_[pos_ + ind];
    Node n;
    if (c == '{' || c == '[') {
      pos_ += ind;
      n = ParseFlow();
    } else {
      n = ParseBlock(ind);
    }
    return n;
  }

 private:
  const std::string& t_;
  std::size_t pos_ = 0;  // always at the start of a line in block context

  [[noreturn]] void Fail(const std::string& msg) const {
    std::size_t line = 1;
    for (std::size_t i = 0; i < pos_ && i < t_.size(); ++i)
      if (t_[i] == '\n') ++line;
    throw std::runtime_error("yaml: " + msg + " (line " + std::to_string(line) + ")");
  }

  void SkipLine() {
    while (pos_ < t_.size() && t_[pos_] != '\n') ++pos_;
    if (pos_ < t_.size()) ++pos_;
  }
  // number of leading spaces of the line starting at pos_
  std::size_t Indent() const {
    std::size_t i = pos_;
    while (i < t_.size() && t_[i] == ' ') ++i;
    return i - pos_;
  }
  bool LineIsBlank() const {
    std::size_t i = pos_;
    while (i < t_.size() && (t_[i] == ' ' || t_[i] == '\t' || t_[i] == '\r')) ++i;
    return i >= t_.size() || t_[i] == '\n' || t_[i] == '#';
  }
  void SkipBlankLines() {
    while (pos_ < t_.size() && LineIsBlank()) SkipLine();
  }

  static std::string Trim(const std::string& s) {
    std::size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
    return s.substr(a, b - a);
  }

  static Node MakeScalar(std::string s) {
    Node n;
    s = Trim(s);
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) {
      n.type = Node::Scalar;
      n.scalar = s.substr(1, s.size() - 2);
      return n;
    }
    if (s.empty() || s == "null" || s == "~" || s == "Null" || s == "NULL") return n;
    n.type = Node::Scalar;
    n.scalar = s;
    return n;
  }

  // ---- block style ----------------------------------------------------------------------------

  bool AtSeqItem(std::size_t at) const {
    return at < t_.size() && t_[at] == '-' && (at + 1 >= t_.size() || t_[at + 1] == ' ' || t_[at + 1] == '\n');
  }

  // Parses the block collection whose entries start at column `ind`; pos_ is at a line start.
  Node ParseBlock(std::size_t ind) {
    SkipBlankLines();
    if (pos_ >= t_.size() || Indent() != ind) return Node();
    if (AtSeqItem(pos_ + ind)) return ParseBlockSeq(ind);
    pos_ += ind;
    return ParseBlockMapFrom(ind);
  }

  Node ParseBlockSeq(std::size_t ind) {
    Node n;
    n.type = Node::Sequence;
    while (true) {
      SkipBlankLines();
      if (pos_ >= t_.size() || Indent() != ind || !AtSeqItem(pos_ + ind)) break;
      pos_ += ind + 1;
      std::size_t extra = 0;
      while (pos_ < t_.size() && t_[pos_] == ' ') {
        ++pos_;
        ++extra;
      }
      const std::size_t item_ind = ind + 1 + extra;
      if (pos_ >= t_.size() || t_[pos_] == '\n' || t_[pos_] == '#') {  // "-" alone: nested block below
        SkipLine();
        SkipBlankLines();
        n.seq.push_back(pos_ < t_.size() && Indent() > ind ? ParseBlock(Indent()) : Node());
        continue;
      }
      if (t_[pos_] == '{' || t_[pos_] == '[') {
        n.seq.push_back(ParseFlow());
        SkipLine();
        continue;
      }
      if (LooksLikeMapEntry()) {
        n.seq.push_back(ParseBlockMapFrom(item_ind));
      } else {
        n.seq.push_back(MakeScalar(RestOfLineNoComment()));
        SkipLine();
      }
    }
    return n;
  }

  // Is the text from pos_ to end of line of the form `key:( |$)`?
  bool LooksLikeMapEntry() const {
    std::size_t i = pos_;
    if (i < t_.size() && (t_[i] == '"' || t_[i] == '\'')) {
      const char q = t_[i++];
      while (i < t_.size() && t_[i] != q && t_[i] != '\n') ++i;
      if (i < t_.size() && t_[i] == q) ++i;
      while (i < t_.size() && t_[i] == ' ') ++i;
      return i < t_.size() && t_[i] == ':';
    }
    for (; i < t_.size() && t_[i] != '\n'; ++i) {
      if (t_[i] == ':' && (i + 1 >= t_.size() || t_[i + 1] == ' ' || t_[i + 1] == '\n' || t_[i + 1] == '\r'))
        return true;
      if (t_[i] == '#' && i > pos_ && t_[i - 1] == ' ') return false;
    }
    return false;
  }

  std::string RestOfLineNoComment() const {
    std::size_t i = pos_;
    std::string out;
    while (i < t_.size() && t_[i] != '\n') {
      if (t_[i] == '#' && (i == pos_ || t_[i - 1] == ' ')) break;
      out.push_back(t_[i++]);
    }
    return out;
  }

  // pos_ is at the first key of a block map whose keys sit at column `ind` (possibly mid-line,
  // right after "- ").
  Node ParseBlockMapFrom(std::size_t ind) {
    Node n;
    n.type = Node::Map;
    while (true) {
      // key
      std::string key;
      if (t_[pos_] == '"' || t_[pos_] == '\'') {
        const char q = t_[pos_++];
        while (pos_ < t_.size() && t_[pos_] != q) key.push_back(t_[pos_++]);
        ++pos_;
        while (pos_ < t_.size() && t_[pos_] == ' ') ++pos_;
        if (pos_ >= t_.size() || t_[pos_] != ':') Fail("expected ':' after quoted key");
      } else {
        while (pos_ < t_.size() && t_[pos_] != '\n' &&
               !(t_[pos_] == ':' && (pos_ + 1 >= t_.size() || t_[pos_ + 1] == ' ' || t_[pos_ + 1] == '\n' ||
                                     t_[pos_ + 1] == '\r')))
          key.push_back(t_[pos_++]);
        if (pos_ >= t_.size() || t_[pos_] != ':') Fail("expected 'key:' in block map");
        key = Trim(key);
      }
      ++pos_;  // ':'
      while (pos_ < t_.size() && t_[pos_] == ' ') ++pos_;
      Node value;
      if (pos_ < t_.size() && (t_[pos_] == '{' || t_[pos_] == '[')) {
        value = ParseFlow();
        SkipLine();
      } else {
        const std::string rest = Trim(RestOfLineNoComment());
        SkipLine();
        if (!rest.empty()) {
          value = MakeScalar(rest);
        } else {
          SkipBlankLines();
          if (pos_ < t_.size()) {
            const std::size_t nind = Indent();
            if (nind > ind)
              value = ParseBlock(nind);
            else if (nind == ind && AtSeqItem(pos_ + nind))
              value = ParseBlockSeq(nind);  // "key:\n- item" at the same indentation
          }
        }
      }
      n.map.emplace_back(key, std::move(value));
      SkipBlankLines();
      if (pos_ >= t_.size() || Indent() != ind || AtSeqItem(pos_ + ind)) break;
      pos_ += ind;
    }
    return n;
  }

  // ---- flow style (may span lines) ---------------------------------------------------------------

  void SkipFlowSpace() {
    while (pos_ < t_.size()) {
      const char c = t_[pos_];
      if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
        ++pos_;
      } else if (c == '#' && (pos_ == 0 || t_[pos_ - 1] == ' ' || t_[pos_ - 1] == '\n')) {
        while (pos_ < t_.size() && t_[pos_] != '\n') ++pos_;
      } else {
        break;
      }
    }
  }

  std::string ParseFlowScalarText(bool is_key) {
    std::string s;
    if (t_[pos_] == '"' || t_[pos_] == '\'') {
      const char q = t_[pos_++];
      while (pos_ < t_.size() && t_[pos_] != q) {
        if (t_[pos_] == '\\' && q == '"' && pos_ + 1 < t_.size()) ++pos_;
        s.push_back(t_[pos_++]);
      }
      if (pos_ >= t_.size()) Fail("unterminated quoted string");
      ++pos_;
      return std::string(1, '\x01') + s;  // marker: quoted (never null)
    }
    while (pos_ < t_.size()) {
      const char c = t_[pos_];
      if (c == ',' || c == '}' || c == ']' || c == '\n') break;
      if (c == ':' && (is_key || pos_ + 1 >= t_.size() || t_[pos_ + 1] == ' ' || t_[pos_ + 1] == '\n')) {
        if (is_key) break;
      }
      s.push_back(c);
      ++pos_;
    }
    return Trim(s);
  }

  static Node FlowScalarNode(const std::string& raw) {
    if (!raw.empty() && raw[0] == '\x01') {
      Node n;
      n.type = Node::Scalar;
      n.scalar = raw.substr(1);
      return n;
    }
    return MakeScalar(raw);
  }

  Node ParseFlow() {
    SkipFlowSpace();
    if (pos_ >= t_.size()) Fail("unexpected end of input in flow collection");
    Node n;
    if (t_[pos_] == '{') {
      n.type = Node::Map;
      ++pos_;
      while (true) {
        SkipFlowSpace();
        if (pos_ >= t_.size()) Fail("unterminated flow map");
        if (t_[pos_] == '}') {
          ++pos_;
          break;
        }
        std::string key = ParseFlowScalarText(true);
        if (!key.empty() && key[0] == '\x01') key = key.substr(1);
        SkipFlowSpace();
        if (pos_ >= t_.size() || t_[pos_] != ':') Fail("expected ':' in flow map");
        ++pos_;
        Node v = ParseFlow();
        n.map.emplace_back(key, std::move(v));
        SkipFlowSpace();
        if (pos_ < t_.size() && t_[pos_] == ',') ++pos_;
      }
    } else if (t_[pos_] == '[') {
      n.type = Node::Sequence;
      ++pos_;
      while (true) {
        SkipFlowSpace();
        if (pos_ >= t_.size()) Fail("unterminated flow sequence");
        if (t_[pos_] == ']') {
          ++pos_;
          break;
        }
        n.seq.push_back(ParseFlow());
        SkipFlowSpace();
        if (pos_ < t_.size() && t_[pos_] == ',') ++pos_;
      }
    } else {
      n = FlowScalarNode(ParseFlowScalarText(false));
    }
    return n;
  }
};

inline Node Load(const std::string& text) { return Parser(text).ParseDocument(); }

inline Node LoadFile(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("yaml: cannot open \"" + path + "\"");
  std::stringstream ss;
  ss << in.rdbuf();
  const std::string text = ss.str();
  return Parser(text).ParseDocument();
}

}  // namespace yaml_lite

#endif  // LINEARHAM_YAML_LITE_
