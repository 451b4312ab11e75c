// Constants, tiny dense containers and sequence helpers of the host side
// (mirrors src/utils.hpp / src/utils.cpp of the reference; Eigen is replaced by Matrix<T>).
#ifndef LINEARHAM_UTILS_
#define LINEARHAM_UTILS_

#include <cmath>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "yaml_lite.hpp"

namespace linearham {

/// The linearham epsilon (src/utils.hpp:20).
const double EPS = 1e-6;
/// The linearham scale factor for dealing with numeric underflow (src/utils.hpp:22).
const double SCALE_FACTOR = std::pow(2, 256);
/// The linearham scale threshold (src/utils.hpp:24).
const double SCALE_THRESHOLD = 1.0 / SCALE_FACTOR;

/// Row-major dense matrix standing in for Eigen::MatrixXd / Eigen::MatrixXi.
template <typename T>
class Matrix {
 public:
  Matrix() = default;
  Matrix(int rows, int cols, T fill = T()) : rows_(rows), cols_(cols), d_((std::size_t)rows * cols, fill) {}
  void setConstant(int rows, int cols, T v) {
    rows_ = rows;
    cols_ = cols;
    d_.assign((std::size_t)rows * cols, v);
  }
  void setZero(int rows, int cols) { setConstant(rows, cols, T()); }
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  std::size_t size() const { return d_.size(); }
  T& operator()(int r, int c) { return d_[(std::size_t)r * cols_ + c]; }
  const T& operator()(int r, int c) const { return d_[(std::size_t)r * cols_ + c]; }
  const T* data() const { return d_.data(); }
  T* data() { return d_.data(); }
  const T* row(int r) const { return d_.data() + (std::size_t)r * cols_; }
  T* row(int r) { return d_.data() + (std::size_t)r * cols_; }
  bool operator==(const Matrix& o) const { return rows_ == o.rows_ && cols_ == o.cols_ && d_ == o.d_; }

 private:
  int rows_ = 0, cols_ = 0;
  std::vector<T> d_;
};

typedef Matrix<double> MatrixXd;
typedef Matrix<int> MatrixXi;
typedef std::vector<double> VectorXd;
typedef std::vector<int> VectorXi;

std::pair<std::vector<std::string>, VectorXd> ParseStringProbMap(const yaml_lite::Node& node);
std::string GetAlphabet(const yaml_lite::Node& root);
int GetAlphabetIndex(const std::string& alphabet, char base);
bool MatchGermlineState(const std::string& state_name, const std::string& gname, int* index);
bool MatchNTIState(const std::string& state_name, const std::string& alphabet, char* base);
std::pair<int, int> FindGermlineStartEnd(const yaml_lite::Node& root, const std::string& gname);
VectorXi ConvertSeqToInts(const std::string& seq_str, const std::string& alphabet);
std::string ConvertIntsToSeq(const VectorXi& seq, const std::string& alphabet);
std::string FixGeneName(std::string name);

/// assert() of the reference is live in its release build (no -DNDEBUG, SConstruct:266); here format
/// violations throw so that a library user gets a message instead of an abort.
inline void Require(bool cond, const std::string& what) {
  if (!cond) throw std::runtime_error("linearham: requirement failed: " + what);
}

/// The environment switches of the host library (test hooks; read once per process).
struct HostOptions {
  bool pipeline_timing = false;  // LH_PIPELINE_TIMING: stage times of RunPipeline / RunASR / main on stderr
  bool host_sampling = false;    // LH_HOST_SAMPLING: HMM::SampleRow on the host even where the device sampler could run
  int host_threads = 0;          // LH_HOST_THREADS=<n>: worker threads per host stage (0: from the affinity mask)
};
const HostOptions& host_options();

/// Wall-clock marks printed to stderr when LH_PIPELINE_TIMING is set (where the host side of a run spends its time).
struct StageTimer {
  bool on;
  double t0;
  static double Now();
  StageTimer();
  void Mark(const char* what);
};

}  // namespace linearham

#endif  // LINEARHAM_UTILS_
