// C ABI of liblinearham_hip.so (see include/linearham_amd.h): family upload, tree scheduling,
// batched evaluation.  Host-side code only; the kernels live in lh_model/lh_prune/lh_forward.hip.
#include <algorithm>
#include <map>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "lh_device.h"

namespace lh {
const DebugOptions& debug_options() {
  static const DebugOptions opts = [] {
    DebugOptions o;
    auto set = [](const char* name) { return std::getenv(name) != nullptr; };
    auto num = [](const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; };
    o.chunk = std::max(256, num("LH_CHUNK", o.chunk));
    o.host_sub = std::max(256, num("LH_HOST_SUB", o.host_sub));
    o.k2b_no_pair = set("LH_K2B_NO_PAIR");
    o.k2b_vd_single = set("LH_K2B_VD_SINGLE");
    o.sample_timing = set("LH_SAMPLE_TIMING");
    o.k1_tile_cap = num("LH_K1_TILE_CAP", 0);
    o.k1_cxx_walk = set("LH_K1_CXX_WALK");
    o.k1_tables = set("LH_K1_TABLES");
    o.k1_stack = set("LH_K1_STACK");
    o.k1_no_tables = set("LH_K1_NO_TABLES");
    o.k1_segments = set("LH_K1_SEGMENTS");
    o.k1_seg_waves = num("LH_K1_SEG_WAVES", o.k1_seg_waves);
    o.k1_no_fuse = set("LH_K1_NO_FUSE");
    return o;
  }();
  return opts;
}
}  // namespace lh

namespace {

thread_local std::string g_error;

int fail(const std::string& msg) {
  g_error = msg;
  return 1;
}

#define LH_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
  } while (0)

struct Workspace {
  int n_cap = 0, R = 0, T = 0;
  double *rates = nullptr, *eig = nullptr, *site_lik = nullptr;
  int32_t* site_scal = nullptr;
  lh::PruneWs prune{};  // K0c's checked / rewritten schedules and K1's scratch area; err_flag is the family's
};

struct ForwardWs {  // K2a -> K2b hand-off, sized by the largest batch seen
  int n_cap = 0;
  double *gem = nullptr, *jem = nullptr, *dxf = nullptr;
  int32_t *gcnt = nullptr, *jrs = nullptr, *dxc = nullptr;
};

struct AsrWs {  // K3's CLV area and the device copies of lh_asr_batch's host arrays (grow-only)
  size_t clv_cap = 0;
  double* clv = nullptr;
  size_t choice_cap = 0;
  uint8_t* choice = nullptr;  // K3a -> K3b when the caller does not ask for the rate categories
  size_t desc_cap = 0;
  void* desc = nullptr;       // K3s -> K3b schedule descriptors
  size_t cap[8] = {0};
  void* ptr[8] = {nullptr};
  double ms = 0;          // K3 time of the profiled launches (lh_profile_enable)
  int64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};

struct Staging {  // device copies of host inputs/outputs for the host-pointer entry points
  size_t cap[10] = {0};
  void* ptr[10] = {nullptr};
};

// lh_eval_batch's host -> device pipeline: two pinned staging slots, a copy stream and a compute stream
struct HostPipe {
  void* pinned[2] = {nullptr, nullptr};
  size_t cap = 0;
  hipStream_t copy = nullptr, comp = nullptr;
  hipEvent_t staged[2] = {nullptr, nullptr};
};

struct EventSet {
  hipEvent_t e[4];
};

}  // namespace

struct lh_family {
  int device = 0;
  lh::DevFamily host{};            // device pointers inside
  lh::DevFamily* dev = nullptr;    // device copy of `host`
  std::vector<void*> allocs;
  char* arena_ptr = nullptr;
  size_t arena_left = 0;
  Workspace ws;
  ForwardWs fws;
  AsrWs asr;
  Staging st;
  HostPipe pipe;
  bool profile = false;
  bool extended = false;  // lh_family_set_extended_range
  bool have_sampler = false;
  lh::DevSampler sampler{};  // device pointers inside (arena)
  const lh::DevSampler* sampler_dev = nullptr;  // its device copy (K4 reads the tables' addresses from memory)
  struct {                   // lh_eval_sample_batch's device buffers (grow-only)
    size_t cap[9] = {0};
    void* ptr[9] = {nullptr};
    void* pinned = nullptr;  // page-locked staging of the six input arrays
    size_t pinned_cap = 0;
  } smp;
  int32_t n_ucol_used = 0;  // (naive base, pattern) pairs some xMSA column has (lh_family_info)
  int32_t* err_flag = nullptr;  // device word (arena): K0c sets it when a schedule is malformed (lh_family_status)
  std::string k1_form;          // the K1 kernel form of the last evaluation (lh_family_prune_form)
  std::vector<EventSet> events;
  double ms[3] = {0, 0, 0};
  int64_t launches = 0;
};

namespace {

// A handle belongs to the device that was current when lh_family_create ran; every entry point makes that device
// current for its duration, so handles of different GPUs can be driven from any thread (one thread per handle).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const lh_family* f) {
    if (f && hipGetDevice(&prev) == hipSuccess && prev != f->device) switched = hipSetDevice(f->device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// All family tables are sub-allocated from a few large device chunks: the forward kernel touches ~45
// small tables per junction row, and one allocation (and page) per table costs TLB reach.
int arena_alloc(lh_family* f, size_t bytes, void** out) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  if (f->arena_left < bytes) {
    const size_t chunk = std::max(bytes, (size_t)8 << 20);
    void* p = nullptr;
    LH_HIP(hipMalloc(&p, chunk));
    f->allocs.push_back(p);
    f->arena_ptr = static_cast<char*>(p);
    f->arena_left = chunk;
  }
  *out = f->arena_ptr;
  f->arena_ptr += bytes;
  f->arena_left -= bytes;
  return 0;
}

template <typename T>
int upload(lh_family* f, const T* src, size_t count, const T** dst) {
  *dst = nullptr;
  void* p = nullptr;
  if (arena_alloc(f, count * sizeof(T), &p)) return 1;  // count == 0: a valid dummy address
  if (count > 0) {
    if (!src) return fail("lh_family_create: null array in descriptor");
    LH_HIP(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
  }
  *dst = static_cast<const T*>(p);
  return 0;
}

// `ucol` translates the caller's xMSA column indices into u-columns (lh_device.h); the sentinel column
// whose emission is 1.0 sits at position n_ucol.
template <typename T>
int upload_vec(lh_family* f, const std::vector<T>& v, const T** out);

// Consensus form of a segment set (DevSegments): only when every gene's factors sit on consecutive alignment
// sites (xmsa_site known), the set spans at most 510 sites, and the form at least halves the factor count.
int upload_consensus(lh_family* f, const lh_segments& s, const std::vector<int32_t>& ucol, int n_ucol,
                     const int32_t* xmsa_site, int scale, lh::DevSegments* d) {
  // test hook: always the factor-by-factor walk (a property of the family being created: read here, so that a test can
  // build both forms in one process)
  const bool off = std::getenv("LH_K2A_DIRECT") != nullptr;
  if (off || !xmsa_site || s.n_genes < 1) return 0;
  const int n = s.n_genes;
  int lo = INT32_MAX, hi = -1;
  long long total = 0;
  for (int g = 0; g < n; ++g) {
    const int a = s.offsets[g], b = s.offsets[g + 1];
    if (a == b) continue;
    for (int j = a + 1; j < b; ++j)
      if (xmsa_site[s.xmsa_inds[j]] != xmsa_site[s.xmsa_inds[j - 1]] + 1) return 0;  // not site-aligned
    lo = std::min(lo, xmsa_site[s.xmsa_inds[a]]);
    hi = std::max(hi, xmsa_site[s.xmsa_inds[b - 1]] + 1);
    total += b - a;
  }
  if (hi < 0 || total < 4096) return 0;  // nothing to gain on a small set: the scan and its barriers cost more
  const int ns = hi - lo;
  if (ns > 510) return 0;
  // consensus column per site: the most frequent u-column among the genes covering it (ties: smallest)
  std::vector<std::map<int32_t, int>> votes(ns);
  for (int g = 0; g < n; ++g)
    for (int j = s.offsets[g]; j < s.offsets[g + 1]; ++j) ++votes[xmsa_site[s.xmsa_inds[j]] - lo][ucol[s.xmsa_inds[j]]];
  std::vector<int32_t> cons(ns, n_ucol);  // uncovered sites (gaps between genes): the sentinel, emission 1
  for (int p = 0; p < ns; ++p) {
    int best = 0;
    for (const auto& kv : votes[p])
      if (kv.second > best) {
        best = kv.second;
        cons[p] = kv.first;
      }
  }
  std::vector<std::vector<uint32_t>> diffs(n);
  std::vector<uint32_t> rng(n, 0);
  size_t max_diff = 0;
  for (int g = 0; g < n; ++g) {
    const int a = s.offsets[g], b = s.offsets[g + 1];
    if (a == b) continue;
    const int first = xmsa_site[s.xmsa_inds[a]] - lo;
    rng[g] = (uint32_t)first | ((uint32_t)(first + (b - a)) << 16);
    for (int j = a; j < b; ++j) {
      const int p = first + (j - a);
      const int32_t u = ucol[s.xmsa_inds[j]];
      if (u != cons[p]) diffs[g].push_back((uint32_t)p | ((uint32_t)(u * scale) << 16));
    }
    max_diff = std::max(max_diff, diffs[g].size());
    rng[g] |= (uint32_t)((diffs[g].size() + 7) / 8) << 25;  // rounds of eight departures (<= 64: at most 510 sites)
  }
  max_diff = (max_diff + 7) & ~(size_t)7;  // the kernel takes the departures eight at a time
  // worth it?  work of the consensus form (scan + per gene a division and its diffs) against the plain walk
  if ((long long)ns + (long long)n * (long long)(max_diff + 8) > total / 2) return 0;
  std::vector<uint16_t> col(ns);
  for (int p = 0; p < ns; ++p) col[p] = (uint16_t)(cons[p] * scale);
  const uint32_t pad = (uint32_t)ns | ((uint32_t)(n_ucol * scale) << 16);
  std::vector<uint32_t> dif(std::max<size_t>(max_diff, 8) * n, pad);
  for (int g = 0; g < n; ++g)
    for (size_t k = 0; k < diffs[g].size(); ++k) dif[k * n + g] = diffs[g][k];
  if (upload_vec(f, col, &d->cons_col)) return 1;
  if (upload_vec(f, rng, &d->cons_rng)) return 1;
  if (upload_vec(f, dif, &d->cons_dif)) return 1;
  d->cons_sites = ns;
  d->cons_diffs = (int32_t)max_diff;
  return 0;
}

int upload_segments(lh_family* f, const lh_segments& s, int n_xmsa, const std::vector<int32_t>& ucol, int n_ucol,
                    const int32_t* xmsa_site, lh::DevSegments* d) {
  const int scale = f->host.idx_byte_offsets ? 8 : 1;  // byte offsets into the LDS emission vector
  if (s.n_genes < 0) return fail("segments: negative gene count");
  d->n_genes = s.n_genes;
  d->cons_sites = 0;
  d->cons_diffs = 0;
  d->cons_col = nullptr;
  d->cons_rng = nullptr;
  d->cons_dif = nullptr;
  if (s.n_genes > 0) {
    if (!s.offsets) return fail("segments: null offsets");
    if (s.offsets[0] != 0) return fail("segments: offsets[0] != 0");
    for (int g = 0; g < s.n_genes; ++g)
      if (s.offsets[g + 1] < s.offsets[g]) return fail("segments: offsets not monotone");
    const int total = s.offsets[s.n_genes];
    for (int j = 0; j < total; ++j)
      if (s.xmsa_inds[j] < 0 || s.xmsa_inds[j] >= n_xmsa) return fail("segments: xMSA index out of range");
    int longest = 0;
    for (int g = 0; g < s.n_genes; ++g) longest = std::max(longest, s.offsets[g + 1] - s.offsets[g]);
    d->n_chunks = (longest + 7) / 8;
    if (n_ucol > 0xfffe) return fail("segments: more than 65534 distinct xMSA columns");
    // [chunk][gene][8] 16-bit indices, sentinel = column n_ucol (em = 1)
    std::vector<uint16_t> t((size_t)d->n_chunks * s.n_genes * 8, (uint16_t)(n_ucol * scale));
    for (int g = 0; g < s.n_genes; ++g)
      for (int j = s.offsets[g]; j < s.offsets[g + 1]; ++j) {
        const int k = j - s.offsets[g];
        t[((size_t)(k / 8) * s.n_genes + g) * 8 + (k % 8)] = (uint16_t)(ucol[s.xmsa_inds[j]] * scale);
      }
    const uint16_t* dev = nullptr;
    if (upload(f, t.data(), t.size(), &dev)) return 1;
    d->inds_c = reinterpret_cast<const uint4*>(dev);
    if (upload_consensus(f, s, ucol, n_ucol, xmsa_site, scale, d)) return 1;
  } else {
    d->n_chunks = 0;
    const uint16_t* dev = nullptr;
    if (upload<uint16_t>(f, nullptr, 0, &dev)) return 1;
    d->inds_c = reinterpret_cast<const uint4*>(dev);
  }
  return 0;
}

int check_idx(const int32_t* a, size_t n, int n_xmsa, bool allow_neg, const char* what) {
  for (size_t i = 0; i < n; ++i)
    if (a[i] >= n_xmsa || (a[i] < 0 && !(allow_neg && a[i] == -1)))
      return fail(std::string("junction: xMSA index out of range in ") + what);
  return 0;
}

// Validates a junction's emission-column indices and marks the columns it uses.
int collect_junction_cols(const lh_junction& j, int n_xmsa, std::vector<int32_t>* used) {
  const size_t W = j.n_rows, nL = j.n_left, nR = j.n_right;
  if (j.n_rows < 1 || j.n_left < 1 || j.n_right < 1) return fail("junction: bad dimensions");
  if (!j.left_xmsa || !j.right_xmsa || !j.nti_xmsa) return fail("lh_family_create: null array in junction descriptor");
  if (check_idx(j.left_xmsa, W * nL, n_xmsa, true, "left_xmsa")) return 1;
  if (check_idx(j.right_xmsa, W * nR, n_xmsa, true, "right_xmsa")) return 1;
  if (check_idx(j.nti_xmsa, W * nR * 4, n_xmsa, false, "nti_xmsa")) return 1;
  auto mark = [&](const int32_t* p, size_t n) {
    for (size_t i = 0; i < n; ++i)
      if (p[i] >= 0) (*used)[p[i]] = 1;
  };
  mark(j.left_xmsa, W * nL);
  mark(j.right_xmsa, W * nR);
  mark(j.nti_xmsa, W * nR * 4);
  return 0;
}

// Copies a [rows][n][inner] table into [rows][n_pad][inner], filling the padding with `fill`.
template <typename T>
std::vector<T> pad_genes(const T* src, size_t rows, size_t n, size_t n_pad, size_t inner, T fill) {
  std::vector<T> t(rows * n_pad * inner, fill);
  for (size_t i = 0; i < rows; ++i)
    for (size_t g = 0; g < n; ++g)
      for (size_t u = 0; u < inner; ++u) t[(i * n_pad + g) * inner + u] = src[(i * n + g) * inner + u];
  return t;
}

template <typename T>
int upload_vec(lh_family* f, const std::vector<T>& v, const T** out) {
  return upload(f, v.data(), v.size(), out);
}

// `remap` translates xMSA column indices into positions of the compact junction-column list; -1 (the
// state does not emit at this site) and padding become the zero sentinel at position n_jcols.
// left_chunks / right_chunks: the 64-gene register chunks K2b's kernel template gives each side of THIS junction
// (lh_forward.hip launch_forward: the V side 1 / 2 / 4 / 8 / 16 by the V alleles, every D or J side 1 / 2 / 4 by the LARGER of
// the D and J sets).  The tables are padded to that width, not to the side's own gene count rounded up: a lane reads
// entry lane + 64 q of a row for every q of its template (round 4: with 65 D and 30 J alleles the D-J junction's J side was
// padded to 64 and read two chunks wide -- the second chunk was the NEXT row's entries, behind the last row whatever
// followed the table; such lanes feed no result, but their values entered the row's ScaleMatrix key, and a tiny one
// scaled the row's real entries to inf; found by tests/dev_tools/random_sweep_pipeline.py --many).
int upload_junction(lh_family* f, const lh_junction& j, const std::vector<int32_t>& remap, int n_jcols,
                    const int32_t* xmsa_site, const std::vector<int32_t>& pat_of_site, int left_chunks, int right_chunks,
                    lh::DevJunction* d) {
  const size_t W = j.n_rows, nL = j.n_left, nR = j.n_right;
  const size_t pL = std::max<size_t>((nL + 63) / 64, (size_t)left_chunks) * 64,
               pR = std::max<size_t>((nR + 63) / 64, (size_t)right_chunks) * 64;
  d->n_rows = j.n_rows;
  d->n_left = j.n_left;
  d->n_right = j.n_right;
  d->left_pad = (int32_t)pL;
  d->right_pad = (int32_t)pR;
  if (!j.enter_trans || !j.enter_lo || !j.left_trans || !j.left_lo || !j.right_gp_nli || !j.right_ntt ||
      !j.right_nlo || !j.right_trans || !j.right_gp_li || !j.exit_nlo || !j.exit_trans || !j.exit_gp_li)
    return fail("lh_family_create: null array in junction descriptor");
  auto cols = [&](const int32_t* src, size_t rows, size_t n, size_t n_pad, size_t inner) {
    std::vector<int32_t> t = pad_genes<int32_t>(src, rows, n, n_pad, inner, -1);
    for (int32_t& x : t) x = x >= 0 ? remap[x] : n_jcols;
    return t;
  };
  std::vector<double> ltr = pad_genes<double>(j.left_trans, W, nL, pL, 1, 0.0);
  for (size_t l = 0; l < nL; ++l) ltr[l] = j.enter_trans[l];  // row 0 is entered from the germline region
  std::vector<double> ntt(pR * 16, 0.0);
  for (size_t r = 0; r < nR; ++r)
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) ntt[r * 16 + b * 4 + a] = j.right_ntt[r * 16 + a * 4 + b];
  if (upload_vec(f, pad_genes<double>(j.enter_lo, 1, nL, pL, 1, 0.0), &d->enter_lo)) return 1;
  if (upload_vec(f, ltr, &d->left_trans)) return 1;
  if (upload_vec(f, pad_genes<double>(j.left_lo, W, nL, pL, 1, 0.0), &d->left_lo)) return 1;
  if (upload_vec(f, cols(j.left_xmsa, W, nL, pL, 1), &d->left_xmsa)) return 1;
  if (upload_vec(f, pad_genes<double>(j.right_gp_nli, 1, nR, pR, 4, 0.0), &d->right_gp_nli)) return 1;
  if (upload_vec(f, ntt, &d->right_ntt)) return 1;
  if (upload_vec(f, pad_genes<double>(j.right_nlo, W, nR, pR, 4, 0.0), &d->right_nlo)) return 1;
  if (upload_vec(f, pad_genes<double>(j.right_trans, W, nR, pR, 1, 0.0), &d->right_trans)) return 1;
  if (upload_vec(f, pad_genes<double>(j.right_gp_li, W, nR, pR, 1, 0.0), &d->right_gp_li)) return 1;
  if (upload_vec(f, cols(j.right_xmsa, W, nR, pR, 1), &d->right_xmsa)) return 1;
  if (upload_vec(f, cols(j.nti_xmsa, W, nR, pR, 4), &d->nti_xmsa)) return 1;
  if (upload_vec(f, pad_genes<double>(j.exit_nlo, 1, nR, pR, 4, 0.0), &d->exit_nlo)) return 1;
  if (upload_vec(f, pad_genes<double>(j.exit_trans, 1, nR, pR, 1, 0.0), &d->exit_trans)) return 1;
  if (upload_vec(f, pad_genes<double>(j.exit_gp_li, 1, nR, pR, 1, 0.0), &d->exit_gp_li)) return 1;
  // pattern of each row's alignment site, through the NTI emission column of the row (always present)
  std::vector<int32_t> row_pat(W, f->host.n_prune);
  if (xmsa_site)
    for (size_t i = 0; i < W; ++i) row_pat[i] = pat_of_site[xmsa_site[j.nti_xmsa[i * nR * 4]]];
  if (upload_vec(f, row_pat, &d->row_pat)) return 1;
  return 0;
}

// bytes of K1 workspace per sample: scratch area (P-matrices, cherry tables) and K0c's schedule arrays
size_t k1_bytes_per_sample(const lh_family* f, int T, int R) {
  const lh::PruneWsSizes z = lh::prune_ws_sizes(T, f->host.msa_mixed_n != 0);
  const size_t n_ops = (size_t)std::max(T - 2, 1);
  return sizeof(double) * R * z.scratch_doubles_per_rate + n_ops * (sizeof(int2) + sizeof(double)) +
         z.tabs_per_sample * sizeof(int4) + sizeof(int4);
}

int ensure_workspace(lh_family* f, int n, int R, int T) {
  Workspace& w = f->ws;
  if (n <= w.n_cap && R == w.R && T == w.T) return 0;
  void** bufs[] = {(void**)&w.rates,         (void**)&w.eig,        (void**)&w.site_lik,   (void**)&w.site_scal,
                   (void**)&w.prune.scratch, (void**)&w.prune.wops, (void**)&w.prune.wlen, (void**)&w.prune.tabs,
                   (void**)&w.prune.hdr};
  for (void** b : bufs) {
    if (*b) LH_HIP(hipFree(*b));
    *b = nullptr;
  }
  w.n_cap = 0;
  const size_t L = f->host.n_prune;
  const int cap = std::max(n, 1);
  LH_HIP(hipMalloc((void**)&w.rates, sizeof(double) * cap * R));
  LH_HIP(hipMalloc((void**)&w.eig, sizeof(double) * cap * 36));
  // K1's scratch area per (sample, rate): the walk's P-matrices and cherry tables; K0c's per-sample schedule arrays
  const lh::PruneWsSizes z = lh::prune_ws_sizes(T, f->host.msa_mixed_n != 0);
  const size_t n_ops = (size_t)std::max(T - 2, 1);
  LH_HIP(hipMalloc((void**)&w.prune.scratch, sizeof(double) * cap * R * z.scratch_doubles_per_rate));
  LH_HIP(hipMalloc((void**)&w.prune.wops, sizeof(int2) * cap * n_ops));
  LH_HIP(hipMalloc((void**)&w.prune.wlen, sizeof(double) * cap * n_ops));
  LH_HIP(hipMalloc((void**)&w.prune.tabs, sizeof(int4) * cap * z.tabs_per_sample));
  LH_HIP(hipMalloc((void**)&w.prune.hdr, sizeof(int4) * cap));
  w.prune.err_flag = f->err_flag;
  LH_HIP(hipMalloc((void**)&w.site_lik, sizeof(double) * cap * R * 5 * std::max(L, (size_t)1)));
  LH_HIP(hipMalloc((void**)&w.site_scal, sizeof(int32_t) * cap * R * std::max(L, (size_t)1)));
  w.n_cap = cap;
  w.R = R;
  w.T = T;
  return 0;
}

int stage(lh_family* f, int slot, size_t bytes, void** out) {
  Staging& s = f->st;
  if (bytes > s.cap[slot]) {
    if (s.ptr[slot]) LH_HIP(hipFree(s.ptr[slot]));
    s.ptr[slot] = nullptr;
    s.cap[slot] = 0;
    LH_HIP(hipMalloc(&s.ptr[slot], bytes));
    s.cap[slot] = bytes;
  }
  *out = s.ptr[slot];
  return 0;
}

// samples per launch group (bounds the workspace: ~150 KB per sample for a 100-tip tree; a multiple of 6144 = whole
// rounds of all three kernels on 256 CUs for configs[2]-like shapes); LH_CHUNK: test hook
static const int kChunk = lh::debug_options().chunk;

int run_forward(lh_family* f, int n, int R, const double* site_lik, const int32_t* site_scal, const double* pi,
                const double* em_in, double* em_out, double* loglik_dev, const lh_eval_outputs* outs,
                size_t sample_offset, hipStream_t stream) {
  double* fwd = (outs && outs->forward) ? outs->forward + sample_offset * f->host.forward_size : nullptr;
  int32_t* sco =
      (outs && outs->scaler_counts) ? outs->scaler_counts + sample_offset * f->host.scaler_size : nullptr;
  ForwardWs& w = f->fws;
  if (n > w.n_cap) {
    // growing the hand-off buffers: earlier launches on other streams may still be using them
    LH_HIP(hipDeviceSynchronize());
    void** bufs[] = {(void**)&w.gem, (void**)&w.jem, (void**)&w.gcnt, (void**)&w.jrs, (void**)&w.dxf, (void**)&w.dxc};
    for (void** b : bufs) {
      if (*b) LH_HIP(hipFree(*b));
      *b = nullptr;
    }
    w.n_cap = 0;
    LH_HIP(hipMalloc((void**)&w.jrs, sizeof(int32_t) * (size_t)n * std::max(f->host.vd.n_rows + f->host.dj.n_rows, 1)));
    LH_HIP(hipMalloc((void**)&w.dxf, sizeof(double) * (size_t)n * 32));
    LH_HIP(hipMalloc((void**)&w.dxc, sizeof(int32_t) * (size_t)n));
    LH_HIP(hipMalloc((void**)&w.gem, sizeof(double) * (size_t)n * std::max<int64_t>(f->host.gem_size, 1)));
    LH_HIP(hipMalloc((void**)&w.jem, sizeof(double) * (size_t)n * std::max(f->host.n_jcols, 1)));
    LH_HIP(hipMalloc((void**)&w.gcnt, sizeof(int32_t) * (size_t)n * 3));
    w.n_cap = n;
  }
  lh::launch_forward(f->host, f->dev, n, R, site_lik, site_scal, pi, em_in, em_out, w.gem, w.gcnt, w.jem, w.jrs, w.dxf, w.dxc,
                     loglik_dev, fwd, sco, f->extended, stream);
  LH_HIP(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

const char* lh_last_error(void) { return g_error.c_str(); }

int lh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int lh_set_device(int32_t device) {
  if (device < 0 || device >= lh_device_count()) return fail("lh_set_device: no such device");
  LH_HIP(hipSetDevice(device));
  return 0;
}

int lh_warmup(void) {
  if (lh_device_count() < 1) return fail("lh_warmup: no HIP device available");
  LH_HIP(hipFree(nullptr));  // creates the primary context of the current device
  return 0;
}

void* lh_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    fail("lh_host_alloc: hipHostMalloc failed");
    return nullptr;
  }
  return p;
}

void lh_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int lh_family_create(const lh_family_desc* desc, lh_family** out) {
  if (!desc || !out) return fail("lh_family_create: null argument");
  *out = nullptr;
  if (desc->abi_version != LH_ABI_VERSION) return fail("lh_family_create: ABI version mismatch");
  if (lh_device_count() < 1)
    return fail("lh_family_create: no HIP device available (this library has no CPU path)");
  if (desc->n_xmsa < 1) return fail("lh_family_create: n_xmsa must be >= 1");
  if (desc->n_seqs < 0 || desc->n_sites < 0) return fail("lh_family_create: negative dimension");
  if ((int64_t)desc->n_seqs * desc->n_sites >= ((int64_t)1 << 31))
    return fail("lh_family_create: MSA larger than 2^31 bytes (K1 addresses it with 32-bit offsets)");
  lh_family* f = new lh_family();
  if (hipGetDevice(&f->device) != hipSuccess) {
    delete f;
    return fail("hipGetDevice failed");
  }
  {
    void* p = nullptr;
    if (arena_alloc(f, sizeof(int32_t), &p) || hipMemset(p, 0, sizeof(int32_t)) != hipSuccess) {
      lh_family_destroy(f);
      return fail("lh_family_create: device allocation failed");
    }
    f->err_flag = static_cast<int32_t*>(p);
  }
  lh::DevFamily& h = f->host;
  h.has_d = desc->has_d ? 1 : 0;
  h.n_seqs = desc->n_seqs;
  h.n_sites = desc->n_sites;
  h.n_xmsa = desc->n_xmsa;
  int rc = 0;
  const size_t C = desc->n_xmsa;
  std::vector<int32_t> ucol(C);  // caller's column -> u-column
  std::vector<int32_t> pat_of_site_all;  // alignment site -> K1 pattern (families with an alignment)
  if (desc->n_seqs > 0) {
    if (!desc->msa || !desc->xmsa_site || !desc->xmsa_naive_base) rc = fail("lh_family_create: null array in descriptor");
    const size_t N = desc->n_seqs, L = desc->n_sites;
    for (size_t i = 0; i < N * L && !rc; ++i)
      if (desc->msa[i] > 4) rc = fail("lh_family_create: msa value out of range");
    for (size_t c = 0; c < C && !rc; ++c)
      if (desc->xmsa_site[c] < 0 || desc->xmsa_site[c] >= desc->n_sites || desc->xmsa_naive_base[c] > 4)
        rc = fail("lh_family_create: xMSA column descriptor out of range");
    if (!rc) {
      // site patterns: identical alignment columns are pruned once (first-appearance order)
      std::map<std::string, int32_t> seen;
      std::vector<int32_t> pat_of_site(L);
      std::vector<size_t> first_site;
      std::string key(N, '\0');
      for (size_t j = 0; j < L; ++j) {
        for (size_t i = 0; i < N; ++i) key[i] = (char)desc->msa[i * L + j];
        auto it = seen.emplace(key, (int32_t)first_site.size());
        if (it.second) first_site.push_back(j);
        pat_of_site[j] = it.first->second;
      }
      // Order: patterns without an N, then patterns with some N, then (at most one) the all-N pattern.
      // The all-N column -- alignment padding -- has likelihood pi_b for naive base b whatever the tree,
      // i.e. emission 1 (sum of pi for naive base N): K1 never sees it (its site dimension is n_prune),
      // K2a writes the constant.
      // If no pattern mixes N with bases, K1 runs the instantiation without N handling.
      const size_t NP = first_site.size();
      std::vector<int> cls(NP, 0);  // 0 clean, 1 mixed, 2 all-N
      for (size_t p = 0; p < NP; ++p) {
        size_t n_n = 0;
        for (size_t i = 0; i < N; ++i) n_n += desc->msa[i * L + first_site[p]] == 4;
        cls[p] = n_n == 0 ? 0 : n_n == N ? 2 : 1;
      }
      std::vector<int32_t> order(NP), new_id(NP);
      for (size_t p = 0; p < NP; ++p) order[p] = (int32_t)p;
      std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cls[a] < cls[b]; });
      for (size_t q = 0; q < NP; ++q) new_id[order[q]] = (int32_t)q;
      for (size_t j = 0; j < L; ++j) pat_of_site[j] = new_id[pat_of_site[j]];
      {
        std::vector<size_t> fs(NP);
        for (size_t q = 0; q < NP; ++q) fs[q] = first_site[order[q]];
        first_site.swap(fs);
      }
      const bool has_all_n = NP > 0 && cls[order[NP - 1]] == 2;
      h.n_pat = (int32_t)NP;
      h.n_prune = (int32_t)(NP - (has_all_n ? 1 : 0));
      h.msa_mixed_n = 0;
      for (int c : cls) h.msa_mixed_n |= c == 1;
      std::vector<uint8_t> pmsa(N * std::max<size_t>(h.n_prune, 1));
      for (size_t i = 0; i < N; ++i)
        for (size_t p = 0; p < (size_t)h.n_prune; ++p) pmsa[i * h.n_prune + p] = desc->msa[i * L + first_site[p]];
      // u-columns: the (naive base, pattern) pairs, numbered by their place in K1's output planes --
      // base * n_prune + pattern, then the five bases of the all-N pattern -- so that K2a fills its emission vector
      // with one pass over the planes and no look-up (lh_device.h).  Pairs no xMSA column has keep u_base = 0xff.
      const int32_t NPr = h.n_prune;
      const size_t n_u = 5 * (size_t)NPr + 5;
      std::vector<int32_t> u_pat(n_u), col_of_ucol(n_u, -1);
      std::vector<uint8_t> u_base(n_u, 0xff);
      for (size_t u = 0; u < n_u; ++u) u_pat[u] = u < 5 * (size_t)NPr ? (int32_t)(u % NPr) : NPr;
      int32_t n_used = 0;
      for (size_t c = 0; c < C; ++c) {
        const int b = desc->xmsa_naive_base[c];
        const int32_t pat = pat_of_site[desc->xmsa_site[c]];
        const int32_t u = pat < NPr ? b * NPr + pat : 5 * NPr + b;
        ucol[c] = u;
        if (col_of_ucol[u] < 0) {
          col_of_ucol[u] = (int32_t)c;
          u_base[u] = (uint8_t)b;
          ++n_used;
        }
      }
      f->n_ucol_used = n_used;
      h.n_ucol = (int32_t)u_pat.size();
      pat_of_site_all = pat_of_site;
      rc = rc || upload(f, pmsa.data(), pmsa.size(), &h.msa);
      h.msa_planes = nullptr;
      if (h.n_prune > 0) {
        // two state bits per pattern, and for alignments that mix N with bases a third plane flagging N (state bits 0 there)
        const size_t np = h.n_prune, nb = (np + 127) / 128, nm = h.msa_mixed_n ? 3 : 2;
        std::vector<uint64_t> planes(N * nb * 2 * nm, 0);
        for (size_t i = 0; i < N; ++i)
          for (size_t b = 0; b < nb; ++b)
            for (size_t s2 = 0; s2 < 2; ++s2)
              for (size_t l = 0; l < 64; ++l) {
                const uint8_t st = pmsa[i * np + std::min(128 * b + 64 * s2 + l, np - 1)];
                uint64_t* m = &planes[((i * nb + b) * 2 + s2) * nm];
                if (st < 4) {
                  m[0] |= (uint64_t)(st & 1) << l;
                  m[1] |= (uint64_t)((st >> 1) & 1) << l;
                } else {
                  m[2] |= (uint64_t)1 << l;  // (reached only when msa_mixed_n: a clean alignment holds no 4)
                }
              }
        rc = rc || upload(f, planes.data(), planes.size(), &h.msa_planes);
      }
      rc = rc || upload(f, pat_of_site.data(), pat_of_site.size(), &h.site_pat);
      rc = rc || upload(f, u_pat.data(), u_pat.size(), &h.u_pat);
      rc = rc || upload(f, u_base.data(), u_base.size(), &h.u_base);
      rc = rc || upload(f, ucol.data(), C, &h.ucol_of_col);
      rc = rc || upload(f, col_of_ucol.data(), col_of_ucol.size(), &h.col_of_ucol);
    }
  } else {
    for (size_t c = 0; c < C; ++c) ucol[c] = (int32_t)c;
    h.n_pat = 0;
    h.n_prune = 0;
    h.msa_mixed_n = 0;
    h.n_ucol = (int32_t)C;
    f->n_ucol_used = (int32_t)C;
    rc = rc || upload<uint8_t>(f, nullptr, 0, &h.msa);
    h.msa_planes = nullptr;
    rc = rc || upload<int32_t>(f, nullptr, 0, &h.site_pat);
    rc = rc || upload<int32_t>(f, nullptr, 0, &h.u_pat);
    rc = rc || upload<uint8_t>(f, nullptr, 0, &h.u_base);
    rc = rc || upload(f, ucol.data(), C, &h.ucol_of_col);
    rc = rc || upload(f, ucol.data(), C, &h.col_of_ucol);
  }
  h.idx_byte_offsets = ((int64_t)h.n_ucol + 1) * 8 <= 0xffff ? 1 : 0;
  const int32_t* seg_site = (desc->n_seqs > 0 && !rc) ? desc->xmsa_site : nullptr;  // alignment site of a column
  rc = rc || upload_segments(f, desc->vpadding, desc->n_xmsa, ucol, h.n_ucol, seg_site, &h.vpadding);
  rc = rc || upload_segments(f, desc->vgerm, desc->n_xmsa, ucol, h.n_ucol, seg_site, &h.vgerm);
  rc = rc || upload_segments(f, desc->jgerm, desc->n_xmsa, ucol, h.n_ucol, seg_site, &h.jgerm);
  rc = rc || upload_segments(f, desc->jpadding, desc->n_xmsa, ucol, h.n_ucol, seg_site, &h.jpadding);
  const size_t nV = desc->vgerm.n_genes, nJ = desc->jgerm.n_genes;
  if (!rc && (desc->vpadding.n_genes != (int)nV || desc->jpadding.n_genes != (int)nJ))
    rc = fail("lh_family_create: padding/germline gene counts differ");
  rc = rc || upload(f, desc->vgerm_gene_prob, nV, &h.vgerm_gene_prob);
  rc = rc || upload(f, desc->vpadding_transition, nV, &h.vpadding_transition);
  rc = rc || upload(f, desc->vgerm_trans_prod, nV, &h.vgerm_trans_prod);
  rc = rc || upload(f, desc->jpadding_transition, nJ, &h.jpadding_transition);
  std::vector<int32_t> remap(desc->n_xmsa, 0);  // first "used" flags, then positions in the compact junction list
  rc = rc || collect_junction_cols(desc->vd, desc->n_xmsa, &remap);
  if (h.has_d) rc = rc || collect_junction_cols(desc->dj, desc->n_xmsa, &remap);
  if (!rc) {
    std::vector<int32_t> jpos(h.n_ucol, -1), jcols;  // junction list = the u-columns the junction rows touch
    std::vector<char> used_u(h.n_ucol, 0);
    for (int c = 0; c < desc->n_xmsa; ++c)
      if (remap[c]) used_u[ucol[c]] = 1;
    for (int u = 0; u < h.n_ucol; ++u)
      if (used_u[u]) {
        jpos[u] = (int32_t)jcols.size();
        jcols.push_back(u);
      }
    for (int c = 0; c < desc->n_xmsa; ++c) remap[c] = remap[c] ? jpos[ucol[c]] : -1;
    h.n_jcols = (int32_t)jcols.size();
    rc = upload(f, jcols.data(), jcols.size(), &h.jcols);
  }
  // the chunk counts of K2b's templates (launch_forward): GA for the V side, GB for every D / J side
  auto round_to = [](int c, std::initializer_list<int> steps) {
    for (int s : steps)
      if (c <= s) return s;
    return c;
  };
  const int ga_t = round_to(((int)nV + 63) / 64, {1, 2, 4, 8, 16});
  const int gb_t = round_to((std::max((int)desc->dgerm.n_genes * (h.has_d ? 1 : 0), (int)nJ) + 63) / 64, {1, 2, 4});
  rc = rc || upload_junction(f, desc->vd, remap, h.n_jcols, seg_site, pat_of_site_all, ga_t, gb_t, &h.vd);
  if (!rc && desc->vd.n_left != (int)nV) rc = fail("lh_family_create: vd.n_left != number of V genes");
  if (h.has_d) {
    rc = rc || upload_segments(f, desc->dgerm, desc->n_xmsa, ucol, h.n_ucol, seg_site, &h.dgerm);
    rc = rc || upload_junction(f, desc->dj, remap, h.n_jcols, seg_site, pat_of_site_all, gb_t, gb_t, &h.dj);
    if (!rc && (desc->vd.n_right != desc->dgerm.n_genes || desc->dj.n_left != desc->dgerm.n_genes ||
                desc->dj.n_right != (int)nJ))
      rc = fail("lh_family_create: junction gene counts do not match the germline regions");
  } else {
    lh_segments empty{0, nullptr, nullptr};
    rc = rc || upload_segments(f, empty, desc->n_xmsa, ucol, h.n_ucol, seg_site, &h.dgerm);
    memset(&h.dj, 0, sizeof(h.dj));
    if (!rc && desc->vd.n_right != (int)nJ) rc = fail("lh_family_create: vd.n_right != number of J genes");
  }
  if (!rc) {
    h.max_genes = std::max({(int)nV, (int)nJ, h.dgerm.n_genes, 1});
    h.forward_size = (int64_t)nV + (int64_t)h.vd.n_rows * (h.vd.n_left + 5 * (int64_t)h.vd.n_right) + nJ;
    h.scaler_size = 1 + h.vd.n_rows + 1;
    if (h.has_d) {
      h.forward_size += h.dgerm.n_genes + (int64_t)h.dj.n_rows * (h.dj.n_left + 5 * (int64_t)h.dj.n_right);
      h.scaler_size += h.dj.n_rows + 1;
    }
    h.gem_size = 2 * (int64_t)nV + h.dgerm.n_genes + 2 * (int64_t)nJ;
    if (lh::forward_lds_bytes(h) > 160 * 1024)
      rc = fail("lh_family_create: family too large for the forward kernels' LDS working set");
    else if (nV > 1024 || h.dgerm.n_genes > 256 || nJ > 256)
      rc = fail("lh_family_create: more than 1024 V genes or 256 D/J genes");
  }
  if (!rc) {
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(lh::DevFamily)) != hipSuccess ||
        hipMemcpy(p, &h, sizeof(lh::DevFamily), hipMemcpyHostToDevice) != hipSuccess)
      rc = fail("lh_family_create: device allocation failed");
    else {
      f->dev = static_cast<lh::DevFamily*>(p);
      f->allocs.push_back(p);
    }
  }
  if (rc) {
    std::string keep = g_error;
    lh_family_destroy(f);
    g_error = keep;
    return 1;
  }
  *out = f;
  return 0;
}

void lh_family_destroy(lh_family* f) {
  if (!f) return;
  DeviceGuard guard(f);
  for (void* p : f->allocs) (void)hipFree(p);
  Workspace& w = f->ws;
  void* bufs[] = {w.rates,    w.eig,       w.site_lik, w.site_scal, w.prune.scratch, w.prune.wops, w.prune.wlen,
                  w.prune.tabs, w.prune.hdr, f->fws.gem, f->fws.jem,  f->fws.gcnt,     f->fws.jrs,   f->fws.dxf,
                  f->fws.dxc};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (void* p : f->st.ptr)
    if (p) (void)hipFree(p);
  if (f->asr.clv) (void)hipFree(f->asr.clv);
  if (f->asr.choice) (void)hipFree(f->asr.choice);
  if (f->asr.desc) (void)hipFree(f->asr.desc);
  for (void* p : f->asr.ptr)
    if (p) (void)hipFree(p);
  for (auto& ev : f->asr.events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  for (EventSet& es : f->events)
    for (hipEvent_t e : es.e) (void)hipEventDestroy(e);
  for (void* p : f->smp.ptr)
    if (p) (void)hipFree(p);
  if (f->smp.pinned) (void)hipHostFree(f->smp.pinned);
  for (void* p : f->pipe.pinned)
    if (p) (void)hipHostFree(p);
  for (hipEvent_t e : f->pipe.staged)
    if (e) (void)hipEventDestroy(e);
  if (f->pipe.copy) (void)hipStreamDestroy(f->pipe.copy);
  if (f->pipe.comp) (void)hipStreamDestroy(f->pipe.comp);
  delete f;
}

int64_t lh_forward_size(const lh_family* f) { return f ? f->host.forward_size : 0; }

const char* lh_family_prune_form(const lh_family* f) { return f ? f->k1_form.c_str() : ""; }

int lh_family_consensus_sets(const lh_family* f) {
  if (!f) return 0;
  const lh::DevFamily& h = f->host;
  return (h.vpadding.cons_sites > 0) | (h.vgerm.cons_sites > 0) << 1 | (h.dgerm.cons_sites > 0) << 2 |
         (h.jgerm.cons_sites > 0) << 3 | (h.jpadding.cons_sites > 0) << 4;
}

int lh_family_info(const lh_family* f, int32_t* n_patterns, int32_t* n_unique_columns) {
  if (!f) return fail("lh_family_info: null family");
  if (n_patterns) *n_patterns = f->host.n_prune;  // the all-N padding pattern, if any, costs nothing
  if (n_unique_columns) *n_unique_columns = f->n_ucol_used;
  return 0;
}
int64_t lh_scaler_size(const lh_family* f) { return f ? f->host.scaler_size : 0; }

int lh_schedule_tree(int32_t T, const int32_t* children, int32_t root, int32_t* ops, int32_t* max_depth) {
  if (T < 3) return fail("lh_schedule_tree: need at least 3 tips (naive + 2 sequences)");
  const int n_nodes = 2 * T - 2, I = T - 2;
  if (!children || !ops) return fail("lh_schedule_tree: null argument");
  if (root < T || root >= n_nodes) return fail("lh_schedule_tree: root must be an inner node");
  // validate: every node except root and tip 0 (naive) appears exactly once as a child
  std::vector<int> seen(n_nodes, 0);
  for (int i = 0; i < 2 * I; ++i) {
    const int c = children[i];
    if (c < 1 || c >= n_nodes) return fail("lh_schedule_tree: child id out of range");
    if (seen[c]++) return fail("lh_schedule_tree: node appears twice as a child");
  }
  if (seen[root]) return fail("lh_schedule_tree: root appears as a child");
  for (int v = 1; v < n_nodes; ++v)
    if (v != root && !seen[v]) return fail("lh_schedule_tree: node is not attached to the tree");
  // subtree tip counts + acyclicity (iterative post-order from root)
  std::vector<int> size(n_nodes, 0), order;
  order.reserve(I);
  {
    std::vector<int> stack{root};
    std::vector<char> visited(n_nodes, 0);
    while (!stack.empty()) {
      const int v = stack.back();
      stack.pop_back();
      if (visited[v]) return fail("lh_schedule_tree: cycle in children array");
      visited[v] = 1;
      if (v >= T) {
        order.push_back(v);
        stack.push_back(children[2 * (v - T)]);
        stack.push_back(children[2 * (v - T) + 1]);
      }
    }
    if ((int)order.size() != I) return fail("lh_schedule_tree: tree does not span all inner nodes");
    for (int k = I - 1; k >= 0; --k) {
      const int v = order[k];
      const int a = children[2 * (v - T)], b = children[2 * (v - T) + 1];
      size[v] = (a < T ? 1 : size[a]) + (b < T ? 1 : size[b]);
    }
  }
  // emit ops: explicit stack of (node, phase)
  struct Frame {
    int v, phase, first, second;
  };
  std::vector<Frame> fs;
  fs.push_back({root, 0, 0, 0});
  int n_out = 0, depth = 0, maxd = 0;
  int n_inner_mats = 0;       // inner-branch P-matrices needed by the ops written so far (K1 packs its prologue by it)
  bool acc_live = false;      // accumulator holds a result that a later op still needs
  bool push_pending = false;  // the next cherry must push the accumulator first
  while (!fs.empty()) {
    Frame& fr = fs.back();
    const int v = fr.v;
    const int a = children[2 * (v - T)], b = children[2 * (v - T) + 1];
    const bool ta = a < T, tb = b < T;
    int32_t* op = ops + 4 * (size_t)n_out;
    if (ta && tb) {
      op[0] = lh::OP_CHERRY;
      op[1] = a;
      op[2] = b;
      op[3] = 0;
      if (push_pending) {
        op[0] |= lh::OP_PUSH_FLAG;
        op[3] = depth++;
        maxd = std::max(maxd, depth);
        push_pending = false;
      }
      ++n_out;
      acc_live = true;
      fs.pop_back();
    } else if (ta != tb) {
      const int tip = ta ? a : b, inner = ta ? b : a;
      if (fr.phase == 0) {
        fr.phase = 1;
        fs.push_back({inner, 0, 0, 0});
      } else {
        op[0] = lh::OP_TIP_ACC | (n_inner_mats << lh::OP_RANK_SHIFT);
        n_inner_mats += 1;
        op[1] = tip;
        op[2] = inner;
        op[3] = 0;
        ++n_out;
        fs.pop_back();
      }
    } else {
      if (fr.phase == 0) {
        fr.first = size[a] >= size[b] ? a : b;  // larger subtree first bounds the stack by log2(T)
        fr.second = size[a] >= size[b] ? b : a;
        fr.phase = 1;
        const int first = fr.first;
        fs.push_back({first, 0, 0, 0});
      } else if (fr.phase == 1) {
        fr.phase = 2;
        push_pending = true;  // first op of the second subtree is a cherry: it pushes `first`
        const int second = fr.second;
        fs.push_back({second, 0, 0, 0});
      } else {
        op[0] = lh::OP_POP_ACC | (n_inner_mats << lh::OP_RANK_SHIFT);
        n_inner_mats += 2;
        op[1] = fr.first;
        op[2] = fr.second;
        op[3] = --depth;
        ++n_out;
        fs.pop_back();
      }
    }
  }
  (void)acc_live;
  if (n_out != I || depth != 0) return fail("lh_schedule_tree: internal scheduling error");
  if (maxd > 16) return fail("lh_schedule_tree: tree needs more than 16 stack slots");
  if (max_depth) *max_depth = maxd;
  return 0;
}

int lh_family_set_extended_range(lh_family* f, int enable) {
  if (!f) return fail("null family");
  f->extended = enable != 0;
  return 0;
}

static int upload_sampler_junction(lh_family* f, const lh_sampler_junction& j, lh::DevSampleJunction* d) {
  const size_t W = j.n_rows, nL = j.n_left, nR = j.n_right;
  if (j.n_rows < 1 || j.n_left < 1 || j.n_right < 1 || j.n_states < 2) return fail("lh_family_set_sampler: bad dimensions");
  if (!j.left_rows || !j.left_dense || !j.left_lo || !j.left_trans || !j.enter_lo || !j.right_dense || !j.right_first ||
      !j.gene_prob || !j.nti_landing_in || !j.nti_transition || !j.nti_landing_out || !j.landing_in || !j.right_trans ||
      !j.exit_nlo || !j.exit_trans || !j.exit_li || !j.prod)
    return fail("lh_family_set_sampler: null array in descriptor");
  // the genes must form two blocks of the dense state vector, each in gene order, with consistent lengths
  int lo_l = INT32_MAX, hi_l = -1, lo_r = INT32_MAX, hi_r = -1;
  for (size_t l = 0; l < nL; ++l) {
    if (j.left_rows[l] < 0 || j.left_rows[l] > j.n_rows || j.left_dense[l] < 0) return fail("lh_family_set_sampler: bad left gene");
    if (l > 0 && j.left_dense[l] < j.left_dense[l - 1] + j.left_rows[l - 1]) return fail("lh_family_set_sampler: left genes out of order");
    lo_l = std::min(lo_l, j.left_dense[l]);
    hi_l = std::max(hi_l, j.left_dense[l] + j.left_rows[l]);
  }
  for (size_t r = 0; r < nR; ++r) {
    if (j.right_first[r] < 0 || j.right_first[r] > j.n_rows || j.right_dense[r] < 0) return fail("lh_family_set_sampler: bad right gene");
    const int len = 4 + (j.n_rows - j.right_first[r]);
    if (r > 0 && j.right_dense[r] < j.right_dense[r - 1] + 4) return fail("lh_family_set_sampler: right genes out of order");
    lo_r = std::min(lo_r, j.right_dense[r]);
    hi_r = std::max(hi_r, j.right_dense[r] + len);
  }
  if (!(hi_r <= lo_l || hi_l <= lo_r) || std::max(hi_l, hi_r) > j.n_states)
    return fail("lh_family_set_sampler: the junction's genes do not form two blocks of its state vector");
  d->n_rows = j.n_rows;
  d->n_left = j.n_left;
  d->n_right = j.n_right;
  d->n_states = j.n_states;
  d->right_first_block = hi_r <= lo_l ? 1 : 0;
  std::vector<int32_t> cls((size_t)j.n_states, 0);
  for (size_t l = 0; l < nL; ++l)
    for (int i = 0; i < j.left_rows[l]; ++i) cls[(size_t)j.left_dense[l] + i] = 0 | (int32_t)(l << 4);
  for (size_t r = 0; r < nR; ++r) {
    for (int b = 0; b < 4; ++b) cls[(size_t)j.right_dense[r] + b] = 1 | (b << 2) | (int32_t)(r << 4);
    for (int i = j.right_first[r]; i < j.n_rows; ++i) cls[(size_t)j.right_dense[r] + 4 + (i - j.right_first[r])] = 2 | (int32_t)(r << 4);
  }
  if (upload(f, cls.data(), cls.size(), &d->state_class)) return 1;
  if (upload(f, j.left_rows, nL, &d->left_rows) || upload(f, j.left_dense, nL, &d->left_dense) ||
      upload(f, j.left_lo, W * nL, &d->left_lo) || upload(f, j.left_trans, W * nL, &d->left_trans) ||
      upload(f, j.enter_lo, nL, &d->enter_lo) || upload(f, j.right_dense, nR, &d->right_dense) ||
      upload(f, j.right_first, nR, &d->right_first) || upload(f, j.gene_prob, nR, &d->gp) ||
      upload(f, j.nti_landing_in, nR * 4, &d->nli) || upload(f, j.nti_transition, nR * 16, &d->ntt) ||
      upload(f, j.nti_landing_out, W * nR * 4, &d->nlo) || upload(f, j.landing_in, W * nR, &d->li) ||
      upload(f, j.right_trans, W * nR, &d->rtrans) || upload(f, j.exit_nlo, nR * 4, &d->exit_nlo) ||
      upload(f, j.exit_trans, nR, &d->exit_trans) || upload(f, j.exit_li, nR, &d->exit_li) ||
      upload(f, j.prod, nR, &d->prod))
    return 1;
  return 0;
}

int lh_family_set_sampler(lh_family* f, const lh_sampler_desc* desc) {
  if (!f || !desc) return fail("lh_family_set_sampler: null argument");
  DeviceGuard guard(f);
  const lh::DevFamily& h = f->host;
  lh::DevSampler s{};
  s.has_d = h.has_d;
  s.n_v = h.vgerm.n_genes;
  s.n_d = h.dgerm.n_genes;
  s.n_j = h.jgerm.n_genes;
  if (desc->vd.n_rows != h.vd.n_rows || desc->vd.n_left != h.vd.n_left || desc->vd.n_right != h.vd.n_right)
    return fail("lh_family_set_sampler: V-D junction dimensions differ from the family's");
  if (upload_sampler_junction(f, desc->vd, &s.vd)) return 1;
  int draws = (s.n_j >= 2) + (s.n_v >= 2) + s.vd.n_rows;
  s.states_per_sample = 2 + s.vd.n_rows;
  if (h.has_d) {
    if (desc->dj.n_rows != h.dj.n_rows || desc->dj.n_left != h.dj.n_left || desc->dj.n_right != h.dj.n_right)
      return fail("lh_family_set_sampler: D-J junction dimensions differ from the family's");
    if (upload_sampler_junction(f, desc->dj, &s.dj)) return 1;
    draws += (s.n_d >= 2) + s.dj.n_rows;
    s.states_per_sample += 1 + s.dj.n_rows;
  }
  s.words_per_sample = 2 * draws;
  f->sampler = s;
  if (upload(f, &s, 1, &f->sampler_dev)) return 1;
  f->have_sampler = true;
  return 0;
}

int32_t lh_sample_words(const lh_family* f) { return f && f->have_sampler ? f->sampler.words_per_sample : 0; }
int32_t lh_sample_states(const lh_family* f) { return f && f->have_sampler ? f->sampler.states_per_sample : 0; }

// Reads (and clears) the handle's asynchronous error word after synchronising its device.
static int check_async_error(lh_family* f, const char* who) {
  int32_t flag = 0;
  LH_HIP(hipDeviceSynchronize());
  LH_HIP(hipMemcpy(&flag, f->err_flag, sizeof(flag), hipMemcpyDeviceToHost));
  if (flag) {
    LH_HIP(hipMemset(f->err_flag, 0, sizeof(flag)));
    return fail(std::string(who) + ": malformed schedule op (use lh_schedule_tree); the affected samples' results are NaN");
  }
  return 0;
}

int lh_family_status(lh_family* f) {
  if (!f) return fail("lh_family_status: null family");
  DeviceGuard guard(f);
  return check_async_error(f, "lh_family_status");
}

int lh_profile_enable(lh_family* f, int enable) {
  if (!f) return fail("null family");
  f->profile = enable != 0;
  return 0;
}

int lh_profile_read(lh_family* f, double* ms_model, double* ms_prune, double* ms_forward, int64_t* n_launches) {
  if (!f) return fail("null family");
  DeviceGuard guard(f);
  for (EventSet& es : f->events) {
    LH_HIP(hipEventSynchronize(es.e[3]));
    for (int k = 0; k < 3; ++k) {
      float ms = 0;
      LH_HIP(hipEventElapsedTime(&ms, es.e[k], es.e[k + 1]));
      f->ms[k] += ms;
    }
    for (hipEvent_t e : es.e) (void)hipEventDestroy(e);
    ++f->launches;
  }
  f->events.clear();
  if (ms_model) *ms_model = f->ms[0];
  if (ms_prune) *ms_prune = f->ms[1];
  if (ms_forward) *ms_forward = f->ms[2];
  if (n_launches) *n_launches = f->launches;
  f->ms[0] = f->ms[1] = f->ms[2] = 0;
  f->launches = 0;
  return 0;
}

int lh_asr_profile_read(lh_family* f, double* ms_sampling, int64_t* n_launches) {
  if (!f) return fail("null family");
  DeviceGuard guard(f);
  AsrWs& a = f->asr;
  for (auto& ev : a.events) {
    LH_HIP(hipEventSynchronize(ev.second));
    float ms = 0;
    LH_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
    a.ms += ms;
    ++a.launches;
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  a.events.clear();
  if (ms_sampling) *ms_sampling = a.ms;
  if (n_launches) *n_launches = a.launches;
  a.ms = 0;
  a.launches = 0;
  return 0;
}

int lh_eval_batch_device(lh_family* f, int32_t n, int32_t T, int32_t max_depth, const int32_t* ops,
                         const double* brlen, const double* er, const double* pi, const double* alpha,
                         int32_t R, double* loglik, const lh_eval_outputs* outs, void* hip_stream) {
  if (!f) return fail("lh_eval_batch: null family");
  DeviceGuard guard(f);
  if (n < 0) return fail("lh_eval_batch: negative batch size");
  if (n == 0) return 0;
  if (f->host.n_seqs < 1) return fail("lh_eval_batch: family was created without an MSA (forward-only)");
  if (T != f->host.n_seqs + 1) return fail("lh_eval_batch: n_tips must equal n_seqs + 1 (naive)");
  if (T < 3) return fail("lh_eval_batch: need at least 3 tips");
  if (R < 1 || R > 64) return fail("lh_eval_batch: num_rates out of range");
  if (max_depth < 0 || max_depth > 16) return fail("lh_eval_batch: max_depth out of range");
  // (launch_prune checks the LDS need of the form it takes -- trees this large with a stack deeper than four slots do not fit)
  if ((size_t)T * 128 > 160 * 1024) return fail("lh_eval_batch: too many tips for the LDS tip table");
  if (!ops || !brlen || !er || !pi || !alpha || !loglik) return fail("lh_eval_batch: null array");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  // launch groups of at most kChunk samples and at most ~16 GB of per-sample workspace
  const size_t per_sample = k1_bytes_per_sample(f, T, R) + sizeof(double) * R * 6 * (size_t)std::max(f->host.n_prune, 1);
  const int by_memory = (int)std::max<size_t>(1024, ((size_t)16 << 30) / per_sample);
  const int chunk = std::min<int>(n, std::min(kChunk, by_memory));
  if (ensure_workspace(f, chunk, R, T)) return 1;
  Workspace& w = f->ws;
  const size_t nodes = 2 * (size_t)T - 2, n_ops = (size_t)T - 2, C = f->host.n_xmsa;
  for (int off = 0; off < n; off += chunk) {
    const int m = std::min(chunk, n - off);
    EventSet es;
    if (f->profile) {
      for (hipEvent_t& e : es.e) LH_HIP(hipEventCreate(&e));
      LH_HIP(hipEventRecord(es.e[0], stream));
    }
    double* rates = (outs && outs->rates) ? outs->rates + (size_t)off * R : w.rates;
    double* em_out = (outs && outs->xmsa_emission) ? outs->xmsa_emission + (size_t)off * C : nullptr;
    lh::launch_model_setup(m, R, er + (size_t)off * 6, pi + (size_t)off * 4, alpha + off, rates, w.eig,
                           stream);
    if (f->profile) LH_HIP(hipEventRecord(es.e[1], stream));
    const int planes = lh::launch_prune(f->host, m, R, T, max_depth, ops + (size_t)off * n_ops * 4,
                                        brlen + (size_t)off * nodes, rates, w.eig, w.prune, pi + (size_t)off * 4,
                                        w.site_lik, w.site_scal, stream);
    if (planes < 0) return fail(std::string("lh_eval_batch: ") + lh::prune_last_error());
    f->k1_form = lh::prune_last_form();
    if (f->profile) LH_HIP(hipEventRecord(es.e[2], stream));
    if (run_forward(f, m, planes, w.site_lik, w.site_scal, pi + (size_t)off * 4, nullptr, em_out, loglik + off, outs, off,
                    stream))
      return 1;
    if (f->profile) {
      LH_HIP(hipEventRecord(es.e[3], stream));
      f->events.push_back(es);
    }
    LH_HIP(hipGetLastError());
  }
  return 0;
}

// One schedule op as lh_schedule_tree writes it (a malformed op would index out of bounds on the device).
static bool valid_op(const int32_t* op, int T, int nodes, int max_depth) {
  const int kind = op[0] & 15;
  const bool push = op[0] & lh::OP_PUSH_FLAG;
  const int rank = op[0] >> lh::OP_RANK_SHIFT;  // bits 5-7 unused
  bool ok = (op[0] & 0xe0) == 0 && op[0] >= 0 && kind <= 2 && rank <= T - 3;
  if (kind == lh::OP_CHERRY) ok = ok && op[1] >= 1 && op[1] < T && op[2] >= 1 && op[2] < T;
  if (kind == lh::OP_TIP_ACC) ok = ok && !push && op[1] >= 1 && op[1] < T && op[2] >= T && op[2] < nodes;
  if (kind == lh::OP_POP_ACC) ok = ok && !push && op[1] >= T && op[1] < nodes && op[2] >= T && op[2] < nodes;
  if (push || kind == lh::OP_POP_ACC) ok = ok && op[3] >= 0 && op[3] < max_depth;
  return ok;
}

// One sample's schedule: every op well-formed, and the rank field of each op the running count of inner-branch matrices
// lh_schedule_tree leaves there (a tip-accumulate op takes one, a pop-accumulate op two, a cherry none; T - 3 in all) --
// the fused K1 prologue files its matrices by that number.  (The kernels check the same thing again on the device:
// schedules may also arrive in device memory.)
// And the stack discipline: a push goes to slot = the number of pending siblings, a pop takes the last one, none is left at
// the end (a schedule that breaks it within the slot range computes a finite, wrong likelihood; the device checks repeat this).
static bool valid_schedule(const int32_t* ops, int T, int nodes, int max_depth) {
  int count = 0, depth = 0;
  for (int k = 0; k < T - 2; ++k) {
    const int32_t* op = ops + (size_t)k * 4;
    if (!valid_op(op, T, nodes, max_depth)) return false;
    const int kind = op[0] & 15, rank = op[0] >> lh::OP_RANK_SHIFT;
    if (kind == lh::OP_CHERRY) {
      if (rank != 0 && rank != count) return false;
      if (((op[0] & lh::OP_PUSH_FLAG) != 0) != (k != 0)) return false;  // the first op has no accumulator to set aside, every later cherry does
      if (k != 0 && op[3] != depth++) return false;
    } else {
      if (rank != count) return false;
      count += kind == lh::OP_POP_ACC ? 2 : 1;
      if (kind == lh::OP_POP_ACC && op[3] != --depth) return false;
    }
  }
  return count == T - 3 && depth == 0;
}

// All schedules of a batch; a few threads when the batch is large (0.19 us per op on one core: 0.4 ms per 2048 samples of
// a 101-tip tree, as much as the device then needs for K0-K2).
static bool valid_schedules(const int32_t* ops, size_t n, int T, int nodes, int max_depth) {
  const size_t n_ops = (size_t)T - 2;
  const int nw = (int)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), (size_t)8, n * n_ops / 65536}));
  std::atomic<bool> bad{false};
  auto work = [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi && !bad; ++i)
      if (!valid_schedule(ops + i * n_ops * 4, T, nodes, max_depth)) bad = true;
  };
  if (nw == 1) {
    work(0, n);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nw; ++t) pool.emplace_back(work, n * t / nw, n * (t + 1) / nw);
    for (std::thread& th : pool) th.join();
  }
  return !bad;
}

// Host pointers in, host pointers out.  The batch moves in sub-chunks through two pinned staging slots:
// while the kernels of one sub-chunk run on the compute stream, a few host threads validate the next
// sub-chunk's schedules and gather its inputs into the other slot, and the copy stream ships it.
int lh_eval_batch(lh_family* f, int32_t n, int32_t T, int32_t max_depth, const int32_t* ops,
                  const double* brlen, const double* er, const double* pi, const double* alpha, int32_t R,
                  double* loglik, const lh_eval_outputs* outs) {
  if (!f) return fail("lh_eval_batch: null family");
  DeviceGuard guard(f);
  if (n <= 0) return n == 0 ? 0 : fail("lh_eval_batch: negative batch size");
  if (T < 3) return fail("lh_eval_batch: need at least 3 tips");
  if (!ops || !brlen || !er || !pi || !alpha || !loglik) return fail("lh_eval_batch: null array");
  const size_t nodes = 2 * (size_t)T - 2, n_ops = (size_t)T - 2;
  void *d_ops, *d_brlen, *d_er, *d_pi, *d_alpha, *d_ll;
  if (stage(f, 0, sizeof(int32_t) * 4 * n_ops * n, &d_ops)) return 1;
  if (stage(f, 1, sizeof(double) * nodes * n, &d_brlen)) return 1;
  if (stage(f, 2, sizeof(double) * 6 * n, &d_er)) return 1;
  if (stage(f, 3, sizeof(double) * 4 * n, &d_pi)) return 1;
  if (stage(f, 4, sizeof(double) * n, &d_alpha)) return 1;
  if (stage(f, 5, sizeof(double) * n, &d_ll)) return 1;
  lh_eval_outputs d_outs{nullptr, nullptr, nullptr, nullptr};
  const size_t C = f->host.n_xmsa, FS = f->host.forward_size, SS = f->host.scaler_size;
  if (outs) {
    if (outs->rates && stage(f, 6, sizeof(double) * R * n, (void**)&d_outs.rates)) return 1;
    if (outs->xmsa_emission && stage(f, 7, sizeof(double) * C * n, (void**)&d_outs.xmsa_emission)) return 1;
    if (outs->forward && stage(f, 8, sizeof(double) * FS * n, (void**)&d_outs.forward)) return 1;
    if (outs->scaler_counts && stage(f, 9, sizeof(int32_t) * SS * n, (void**)&d_outs.scaler_counts)) return 1;
  }

  // per-sample bytes of the five input arrays, in the order they sit in a staging slot
  const size_t bytes[5] = {sizeof(int32_t) * 4 * n_ops, sizeof(double) * nodes, sizeof(double) * 6,
                           sizeof(double) * 4, sizeof(double)};
  const char* src[5] = {(const char*)ops, (const char*)brlen, (const char*)er, (const char*)pi, (const char*)alpha};
  char* dst[5] = {(char*)d_ops, (char*)d_brlen, (char*)d_er, (char*)d_pi, (char*)d_alpha};
  size_t per_sample = 0;
  for (size_t b : bytes) per_sample += b;
  const int kSub = lh::debug_options().host_sub;  // (default 12 288: whole rounds of all kernels for configs[2]-like shapes)
  const int sub = std::min<int>(n, kSub);
  HostPipe& hp = f->pipe;
  if (!hp.copy) {
    LH_HIP(hipStreamCreateWithFlags(&hp.copy, hipStreamNonBlocking));
    LH_HIP(hipStreamCreateWithFlags(&hp.comp, hipStreamNonBlocking));
    for (hipEvent_t& e : hp.staged) LH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  if (hp.cap < per_sample * sub) {
    for (void*& p : hp.pinned) {
      if (p) LH_HIP(hipHostFree(p));
      p = nullptr;
    }
    hp.cap = 0;
    for (void*& p : hp.pinned) LH_HIP(hipHostMalloc(&p, per_sample * sub, hipHostMallocDefault));
    hp.cap = per_sample * sub;
  }
  LH_HIP(hipDeviceSynchronize());  // the staging buffers may still be in use by an earlier device-pointer call

  const int n_workers = (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 8u));
  int rc = 0;
  int slot = 0;
  bool slot_used[2] = {false, false};
  for (int off = 0; off < n && !rc; off += sub, slot ^= 1) {
    const int m = std::min(sub, n - off);
    if (slot_used[slot]) LH_HIP(hipEventSynchronize(hp.staged[slot]));  // its last copy has left the slot
    char* base = static_cast<char*>(hp.pinned[slot]);
    char* part[5];
    {
      size_t o = 0;
      for (int a = 0; a < 5; ++a) {
        part[a] = base + o;
        o += bytes[a] * m;
      }
    }
    std::atomic<bool> bad{false};
    const int nw = std::max(1, std::min(n_workers, m / 256));
    auto in_threads = [&](auto&& fn) {
      if (nw == 1) {
        fn(0, m);
      } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nw; ++t)
          pool.emplace_back(fn, (int)((long long)m * t / nw), (int)((long long)m * (t + 1) / nw));
        for (std::thread& th : pool) th.join();
      }
    };
    // the sub-chunk's inputs into the pinned slot ...
    in_threads([&](int lo, int hi) {
      for (int a = 0; a < 5; ++a)
        memcpy(part[a] + bytes[a] * lo, src[a] + bytes[a] * ((size_t)off + lo), bytes[a] * (size_t)(hi - lo));
    });
    for (int a = 0; a < 5; ++a)
      if (hipMemcpyAsync(dst[a] + bytes[a] * off, part[a], bytes[a] * m, hipMemcpyHostToDevice, hp.copy) !=
          hipSuccess)
        rc = fail("lh_eval_batch: hipMemcpyAsync failed");
    if (rc) break;
    LH_HIP(hipEventRecord(hp.staged[slot], hp.copy));
    slot_used[slot] = true;
    LH_HIP(hipStreamWaitEvent(hp.comp, hp.staged[slot], 0));
    lh_eval_outputs o{d_outs.rates ? d_outs.rates + (size_t)off * R : nullptr,
                      d_outs.xmsa_emission ? d_outs.xmsa_emission + (size_t)off * C : nullptr,
                      d_outs.forward ? d_outs.forward + (size_t)off * FS : nullptr,
                      d_outs.scaler_counts ? d_outs.scaler_counts + (size_t)off * SS : nullptr};
    rc = lh_eval_batch_device(f, m, T, max_depth, (const int32_t*)(dst[0] + bytes[0] * off),
                              (const double*)(dst[1] + bytes[1] * off), (const double*)(dst[2] + bytes[2] * off),
                              (const double*)(dst[3] + bytes[3] * off), (const double*)(dst[4] + bytes[4] * off), R,
                              (double*)d_ll + off, &o, hp.comp);
    if (rc) break;
    // ... and its schedules checked on the host while the device works on them (the kernels make the same checks: a
    // malformed op is a NaN and an error code there, never an out-of-bounds access; a refused batch hands nothing back)
    in_threads([&](int lo, int hi) {
      for (int i = lo; i < hi && !bad; ++i)
        if (!valid_schedule(ops + ((size_t)off + i) * n_ops * 4, T, (int)nodes, max_depth)) bad = true;
    });
    if (bad) {
      (void)hipDeviceSynchronize();
      (void)check_async_error(f, "lh_eval_batch");  // (the device found it too: one report is enough)
      rc = fail("lh_eval_batch: malformed schedule op (use lh_schedule_tree)");
    }
  }
  if (hipDeviceSynchronize() != hipSuccess && !rc) rc = fail("lh_eval_batch: device synchronisation failed");
  if (!rc && check_async_error(f, "lh_eval_batch")) rc = 1;  // K0c's verdict on the schedules as the device saw them
  if (rc) return 1;
  LH_HIP(hipMemcpy(loglik, d_ll, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (outs) {
    if (outs->rates) LH_HIP(hipMemcpy(outs->rates, d_outs.rates, sizeof(double) * R * n, hipMemcpyDeviceToHost));
    if (outs->xmsa_emission)
      LH_HIP(hipMemcpy(outs->xmsa_emission, d_outs.xmsa_emission, sizeof(double) * C * n, hipMemcpyDeviceToHost));
    if (outs->forward)
      LH_HIP(hipMemcpy(outs->forward, d_outs.forward, sizeof(double) * FS * n, hipMemcpyDeviceToHost));
    if (outs->scaler_counts)
      LH_HIP(hipMemcpy(outs->scaler_counts, d_outs.scaler_counts, sizeof(int32_t) * SS * n,
                       hipMemcpyDeviceToHost));
  }
  return 0;
}

int lh_eval_sample_batch_device(lh_family* f, int32_t n, int32_t T, int32_t max_depth, const int32_t* ops,
                                const double* brlen, const double* er, const double* pi, const double* alpha, int32_t R,
                                const uint32_t* words, double* loglik, double* rates, int32_t* states, void* hip_stream) {
  if (!f) return fail("lh_eval_sample_batch_device: null family");
  DeviceGuard guard(f);
  if (!f->have_sampler) return fail("lh_eval_sample_batch_device: lh_family_set_sampler has not been called");
  if (n <= 0) return n == 0 ? 0 : fail("lh_eval_sample_batch_device: negative batch size");
  if (!words || !states) return fail("lh_eval_sample_batch_device: null array");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const size_t FS = f->host.forward_size;
  void* d_fwd;  // the forward arrays never leave the device (grow-only slot; a growing call waits for earlier work)
  if (sizeof(double) * FS * n > f->st.cap[8]) LH_HIP(hipDeviceSynchronize());
  if (stage(f, 8, sizeof(double) * FS * n, &d_fwd)) return 1;
  lh_eval_outputs outs{rates, nullptr, (double*)d_fwd, nullptr};
  if (lh_eval_batch_device(f, n, T, max_depth, ops, brlen, er, pi, alpha, R, loglik, &outs, hip_stream)) return 1;
  const lh::DevSampler& smp = f->sampler;
  lh::launch_sample(smp, f->sampler_dev, n, (const double*)d_fwd, FS, words, smp.words_per_sample, states, stream);
  LH_HIP(hipGetLastError());
  return 0;
}

int lh_eval_sample_batch(lh_family* f, int32_t n, int32_t T, int32_t max_depth, const int32_t* ops,
                         const double* brlen, const double* er, const double* pi, const double* alpha, int32_t R,
                         const uint32_t* words, double* loglik, double* rates, int32_t* states) {
  if (!f) return fail("lh_eval_sample_batch: null family");
  DeviceGuard guard(f);
  if (!f->have_sampler) return fail("lh_eval_sample_batch: lh_family_set_sampler has not been called");
  if (n <= 0) return n == 0 ? 0 : fail("lh_eval_sample_batch: negative batch size");
  if (T < 3) return fail("lh_eval_sample_batch: need at least 3 tips");
  if (!ops || !brlen || !er || !pi || !alpha || !words || !loglik || !states) return fail("lh_eval_sample_batch: null array");
  const size_t nodes = 2 * (size_t)T - 2, n_ops = (size_t)T - 2, FS = f->host.forward_size;
  static const bool timing = lh::debug_options().sample_timing;  // stage times of every call, on stderr
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t0 = now();
  auto t1 = now();
  const lh::DevSampler& smp = f->sampler;
  const size_t bytes[9] = {sizeof(int32_t) * 4 * n_ops * n, sizeof(double) * nodes * n, sizeof(double) * 6 * n,
                           sizeof(double) * 4 * n, sizeof(double) * n, sizeof(uint32_t) * smp.words_per_sample * (size_t)n,
                           sizeof(double) * n /* loglik */, sizeof(double) * R * n /* rates */,
                           sizeof(int32_t) * smp.states_per_sample * (size_t)n};
  LH_HIP(hipDeviceSynchronize());  // earlier calls may still be using the buffers
  void* d[9];
  for (int a = 0; a < 9; ++a) {
    if (bytes[a] > f->smp.cap[a]) {
      if (f->smp.ptr[a]) LH_HIP(hipFree(f->smp.ptr[a]));
      f->smp.ptr[a] = nullptr;
      f->smp.cap[a] = 0;
      LH_HIP(hipMalloc(&f->smp.ptr[a], bytes[a]));
      f->smp.cap[a] = bytes[a];
    }
    d[a] = f->smp.ptr[a];
  }
  void* d_fwd;  // the forward arrays never leave the device
  if (stage(f, 8, sizeof(double) * FS * n, &d_fwd)) return 1;
  // The caller's arrays are ordinary pageable memory: copied from there, every transfer has the driver lock and
  // unlock their pages, which stalls for milliseconds whenever other threads of the process are busy allocating
  // (RunPipeline's formatting workers are).  One memcpy into a page-locked slot costs a fraction of that.
  size_t in_bytes = 0;
  for (int a = 0; a < 6; ++a) in_bytes += (bytes[a] + 63) & ~(size_t)63;
  if (in_bytes > f->smp.pinned_cap) {
    if (f->smp.pinned) LH_HIP(hipHostFree(f->smp.pinned));
    f->smp.pinned = nullptr;
    f->smp.pinned_cap = 0;
    LH_HIP(hipHostMalloc(&f->smp.pinned, in_bytes, hipHostMallocDefault));
    f->smp.pinned_cap = in_bytes;
  }
  auto t2 = now();
  const void* src[6] = {ops, brlen, er, pi, alpha, words};
  {
    char* slot = static_cast<char*>(f->smp.pinned);
    for (int a = 0; a < 6; ++a) {
      memcpy(slot, src[a], bytes[a]);
      LH_HIP(hipMemcpyAsync(d[a], slot, bytes[a], hipMemcpyHostToDevice, nullptr));
      slot += (bytes[a] + 63) & ~(size_t)63;
    }
  }
  auto t3 = now();
  lh_eval_outputs outs{(double*)d[7], nullptr, (double*)d_fwd, nullptr};
  if (lh_eval_batch_device(f, n, T, max_depth, (const int32_t*)d[0], (const double*)d[1], (const double*)d[2],
                           (const double*)d[3], (const double*)d[4], R, (double*)d[6], &outs, nullptr))
    return 1;
  if (timing) LH_HIP(hipDeviceSynchronize());
  auto t4 = now();
  lh::launch_sample(smp, f->sampler_dev, n, (const double*)d_fwd, FS, (const uint32_t*)d[5], smp.words_per_sample, (int32_t*)d[8], nullptr);
  LH_HIP(hipGetLastError());
  if (timing) LH_HIP(hipDeviceSynchronize());
  auto t5 = now();
  // The schedules are checked on the host WHILE the device works on them: the kernels make the same checks themselves (a
  // malformed op costs that sample a NaN and the handle an error code, never an out-of-bounds access), so nothing is
  // risked by the order, and a refused batch hands nothing back.
  if (!valid_schedules(ops, (size_t)n, T, (int)nodes, max_depth)) {
    (void)hipDeviceSynchronize();
    (void)check_async_error(f, "lh_eval_sample_batch");  // (the device found it too: one report is enough)
    return fail("lh_eval_sample_batch: malformed schedule op (use lh_schedule_tree)");
  }
  auto t6 = now();
  LH_HIP(hipMemcpy(loglik, d[6], bytes[6], hipMemcpyDeviceToHost));
  if (rates) LH_HIP(hipMemcpy(rates, d[7], bytes[7], hipMemcpyDeviceToHost));
  LH_HIP(hipMemcpy(states, d[8], bytes[8], hipMemcpyDeviceToHost));
  if (check_async_error(f, "lh_eval_sample_batch")) return 1;
  if (timing) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    std::fprintf(stderr,
                 "[lh_eval_sample_batch] n=%d: buffers %.2f ms, copies in %.2f, evaluation %.2f, sampling %.2f, check ops (beside "
                 "the device) %.2f, copies out %.2f\n",
                 n, ms(t0, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, t6), ms(t6, now()));
  }
  return 0;
}

int lh_asr_batch_device(lh_family* f, int32_t n, int32_t T, int32_t max_depth, const int32_t* ops,
                        const double* brlen, const double* er, const double* pi, const double* rates, int32_t R,
                        const uint8_t* naive, uint64_t seed, uint64_t first_sample, uint8_t* anc,
                        uint8_t* rate_choice, void* hip_stream) {
  if (!f) return fail("lh_asr_batch: null family");
  DeviceGuard guard(f);
  if (n < 0) return fail("lh_asr_batch: negative batch size");
  if (n == 0) return 0;
  if (f->host.n_seqs < 1) return fail("lh_asr_batch: family was created without an MSA");
  if (T != f->host.n_seqs + 1) return fail("lh_asr_batch: n_tips must equal n_seqs + 1 (naive)");
  if (T < 3) return fail("lh_asr_batch: need at least 3 tips");
  if (R < 1 || R > 64) return fail("lh_asr_batch: num_rates out of range");
  if (max_depth < 0 || max_depth > 16) return fail("lh_asr_batch: max_depth out of range");
  if ((size_t)T * 128 > 160 * 1024) return fail("lh_asr_batch: too many tips for the LDS tip table");
  if (lh::asr_lds_bytes(T, f->host.n_sites, R, f->host.n_prune) > 160 * 1024)
    return fail("lh_asr_batch: tree / alignment too large for the sampling kernel's LDS tables");
  if (!ops || !brlen || !er || !pi || !rates || !naive || !anc) return fail("lh_asr_batch: null array");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const size_t nodes = 2 * (size_t)T - 2, n_ops = (size_t)T - 2, L = f->host.n_sites;
  // launch groups: at most ~8 GB of CLV area (32 B per inner node and site) plus K1's workspace
  const size_t clv_per_sample = sizeof(double) * n_ops * 4 * lh::asr_slots((int)L, R);
  const size_t k1_per_sample = k1_bytes_per_sample(f, T, R) + sizeof(double) * R * 6 * (size_t)std::max(f->host.n_prune, 1);
  const int by_memory = (int)std::max<size_t>(64, ((size_t)8 << 30) / (clv_per_sample + k1_per_sample));
  const int chunk = std::min<int>(n, std::min(8192, by_memory));
  if (ensure_workspace(f, chunk, R, T)) return 1;
  AsrWs& aw = f->asr;
  if (aw.clv_cap < clv_per_sample * chunk) {
    LH_HIP(hipDeviceSynchronize());
    if (aw.clv) LH_HIP(hipFree(aw.clv));
    aw.clv = nullptr;
    aw.clv_cap = 0;
    LH_HIP(hipMalloc((void**)&aw.clv, clv_per_sample * chunk));
    aw.clv_cap = clv_per_sample * chunk;
  }
  if (aw.desc_cap < lh::asr_desc_bytes(T) * chunk) {
    LH_HIP(hipDeviceSynchronize());
    if (aw.desc) LH_HIP(hipFree(aw.desc));
    aw.desc = nullptr;
    aw.desc_cap = 0;
    LH_HIP(hipMalloc(&aw.desc, lh::asr_desc_bytes(T) * chunk));
    aw.desc_cap = lh::asr_desc_bytes(T) * chunk;
  }
  if (!rate_choice && aw.choice_cap < L * (size_t)chunk) {
    LH_HIP(hipDeviceSynchronize());
    if (aw.choice) LH_HIP(hipFree(aw.choice));
    aw.choice = nullptr;
    aw.choice_cap = 0;
    LH_HIP(hipMalloc((void**)&aw.choice, L * (size_t)chunk));
    aw.choice_cap = L * (size_t)chunk;
  }
  Workspace& w = f->ws;
  for (int off = 0; off < n; off += chunk) {
    const int m = std::min(chunk, n - off);
    const double* r_m = rates + (size_t)off * R;
    const double* pi_m = pi + (size_t)off * 4;
    const int32_t* ops_m = ops + (size_t)off * n_ops * 4;
    const double* bl_m = brlen + (size_t)off * nodes;
    lh::launch_gtr_setup(m, er + (size_t)off * 6, pi_m, w.eig, stream);
    // per-rate planes: K1 must not mix the categories here
    const int planes = lh::launch_prune(f->host, m, R, T, max_depth, ops_m, bl_m, r_m, w.eig, w.prune, pi_m,
                                        w.site_lik, w.site_scal, stream, false);
    if (planes < 0) return fail(std::string("lh_asr_batch: ") + lh::prune_last_error());
    f->k1_form = lh::prune_last_form();
    if (planes != R && f->host.n_prune > 0) return fail("lh_asr_batch: internal error (rate planes were mixed)");
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (f->profile) {
      LH_HIP(hipEventCreate(&ev.first));
      LH_HIP(hipEventCreate(&ev.second));
      LH_HIP(hipEventRecord(ev.first, stream));
    }
    if (lh::launch_asr(f->host, m, R, T, ops_m, bl_m, r_m, w.eig, pi_m, w.site_lik, w.site_scal,
                       naive + (size_t)off * L, seed, first_sample + (uint64_t)off, aw.clv, aw.desc,
                       anc + (size_t)off * n_ops * L,
                       rate_choice ? rate_choice + (size_t)off * L : aw.choice, w.prune.hdr, stream))
      return fail("lh_asr_batch: launch failed");
    if (f->profile) {
      LH_HIP(hipEventRecord(ev.second, stream));
      aw.events.push_back(ev);
    }
    LH_HIP(hipGetLastError());
  }
  return 0;
}

static int asr_stage(lh_family* f, int slot, size_t bytes, void** out) {
  AsrWs& a = f->asr;
  if (bytes > a.cap[slot]) {
    if (a.ptr[slot]) LH_HIP(hipFree(a.ptr[slot]));
    a.ptr[slot] = nullptr;
    a.cap[slot] = 0;
    LH_HIP(hipMalloc(&a.ptr[slot], bytes));
    a.cap[slot] = bytes;
  }
  *out = a.ptr[slot];
  return 0;
}

int lh_asr_batch(lh_family* f, int32_t n, int32_t T, int32_t max_depth, const int32_t* ops, const double* brlen,
                 const double* er, const double* pi, const double* rates, int32_t R, const uint8_t* naive,
                 uint64_t seed, uint64_t first_sample, uint8_t* anc, uint8_t* rate_choice) {
  if (!f) return fail("lh_asr_batch: null family");
  DeviceGuard guard(f);
  if (n < 0) return fail("lh_asr_batch: negative batch size");
  if (n == 0) return 0;
  if (T < 3 || T != f->host.n_seqs + 1) return fail("lh_asr_batch: n_tips must equal n_seqs + 1 (naive)");
  if (R < 1 || R > 64) return fail("lh_asr_batch: num_rates out of range");
  if (max_depth < 0 || max_depth > 16) return fail("lh_asr_batch: max_depth out of range");
  if (!ops || !brlen || !er || !pi || !rates || !naive || !anc) return fail("lh_asr_batch: null array");
  const size_t nodes = 2 * (size_t)T - 2, n_ops = (size_t)T - 2, L = f->host.n_sites;
  if (!valid_schedules(ops, (size_t)n, T, (int)nodes, max_depth))
    return fail("lh_asr_batch: malformed schedule op (use lh_schedule_tree)");
  for (size_t k = 0; k < (size_t)n * L; ++k)
    if (naive[k] > 4) return fail("lh_asr_batch: naive base out of range");
  LH_HIP(hipDeviceSynchronize());
  // sub-batches bound the device copy of the output (anc: (T-2) * L bytes per sample)
  const int sub = (int)std::max<size_t>(1, std::min<size_t>(n, ((size_t)1 << 30) / std::max<size_t>(n_ops * L, 1)));
  for (int off = 0; off < n; off += sub) {
    const int m = std::min(sub, n - off);
    const size_t bytes[7] = {sizeof(int32_t) * 4 * n_ops * m, sizeof(double) * nodes * m, sizeof(double) * 6 * m,
                             sizeof(double) * 4 * m,          sizeof(double) * R * m,     L * m,
                             n_ops * L * m};
    const void* src[6] = {ops + (size_t)off * n_ops * 4, brlen + (size_t)off * nodes, er + (size_t)off * 6,
                          pi + (size_t)off * 4,          rates + (size_t)off * R,     naive + (size_t)off * L};
    void* d[8];
    for (int a = 0; a < 7; ++a)
      if (asr_stage(f, a, bytes[a], &d[a])) return 1;
    if (asr_stage(f, 7, L * m, &d[7])) return 1;
    for (int a = 0; a < 6; ++a) LH_HIP(hipMemcpy(d[a], src[a], bytes[a], hipMemcpyHostToDevice));
    if (lh_asr_batch_device(f, m, T, max_depth, (const int32_t*)d[0], (const double*)d[1], (const double*)d[2],
                            (const double*)d[3], (const double*)d[4], R, (const uint8_t*)d[5], seed,
                            first_sample + (uint64_t)off, (uint8_t*)d[6], (uint8_t*)d[7], nullptr))
      return 1;
    if (check_async_error(f, "lh_asr_batch")) return 1;
    LH_HIP(hipMemcpy(anc + (size_t)off * n_ops * L, d[6], bytes[6], hipMemcpyDeviceToHost));
    if (rate_choice) LH_HIP(hipMemcpy(rate_choice + (size_t)off * L, d[7], L * m, hipMemcpyDeviceToHost));
  }
  return 0;
}

int lh_forward_batch(lh_family* f, int32_t n, const double* em, double* loglik, const lh_eval_outputs* outs) {
  if (!f) return fail("lh_forward_batch: null family");
  DeviceGuard guard(f);
  if (n <= 0) return n == 0 ? 0 : fail("lh_forward_batch: negative batch size");
  if (!em || !loglik) return fail("lh_forward_batch: null array");
  const size_t C = f->host.n_xmsa, FS = f->host.forward_size, SS = f->host.scaler_size;
  void *d_em, *d_ll;
  if (stage(f, 7, sizeof(double) * C * n, &d_em)) return 1;
  if (stage(f, 5, sizeof(double) * n, &d_ll)) return 1;
  LH_HIP(hipMemcpy(d_em, em, sizeof(double) * C * n, hipMemcpyHostToDevice));
  lh_eval_outputs d_outs{nullptr, nullptr, nullptr, nullptr};
  if (outs) {
    if (outs->forward && stage(f, 8, sizeof(double) * FS * n, (void**)&d_outs.forward)) return 1;
    if (outs->scaler_counts && stage(f, 9, sizeof(int32_t) * SS * n, (void**)&d_outs.scaler_counts)) return 1;
  }
  if (run_forward(f, n, 1, nullptr, nullptr, nullptr, (const double*)d_em, nullptr, (double*)d_ll, &d_outs, 0, nullptr))
    return 1;
  LH_HIP(hipDeviceSynchronize());
  LH_HIP(hipMemcpy(loglik, d_ll, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (outs) {
    if (outs->forward)
      LH_HIP(hipMemcpy(outs->forward, d_outs.forward, sizeof(double) * FS * n, hipMemcpyDeviceToHost));
    if (outs->scaler_counts)
      LH_HIP(hipMemcpy(outs->scaler_counts, d_outs.scaler_counts, sizeof(int32_t) * SS * n,
                       hipMemcpyDeviceToHost));
  }
  return 0;
}

}  // extern "C"
