// dib_ctw.cpp - infinite-depth context-tree weighting (CTW) entropy-rate estimator, host C++ (include/dib_ctw.h).
//
// What it computes is what the reference's Cython extension computes (reference chaos/cppctw.cpp): a suffix tree of
// the contexts seen so far with lazily expanded "tail" leaves (cppctw.cpp:117-125), per-node symbol counts, a
// Krichevsky-Trofimov / Dirichlet(beta = 1/|A|) local code length (cppctw.cpp:57-64) and the CTW mixture
// L_w = 1 + min(L_children, L_local) - log2(1 + 2^-|L_local - L_children|) (cppctw.cpp:74-78), rate = L_w(root) / n.
//
// How it is built differs: the reference allocates one heap object with two std::vectors per node, keeps alphabet
// size / beta in static members (not thread-safe) and evaluates the mixture by recursion.  Here a tree is one arena
// (flat child / count tables indexed by node id - children are always created after their parent, so a single
// reverse sweep over the ids is a valid post-order: no recursion, no pointer chasing), lgamma(count + beta) is
// memoised per tree, all state is per call, and a batch entry point spreads independent sequences over host threads
// (the chaos notebook estimates 75 sequences per measurement partition).  Arithmetic order is kept identical to the
// reference expression by expression, so results are bit-identical on the same libm.
#include "../../include/dib_ctw.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <limits>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxDepth = 512;  // reference chaos/cppctw.cpp:13 (contexts longer than this are not extended)

class ContextTree {
 public:
  explicit ContextTree(int alphabet) : A_(alphabet), beta_(1. / alphabet) { add_node(-1, -1); }

  // reference chaos/cppctw.cpp:104-152 (SuffixTree::process_sequence)
  void consume(const int8_t* seq, int64_t n) {
    for (int64_t t = 0; t < n; ++t) {
      const int s = seq[t];
      int32_t node = 0;
      ++count_[s];
      for (int64_t c = t - 1; c >= 0; --c) {
        if (tail_pos_[node] > 0) {
          // a leaf that remembers where its one occurrence continues: materialise the next context symbol first
          const int32_t pos = tail_pos_[node] - 1;
          const int8_t sym = tail_sym_[node];
          const int32_t kid = add_node(pos, sym);
          child_[(size_t)node * A_ + seq[pos]] = kid;
          ++count_[(size_t)kid * A_ + sym];
          tail_pos_[node] = -1;
          tail_sym_[node] = -1;
        }
        const int ctx = seq[c];
        const int32_t next = child_[(size_t)node * A_ + ctx];
        if (next < 0) {
          if (t - c > kMaxDepth) break;
          const int32_t kid = (c > 0) ? add_node((int32_t)c, (int8_t)s) : add_node(-1, -1);
          child_[(size_t)node * A_ + ctx] = kid;
          ++count_[(size_t)kid * A_ + s];
          break;
        }
        node = next;
        ++count_[(size_t)node * A_ + s];
      }
    }
  }

  // reference chaos/cppctw.cpp:55-82 (update_code_lengths) + :98-102 (estimate_entropy, float return)
  double rate(int64_t n) {
    const int64_t nodes = (int64_t)tail_pos_.size();
    std::vector<double> weighted((size_t)nodes);
    const double ab = A_ * beta_;
    const double lg_ab = lgam(ab), lg_b = lgam(beta_), ln2 = std::log(2);
    for (int64_t v = nodes - 1; v >= 0; --v) {  // ids grow from parent to child: reverse order = post-order
      const int32_t* cnt = &count_[(size_t)v * A_];
      double total = 0.;
      for (int i = 0; i < A_; ++i) total += cnt[i];
      double local = lgam(total + ab) - lg_ab;
      for (int i = 0; i < A_; ++i) local -= lgam_count(cnt[i]) - lg_b;
      local /= ln2;
      double kids = 0;
      bool any = false;
      const int32_t* ch = &child_[(size_t)v * A_];
      for (int i = 0; i < A_; ++i)
        if (ch[i] >= 0) {
          any = true;
          kids += weighted[(size_t)ch[i]];
        }
      weighted[(size_t)v] = (any && total > 1)
                                ? 1 + std::min(kids, local) - std::log2(1 + std::pow(2, -std::abs(local - kids)))
                                : local;
    }
    const float r = (float)(weighted[0] / (int)n);
    return r;
  }

  int64_t nodes() const { return (int64_t)tail_pos_.size(); }

 private:
  int32_t add_node(int32_t tail_pos, int8_t tail_sym) {
    const int32_t id = (int32_t)tail_pos_.size();
    tail_pos_.push_back(tail_pos);
    tail_sym_.push_back(tail_sym);
    child_.insert(child_.end(), (size_t)A_, -1);
    count_.insert(count_.end(), (size_t)A_, 0);
    return id;
  }
  static double lgam(double x) {
    int sign;
    return ::lgamma_r(x, &sign);  // same value as lgamma(), without the write to the global signgam
  }
  double lgam_count(int32_t c) {  // lgamma(c + beta), memoised for the small counts that dominate the tree
    if (c >= kMemo) return lgam(c + beta_);
    if ((int)memo_.size() <= c) {
      const int old = (int)memo_.size();
      memo_.resize((size_t)c + 1);
      for (int k = old; k <= c; ++k) memo_[(size_t)k] = lgam(k + beta_);
    }
    return memo_[(size_t)c];
  }

  static constexpr int kMemo = 1 << 16;
  const int A_;
  const double beta_;
  std::vector<int32_t> child_, count_, tail_pos_;
  std::vector<int8_t> tail_sym_;
  std::vector<double> memo_;
};

int run_one(const int8_t* seq, int64_t n, int alphabet, double* rate, int64_t* nodes) {
  if (!seq && n > 0) return DIB_CTW_E_ARG;
  if (n < 0 || n > std::numeric_limits<int32_t>::max() || alphabet < 1 || alphabet > 127) return DIB_CTW_E_ARG;
  for (int64_t i = 0; i < n; ++i)
    if (seq[i] < 0 || seq[i] >= alphabet) return DIB_CTW_E_ARG;
  try {
    ContextTree tree(alphabet);
    tree.consume(seq, n);
    if (rate) *rate = tree.rate(n);
    if (nodes) *nodes = tree.nodes();
  } catch (const std::bad_alloc&) {
    return DIB_CTW_E_NOMEM;
  }
  return DIB_CTW_OK;
}

}  // namespace

extern "C" {

const char* dib_ctw_version(void) { return "dib_ctw 0.1 (arena suffix tree, iterative CTW mixture, threaded batch)"; }

int dib_ctw_estimate_entropy(const int8_t* seq, int64_t n, int alphabet_size, double* rate_out) {
  if (!rate_out) return DIB_CTW_E_ARG;
  return run_one(seq, n, alphabet_size, rate_out, nullptr);
}

int dib_ctw_node_count(const int8_t* seq, int64_t n, int alphabet_size, int64_t* nodes_out) {
  if (!nodes_out) return DIB_CTW_E_ARG;
  return run_one(seq, n, alphabet_size, nullptr, nodes_out);
}

int dib_ctw_estimate_entropy_batch(const int8_t* seqs, const int64_t* offsets, int n_seq, int alphabet_size, int threads,
                                   double* rates_out) {
  if (!offsets || !rates_out || n_seq < 0) return DIB_CTW_E_ARG;
  if (n_seq == 0) return DIB_CTW_OK;
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, n_seq));
  std::atomic<int> next(0), first_err(DIB_CTW_OK);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n_seq) return;
      const int64_t b = offsets[i], e = offsets[i + 1];
      int rc = (e < b) ? DIB_CTW_E_ARG : run_one(seqs ? seqs + b : nullptr, e - b, alphabet_size, &rates_out[i], nullptr);
      if (rc != DIB_CTW_OK) {
        rates_out[i] = std::numeric_limits<double>::quiet_NaN();
        int expect = DIB_CTW_OK;
        first_err.compare_exchange_strong(expect, rc);
      }
    }
  };
  std::vector<std::thread> pool;
  try {
    for (int k = 1; k < nt; ++k) pool.emplace_back(work);
  } catch (...) {  // could not start a thread: the calling thread does the rest
  }
  work();
  for (auto& th : pool) th.join();
  return first_err.load();
}

}  // extern "C"
