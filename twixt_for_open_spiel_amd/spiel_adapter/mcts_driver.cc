// mcts_driver.cc -- batched-leaf UCT self-play on the HIP engine, host side in C++
// (BASELINE config 5: board_size=12, max_simulations=20000, batched leaf ApplyAction /
// IsTerminal / play-outs on one MI355X).  The same search as twixt_for_open_spiel_amd/mcts.py
// (same selection order, virtual loss, tie-breaks and arithmetic), so both drivers play
// the same games; the Python one is the readable reference, this one removes the
// interpreter from the loop.
//
//   g++ -O2 -fopenmp -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include mcts_driver.cc
//     -Ltwixt_for_open_spiel_amd -ltwixt_hip -L/opt/rocm/lib -lamdhip64 -o mcts_driver
//   ./mcts_driver [board_size=12] [max_simulations=20000] [moves=0 (whole game)] [leaves=2048] [seed=1] [games=1]
//
// The leaves of a round trip go out in two half-batches ("lanes"): while the device works on one, the
// host selects the leaves of the other and backs the previous one up -- the order
// launch(0), launch(1), finish(0), launch(2), finish(1), ... is fixed, so the search stays deterministic
// (mcts.py BatchedMCTS.search follows the same order).
//
// games > 1: that many independent self-play games at once (self-play data generation).  Every
// game has its own tree and root slot; the `leaves` of a device round trip are shared evenly,
// the trees are worked on by one host thread each (OpenMP), the device work of all games goes
// out as one batch.  games = 1 is exactly the single search above.
#include <hip/hip_runtime_api.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "twixt_hip.h"

#define TWX(x) do { if ((x) != TWX_OK) { std::fprintf(stderr, "%s: %s\n", #x, twx_last_error()); std::exit(1); } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// Bump allocator for one search tree: nodes and their arrays come out of 1 MiB blocks that
// are recycled for the next move's tree, so building and dropping a tree of a few 10^4 nodes
// costs no malloc / free traffic (which dominated once several trees were worked on at once).
class Arena {
 public:
  void *Alloc(size_t bytes) {
    bytes = (bytes + 15) & ~(size_t)15;
    if (blocks_.empty() || used_ + bytes > kBlock) {
      if (++cur_ >= (long)blocks_.size()) blocks_.emplace_back(new char[std::max(kBlock, bytes)]);
      else if (bytes > kBlock) { blocks_[cur_].reset(new char[bytes]); }
      used_ = 0;
    }
    void *p = blocks_[cur_].get() + used_;
    used_ += bytes;
    return p;
  }
  template <class T>
  T *Array(size_t n) { return static_cast<T *>(Alloc(n * sizeof(T))); }
  void Reset() { cur_ = -1; used_ = kBlock; }  // keep the blocks

 private:
  static constexpr size_t kBlock = 1 << 20;
  std::vector<std::unique_ptr<char[]>> blocks_;
  long cur_ = -1;
  size_t used_ = kBlock;
};

// A node keeps its children's statistics in its own arrays (cn / cw, indexed like `legal`),
// so that the UCT scan of a node reads contiguous memory; a child reaches its own counters
// through (parent, slot).  Children are created in ascending action order -- an uncreated
// child scores 1e9 in mcts.py and the first maximum wins -- so `created` children are
// always the prefix legal[0..created) and the next new child is found without a scan.
// Everything lives in the tree's arena; nothing is freed node by node.
struct Node {
  Node *parent;
  int slot;           // index in the parent's arrays (-1 at the root)
  int action, mover;  // mover: who played `action` (-1 at the root)
  int player = -1;    // to move here, known once expanded
  bool expanded = false;
  int terminal = 0;   // result if the game is over here
  double root_n = 0.0, root_w = 0.0;  // the root's own counters
  size_t nlegal = 0, created = 0;
  int *legal = nullptr;
  double *cn = nullptr, *cw = nullptr;  // per child: visits, summed return for this node's player
  Node **child = nullptr;
  Node(Node *p, int sl, int a, int m) : parent(p), slot(sl), action(a), mover(m) {}
  double &n() { return parent ? parent->cn[slot] : root_n; }
  double &w() { return parent ? parent->cw[slot] : root_w; }
  void Expand(Arena *arena, const std::vector<int> &la, int to_move) {
    expanded = true;
    player = to_move;
    nlegal = la.size();
    legal = arena->Array<int>(nlegal);
    cn = arena->Array<double>(nlegal);
    cw = arena->Array<double>(nlegal);
    child = arena->Array<Node *>(nlegal);
    for (size_t i = 0; i < nlegal; i++) { legal[i] = la[i]; cn[i] = 0.0; cw[i] = 0.0; child[i] = nullptr; }
  }
};

struct Leaf { int result, player; std::vector<int> legal; int final_result; };

// wall-clock shares of a run (printed at the end): tree work on the host vs device round trips
static double g_t[7];  // select, queueing a lane, waiting for a lane, parsing its results (CPU evaluator: play-outs), backup, new trees, batch assembly
struct Timer {
  int slot;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit Timer(int s) : slot(s) {}
  ~Timer() { g_t[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// The evaluator can be swapped at build time: tests/mcts_cpu_evaluator.h plugs the CPU checker
// in behind the same interface, so that the identical search can be timed and compared on the
// host (tests/test_mcts.py, tools/mcts_bench.py).  The product build has no such header.
#ifdef MCTS_EVALUATOR_HEADER
#include MCTS_EVALUATOR_HEADER
#else
// slots 0..L-1 are scratch leaves (lane l owns slots l*L/2 ...), slots L..L+G-1 the roots of the G games
// (mcts.py: DeviceEvaluator).  A lane's inputs and outputs live in mapped pinned host memory that the
// kernels read and write in place: Launch() only enqueues, Wait() waits for the lane's event.
// The leaf slots are split between the lanes only while each half still holds a leaf of every game
// (L >= 2G); below that both lanes use ALL L slots, one evaluation after the other (the stream is in
// order and a lane's results have left the slots when the next evaluation overwrites them), so a lane
// never reaches into the root slots L.. (the same rule: LaneSlots in tests/mcts_cpu_evaluator.h,
// BatchedMCTS.search in mcts.py).
class Device {
 public:
  Device(int n, int leaves, int games, int threads)
      : n_(n), L_(leaves), G_(games), W_(twx_mask_words(n)), threads_(threads) {
    const int cap = L_ + G_;
    TWX(twx_create(n, cap, 0, 0, &env_));
    HIP(hipMalloc(&d_ids_, 4 * cap));
    HIP(hipMalloc(&d_act_, 4 * cap));
    HIP(hipMalloc(&d_st_, cap));
    HIP(hipMalloc(&d_res_, cap));
    HIP(hipMalloc(&d_cp_, cap));
    HIP(hipMalloc(&d_mask_, 4 * (size_t)W_ * cap));
    HIP(hipMalloc(&d_iota_, 4 * cap));
    std::vector<int32_t> iota(cap);
    for (int i = 0; i < cap; i++) iota[i] = i;
    HIP(hipMemcpy(d_iota_, iota.data(), 4 * cap, hipMemcpyHostToDevice));
    for (auto &ln : lane_) HIP(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
  }
  ~Device() {
    for (auto &ln : lane_) { if (ln.pin) (void)hipHostFree(ln.pin); (void)hipEventDestroy(ln.done); }
    twx_destroy(env_);
  }
  int LaneLeaves() const { return L_ >= 2 * G_ ? L_ / 2 : L_; }               // leaves one lane takes
  int LaneBase(int lane) const { return L_ >= 2 * G_ ? lane * (L_ / 2) : 0; }  // its first slot

  void ResetRoots() { TWX(twx_reset_boards(env_, d_iota_ + L_, G_, nullptr)); }
  std::vector<Leaf> RootInfos() {
    TWX(twx_status_boards(env_, d_iota_ + L_, G_, d_res_, nullptr, d_cp_, nullptr));
    TWX(twx_legal_mask_boards(env_, d_iota_ + L_, G_, d_mask_, nullptr, nullptr));
    std::vector<uint8_t> res(G_);
    std::vector<int8_t> cp(G_);
    std::vector<uint32_t> mask((size_t)W_ * G_);
    HIP(hipMemcpy(res.data(), d_res_, G_, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(cp.data(), d_cp_, G_, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(mask.data(), d_mask_, 4 * (size_t)W_ * G_, hipMemcpyDeviceToHost));
    return Parse(res.data(), cp.data(), mask.data(), G_);
  }
  // one move per listed game; returns nothing: the next RootInfos() tells the results
  void AdvanceRoots(const std::vector<int> &games, const std::vector<int> &actions) {
    if (games.empty()) return;
    std::vector<int32_t> ids(games.size()), acts(actions.begin(), actions.end());
    for (size_t i = 0; i < games.size(); i++) ids[i] = L_ + games[i];
    std::vector<uint8_t> st(games.size());
    HIP(hipMemcpy(d_ids_, ids.data(), 4 * ids.size(), hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_act_, acts.data(), 4 * acts.size(), hipMemcpyHostToDevice));
    TWX(twx_step_boards(env_, d_ids_, (int64_t)ids.size(), d_act_, d_st_, nullptr));
    HIP(hipMemcpy(st.data(), d_st_, st.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < st.size(); i++)
      if (st[i] & TWX_ST_ILLEGAL) { std::fprintf(stderr, "illegal root move %d\n", actions[i]); std::exit(1); }
  }
  // paths[i]: actions from the root of game game_of[i] to leaf i of this lane (slot base + i).  Queues the
  // whole evaluation -- roots -> slots, the paths level by level, status and legal masks of the leaves,
  // the play-outs -- and returns; the lane's buffers must not be touched before Wait(lane).
  void Launch(int lane, const std::vector<std::vector<int>> &paths, const std::vector<int> &game_of, uint32_t seed) {
    Timer tm(1);
    Lane &ln = lane_[lane];
    const int k = (int)paths.size(), base = LaneBase(lane);
    if (k < 0 || base + k > L_) {  // the slots from L_ on hold the games' roots
      std::fprintf(stderr, "lane %d: %d leaves from slot %d do not fit %d leaf slots\n", lane, k, base, L_);
      std::exit(1);
    }
    size_t steps = 0, depth = 0;
    for (auto &p : paths) { steps += p.size(); depth = std::max(depth, p.size()); }
    // layout of the lane's block: src[k] | rows[steps] | acts[steps] | res[k] cp[k] st[k] (bytes) | mask[k*W]
    const size_t o_rows = 4 * (size_t)k, o_acts = o_rows + 4 * steps, o_res = o_acts + 4 * steps, o_cp = o_res + k,
                 o_st = o_cp + k, o_mask = (o_st + k + 15) / 16 * 16, bytes = o_mask + 4 * (size_t)W_ * k;
    if (bytes > ln.bytes) {
      if (ln.pin) HIP(hipHostFree(ln.pin));
      ln.bytes = bytes + bytes / 2;
      HIP(hipHostMalloc(&ln.pin, ln.bytes, hipHostMallocMapped));
      void *d = nullptr;
      HIP(hipHostGetDevicePointer(&d, ln.pin, 0));
      ln.dev = static_cast<char *>(d);
    }
    ln.k = k; ln.o_res = o_res; ln.o_cp = o_cp; ln.o_st = o_st; ln.o_mask = o_mask;
    char *h = static_cast<char *>(ln.pin);
    int32_t *src = reinterpret_cast<int32_t *>(h), *rows = reinterpret_cast<int32_t *>(h + o_rows),
            *acts = reinterpret_cast<int32_t *>(h + o_acts);
    for (int i = 0; i < k; i++) src[i] = L_ + game_of[i];
    const int32_t *slots = d_iota_ + base;
    TWX(twx_copy_boards(env_, slots, env_, reinterpret_cast<const int32_t *>(ln.dev), k, nullptr));  // roots -> slots
    size_t at = 0;
    for (size_t t = 0; t < depth; t++) {
      const size_t first = at;
      for (int i = 0; i < k; i++)
        if (paths[i].size() > t) { rows[at] = base + i; acts[at] = paths[i][t]; at++; }
      TWX(twx_step_boards(env_, reinterpret_cast<const int32_t *>(ln.dev + o_rows) + first, (int64_t)(at - first),
                          reinterpret_cast<const int32_t *>(ln.dev + o_acts) + first, nullptr, nullptr));
    }
    // (the play-outs do not touch the query results)
    TWX(twx_status_boards(env_, slots, k, reinterpret_cast<uint8_t *>(ln.dev + o_res), nullptr,
                          reinterpret_cast<int8_t *>(ln.dev + o_cp), nullptr));
    TWX(twx_legal_mask_boards(env_, slots, k, reinterpret_cast<uint32_t *>(ln.dev + o_mask), nullptr, nullptr));
    TWX(twx_random_rollout(env_, slots, k, seed, n_ * n_ - 3, reinterpret_cast<uint8_t *>(ln.dev + o_st), nullptr));  // terminal leaves stay
    HIP(hipEventRecord(ln.done, nullptr));
  }
  std::vector<Leaf> Wait(int lane) {
    Lane &ln = lane_[lane];
    {
      Timer tm(2);
      HIP(hipEventSynchronize(ln.done));
    }
    Timer tm(3);
    const char *h = static_cast<const char *>(ln.pin);
    std::vector<Leaf> out = Parse(reinterpret_cast<const uint8_t *>(h + ln.o_res), reinterpret_cast<const int8_t *>(h + ln.o_cp),
                                  reinterpret_cast<const uint32_t *>(h + ln.o_mask), ln.k);
    const uint8_t *st = reinterpret_cast<const uint8_t *>(h + ln.o_st);
    for (int i = 0; i < ln.k; i++) out[i].final_result = st[i] & TWX_ST_RESULT_MASK;
    return out;
  }

 private:
  std::vector<Leaf> Parse(const uint8_t *res, const int8_t *cp, const uint32_t *mask, int k) {
    std::vector<Leaf> out(k);
#pragma omp parallel for schedule(static) num_threads(threads_) if (threads_ > 1 && k >= 1024)
    for (int i = 0; i < k; i++) {
      out[i].result = res[i];
      out[i].player = cp[i];
      out[i].final_result = res[i];
      if (res[i] == 0) {
        out[i].legal.reserve(n_ * (n_ - 2));
        for (int w = 0; w < W_; w++)  // ascending action ids: LegalActions() order (tw.h:86-90)
          for (uint32_t m = mask[(size_t)i * W_ + w]; m; m &= m - 1) out[i].legal.push_back(32 * w + __builtin_ctz(m));
      }
    }
    return out;
  }
  struct Lane {
    void *pin = nullptr;
    char *dev = nullptr;
    size_t bytes = 0, o_res = 0, o_cp = 0, o_st = 0, o_mask = 0;
    int k = 0;
    hipEvent_t done{};
  };
  int n_, L_, G_, W_, threads_;
  twx_env *env_ = nullptr;
  int32_t *d_ids_, *d_act_, *d_iota_;
  uint8_t *d_st_, *d_res_;
  int8_t *d_cp_;
  uint32_t *d_mask_;
  Lane lane_[2];
};
#endif  // MCTS_EVALUATOR_HEADER

static void Returns(int result, double r[2]) {  // twixt.h:50-63
  r[0] = result == 1 ? 1.0 : result == 2 ? -1.0 : 0.0;
  r[1] = -r[0];
}

// UCT values of all children of a node: val[i] = cw[i]/cn[i] + c*sqrt(logn/cn[i]).  IEEE
// division, square root, multiplication and addition round the same in scalar and vector
// form and nothing is contracted into an FMA (the AVX2 target has none), so the vector
// path returns the bits mcts.py computes.
static void UctValuesScalar(const double *cn, const double *cw, double *val, size_t i, size_t m, double logn,
                            double c) {
  for (; i < m; i++) val[i] = cw[i] / cn[i] + c * std::sqrt(logn / cn[i]);
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void UctValuesAvx2(const double *cn, const double *cw, double *val, size_t m,
                                                          double logn, double c) {
  const __m256d vl = _mm256_set1_pd(logn), vc = _mm256_set1_pd(c);
  size_t i = 0;
  for (; i + 4 <= m; i += 4) {
    const __m256d n = _mm256_loadu_pd(cn + i), w = _mm256_loadu_pd(cw + i);
    const __m256d u = _mm256_mul_pd(vc, _mm256_sqrt_pd(_mm256_div_pd(vl, n)));
    _mm256_storeu_pd(val + i, _mm256_add_pd(_mm256_div_pd(w, n), u));
  }
  UctValuesScalar(cn, cw, val, i, m, logn, c);
}
#endif
static void UctValues(const double *cn, const double *cw, double *val, size_t m, double logn, double c) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return UctValuesAvx2(cn, cw, val, m, logn, c);
#endif
  UctValuesScalar(cn, cw, val, 0, m, logn, c);
}

// the search tree of one game for one move (mcts.py BatchedMCTS)
class Mcts {
 public:
  explicit Mcts(double c = 2.0) : c_(c) {}
  void Begin(int root_player, const std::vector<int> &root_legal) {
    arena_.Reset();  // drops the previous move's tree
    root_ = new (arena_.Alloc(sizeof(Node))) Node(nullptr, -1, -1, -1);
    root_->Expand(&arena_, root_legal, root_player);
  }
  // k descents with virtual loss; the leaves and their paths are appended to the outputs
  void SelectBatch(long k, std::vector<Node *> *leaves, std::vector<std::vector<int>> *paths) {
    for (long i = 0; i < k; i++) {
      paths->emplace_back();
      leaves->push_back(Select(root_, &paths->back()));
    }
  }
  void ApplyBatch(Node *const *leaves, Leaf *out, long k) {
    for (long i = 0; i < k; i++) {
      Node *leaf = leaves[i];
      if (out[i].result != 0) leaf->terminal = out[i].result;
      else if (!leaf->expanded) leaf->Expand(&arena_, out[i].legal, out[i].player);
      Backup(leaf, out[i].result != 0 ? out[i].result : out[i].final_result);
    }
  }
  int Best(long *best_visits) const {
    int best = -1;
    long bv = -1;
    for (size_t i = 0; i < root_->created; i++)  // ascending action id; first maximum wins
      if ((long)root_->cn[i] > bv) { bv = (long)root_->cn[i]; best = root_->legal[i]; }
    *best_visits = bv;
    return best;
  }

 private:
  // mcts.py BatchedMCTS._select: same order, same double arithmetic (w/n + c*sqrt(log N / n)).
  // The values are computed first (vectorised, UctValues), the arg-max scan follows.
  size_t BestChild(Node *node) {
    const size_t m = node->nlegal;
    if (node->created < m) return node->created;  // the first uncreated child scores 1e9
    const double logn = std::log(node->n());
    val_.resize(m);
    double *val = val_.data();
    UctValues(node->cn, node->cw, val, m, logn, c_);
    size_t best = 0;
    double best_v = -1e30;
    for (size_t i = 0; i < m; i++)
      if (val[i] > best_v) { best = i; best_v = val[i]; }
    return best;
  }
  Node *Select(Node *root, std::vector<int> *path) {
    Node *node = root;
    for (;;) {
      node->n() += 1.0;
      node->w() -= 1.0;  // virtual loss
      if (node->terminal != 0 || !node->expanded) return node;
      const size_t i = BestChild(node);
      if (!node->child[i]) {
        node->child[i] = new (arena_.Alloc(sizeof(Node))) Node(node, (int)i, node->legal[i], node->player);
        node->created = i + 1;
      }
      node = node->child[i];
      path->push_back(node->action);
    }
  }
  void Backup(Node *leaf, int result) {
    double r[2];
    Returns(result, r);
    for (Node *node = leaf; node; node = node->parent) {
      node->w() += 1.0;
      if (node->mover >= 0) node->w() += r[node->mover];
    }
  }
  double c_;
  Arena arena_;
  Node *root_ = nullptr;
  std::vector<double> val_;
};

// host threads for the trees: the cores this process may really use (the cgroup quota caps
// what the affinity mask shows: an oversubscribed OpenMP team spins itself to a standstill)
static int HostCores() {
  int cores = (int)std::thread::hardware_concurrency();
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32];
    long period = 0;
    if (std::fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && q[0] != 'm')
      cores = std::min(cores, std::max(1, (int)((std::atol(q) + period / 2) / period)));
    std::fclose(f);
  }
  return std::max(1, cores);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 12;
  const long sims = argc > 2 ? std::atol(argv[2]) : 20000;
  const int max_moves = argc > 3 ? std::atoi(argv[3]) : 0;
  const int leaves = argc > 4 ? std::atoi(argv[4]) : 2048;
  const uint32_t seed = argc > 5 ? (uint32_t)std::atol(argv[5]) : 1u;
  const int games = argc > 6 ? std::atoi(argv[6]) : 1;
  if (games < 1 || leaves < games) { std::fprintf(stderr, "need 1 <= games <= leaves\n"); return 1; }
  const int threads = std::min(games, HostCores());
  Device dev(n, leaves, games, threads);
  dev.ResetRoots();
  std::vector<Mcts> mcts(games);
  std::vector<int> result(games, 0), moves(games, 0);
  const auto t0 = std::chrono::steady_clock::now();
  long total = 0, batches = 0;
  for (;;) {
    const std::vector<Leaf> info = dev.RootInfos();
    std::vector<int> live;
    for (int g = 0; g < games; g++) {
      result[g] = info[g].result;
      if (result[g] == 0 && !(max_moves > 0 && moves[g] >= max_moves)) live.push_back(g);
    }
    if (live.empty()) break;
    const int nl = (int)live.size();
    {
      Timer tm(5);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (nl > 1)
      for (int j = 0; j < nl; j++) mcts[live[j]].Begin(info[live[j]].player, info[live[j]].legal);
    }
    // two lanes in flight: launch(0), launch(1), finish(0), launch(2), finish(1), ...
    struct InFlight { int lane; long k; std::vector<std::vector<Node *>> leaf; };
    std::vector<InFlight> flying;  // oldest first, at most two
    const long per_lane = std::max<long>(1, dev.LaneLeaves() / games);
    long launched = 0;
    int next_lane = 0;
    auto launch = [&]() {
      InFlight f;
      f.lane = next_lane;
      next_lane ^= 1;
      f.k = std::min<long>(per_lane, sims - launched);
      f.leaf.resize(nl);
      std::vector<std::vector<std::vector<int>>> path(nl);
      {
        Timer tm(0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (nl > 1)
        for (int j = 0; j < nl; j++) mcts[live[j]].SelectBatch(f.k, &f.leaf[j], &path[j]);
      }
      std::vector<std::vector<int>> paths;
      std::vector<int> game_of;
      {
        Timer tm(6);
        paths.reserve(nl * f.k);
        game_of.reserve(nl * f.k);
        for (int j = 0; j < nl; j++)
          for (long i = 0; i < f.k; i++) { paths.push_back(std::move(path[j][i])); game_of.push_back(live[j]); }
      }
      dev.Launch(f.lane, paths, game_of, seed + (uint32_t)batches++);
      launched += f.k;
      flying.push_back(std::move(f));
    };
    launch();
    while (!flying.empty()) {
      if (launched < sims) launch();
      InFlight f = std::move(flying.front());
      flying.erase(flying.begin());
      std::vector<Leaf> out = dev.Wait(f.lane);
      Timer tm(4);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (nl > 1)
      for (int j = 0; j < nl; j++) mcts[live[j]].ApplyBatch(f.leaf[j].data(), out.data() + (size_t)j * f.k, f.k);
    }
    std::vector<int> acts;
    for (int g : live) {
      long visits = 0;
      const int a = mcts[g].Best(&visits);
      if (games == 1) std::printf("move %d: player %d plays %d (%ld visits)\n", moves[g], info[g].player, a, visits);
      else std::printf("game %d move %d: player %d plays %d (%ld visits)\n", g, moves[g], info[g].player, a, visits);
      acts.push_back(a);
      moves[g]++;
      total += sims;
    }
    dev.AdvanceRoots(live, acts);
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int g = 0; g < games; g++)
    std::printf("%sresult %d after %d moves\n", games == 1 ? "" : ("game " + std::to_string(g) + ": ").c_str(), result[g],
                moves[g]);
  std::printf("%ld simulations in %.2f s = %.0f simulations/s (%d game%s, %d leaves per round trip, %d host thread%s)\n",
              total, dt, total / dt, games, games == 1 ? "" : "s", leaves, threads, threads == 1 ? "" : "s");
  std::printf("time: select %.2f s, queueing the lanes %.2f s, waiting for them %.2f s, results / play-outs %.2f s, "
              "backup %.2f s, new trees %.2f s, batch assembly %.2f s\n", g_t[0], g_t[1], g_t[2], g_t[3], g_t[4], g_t[5],
              g_t[6]);
  return 0;
}
