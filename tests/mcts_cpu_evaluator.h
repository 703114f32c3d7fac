// tests/mcts_cpu_evaluator.h -- TEST INFRASTRUCTURE: the batched-leaf search of
// spiel_adapter/mcts_driver.cc with its leaves evaluated by the CPU oracle instead of the
// HIP engine (build mcts_driver.cc with -DMCTS_EVALUATOR_HEADER='"mcts_cpu_evaluator.h"').
// Same interface as the driver's `Device`, same slot numbering and policy keys (leaf slot i
// plays out with board id i, episode 0), so the two builds must print the same moves and
// visit counts; timed, it is "the CPU engine doing the same search" of SURVEY 8(f3): every
// leaf is a copy of the root (Clone, tw.h:80-82), replayed along its path with
// orc_apply (tw.h:93-104) and played out with the random policy, OpenMP over the leaves.
#include <cstring>

#include "twixt_oracle.h"

class Device {
 public:
  Device(int n, int leaves, int games, int threads) : n_(n), L_(leaves), roots_(games), threads_(threads) {
    const char *t = std::getenv("MCTS_CPU_THREADS");  // leaves are independent: all host cores by default
    eval_threads_ = t ? std::atoi(t) : 0;
  }
  void ResetRoots() {
    for (auto &r : roots_) orc_init(&r, n_);
  }
  std::vector<Leaf> RootInfos() {
    std::vector<Leaf> out(roots_.size());
    for (size_t g = 0; g < roots_.size(); g++) Describe(roots_[g], &out[g]);
    return out;
  }
  void AdvanceRoots(const std::vector<int> &games, const std::vector<int> &actions) {
    for (size_t i = 0; i < games.size(); i++)
      if (orc_apply(&roots_[games[i]], actions[i]) != 0) {
        std::fprintf(stderr, "illegal root move %d\n", actions[i]);
        std::exit(1);
      }
  }
  // (the driver's rule: the lanes split the leaf slots only while each half holds a leaf of every game)
  int Games() const { return (int)roots_.size(); }
  int LaneLeaves() const { return L_ >= 2 * Games() ? L_ / 2 : L_; }
  int LaneBase(int lane) const { return L_ >= 2 * Games() ? lane * (L_ / 2) : 0; }
  // the driver's two-lane interface: here the lane is evaluated on the spot and handed over by Wait()
  void Launch(int lane, const std::vector<std::vector<int>> &paths, const std::vector<int> &game_of, uint32_t seed) {
    const int k = (int)paths.size(), base = LaneBase(lane);
    if (base + k > L_) {
      std::fprintf(stderr, "lane %d: %d leaves from slot %d do not fit %d leaf slots\n", lane, k, base, L_);
      std::exit(1);
    }
    std::vector<Leaf> &out = lane_out_[lane];
    out.assign(k, Leaf());
    Timer tm(3);
    const int nt = eval_threads_ > 0 ? eval_threads_ : HostCoresForEval();
    if (std::getenv("MCTS_CPU_DEBUG")) std::fprintf(stderr, "eval threads %d\n", nt);
#pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
    for (int i = 0; i < k; i++) {
      orc_state s;
      std::memcpy(&s, &roots_[game_of[i]], sizeof s);  // Clone
      for (int a : paths[i])
        if (orc_apply(&s, a) != 0) {
          std::fprintf(stderr, "tree path contains an illegal move\n");
          std::exit(1);
        }
      Describe(s, &out[i]);
      if (s.result == ORC_OPEN) {
        uint32_t episode = 0;
        orc_rollout(&s, seed, (uint32_t)(base + i), &episode, n_ * n_ - 3, 0, nullptr, nullptr);  // board id = leaf slot
        out[i].final_result = s.result;
      }
    }
  }
  std::vector<Leaf> Wait(int lane) { return std::move(lane_out_[lane]); }

 private:
  static int HostCoresForEval() {
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
  void Describe(const orc_state &s, Leaf *l) const {
    l->result = s.result;
    l->player = s.current_player;
    l->final_result = s.result;
    l->legal.clear();
    if (s.result == ORC_OPEN) {
      int buf[ORC_MAX_LEGAL];
      const int k = orc_legal_actions(&s, buf);
      l->legal.assign(buf, buf + k);
    }
  }
  int n_, L_;
  std::vector<orc_state> roots_;
  int threads_, eval_threads_ = 0;
  std::vector<Leaf> lane_out_[2];
};
