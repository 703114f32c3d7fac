// twixt.h -- open_spiel game plug-in "twixt" on top of libtwixt_hip.so.
//
// Drop-in for the reference's open_spiel/games/twixt/twixt.h + twixt.cc
// (stevens68/TwixT_for_open_spiel): same registration, parameters, State / Game
// virtuals, strings and error texts; twixtboard.{h,cc} and twixtcell.h are replaced by a
// slot in a device batch reached through the C ABI (include/twixt_hip.h).  Every member
// cites the reference member it mirrors (tw.h = twixt.h, tw.cc = twixt.cc).
//
// One-state-at-a-time calls (example, mcts_example, pyspiel) work unchanged but cost one
// kernel launch each; throughput callers use the batched C ABI directly.
#ifndef OPEN_SPIEL_GAMES_TWIXT_TWIXT_H_
#define OPEN_SPIEL_GAMES_TWIXT_TWIXT_H_

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "open_spiel/spiel.h"
#include "twixt_hip.h"

namespace open_spiel {
namespace twixt {

inline constexpr int kNumPlayers = 2;                 // twixtcell.h:54
inline constexpr Player kRedPlayer = 0, kBluePlayer = 1;  // twixtcell.h:52-53
inline constexpr int kNumPlanes = 12;                 // twixtboard.h:46
inline constexpr int kMinBoardSize = 5, kMaxBoardSize = 24, kDefaultBoardSize = 8;  // twixtboard.h:32-34
inline constexpr bool kDefaultAnsiColorOutput = true;  // twixtboard.h:36

// Pool of board slots inside one twx_env, shared by all states of a game object.
// Threading: open_spiel's MCTS and evaluators clone and use States from several threads.  The
// pool has ONE set of scratch buffers and a twx_env is used by one thread at a time
// (include/twixt_hip.h), so every method takes the pool's mutex: States of one game may be
// used from any thread, their device calls are serialised.  The pool lives on HIP device
// TWX_ADAPTER_DEVICE (default 0) and holds TWX_ADAPTER_SLOTS live states (default 4096).
class DeviceBoards {
 public:
  DeviceBoards(int board_size, int capacity, int device = 0);
  ~DeviceBoards();
  DeviceBoards(const DeviceBoards &) = delete;
  DeviceBoards &operator=(const DeviceBoards &) = delete;

  int32_t Acquire();              // a slot in the initial position
  void Release(int32_t slot);
  void Copy(int32_t dst, int32_t src);                       // Clone
  uint8_t Step(int32_t slot, Action action);                 // status byte of twx_step_boards
  std::vector<Action> LegalActions(int32_t slot);            // ascending
  void Observation(int32_t slot, float *out);                // 12*N*(N-2) floats
  std::string ToString(int32_t slot, bool ansi, bool swapped, int result);
  int board_size() const { return n_; }

 private:
  int n_, capacity_, device_, mask_words_;
  std::mutex mu_;
  twx_env *env_ = nullptr;
  std::vector<int32_t> free_;
  // One block of pinned host memory that the device reads and writes in place (hipHostMallocMapped):
  // board ids and the action go in, status / legal mask / tensor / cells come out, so a State call is
  // one launch and one stream synchronisation with no staging copies.
  struct Scratch {
    int32_t ids[2];
    int32_t action;
    uint8_t status[4];
  };
  void *pin_ = nullptr;            // the block as the host sees it
  char *dev_ = nullptr;            // ... and as the device does
  size_t off_mask_ = 0, off_obs_ = 0, off_cells_ = 0;
  Scratch *scratch() { return static_cast<Scratch *>(pin_); }
  template <typename T> T *host_at(size_t off) { return reinterpret_cast<T *>(static_cast<char *>(pin_) + off); }
  template <typename T> T *dev_at(size_t off) { return reinterpret_cast<T *>(dev_ + off); }
  void Finish();                   // waits for the launch (default stream) that fills the block
};

class TwixTState : public State {
 public:
  explicit TwixTState(std::shared_ptr<const Game> game);  // tw.cc:62-65
  TwixTState(const TwixTState &other);                    // tw.h:35 (Clone: a new slot)
  TwixTState &operator=(const TwixTState &) = delete;
  ~TwixTState() override;

  Player CurrentPlayer() const override { return current_player_; }  // tw.h:38
  std::string ActionToString(Player player, Action action) const override;  // tw.cc:67-74
  std::string ToString() const override;                                 // tw.h:43
  bool IsTerminal() const override { return result_ != 0; }              // tw.h:45-48
  std::vector<double> Returns() const override;                          // tw.h:50-63
  std::string InformationStateString(Player player) const override;     // tw.h:65-69
  std::string ObservationString(Player player) const override;          // tw.h:71-75
  void ObservationTensor(Player player, absl::Span<float> values) const override;  // tw.cc:101-132
  std::unique_ptr<State> Clone() const override;                         // tw.h:80-82
  void UndoAction(Player, Action) override {}                            // tw.h:84
  std::vector<Action> LegalActions() const override;                     // tw.h:86-90

 protected:
  void DoApplyAction(Action action) override;                            // tw.h:93-104

 private:
  std::shared_ptr<DeviceBoards> boards_;
  int32_t slot_;
  Player current_player_ = kRedPlayer;  // tw.h:107
  int result_ = 0;
  bool swapped_ = false;
  bool ansi_color_output_;
};

class TwixTGame : public Game {
 public:
  explicit TwixTGame(const GameParameters &params);  // tw.cc:134-145

  std::unique_ptr<State> NewInitialState() const override {  // tw.h:118-120
    return std::unique_ptr<State>(new TwixTState(shared_from_this()));
  }
  int NumDistinctActions() const override { return board_size_ * board_size_; }  // tw.h:122-124
  int NumPlayers() const override { return kNumPlayers; }
  double MinUtility() const override { return -1.0; }
  absl::optional<double> UtilitySum() const override { return 0.0; }
  double MaxUtility() const override { return 1.0; }
  std::vector<int> ObservationTensorShape() const override {  // tw.h:131-134 (without the static)
    return {kNumPlanes, board_size_, board_size_ - 2};
  }
  int MaxGameLength() const override { return board_size_ * board_size_ - 4 + 1; }  // tw.h:136-139
  bool ansi_color_output() const { return ansi_color_output_; }
  int board_size() const { return board_size_; }
  std::shared_ptr<DeviceBoards> boards() const;  // created on first use

 private:
  bool ansi_color_output_;
  int board_size_;
  mutable std::once_flag boards_once_;
  mutable std::shared_ptr<DeviceBoards> boards_;
};

}  // namespace twixt
}  // namespace open_spiel

#endif  // OPEN_SPIEL_GAMES_TWIXT_TWIXT_H_
