// twixt.cc -- see twixt.h.  Host C++ only; all board work goes through the C ABI.
#include "twixt.h"

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "open_spiel/spiel_utils.h"

namespace open_spiel {
namespace twixt {
namespace {

// Facts about the game: identical to the reference's kGameType (tw.cc:35-52).
const GameType kGameType{
    /*short_name=*/"twixt",
    /*long_name=*/"TwixT",
    GameType::Dynamics::kSequential,
    GameType::ChanceMode::kDeterministic,
    GameType::Information::kPerfectInformation,
    GameType::Utility::kZeroSum,
    GameType::RewardModel::kTerminal,
    /*max_num_players=*/2,
    /*min_num_players=*/2,
    /*provides_information_state_string=*/true,
    /*provides_information_state_tensor=*/false,
    /*provides_observation_string=*/true,
    /*provides_observation_tensor=*/true,
    /*parameter_specification=*/
    {{"board_size", GameParameter(kDefaultBoardSize)},
     {"ansi_color_output", GameParameter(kDefaultAnsiColorOutput)}},
};

std::unique_ptr<Game> Factory(const GameParameters &params) {  // tw.cc:54-56
  return std::unique_ptr<Game>(new TwixTGame(params));
}

REGISTER_SPIEL_GAME(kGameType, Factory);  // tw.cc:58

void Twx(int rc) {
  if (rc != TWX_OK) SpielFatalError(std::string("twixt_hip: ") + twx_last_error());
}
void Hip(hipError_t e) {
  if (e != hipSuccess) SpielFatalError(std::string("twixt_hip: ") + hipGetErrorString(e));
}

// the pool's scratch buffers live on the pool's device, whatever the caller's current one is
class OnDevice {
 public:
  explicit OnDevice(int device) {
    Hip(hipGetDevice(&prev_));
    if (prev_ != device) {
      Hip(hipSetDevice(device));
      changed_ = true;
    }
  }
  ~OnDevice() {
    if (changed_) (void)hipSetDevice(prev_);
  }

 private:
  int prev_ = 0;
  bool changed_ = false;
};

}  // namespace

// ---------------------------------------------------------------- DeviceBoards
DeviceBoards::DeviceBoards(int board_size, int capacity, int device)
    : n_(board_size), capacity_(capacity), device_(device) {
  OnDevice guard(device_);
  Twx(twx_create(board_size, capacity, device_, /*first_board_id=*/0, &env_));
  mask_words_ = twx_mask_words(board_size);
  free_.reserve(capacity);
  for (int i = capacity - 1; i >= 0; --i) free_.push_back(i);
  auto up = [](size_t v) { return (v + 63) / 64 * 64; };
  off_mask_ = up(sizeof(Scratch));
  off_obs_ = off_mask_ + up(4 * (size_t)mask_words_);
  off_cells_ = off_obs_ + up(sizeof(float) * kNumPlanes * n_ * (n_ - 2));
  const size_t bytes = off_cells_ + up(4 * (size_t)n_ * n_);
  Hip(hipHostMalloc(&pin_, bytes, hipHostMallocMapped));
  void *d = nullptr;
  Hip(hipHostGetDevicePointer(&d, pin_, 0));
  dev_ = static_cast<char *>(d);
}

DeviceBoards::~DeviceBoards() {
  OnDevice guard(device_);
  twx_destroy(env_);
  if (pin_) (void)hipHostFree(pin_);
}

void DeviceBoards::Finish() { Hip(hipStreamSynchronize(nullptr)); }

int32_t DeviceBoards::Acquire() {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  if (free_.empty())
    SpielFatalError("twixt: out of device board slots (raise TWX_ADAPTER_SLOTS, now " +
                    std::to_string(capacity_) + ")");
  const int32_t slot = free_.back();
  free_.pop_back();
  scratch()->ids[0] = slot;
  Twx(twx_reset_boards(env_, dev_at<int32_t>(offsetof(Scratch, ids)), 1, nullptr));  // Board::Board
  Finish();  // (the id is read by the launch: the block must not change under it)
  return slot;
}

void DeviceBoards::Release(int32_t slot) {
  std::lock_guard<std::mutex> lock(mu_);
  free_.push_back(slot);
}

void DeviceBoards::Copy(int32_t dst, int32_t src) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  scratch()->ids[0] = dst;
  scratch()->ids[1] = src;
  const int32_t *ids = dev_at<int32_t>(offsetof(Scratch, ids));
  Twx(twx_copy_boards(env_, ids, env_, ids + 1, 1, nullptr));
  Finish();
}

uint8_t DeviceBoards::Step(int32_t slot, Action action) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  scratch()->ids[0] = slot;
  // an action outside int32 can never be legal: any out-of-range value reports "illegal"
  scratch()->action = (action < 0 || action > 0x7fffffff) ? -1 : static_cast<int32_t>(action);
  Twx(twx_step_boards(env_, dev_at<int32_t>(offsetof(Scratch, ids)), 1, dev_at<int32_t>(offsetof(Scratch, action)),
                      dev_at<uint8_t>(offsetof(Scratch, status)), nullptr));
  Finish();
  return scratch()->status[0];
}

std::vector<Action> DeviceBoards::LegalActions(int32_t slot) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  scratch()->ids[0] = slot;
  Twx(twx_legal_mask_boards(env_, dev_at<int32_t>(offsetof(Scratch, ids)), 1, dev_at<uint32_t>(off_mask_), nullptr,
                            nullptr));
  Finish();
  const uint32_t *mask = host_at<uint32_t>(off_mask_);
  std::vector<Action> out;
  for (int w = 0; w < mask_words_; ++w)  // ascending, like the reference's lists (twixtboard.cc:260-275)
    for (uint32_t m = mask[w]; m; m &= m - 1) out.push_back(32 * w + __builtin_ctz(m));
  return out;
}

void DeviceBoards::Observation(int32_t slot, float *out) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  scratch()->ids[0] = slot;
  Twx(twx_observation_boards(env_, dev_at<int32_t>(offsetof(Scratch, ids)), 1, dev_at<float>(off_obs_), nullptr));
  Finish();
  std::memcpy(out, host_at<float>(off_obs_), sizeof(float) * kNumPlanes * n_ * (n_ - 2));
}

std::string DeviceBoards::ToString(int32_t slot, bool ansi, bool swapped, int result) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  scratch()->ids[0] = slot;
  Twx(twx_export_cells(env_, dev_at<int32_t>(offsetof(Scratch, ids)), 1, dev_at<uint8_t>(off_cells_), nullptr));
  Finish();
  const uint8_t *cells = host_at<uint8_t>(off_cells_);
  std::string s(twx_cells_to_string(cells, n_, ansi, swapped, result, nullptr, 0) + 1, '\0');
  twx_cells_to_string(cells, n_, ansi, swapped, result, &s[0], static_cast<int>(s.size()));
  s.pop_back();
  return s;
}

// ------------------------------------------------------------------ TwixTState
TwixTState::TwixTState(std::shared_ptr<const Game> game) : State(game) {
  const TwixTGame &parent_game = static_cast<const TwixTGame &>(*game);
  boards_ = parent_game.boards();
  ansi_color_output_ = parent_game.ansi_color_output();
  slot_ = boards_->Acquire();
}

TwixTState::TwixTState(const TwixTState &other)
    : State(other),
      boards_(other.boards_),
      slot_(other.boards_->Acquire()),
      current_player_(other.current_player_),
      result_(other.result_),
      swapped_(other.swapped_),
      ansi_color_output_(other.ansi_color_output_) {
  boards_->Copy(slot_, other.slot_);
}

TwixTState::~TwixTState() { boards_->Release(slot_); }

std::string TwixTState::ActionToString(Player player, Action action) const {
  char buf[16] = {0};
  const int n = boards_->board_size();
  if (twx_action_to_string(n, player, action, buf, sizeof buf) >= 0) return buf;
  // outside [0, N*N): the reference formats whatever position the id decodes to (tw.cc:67-74)
  std::string s = (player == kRedPlayer) ? "x" : "o";
  s += static_cast<char>(static_cast<int>('a') + static_cast<int>(action / n));
  s.append(std::to_string(n - static_cast<int>(action % n)));
  return s;
}

std::string TwixTState::ToString() const {
  return boards_->ToString(slot_, ansi_color_output_, swapped_, result_);
}

std::vector<double> TwixTState::Returns() const {
  if (result_ == 1) return {1.0, -1.0};
  if (result_ == 2) return {-1.0, 1.0};
  return {0.0, 0.0};
}

std::string TwixTState::InformationStateString(Player player) const {
  SPIEL_CHECK_GE(player, 0);
  SPIEL_CHECK_LT(player, kNumPlayers);
  return ToString();
}

std::string TwixTState::ObservationString(Player player) const {
  SPIEL_CHECK_GE(player, 0);
  SPIEL_CHECK_LT(player, kNumPlayers);
  return ToString();
}

void TwixTState::ObservationTensor(Player player, absl::Span<float> values) const {
  SPIEL_CHECK_GE(player, 0);
  SPIEL_CHECK_LT(player, kNumPlayers);
  const int n = boards_->board_size();
  SPIEL_CHECK_GE(static_cast<int>(values.size()), kNumPlanes * n * (n - 2));
  boards_->Observation(slot_, values.data());
}

std::unique_ptr<State> TwixTState::Clone() const { return std::unique_ptr<State>(new TwixTState(*this)); }

std::vector<Action> TwixTState::LegalActions() const {
  if (IsTerminal()) return {};
  return boards_->LegalActions(slot_);
}

void TwixTState::DoApplyAction(Action action) {
  const uint8_t st = boards_->Step(slot_, action);
  if (st & TWX_ST_ILLEGAL) SpielFatalError("Not a legal action: " + std::to_string(action));  // tw.h:95-97
  result_ = st & TWX_ST_RESULT_MASK;
  swapped_ = (st & TWX_ST_SWAPPED) != 0;
  current_player_ = result_ == 0 ? 1 - current_player_ : kTerminalPlayerId;  // tw.h:99-103
}

// ------------------------------------------------------------------- TwixTGame
TwixTGame::TwixTGame(const GameParameters &params)
    : Game(kGameType, params),
      ansi_color_output_(ParameterValue<bool>("ansi_color_output", kDefaultAnsiColorOutput)),
      board_size_(ParameterValue<int>("board_size", kDefaultBoardSize)) {
  if (board_size_ < kMinBoardSize || board_size_ > kMaxBoardSize) {  // tw.cc:139-144
    SpielFatalError("board_size out of range [" + std::to_string(kMinBoardSize) + ".." +
                    std::to_string(kMaxBoardSize) + "]: " + std::to_string(board_size_));
  }
}

std::shared_ptr<DeviceBoards> TwixTGame::boards() const {
  std::call_once(boards_once_, [this] {
    const char *v = std::getenv("TWX_ADAPTER_SLOTS");
    const int capacity = (v && std::atoi(v) > 0) ? std::atoi(v) : 4096;
    const char *d = std::getenv("TWX_ADAPTER_DEVICE");
    boards_ = std::make_shared<DeviceBoards>(board_size_, capacity, d ? std::atoi(d) : 0);
  });
  return boards_;
}

}  // namespace twixt
}  // namespace open_spiel
