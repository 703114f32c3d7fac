// twixt.cc -- see twixt.h.  Host C++ only; all board work goes through the C ABI.
#include "twixt.h"

#include <hip/hip_runtime_api.h>

#include <cstdlib>
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
  Hip(hipMalloc(&d_ids_, 4));
  Hip(hipMalloc(&d_ids2_, 4));
  Hip(hipMalloc(&d_action_, 4));
  Hip(hipMalloc(&d_status_, 4));
  Hip(hipMalloc(&d_mask_, 4 * mask_words_ + 4));
  Hip(hipMalloc(&d_obs_, sizeof(float) * kNumPlanes * n_ * (n_ - 2)));
  Hip(hipMalloc(&d_cells_, 4 * n_ * n_));
}

DeviceBoards::~DeviceBoards() {
  OnDevice guard(device_);
  for (void *p : {d_ids_, d_ids2_, d_action_, d_status_, d_mask_, d_obs_, d_cells_})
    if (p) (void)hipFree(p);
  twx_destroy(env_);
}

void DeviceBoards::PutId(void *dst, int32_t slot) { Hip(hipMemcpy(dst, &slot, 4, hipMemcpyHostToDevice)); }

int32_t DeviceBoards::Acquire() {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  if (free_.empty())
    SpielFatalError("twixt: out of device board slots (raise TWX_ADAPTER_SLOTS, now " +
                    std::to_string(capacity_) + ")");
  const int32_t slot = free_.back();
  free_.pop_back();
  PutId(d_ids_, slot);
  Twx(twx_reset_boards(env_, static_cast<const int32_t *>(d_ids_), 1, nullptr));  // Board::Board
  return slot;
}

void DeviceBoards::Release(int32_t slot) {
  std::lock_guard<std::mutex> lock(mu_);
  free_.push_back(slot);
}

void DeviceBoards::Copy(int32_t dst, int32_t src) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  PutId(d_ids_, dst);
  PutId(d_ids2_, src);
  Twx(twx_copy_boards(env_, static_cast<const int32_t *>(d_ids_), env_, static_cast<const int32_t *>(d_ids2_), 1,
                      nullptr));
}

uint8_t DeviceBoards::Step(int32_t slot, Action action) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  PutId(d_ids_, slot);
  // an action outside int32 can never be legal: any out-of-range value reports "illegal"
  const int32_t a = (action < 0 || action > 0x7fffffff) ? -1 : static_cast<int32_t>(action);
  Hip(hipMemcpy(d_action_, &a, 4, hipMemcpyHostToDevice));
  Twx(twx_step_boards(env_, static_cast<const int32_t *>(d_ids_), 1, static_cast<const int32_t *>(d_action_),
                      static_cast<uint8_t *>(d_status_), nullptr));
  uint8_t st = 0;
  Hip(hipMemcpy(&st, d_status_, 1, hipMemcpyDeviceToHost));
  return st;
}

std::vector<Action> DeviceBoards::LegalActions(int32_t slot) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  PutId(d_ids_, slot);
  Twx(twx_legal_mask_boards(env_, static_cast<const int32_t *>(d_ids_), 1, static_cast<uint32_t *>(d_mask_),
                            nullptr, nullptr));
  std::vector<uint32_t> mask(mask_words_);
  Hip(hipMemcpy(mask.data(), d_mask_, 4 * mask_words_, hipMemcpyDeviceToHost));
  std::vector<Action> out;
  for (int a = 0; a < n_ * n_; ++a)  // ascending, like the reference's lists (twixtboard.cc:260-275)
    if ((mask[a >> 5] >> (a & 31)) & 1u) out.push_back(a);
  return out;
}

void DeviceBoards::Observation(int32_t slot, float *out) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  PutId(d_ids_, slot);
  Twx(twx_observation_boards(env_, static_cast<const int32_t *>(d_ids_), 1, static_cast<float *>(d_obs_), nullptr));
  Hip(hipMemcpy(out, d_obs_, sizeof(float) * kNumPlanes * n_ * (n_ - 2), hipMemcpyDeviceToHost));
}

std::string DeviceBoards::ToString(int32_t slot, bool ansi, bool swapped, int result) {
  std::lock_guard<std::mutex> lock(mu_);
  OnDevice guard(device_);
  PutId(d_ids_, slot);
  Twx(twx_export_cells(env_, static_cast<const int32_t *>(d_ids_), 1, static_cast<uint8_t *>(d_cells_), nullptr));
  std::vector<uint8_t> cells(4 * n_ * n_);
  Hip(hipMemcpy(cells.data(), d_cells_, cells.size(), hipMemcpyDeviceToHost));
  std::string s(twx_cells_to_string(cells.data(), n_, ansi, swapped, result, nullptr, 0) + 1, '\0');
  twx_cells_to_string(cells.data(), n_, ansi, swapped, result, &s[0], static_cast<int>(s.size()));
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
