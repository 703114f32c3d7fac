// Minimal model of the open_spiel core API surface that the twixt plug-in uses
// (SURVEY.md 8(b) lists it).  Test scaffolding for OUR adapter
// (twixt_for_open_spiel_amd/spiel_adapter): it lets the adapter be compiled and driven
// where open_spiel is not installed.  It is not open_spiel and is never used to build
// the reference.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "absl/types/optional.h"
#include "absl/types/span.h"

namespace open_spiel {

using Action = int64_t;
using Player = int;
constexpr Player kTerminalPlayerId = -4;

using ErrorHandler = void (*)(const std::string &);
inline ErrorHandler &ErrorHandlerRef() {
  static ErrorHandler h = nullptr;
  return h;
}
inline void SetErrorHandler(ErrorHandler h) { ErrorHandlerRef() = h; }
[[noreturn]] inline void SpielFatalError(const std::string &msg) {
  if (ErrorHandlerRef()) ErrorHandlerRef()(msg);
  std::cerr << "Spiel Fatal Error: " << msg << std::endl;
  std::exit(1);
}
#define SPIEL_CHECK_GE(a, b) \
  do { if (!((a) >= (b))) open_spiel::SpielFatalError(std::string("CHECK_GE failed: ") + #a + " >= " + #b); } while (0)
#define SPIEL_CHECK_LT(a, b) \
  do { if (!((a) < (b))) open_spiel::SpielFatalError(std::string("CHECK_LT failed: ") + #a + " < " + #b); } while (0)

class GameParameter {
 public:
  GameParameter() = default;
  explicit GameParameter(int v, bool mandatory = false) : is_int_(true), int_(v) { (void)mandatory; }
  explicit GameParameter(bool v, bool mandatory = false) : is_bool_(true), bool_(v) { (void)mandatory; }
  template <typename T>
  T value() const;
 private:
  bool is_int_ = false, is_bool_ = false;
  int int_ = 0;
  bool bool_ = false;
};
template <>
inline int GameParameter::value<int>() const { return int_; }
template <>
inline bool GameParameter::value<bool>() const { return bool_; }
using GameParameters = std::map<std::string, GameParameter>;

struct GameType {
  std::string short_name, long_name;
  enum class Dynamics { kSimultaneous, kSequential };
  Dynamics dynamics;
  enum class ChanceMode { kDeterministic, kExplicitStochastic, kSampledStochastic };
  ChanceMode chance_mode;
  enum class Information { kOneShot, kPerfectInformation, kImperfectInformation };
  Information information;
  enum class Utility { kZeroSum, kConstantSum, kGeneralSum, kIdentical };
  Utility utility;
  enum class RewardModel { kRewards, kTerminal };
  RewardModel reward_model;
  int max_num_players, min_num_players;
  bool provides_information_state_string, provides_information_state_tensor, provides_observation_string,
      provides_observation_tensor;
  GameParameters parameter_specification;
};

class State;

class Game : public std::enable_shared_from_this<Game> {
 public:
  Game(GameType type, GameParameters params) : type_(std::move(type)), params_(std::move(params)) {}
  virtual ~Game() = default;
  virtual std::unique_ptr<State> NewInitialState() const = 0;
  virtual int NumDistinctActions() const = 0;
  virtual int NumPlayers() const = 0;
  virtual double MinUtility() const = 0;
  virtual double MaxUtility() const = 0;
  virtual absl::optional<double> UtilitySum() const { return absl::optional<double>(); }
  virtual std::vector<int> ObservationTensorShape() const = 0;
  virtual int MaxGameLength() const = 0;
  const GameType &GetType() const { return type_; }
  template <typename T>
  T ParameterValue(const std::string &key, T dflt) const {
    auto it = params_.find(key);
    return it == params_.end() ? dflt : it->second.value<T>();
  }
 private:
  GameType type_;
  GameParameters params_;
};

class State {
 public:
  explicit State(std::shared_ptr<const Game> game) : game_(std::move(game)) {}
  State(const State &) = default;
  virtual ~State() = default;
  virtual Player CurrentPlayer() const = 0;
  virtual std::string ActionToString(Player, Action) const = 0;
  virtual std::string ToString() const = 0;
  virtual bool IsTerminal() const = 0;
  virtual std::vector<double> Returns() const = 0;
  virtual std::string InformationStateString(Player) const = 0;
  virtual std::string ObservationString(Player) const = 0;
  virtual void ObservationTensor(Player, absl::Span<float>) const = 0;
  virtual std::unique_ptr<State> Clone() const = 0;
  virtual void UndoAction(Player, Action) {}
  virtual std::vector<Action> LegalActions() const = 0;
  void ApplyAction(Action a) {
    DoApplyAction(a);
    history_.push_back(a);
  }
  double PlayerReturn(Player p) const { return Returns()[p]; }
  const std::vector<Action> &History() const { return history_; }
  std::shared_ptr<const Game> GetGame() const { return game_; }
 protected:
  virtual void DoApplyAction(Action) = 0;
 private:
  std::shared_ptr<const Game> game_;
  std::vector<Action> history_;
};

using GameFactory = std::function<std::unique_ptr<Game>(const GameParameters &)>;
struct Registry {
  static std::map<std::string, std::pair<GameType, GameFactory>> &Games() {
    static std::map<std::string, std::pair<GameType, GameFactory>> g;
    return g;
  }
};
struct GameRegisterer {
  GameRegisterer(const GameType &t, GameFactory f) { Registry::Games()[t.short_name] = {t, std::move(f)}; }
};
#define REGISTER_SPIEL_GAME(type, factory) static ::open_spiel::GameRegisterer registerer_##factory(type, factory)

inline std::shared_ptr<const Game> LoadGame(const std::string &name, const GameParameters &params = {}) {
  auto it = Registry::Games().find(name);
  if (it == Registry::Games().end()) SpielFatalError("Unknown game '" + name + "'.");
  const GameParameters &spec = it->second.first.parameter_specification;
  for (const auto &kv : params) {
    if (spec.find(kv.first) == spec.end()) {
      std::string avail;
      for (const auto &s : spec) avail += (avail.empty() ? "" : ", ") + s.first;
      SpielFatalError("Unknown parameter '" + kv.first + "'. Available parameters are: " + avail);
    }
  }
  return std::shared_ptr<const Game>(it->second.second(params));
}

}  // namespace open_spiel
