#pragma once
#include <optional>
namespace absl {
template <typename T>
using optional = std::optional<T>;
}
