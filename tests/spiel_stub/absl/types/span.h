// Minimal stand-in for absl::Span used to compile-check OUR adapter where open_spiel /
// abseil are not installed (tests only; never used to build the reference).
#pragma once
#include <cstddef>
#include <vector>
namespace absl {
template <typename T>
class Span {
 public:
  Span(T *p, size_t n) : p_(p), n_(n) {}
  Span(std::vector<T> &v) : p_(v.data()), n_(v.size()) {}
  T *data() const { return p_; }
  size_t size() const { return n_; }
  T &operator[](size_t i) const { return p_[i]; }
 private:
  T *p_;
  size_t n_;
};
template <typename T>
Span<T> MakeSpan(std::vector<T> &v) { return Span<T>(v); }
}  // namespace absl
