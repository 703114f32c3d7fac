// Host-only dump of the compile-time tables in twixt_geom.h (the ones the HIP
// kernels use), so that tests/test_geometry.py can compare them with the oracle.
#include <cstdio>

#include "../twixt_for_open_spiel_amd/csrc/twixt_geom.h"

template <int N>
void dump() {
  using G = twx::Geo<N>;
  std::printf("N %d %d %d\n", N, G::W, G::SW);
  for (int p = 0; p < 2; p++) {
    std::printf("region %d", p);
    for (int i = 0; i < G::W; i++) std::printf(" %u", G::tab.region[p][i]);
    std::printf("\n");
  }
  std::printf("init");
  for (int i = 0; i < G::SW; i++) std::printf(" %u", G::tab.init[i]);
  std::printf("\n");
  if constexpr (N > 5) dump<N - 1>();
}

int main() {
  for (int d = 0; d < 8; d++) {
    std::printf("cross %d", d);
    for (int i = 0; i < twx::kWinCols; i++) std::printf(" %u", twx::kCross.m[d][i]);
    std::printf("\n");
  }
  dump<24>();
  return 0;
}
