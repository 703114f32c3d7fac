// Host text renderer (csrc/twixt_text.h, product code) under ASan + UBSan: every board size,
// ANSI on/off, every result, with cell bytes of the shape twx_export_cells produces
// ({colour 0..3, links, blocked, flags}; corners off-board) but arbitrary link bits, so the
// neighbour probes at the board's edges are exercised.  tests/test_text_sanitizers.py.
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../twixt_for_open_spiel_amd/csrc/twixt_text.h"

int main() {
  uint64_t rng = 12345, sum = 0;
  auto next = [&]() { rng = rng * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(rng >> 33); };
  for (int n = 5; n <= 24; n++) {
    std::vector<uint8_t> cells((size_t)n * n * 4);
    for (int rep = 0; rep < 8; rep++) {
      for (int x = 0; x < n; x++)
        for (int y = 0; y < n; y++) {
          uint8_t *c = &cells[(size_t)(x * n + y) * 4];
          const bool corner = (x == 0 || x == n - 1) && (y == 0 || y == n - 1);
          c[0] = corner ? 3 : (uint8_t)(next() % 3);
          c[1] = (uint8_t)next();
          c[2] = (uint8_t)next();
          c[3] = (uint8_t)(next() & 15);
        }
      for (int ansi = 0; ansi < 2; ansi++)
        for (int result = 0; result < 4; result++) {
          const std::string s = twx::text::BoardToString(cells.data(), n, ansi != 0, (rep & 1) != 0, result);
          sum += s.size();
        }
    }
    for (int a = 0; a < n * n; a++)
      for (int p = 0; p < 2; p++) sum += twx::text::ActionToString(n, p, a).size();
  }
  std::printf("text_sanitize ok %llu\n", (unsigned long long)sum);
  return 0;
}
