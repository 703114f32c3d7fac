/* Runs the oracle's entry points under AddressSanitizer + UndefinedBehaviorSanitizer
 * (CPU build only; `make -C oracle sanitize`).  Test infrastructure: the oracle is the
 * checker of the HIP engine, so it must itself be free of out-of-bounds and undefined
 * arithmetic.  Exercises every board size, the KAT games (plain and forced swap), text,
 * observation tensor, exports and the policy roll-outs with auto-reset. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "twixt_oracle.h"

int main(void) {
  static orc_state st;
  static float obs[12 * 24 * 22];
  static char text[1 << 16];
  static uint8_t cells[24 * 24 * 4];
  static uint32_t planes[8 * 18 + 4];
  int la[ORC_MAX_LEGAL];
  uint64_t sum = 0;
  for (int n = 5; n <= 24; n++) {
    for (int forced = 0; forced < 2; forced++) {
      orc_kat k;
      orc_kat_game(n, (uint64_t)(n * 7 + forced), forced, &k);
      sum += k.hist_fnv ^ k.step_fnv ^ k.final_fnv;
    }
    orc_init(&st, n);
    uint64_t rng = (uint64_t)n;
    while (!orc_is_terminal(&st)) {
      const int cnt = orc_legal_actions(&st, la);
      rng = rng * 6364136223846793005ULL + 1442695040888963407ULL;
      const int a = la[(rng >> 33) % (uint64_t)cnt];
      char as[32];
      orc_action_to_string(&st, st.current_player, a, as);
      if (orc_apply(&st, a) != 0) { fprintf(stderr, "legal action refused\n"); return 1; }
      if (orc_apply(&st, a) == 0 && st.move_counter != 2) { fprintf(stderr, "occupied cell accepted\n"); return 1; }
      orc_observation(&st, obs);
      sum += (uint64_t)orc_to_string(&st, (int)(rng & 1), text, (int)sizeof text);
      orc_export_cells(&st, cells);
      orc_export_planes(&st, planes);
      sum += planes[0] + cells[5] + (uint64_t)obs[3];
    }
    if (orc_apply(&st, -1) == 0 || orc_apply(&st, n * n) == 0) { fprintf(stderr, "out of range accepted\n"); return 1; }
    sum += (uint64_t)orc_rollout_batch(n, 11u, 1000u, 64, n * n, 1, 2, NULL, NULL, NULL, NULL);
    uint64_t dg = 0;
    sum += (uint64_t)orc_bench(n, 3u, 128, 40, 2, &dg) + dg;
  }
  printf("sanitize_check ok %llu\n", (unsigned long long)sum);
  return 0;
}
