// tests/oracle_invariants.cc -- independent pin of the CPU oracle on every board size
// (re-expressed; test infrastructure).
//
// The reference holds per-step fixtures only for N=8 (playthrough.txt) and N=5/8 answers
// (twixt_test.cc); above that the oracle would rest on the SURVEY KAT integers alone.  This
// program closes the gap without the reference: it plays random games on the oracle and, after
// EVERY move, recomputes the observable state from first principles with code that shares
// nothing with oracle/twixt_oracle.c (no blocker table, no incremental flags, no legal lists):
//
//  (i)   border flags (tb.cc:538-546, 557-588): every peg of the mover's colour has START / END
//        exactly when a BFS over the link bits from it reaches a peg on the owner's START / END
//        line (checked while the game is open: SURVEY A.6);
//  (ii)  result (tb.cc:192-207): win iff the new peg's link component touches both of the
//        mover's lines, else draw iff the opponent has no legal cell, else open;
//  (iii) legal lists (tb.cc:252-276, 450-455, 478-492, 633-640; tw.h:86-90) = region & ~occupied
//        in ascending order, with the move-0/1 exceptions of SURVEY A.4;
//  (iv)  link scan (tb.cc:501-555): for each of the 8 knight neighbours of the new peg that
//        carries the mover's colour, the link exists iff NO existing link segment properly
//        crosses it, else both ends get the blocked bit -- "properly crosses" is evaluated on
//        integer segment geometry (twixt_geom.h::crosses), not on the 72-entry table;
//  (v)   link bits are symmetric, join same-colour pegs only, and at the end of a game no two
//        links on the board cross.
//
// usage: oracle_invariants [games_scale]   exit code 0 = all invariants held.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../oracle/twixt_oracle.h"
#include "../twixt_for_open_spiel_amd/csrc/twixt_geom.h"

namespace {

struct Rng {
  uint64_t s;
  uint32_t next() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (uint32_t)(s >> 20);
  }
};

long g_checks[6];

#define REQUIRE(cond, ...)                                  \
  do {                                                      \
    if (!(cond)) {                                          \
      fprintf(stderr, "INVARIANT FAILED (%s:%d): ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);                         \
      fprintf(stderr, "\n");                                \
      return false;                                         \
    }                                                       \
  } while (0)

bool on_board(int n, int x, int y) { return twx::on_board(n, x, y); }

// owner's border lines: red (0) y = 0 / n-1, blue (1) x = 0 / n-1   (tb.cc:223-231)
bool on_line(int n, int player, int border, int x, int y) {
  const int v = player == 0 ? y : x;
  return border == 0 ? v == 0 : v == n - 1;
}

// component labels of `colour` pegs over the link bits; for each label whether it touches the
// owner's START / END line
struct Components {
  int label[ORC_MAX_N][ORC_MAX_N];
  std::vector<int> touch;  // bit0 START, bit1 END
  void build(const orc_state &s, int colour) {
    const int n = s.n;
    touch.clear();
    for (int x = 0; x < n; x++)
      for (int y = 0; y < n; y++) label[x][y] = -1;
    std::vector<int> stack;
    for (int x0 = 0; x0 < n; x0++)
      for (int y0 = 0; y0 < n; y0++) {
        if (s.cell[x0][y0].color != colour || label[x0][y0] >= 0) continue;
        const int id = (int)touch.size();
        touch.push_back(0);
        label[x0][y0] = id;
        stack.assign(1, x0 * 32 + y0);
        while (!stack.empty()) {
          const int x = stack.back() / 32, y = stack.back() % 32;
          stack.pop_back();
          if (on_line(n, colour, 0, x, y)) touch[id] |= 1;
          if (on_line(n, colour, 1, x, y)) touch[id] |= 2;
          for (int d = 0; d < 8; d++)
            if (s.cell[x][y].links & (1 << d)) {
              const int tx = x + twx::kDX[d], ty = y + twx::kDY[d];
              if (on_board(n, tx, ty) && label[tx][ty] < 0) {
                label[tx][ty] = id;
                stack.push_back(tx * 32 + ty);
              }
            }
        }
      }
  }
};

// (iii) the legal set of `player` derived from the cells alone, ascending action ids
void derived_legal(const orc_state &s, int player, std::vector<int> *out) {
  const int n = s.n;
  out->clear();
  if (s.result != ORC_OPEN) return;  // tw.h:87-88
  for (int x = 0; x < n; x++)
    for (int y = 0; y < n; y++) {
      if (!twx::in_region(n, player, x, y)) continue;
      const int c = s.cell[x][y].color;
      bool free = c == ORC_EMPTY;
      // move 1: red's first peg stays in blue's list -- choosing it is the swap (tb.cc:485-488)
      if (s.move_counter == 1 && player == 1 && c == ORC_RED) free = true;
      if (free) out->push_back(x * n + y);
    }
}

// does any link on the board properly cross the segment p -> p + off[dir]?
bool crossed(const orc_state &s, int px, int py, int dir) {
  const int n = s.n;
  for (int fx = px - 4; fx <= px + 4; fx++)
    for (int fy = py - 4; fy <= py + 4; fy++) {
      if (!on_board(n, fx, fy)) continue;
      for (int e = 0; e < 4; e++)  // every physical link once: its eastward direction at its west end
        if ((s.cell[fx][fy].links & (1 << e)) && twx::crosses(dir, fx - px, fy - py, e)) return true;
    }
  return false;
}

bool check_after_move(const orc_state &before, const orc_state &s, int player, int action) {
  const int n = s.n;
  // where the peg landed: the action's cell, or its rotation on a swap (tb.cc:462-473)
  int px = action / n, py = action % n;
  const bool swap = before.move_counter == 1 && px == before.m1x && py == before.m1y;
  if (swap) {
    const int ox = px;
    px = py;
    py = n - 1 - ox;
    // (the centre cell of an odd board rotates onto itself and is blue afterwards)
    const bool same = px == before.m1x && py == before.m1y;
    REQUIRE(s.swapped == 1 && s.cell[before.m1x][before.m1y].color == (same ? ORC_BLUE : ORC_EMPTY),
            "swap did not vacate move one");
  }
  REQUIRE(s.cell[px][py].color == player, "peg of player %d missing at (%d,%d)", player, px, py);
  REQUIRE(s.move_counter == before.move_counter + 1, "move counter");

  // (iv) the link scan at the new peg, from segment geometry on the links that existed before
  int want_links = 0, want_blocked = 0;
  for (int d = 0; d < 8; d++) {
    const int tx = px + twx::kDX[d], ty = py + twx::kDY[d];
    if (!on_board(n, tx, ty) || s.cell[tx][ty].color != player) continue;
    // (links made in this move share the new peg as an endpoint and never *properly* cross each
    // other, so evaluating on `before` or on `s` is the same; `before` is the definition)
    if (crossed(before, px, py, d)) want_blocked |= 1 << d; else want_links |= 1 << d;
    g_checks[3]++;
  }
  REQUIRE((s.cell[px][py].links & 0xff) == want_links, "N=%d move %d at (%d,%d): links %02x, geometry says %02x", n,
          s.move_counter, px, py, s.cell[px][py].links & 0xff, want_links);
  REQUIRE((s.cell[px][py].blocked & 0xff) == want_blocked, "N=%d move %d at (%d,%d): blocked %02x, geometry says %02x",
          n, s.move_counter, px, py, s.cell[px][py].blocked & 0xff, want_blocked);
  for (int d = 0; d < 8; d++) {  // the other end carries the opposite direction (tb.cc:532-533, 550-551)
    const int tx = px + twx::kDX[d], ty = py + twx::kDY[d], od = (d + 4) % 8;
    if (want_links & (1 << d)) REQUIRE(s.cell[tx][ty].links & (1 << od), "missing back link");
    if (want_blocked & (1 << d)) REQUIRE(s.cell[tx][ty].blocked & (1 << od), "missing back blocked bit");
  }
  // (v) symmetry + same colour, whole board
  for (int x = 0; x < n; x++)
    for (int y = 0; y < n; y++) {
      const orc_cell &c = s.cell[x][y];
      if (c.color != ORC_RED && c.color != ORC_BLUE) {
        REQUIRE((c.links & 0xff) == 0, "link bits on a cell without a peg (%d,%d)", x, y);
        continue;
      }
      for (int d = 0; d < 8; d++)
        if (c.links & (1 << d)) {
          const int tx = x + twx::kDX[d], ty = y + twx::kDY[d];
          REQUIRE(on_board(n, tx, ty) && s.cell[tx][ty].color == c.color &&
                      (s.cell[tx][ty].links & (1 << ((d + 4) % 8))),
                  "asymmetric link at (%d,%d) dir %d", x, y, d);
        }
    }

  // (ii) result from connectivity
  Components comp;
  comp.build(s, player);
  const int t = comp.touch[comp.label[px][py]];
  int want_result = ORC_OPEN;
  if (t == 3) {
    want_result = player == 0 ? ORC_RED_WIN : ORC_BLUE_WIN;
  } else {
    orc_state open = s;  // the opponent's cells-derived legal set, as if the game were open
    open.result = ORC_OPEN;
    std::vector<int> opp;
    derived_legal(open, 1 - player, &opp);
    if (opp.empty()) want_result = ORC_DRAW;
  }
  REQUIRE(s.result == want_result, "N=%d move %d: result %d, connectivity says %d", n, s.move_counter, s.result,
          want_result);
  g_checks[1]++;
  REQUIRE(s.current_player == (s.result == ORC_OPEN ? 1 - player : ORC_TERMINAL_PLAYER), "current player");

  // (i) flags of every peg of the mover == BFS connectivity (while the game is open)
  if (s.result == ORC_OPEN) {
    for (int x = 0; x < n; x++)
      for (int y = 0; y < n; y++) {
        if (s.cell[x][y].color != player) continue;
        const int tt = comp.touch[comp.label[x][y]];
        REQUIRE((s.cell[x][y].ltb[player][ORC_START] != 0) == ((tt & 1) != 0) &&
                    (s.cell[x][y].ltb[player][ORC_END] != 0) == ((tt & 2) != 0),
                "N=%d move %d: peg (%d,%d) of player %d has flags %d%d, BFS says %d%d", n, s.move_counter, x, y,
                player, s.cell[x][y].ltb[player][0], s.cell[x][y].ltb[player][1], tt & 1, (tt >> 1) & 1);
        g_checks[0]++;
      }
  }

  // (iii) legal lists of the player to move (and of the other one as the board keeps it)
  std::vector<int> want;
  int got[ORC_MAX_LEGAL];
  derived_legal(s, s.result == ORC_OPEN ? 1 - player : player, &want);
  const int k = orc_legal_actions(&s, got);
  REQUIRE(k == (int)want.size() && (k == 0 || memcmp(got, want.data(), sizeof(int) * k) == 0),
          "N=%d move %d: legal list differs from region & ~occupied (%d vs %zu entries)", n, s.move_counter, k,
          want.size());
  g_checks[2]++;
  return true;
}

bool no_crossing_links(const orc_state &s) {
  const int n = s.n;
  for (int x = 0; x < n; x++)
    for (int y = 0; y < n; y++)
      for (int d = 0; d < 4; d++)
        if (s.cell[x][y].links & (1 << d)) {
          REQUIRE(!crossed(s, x, y, d), "two links cross at the end of a game: (%d,%d) dir %d", x, y, d);
          g_checks[4]++;
        }
  return true;
}

bool play(int n, uint64_t seed, bool forced_swap, long *moves, int hist[4]) {
  static orc_state s, before;
  Rng rng{seed * 0x9E3779B97F4A7C15ull + 0x1234567ull};
  orc_init(&s, n);
  int legal[ORC_MAX_LEGAL];
  {  // (iii) at move 0 as well
    std::vector<int> want;
    derived_legal(s, 0, &want);
    const int k = orc_legal_actions(&s, legal);
    REQUIRE(k == n * (n - 2) && k == (int)want.size() && memcmp(legal, want.data(), sizeof(int) * k) == 0,
            "initial legal list");
  }
  int first = -1;
  while (s.result == ORC_OPEN) {
    const int k = orc_legal_actions(&s, legal);
    REQUIRE(k > 0, "open game without a legal action");
    int a = legal[rng.next() % (uint32_t)k];
    if (s.move_counter == 0) first = a;
    if (forced_swap && s.move_counter == 1)
      for (int i = 0; i < k; i++)
        if (legal[i] == first) a = first;  // blue repeats red's action whenever it may
    before = s;
    const int player = s.current_player;
    REQUIRE(orc_apply(&s, a) == 0, "legal action refused");
    if (!check_after_move(before, s, player, a)) return false;
    (*moves)++;
    REQUIRE(s.move_counter <= n * n - 3, "game longer than MaxGameLength");  // tw.h:136-139
  }
  hist[s.result]++;
  g_checks[5] += s.swapped;
  return no_crossing_links(s);
}

// The checker must have teeth: corrupt one observable of a correct state at a time and expect
// check_after_move to refuse it.  Returns the number of corruptions that went UNNOTICED.
int selftest() {
  static orc_state s, before, bad;
  int missed = 0, tried = 0;
  FILE *const quiet = freopen("/dev/null", "w", stderr);
  (void)quiet;
  for (int n : {6, 12, 24}) {
    Rng rng{(uint64_t)n * 77u + 5u};
    orc_init(&s, n);
    int legal[ORC_MAX_LEGAL];
    while (s.result == ORC_OPEN) {
      const int k = orc_legal_actions(&s, legal);
      const int a = legal[rng.next() % (uint32_t)k];
      before = s;
      const int player = s.current_player;
      orc_apply(&s, a);
      const int px = a / n, py = a % n;
      if (s.swapped && before.move_counter == 1) continue;
      if (!check_after_move(before, s, player, a)) return -1;  // the clean state must pass
      auto expect_caught = [&](const char *what) {
        tried++;
        if (check_after_move(before, bad, player, a)) {
          missed++;
          fprintf(stdout, "selftest: corruption not noticed: %s (N=%d move %d)\n", what, n, s.move_counter);
        }
      };
      if (s.result == ORC_OPEN) {
        bad = s; bad.cell[px][py].ltb[player][ORC_START] ^= 1; expect_caught("START flag of the new peg");
        bad = s; bad.cell[px][py].ltb[player][ORC_END] ^= 1; expect_caught("END flag of the new peg");
        bad = s; bad.result = player == 0 ? ORC_RED_WIN : ORC_BLUE_WIN; bad.current_player = ORC_TERMINAL_PLAYER;
        expect_caught("result");
        bad = s; bad.nlegal[1 - player] -= 1; expect_caught("legal list one short");
      }
      for (int d = 0; d < 8; d++) {
        const int tx = px + twx::kDX[d], ty = py + twx::kDY[d];
        if (!on_board(n, tx, ty) || s.cell[tx][ty].color != player) continue;
        bad = s;  // turn a link into a blocked pair or the other way round
        bad.cell[px][py].links ^= 1 << d; bad.cell[tx][ty].links ^= 1 << ((d + 4) % 8);
        bad.cell[px][py].blocked ^= 1 << d; bad.cell[tx][ty].blocked ^= 1 << ((d + 4) % 8);
        expect_caught("link <-> blocked");
      }
    }
  }
  printf("selftest: %d corruptions tried, %d unnoticed\n", tried, missed);
  return tried > 100 ? missed : -1;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "--selftest")) return selftest() == 0 ? 0 : 1;
  const double scale = argc > 1 ? atof(argv[1]) : 1.0;
  const int sizes[] = {5, 6, 7, 8, 12, 24};
  const int games[] = {600, 500, 400, 400, 200, 120};  // 2,220 games at scale 1
  long total_games = 0, total_moves = 0;
  for (int i = 0; i < 6; i++) {
    const int n = sizes[i], g = (int)(games[i] * scale + 0.5);
    long moves = 0;
    int hist[4] = {0, 0, 0, 0};
    for (int j = 0; j < g; j++)
      if (!play(n, (uint64_t)n * 100000 + j, j % 3 == 2, &moves, hist)) {
        fprintf(stderr, "game N=%d #%d (forced swap %d)\n", n, j, j % 3 == 2);
        return 1;
      }
    printf("N=%2d: %4d games, %7ld moves, results red/blue/draw = %d/%d/%d\n", n, g, moves, hist[1], hist[2], hist[3]);
    total_games += g;
    total_moves += moves;
  }
  printf("OK games=%ld moves=%ld peg_flag_checks=%ld result_checks=%ld legal_checks=%ld link_scan_checks=%ld "
         "final_link_checks=%ld swapped_games=%ld\n",
         total_games, total_moves, g_checks[0], g_checks[1], g_checks[2], g_checks[3], g_checks[4], g_checks[5]);
  return 0;
}
