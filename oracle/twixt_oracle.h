/*
 * oracle/twixt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference TwixT engine
 * (stevens68/TwixT_for_open_spiel: open_spiel/games/twixt/twixtboard.cc,
 * twixtboard.h, twixtcell.h, twixt.h, twixt.cc).  It exists only so that the
 * HIP path can be checked bit-for-bit against the reference's algorithm.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product library (libtwixt_hip.so) never links or calls it.
 *
 * Pinning: the reference itself cannot be compiled in this image without a
 * stand-in for open_spiel/spiel.h (twixtboard.h:27, twixtcell.h:19), which the
 * build rules forbid, so this restatement is pinned by
 *   - the reference's playthrough fixture (playthrough.txt: 35 actions, 9 fully
 *     printed states: strings, tensors, legal actions, terminal returns),
 *   - the known answers in twixt_test.cc:108-199 (swap, legal counts, draw),
 *   - the 20-game KAT table of SURVEY.md section 8(c), which was produced from
 *     the compiled reference (LCG policy; the moves / result / swapped / links
 *     columns are matched; its hex digests could not be reproduced).
 * See tests/test_oracle_*.py.
 *
 * Every function cites the reference lines it follows ("tb.cc" =
 * twixtboard.cc, "tb.h" = twixtboard.h, "tc.h" = twixtcell.h, "tw.h" =
 * twixt.h, "tw.cc" = twixt.cc).
 */
#ifndef TWIXT_ORACLE_H_
#define TWIXT_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MIN_N 5  /* tb.h:32 */
#define ORC_MAX_N 24 /* tb.h:33 */
#define ORC_MAX_LEGAL (ORC_MAX_N * (ORC_MAX_N - 2))
#define ORC_TERMINAL_PLAYER (-4) /* kTerminalPlayerId, playthrough.txt:678 */

enum { ORC_OPEN = 0, ORC_RED_WIN = 1, ORC_BLUE_WIN = 2, ORC_DRAW = 3 }; /* tb.h:48 */
enum { ORC_RED = 0, ORC_BLUE = 1, ORC_EMPTY = 2, ORC_OFFBOARD = 3 };   /* tb.h:50 */
enum { ORC_START = 0, ORC_END = 1 };                                   /* tc.h:50 */

/* tc.h:70-109 (neighbour table omitted: neighbour = position + offset) */
typedef struct {
  int color;
  int links;                  /* 8 bits, bit = Compass direction */
  int blocked;                /* 8 bits, blocked_neighbors_ */
  unsigned char ltb[2][2];    /* linked_to_border_[player][border] */
} orc_cell;

/* tb.h:75-82 (Board) + tw.h:107 (current_player_) */
typedef struct {
  int n;
  int move_counter;
  int swapped;
  int m1x, m1y; /* move_one_ */
  int result;
  int current_player;
  orc_cell cell[ORC_MAX_N][ORC_MAX_N]; /* cell[x][y] */
  int nlegal[2];
  int legal[2][ORC_MAX_LEGAL]; /* ascending action ids */
} orc_state;

/* ---- L2: Board ---- */
int orc_size_ok(int n);                                   /* tw.cc:139-144 */
void orc_init(orc_state *s, int n);                       /* tb.cc:168-174 */
void orc_board_apply(orc_state *s, int player, int action); /* tb.cc:457-499 */
int orc_off_board(int n, int x, int y);                   /* tb.cc:625-631 */
int orc_on_border(int n, int player, int x, int y);       /* tb.cc:615-623 */

/* ---- L3: State ---- */
int orc_is_terminal(const orc_state *s);                  /* tw.h:45-48 */
int orc_legal_actions(const orc_state *s, int *out);      /* tw.h:86-90; returns count */
int orc_apply(orc_state *s, int action);                  /* tw.h:93-104; 0 ok, -1 "Not a legal action" */
void orc_returns(const orc_state *s, double out[2]);      /* tw.h:50-63 */
void orc_observation(const orc_state *s, float *out);     /* tw.cc:76-132; 12*N*(N-2) floats */
int orc_to_string(const orc_state *s, int ansi, char *buf, int cap); /* tb.cc:278-335 */
int orc_action_to_string(const orc_state *s, int player, int action, char *buf); /* tw.cc:67-74 */

/* ---- views used by the parity tests ---- */
/* per cell x-major then y: {colour, links&0xff, blocked&0xff, flags} with
 * flags = ltb[0][0] | ltb[0][1]<<1 | ltb[1][0]<<2 | ltb[1][1]<<3 */
void orc_export_cells(const orc_state *s, uint8_t *out /* n*n*4 */);

/* Packed bit-plane view shared with the HIP engine's twx_export_state
 * (include/twixt_hip.h): 8 planes {red, blue, link NNE, ENE, ESE, SSE at the
 * west endpoint, START flag, END flag} of W = ceil(n*n/32) dwords, bit
 * (x*n+y); then 4 scalar dwords {move_counter | move_one_action<<16,
 * result | swapped<<8, 0, 0}.  Returns number of dwords written. */
int orc_export_planes(const orc_state *s, uint32_t *out);

/* ---- policies / drivers ---- */
/* SURVEY.md 8(c) KAT: LCG policy + FNV-1a-64 digests. */
typedef struct {
  int moves, result, swapped, links;
  uint64_t hist_fnv, step_fnv, final_fnv;
} orc_kat;
void orc_kat_game(int n, uint64_t seed, int forced_swap, orc_kat *out);

/* The counter-based policy shared (as a specification) with the HIP engine:
 * h = lowbias32(seed ^ board*0x9E3779B1 ^ episode*0xC2B2AE3D ^ move*0x85EBCA77),
 * k = (h * count) >> 32, action = k-th legal action in ascending order. */
uint32_t orc_policy_hash(uint32_t seed, uint32_t board, uint32_t episode, uint32_t move);

/* Play `nsteps` env steps of the random policy on one board (global id
 * `board`), auto-resetting finished games exactly as twx_step_random does
 * (a terminal board is re-initialised, episode+1, and plays move 0 in the same
 * step).  `episode` is in/out.  If actions_out != NULL the chosen action of
 * every step is stored.  If auto_reset == 0, steps on a terminal board are
 * no-ops (action -1).  Returns env steps actually played. */
int64_t orc_rollout(orc_state *s, uint32_t seed, uint32_t board, uint32_t *episode,
                    int nsteps, int auto_reset, int32_t *actions_out, uint8_t *status_out);

/* `nboards` boards (global ids first_board..) x `nsteps` steps, OpenMP over
 * boards.  Any output may be NULL: planes_out[nboards][8W+4] (orc_export_planes
 * with s2 = episode), cells_out[nboards][n*n*4], actions_out / status_out
 * [nboards][nsteps].  Returns env steps played. */
int64_t orc_rollout_batch(int n, uint32_t seed, uint32_t first_board, int nboards, int nsteps,
                          int auto_reset, int threads, uint32_t *planes_out, uint8_t *cells_out,
                          int32_t *actions_out, uint8_t *status_out);

/* the restated blocker table, for tests/test_geometry.py: out = {dx, dy, dir} */
void orc_block_entry(int dir, int i, int out[3]);

/* cpu_baseline leg: `nboards` boards x `nsteps` steps on `threads` OpenMP
 * threads; returns total env steps, writes a FNV digest over final results. */
int64_t orc_bench(int n, uint32_t seed, int nboards, int nsteps, int threads, uint64_t *digest);

#ifdef __cplusplus
}
#endif
#endif /* TWIXT_ORACLE_H_ */
