/*
 * include/twixt_hip.h -- C ABI of libtwixt_hip.so, the MI355X (gfx950) batched
 * TwixT board engine.
 *
 * This is the drop-in boundary for the reference's hot path: each entry point
 * replaces one piece of the reference's Board / TwixTState interface
 * (stevens68/TwixT_for_open_spiel, open_spiel/games/twixt/), batched over B
 * independent boards that live in HBM as packed bit-planes.  Citations:
 * tb.h = twixtboard.h, tb.cc = twixtboard.cc, tw.h = twixt.h, tw.cc = twixt.cc.
 *
 * Conventions
 *   - plain C, no exceptions, no torch types.  Every function returns TWX_OK
 *     (0) or a negative TWX_E_* code; twx_last_error() gives the text.
 *   - all `*_dev` pointers are DEVICE pointers owned by the caller (e.g. the
 *     storage of a torch tensor, or hipMalloc memory of the C++ adapter).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Calls are asynchronous on that stream; the handle is re-entrant, there is
 *     no global mutable state (the reference's static BlockerMap, tb.h:150, has
 *     no counterpart).
 *   - a handle's work always runs on the device it was created on; entry points select
 *     that device for the duration of the call and put the caller's current device back.
 *   - a handle is used from one host thread and one stream at a time (calls on it are
 *     ordered by that stream).  Between compactions the board state is log-structured
 *     (DESIGN.md): the step functions carry a host-side journal position as a launch
 *     argument, so a sequence of steps must not be replayed from a captured hipGraph
 *     (the step functions and twx_status / twx_status_boards, which read the journal up
 *     to that position, return TWX_E_ARG on a capturing stream)
 *     unless the handle was created with TWX_GRAPH_SAFE=1 in the environment (planes
 *     rewritten in place every step, fixed batch order: every launch argument is then
 *     constant; meant for small, launch-bound batches such as search leaves);
 *     every other entry point first materialises the journal and is unaffected.
 *   - tuning knobs read once by twx_create from the environment: TWX_JOURNAL_PERIOD
 *     (1..8; default by board size: 1 for N <= 9, 2 for N = 10..12, 4 for N = 13..17, 8 above;
 *     1 = rewrite the planes every step), TWX_WINDOW (0/1, default 1: boards of N >= 17 live in
 *     the split layout and step through the windowed kernels; 0: chunk layout, round 2's kernels),
 *     TWX_PAIR (0/1, default 1; 0: chunk layout, one lane per board), TWX_SMALL_BATCH_REWRITE (0/1,
 *     default 1: batches of N >= 17 too small to fill the chip rewrite their boards every step
 *     instead of journalling -- one wave's latency decides there), TWX_PINGPONG (0/1, default 1:
 *     alternate the batch order between launches), TWX_WAVES_PER_CU (default: occupancy),
 *     TWX_DYNAMIC (0/1, default 1: the last rounds of a launch are handed out by ticket
 *     counters instead of a fixed wave -> batch map; like the journal position, the ticket
 *     base is a launch argument kept by the host), TWX_DYNAMIC_STATIC (rounds in fixed
 *     order before the tickets start; 0 = automatic).
 *   - action ids, players, results are the reference's: action = x*N + y
 *     (tb.cc:599-605), player 0 = red / 1 = blue (tc.h:52-53),
 *     result 0 open / 1 red win / 2 blue win / 3 draw (tb.h:48).
 *
 * Board state (per board, see DESIGN.md "Data layout"): 8 bit-planes of
 * W = ceil(N*N/32) dwords, bit index = action id:
 *   0 red pegs, 1 blue pegs, 2..5 links NNE/ENE/ESE/SSE stored at the west
 *   endpoint, 6 START-border flag, 7 END-border flag          (tc.h:98-108)
 * followed by 4 scalar dwords:
 *   s0 = move_counter | move_one_action << 16                  (tb.h:75-77)
 *   s1 = result | swapped << 8                                 (tb.h:76,78)
 *        (bits 16..19 are internal: exported as 0, ignored on import)
 *   s2 = episode (number of auto-resets, feeds the policy hash)
 *   s3 = 0 (reserved)
 * twx_state_words(N) = 8*W + 4.  twx_export_state / twx_import_state use this
 * order with one board's words contiguous.
 */
#ifndef TWIXT_HIP_H_
#define TWIXT_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TWX_OK 0
#define TWX_E_ARG (-1)     /* bad argument (NULL, range)                         */
#define TWX_E_SIZE (-2)    /* board_size out of range [5..24]  (tw.cc:139-144)   */
#define TWX_E_DEVICE (-3)  /* HIP runtime error                                  */
#define TWX_E_NOMEM (-4)   /* device allocation failed                           */

#define TWX_MIN_BOARD_SIZE 5   /* tb.h:32 */
#define TWX_MAX_BOARD_SIZE 24  /* tb.h:33 */

/* status byte written per board by twx_step / twx_step_random */
#define TWX_ST_RESULT_MASK 0x03 /* result after the step (tb.h:48)               */
#define TWX_ST_ILLEGAL 0x04     /* "Not a legal action" (tw.h:95-97): untouched  */
#define TWX_ST_RESET 0x10       /* board was auto-reset before this move         */
#define TWX_ST_SWAPPED 0x20     /* swapped_ (tb.h:76)                            */
#define TWX_ST_NOOP 0x40        /* terminal board skipped (random mode, no reset)*/

typedef struct twx_env twx_env; /* opaque: owns the device state of B boards */

/* Board::Board(int,bool) x batch (tb.cc:168-174) + TwixTGame range check
 * (tw.cc:139-144).  Allocates B boards on HIP device `device`, all in the
 * initial position.  `first_board_id` is the global id of board 0 (the batch
 * shard offset on multi-GPU runs; it only feeds the random policy's hash).  Board ids
 * are 32-bit policy keys: first_board_id + batch > 2^32 is refused with TWX_E_ARG. */
int twx_create(int board_size, int64_t batch, int device, uint64_t first_board_id, twx_env **out);
int twx_destroy(twx_env *env);

int twx_board_size(const twx_env *env);
int64_t twx_batch(const twx_env *env);
int twx_state_words(int board_size);   /* 8*W + 4, or TWX_E_SIZE */
int twx_mask_words(int board_size);    /* W,       or TWX_E_SIZE */

/* TwixTGame::NewInitialState for every board / for the listed boards
 * (tw.h:118-120, tb.cc:209-276).  ids_dev: n int32 board indices. */
int twx_reset(twx_env *env, void *stream);
int twx_reset_boards(twx_env *env, const int32_t *ids_dev, int64_t n, void *stream);

/* TwixTState::DoApplyAction for every board (tw.h:93-104 -> tb.cc:457-499):
 * legality check, swap handling, peg + 8-direction link scan with crossing
 * test, border-flag propagation, result.  actions_dev[B] int32.  A board whose
 * action is not in its legal set (or that is terminal) is left untouched and
 * gets TWX_ST_ILLEGAL.  status_dev[B] may be NULL. */
int twx_step(twx_env *env, const int32_t *actions_dev, uint8_t *status_dev, void *stream);
/* same for the n DISTINCT boards listed in ids_dev: actions_dev[n], status_dev[n] (row i
 * belongs to board ids_dev[i]); every other board is untouched.  This is the batched-leaf
 * form: an MCTS driver steps exactly the states it expanded. */
int twx_step_boards(twx_env *env, const int32_t *ids_dev, int64_t n, const int32_t *actions_dev,
                    uint8_t *status_dev, void *stream);

/* One random-policy env step per board, fused: LegalActions (tw.h:86-90) ->
 * uniform pick -> DoApplyAction.  action = k-th legal action (ascending),
 * k = (hash32(seed, first_board_id + b, episode, move_counter) * count) >> 32
 * (see DESIGN.md "Policy").  auto_reset != 0: a terminal board is re-initialised
 * (episode+1) and plays move 0 in the same call, so every call performs exactly
 * B env steps.  actions_out_dev[B] (int32, -1 for skipped) and status_dev[B]
 * may be NULL. */
int twx_step_random(twx_env *env, uint32_t seed, int auto_reset, int32_t *actions_out_dev,
                    uint8_t *status_dev, void *stream);
int twx_step_random_boards(twx_env *env, const int32_t *ids_dev, int64_t n, uint32_t seed, int auto_reset,
                           int32_t *actions_out_dev, uint8_t *status_dev, void *stream);

/* A whole random play-out per board in ONE launch (what RandomRolloutEvaluator / example do
 * with LegalActions + ApplyAction in a loop): every board (ids_dev == NULL, n = 0 or B) or
 * every listed board plays the random policy of twx_step_random, without auto-reset, until
 * it is terminal or max_steps moves were made; the state stays on chip between moves.
 * The boards end up exactly as after max_steps calls of twx_step_random(auto_reset = 0).
 * status_dev[n or B] (may be NULL) gets result | TWX_ST_SWAPPED, plus TWX_ST_NOOP for a board
 * that was already terminal when the call began (it made no move); the byte of a board never
 * depends on the other boards of the call. */
int twx_random_rollout(twx_env *env, const int32_t *ids_dev, int64_t n, uint32_t seed, int max_steps,
                       uint8_t *status_dev, void *stream);

/* TwixTState::LegalActions as a mask (tw.h:86-90, tb.h:62-64): mask_dev[B][W],
 * bit a set <=> action a legal for the player to move; all zero if terminal.
 * count_dev[B] (int32, may be NULL) = number of legal actions. */
int twx_legal_mask(const twx_env *env, uint32_t *mask_dev, int32_t *count_dev, void *stream);
/* same for the n boards listed in ids_dev (NULL = boards 0..n-1): mask_dev[n][W], count_dev[n] */
int twx_legal_mask_boards(const twx_env *env, const int32_t *ids_dev, int64_t n, uint32_t *mask_dev,
                          int32_t *count_dev, void *stream);

/* Board::result / move_counter (tb.h:60-61), TwixTState::CurrentPlayer
 * (tw.h:38; -4 = kTerminalPlayerId when terminal).  Any pointer may be NULL. */
int twx_status(const twx_env *env, uint8_t *result_dev, int32_t *move_counter_dev,
               int8_t *current_player_dev, void *stream);
int twx_status_boards(const twx_env *env, const int32_t *ids_dev, int64_t n, uint8_t *result_dev,
                      int32_t *move_counter_dev, int8_t *current_player_dev, void *stream);

/* TwixTState::ObservationTensor x batch (tw.cc:76-132): out_dev[B][12][N][N-2]
 * float32, identical for both observers. */
int twx_observation(const twx_env *env, float *out_dev, void *stream);
int twx_observation_boards(const twx_env *env, const int32_t *ids_dev, int64_t n, float *out_dev,
                           void *stream);

/* TwixTState::Clone (tw.h:80-82): dst board dst_ids[i] := src board
 * src_ids[i].  dst and src may be the same handle but must have equal N. */
int twx_copy_boards(twx_env *dst, const int32_t *dst_ids_dev, const twx_env *src,
                    const int32_t *src_ids_dev, int64_t n, void *stream);

/* Raw state of n boards, board-contiguous (layout in the header comment).
 * ids_dev == NULL means boards 0..n-1. */
int twx_export_state(const twx_env *env, const int32_t *ids_dev, int64_t n, uint32_t *out_dev,
                     void *stream);
int twx_import_state(twx_env *env, const int32_t *ids_dev, int64_t n, const uint32_t *in_dev,
                     void *stream);

/* Reference-shaped Cell view (tc.h:98-108) of n boards, for renderers and
 * parity checks: out_dev[n][N*N][4] bytes, cell index x*N+y:
 *   [0] colour 0 red / 1 blue / 2 empty / 3 off-board (tb.h:50)
 *   [1] links_ (8 Compass bits)   [2] blocked_neighbors_ (8 bits)
 *   [3] border flags: bit0 red START, bit1 red END, bit2 blue START, bit3 blue END */
int twx_export_cells(const twx_env *env, const int32_t *ids_dev, int64_t n, uint8_t *out_dev,
                     void *stream);

/* Host-only text (no device work).  Board::ToString (tb.cc:278-335) from the HOST copy of
 * one board's twx_export_cells bytes; returns the full length (like snprintf), writes at
 * most cap-1 characters + NUL.  TwixTState::ActionToString (tw.cc:67-74). */
int twx_cells_to_string(const uint8_t *cells_host, int board_size, int ansi_color_output, int swapped,
                        int result, char *buf, int cap);
int twx_action_to_string(int board_size, int player, int64_t action, char *buf, int cap);

const char *twx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* TWIXT_HIP_H_ */
