// twixt_geom.h -- compile-time geometry of a TwixT board of side N, shared by
// the HIP kernels and the host side of libtwixt_hip.so.
//
// Everything here is derived from first principles (knight-move offsets,
// segment intersection), not from the reference's tables:
//   * Compass numbering / offsets            -> twixtcell.h:58-68, twixtboard.cc:38-144
//   * regions, border lines, corners         -> twixtboard.cc:223-231, 252-276, 615-631
//   * "link A blocks link B"                 -> the BlockerMap built at twixtboard.cc:176-190
//     is exactly "the two segments properly cross" (SURVEY.md section 7); here it
//     becomes 8x5 constant bit masks over a 5-column x 7-row window of the four
//     link planes (tests/test_geometry.py checks them against the oracle's
//     restated table).
#ifndef TWIXT_GEOM_H_
#define TWIXT_GEOM_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define TWX_HD __host__ __device__
#else
#define TWX_HD
#endif

namespace twx {

// Compass: NNE ENE ESE SSE SSW WSW WNW NNW
constexpr int kDX[8] = {1, 2, 2, 1, -1, -2, -2, -1};
constexpr int kDY[8] = {2, 1, -1, -2, -2, -1, 1, 2};

constexpr int kPlaneRed = 0, kPlaneBlue = 1, kPlaneLink0 = 2, kPlaneStart = 6, kPlaneEnd = 7;
constexpr int kNumPlanes = 8;

constexpr int plane_words(int n) { return (n * n + 31) / 32; }
constexpr int state_words(int n) { return kNumPlanes * plane_words(n) + 4; }
constexpr int state_chunks(int n) { return state_words(n) / 4; }  // 16-byte chunks

// Redo journal (DESIGN.md "Log-structured state"): kJournalSlots 32-bit move
// records per board, stored in kJournalChunks extra chunks after the state.
#ifndef TWX_JOURNAL_SLOTS
#define TWX_JOURNAL_SLOTS 8
#endif
constexpr int kJournalSlots = TWX_JOURNAL_SLOTS;  // a multiple of 4 (whole 16-byte chunks)
constexpr int kJournalChunks = kJournalSlots / 4;
constexpr int total_chunks(int n) { return state_chunks(n) + kJournalChunks; }
// record of one move: bit31 valid | cell[9:0] | player[10] | link mask[18:11] | gain START[19] |
//   gain END[20] | swap[21] | result after the move[23:22] | single-cell fill: direction[26:24]
//   of the linked neighbour that gains START[27] / END[28] (the flood-fill shortcut of the step
//   kernels: what ExploreLocalGraph, tb.cc:573-588, does when it flags exactly one cell).
// The scalars chunk is rewritten only together with the whole board (compaction, reset, subset /
// roll-out launches); between two such writes the true scalars are the stored ones advanced by
// the records, in slot order: journal_scalars().
constexpr uint32_t kJValid = 0x80000000u;
constexpr int kJResultShift = 22, kJFillDirShift = 24;
constexpr uint32_t kJGainStart = 1u << 19, kJGainEnd = 1u << 20, kJSwap = 1u << 21, kJFillStart = 1u << 27,
                   kJFillEnd = 1u << 28;
// (dx + 2, dy + 2) of the 8 directions, 4 bits each: a direction held in a register -> its offset
constexpr uint32_t kDxPack = 0x10013443u, kDyPack = 0x43100134u;

constexpr bool is_corner(int n, int x, int y) {
  return (x == 0 || x == n - 1) && (y == 0 || y == n - 1);
}
constexpr bool on_board(int n, int x, int y) {
  return x >= 0 && x < n && y >= 0 && y < n && !is_corner(n, x, y);
}
// cells a player may ever occupy: everything except the opponent's border lines
constexpr bool in_region(int n, int player, int x, int y) {
  return on_board(n, x, y) && (player == 0 ? (x >= 1 && x <= n - 2) : (y >= 1 && y <= n - 2));
}

constexpr long orient(int ax, int ay, int bx, int by, int cx, int cy) {
  return (long)(bx - ax) * (cy - ay) - (long)(by - ay) * (cx - ax);
}
// does the link leaving (0,0) in direction `dir` properly cross the link that
// leaves (fx,fy) in canonical direction e (0..3)?
constexpr bool crosses(int dir, int fx, int fy, int e) {
  const int tx = kDX[dir], ty = kDY[dir];
  const int gx = fx + kDX[e], gy = fy + kDY[e];
  const long o1 = orient(0, 0, tx, ty, fx, fy), o2 = orient(0, 0, tx, ty, gx, gy);
  const long o3 = orient(fx, fy, gx, gy, 0, 0), o4 = orient(fx, fy, gx, gy, tx, ty);
  return ((o1 > 0 && o2 < 0) || (o1 < 0 && o2 > 0)) && ((o3 > 0 && o4 < 0) || (o3 < 0 && o4 > 0));
}

// Crossing window: columns fx = -3..+1 (index i = fx+3), rows fy = -3..+3
// (bit j = fy+3), plane e in byte e: bit (8*e + j) of word i.
constexpr int kWinCols = 5;
struct CrossMasks {
  uint32_t m[8][kWinCols];
  int count[8];
};
constexpr CrossMasks make_cross_masks() {
  CrossMasks c{};
  for (int dir = 0; dir < 8; dir++) {
    c.count[dir] = 0;
    for (int i = 0; i < kWinCols; i++) c.m[dir][i] = 0;
    for (int fx = -6; fx <= 6; fx++)
      for (int fy = -6; fy <= 6; fy++)
        for (int e = 0; e < 4; e++)
          if (crosses(dir, fx, fy, e)) {
            c.count[dir]++;
            if (fx >= -3 && fx <= 1 && fy >= -3 && fy <= 3)
              c.m[dir][fx + 3] |= 1u << (8 * e + fy + 3);
            else
              c.count[dir] += 1000;  // would fall outside the window: must not happen
          }
  }
  return c;
}
constexpr CrossMasks kCross = make_cross_masks();
static_assert(kCross.count[0] == 9 && kCross.count[1] == 9 && kCross.count[2] == 9 &&
                  kCross.count[3] == 9 && kCross.count[4] == 9 && kCross.count[5] == 9 &&
                  kCross.count[6] == 9 && kCross.count[7] == 9,
              "every link direction has exactly 9 crossing links, all inside the window");

// Per-N constant planes.
template <int N>
struct Geo {
  static constexpr int W = plane_words(N);
  static constexpr int SW = state_words(N);
  static constexpr int CH = state_chunks(N);   // planes + scalars
  static constexpr int CHT = total_chunks(N);  // + journal
  struct Tab {
    uint32_t region[2][W];
    uint32_t init[SW];  // the initial position (tb.cc:209-236)
  };
  static constexpr Tab make() {
    Tab t{};
    for (int x = 0; x < N; x++)
      for (int y = 0; y < N; y++) {
        const int i = x * N + y;
        const uint32_t bit = 1u << (i & 31);
        if (in_region(N, 0, x, y)) t.region[0][i >> 5] |= bit;
        if (in_region(N, 1, x, y)) t.region[1][i >> 5] |= bit;
        if (on_board(N, x, y)) {
          if (x == 0 || y == 0) t.init[kPlaneStart * W + (i >> 5)] |= bit;
          if (x == N - 1 || y == N - 1) t.init[kPlaneEnd * W + (i >> 5)] |= bit;
        }
      }
    return t;
  }
  static constexpr Tab tab = make();
};

// Runtime version of the same tables (host side: uploaded once per handle for
// the layout-agnostic utility kernels).
inline void fill_tables(int n, uint32_t *region /*[2][W]*/, uint32_t *init /*[SW]*/) {
  const int W = plane_words(n), SW = state_words(n);
  for (int i = 0; i < 2 * W; i++) region[i] = 0;
  for (int i = 0; i < SW; i++) init[i] = 0;
  for (int x = 0; x < n; x++)
    for (int y = 0; y < n; y++) {
      const int i = x * n + y;
      const uint32_t bit = 1u << (i & 31);
      if (in_region(n, 0, x, y)) region[0 * W + (i >> 5)] |= bit;
      if (in_region(n, 1, x, y)) region[1 * W + (i >> 5)] |= bit;
      if (on_board(n, x, y)) {
        if (x == 0 || y == 0) init[kPlaneStart * W + (i >> 5)] |= bit;
        if (x == n - 1 || y == n - 1) init[kPlaneEnd * W + (i >> 5)] |= bit;
      }
    }
}

// The random policy's hash (DESIGN.md "Policy"): lowbias32 finaliser over the
// xor of the four keys times odd constants.
TWX_HD inline uint32_t policy_hash(uint32_t seed, uint32_t board, uint32_t episode, uint32_t move) {
  uint32_t h = seed ^ (board * 0x9E3779B1u) ^ (episode * 0xC2B2AE3Du) ^ (move * 0x85EBCA77u);
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}

// s1 of the stored scalars also carries the board's journal BASE in bits 16..19 while a journal
// period runs: a board that was reset at journal position j owns only the records of slots >= j
// (the older ones belong to its previous game and share their lines with other boards' records,
// so they cannot be cleared one by one); every whole-board write outside a period stores base 0.
constexpr int kJBaseShift = 16;
// scalars after the move a journal record describes (move_counter, move_one, result, swapped:
// tb.h:75-78 as Board::ApplyAction, tb.cc:457-499, leaves them); an empty record changes nothing
TWX_HD inline void journal_scalars(uint32_t e, uint32_t &counter, uint32_t &move_one, uint32_t &result,
                                   uint32_t &swapped) {
  if (e & kJValid) {
    if (counter == 0u) move_one = e & 1023u;  // tb.cc:485-488
    counter += 1u;
    result = (e >> kJResultShift) & 3u;
    swapped |= (e >> 21) & 1u;
  }
}

}  // namespace twx
#endif  // TWIXT_GEOM_H_
