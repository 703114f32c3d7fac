// tests/layout_check.hip -- host-side check of the engine's address arithmetic (CPU test, no GPU):
// chunk_index / journal_word_index (where a board's chunks and journal records live in the tiled
// state array) and reversed_batch (the backward batch order that keeps every batch on its XCD label),
// compiled from the product header.  Prints "OK" or the first violation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <set>
#include <vector>

#include "../twixt_for_open_spiel_amd/csrc/twixt_geom.h"
#include "../twixt_for_open_spiel_amd/csrc/twixt_device_common.h"

#define CHECK(c, ...) do { if (!(c)) { printf("FAIL %s:%d %s: ", __FILE__, __LINE__, #c); printf(__VA_ARGS__); printf("\n"); return 1; } } while (0)

int main() {
  // ---- state layout: every (board, chunk) and every (board, journal slot) has its own place inside
  //      the board's tile; the journal of a tile is its last two chunk rows, [slot][board] ----
  for (int n = 5; n <= 24; n++) {
    const int W = (n * n + 31) / 32, CH = 2 * W + 1, CHT = CH + twx::kJournalChunks, ts = tile_shift(CHT);
    CHECK(ts == (n >= 19 ? 5 : 6), "n=%d ts=%d", n, ts);
    const long TB = 1L << ts, boards = 5 * TB;  // five tiles
    std::vector<int> owner((size_t)boards * CHT * 4, 0);  // per dword of the state array: times claimed
    for (long b = 0; b < boards; b++) {
      for (int c = 0; c < CH; c++) {
        const long ci = chunk_index(c, b, ts, CHT);
        CHECK(ci >= 0 && ci < boards * CHT, "n=%d chunk index %ld", n, ci);
        CHECK(ci / (CHT * TB) == b / TB, "n=%d: chunk %d of board %ld outside its tile", n, c, b);
        CHECK((ci % (CHT * TB)) / TB == c && ci % TB == b % TB, "n=%d: chunk row / slot of board %ld", n, b);
        for (int k = 0; k < 4; k++) owner[ci * 4 + k]++;
      }
      for (int s = 0; s < twx::kJournalSlots; s++) {
        const long wi = journal_word_index(s, b, ts, CHT);
        CHECK(wi >= 0 && wi < boards * CHT * 4, "n=%d journal word %ld", n, wi);
        const long tile = wi / (CHT * TB * 4), in_tile = wi % (CHT * TB * 4);
        CHECK(tile == b / TB, "n=%d: record of board %ld outside its tile", n, b);
        CHECK(in_tile >= (long)CH * TB * 4, "n=%d: record inside the planes", n);
        CHECK((in_tile - (long)CH * TB * 4) / TB == s && in_tile % TB == b % TB, "n=%d: [slot][board] order", n);
        owner[wi]++;
      }
    }
    for (size_t i = 0; i < owner.size(); i++) CHECK(owner[i] == 1, "n=%d: dword %zu claimed %d times", n, i, owner[i]);
  }
  // ---- split layout (the big boards, step_window_kernel): every word of every plane, every scalar and
  //      every journal record of a board has its own dword inside the board's tile; pegs / scalars /
  //      journal in [row][board] rows of 512 B, links + flags board-major behind them ----
  for (int n = 17; n <= 24; n++) {
    const int W = (n * n + 31) / 32, RP = split_peg_rows(W), RA = split_rows(W), LFB = split_lf_bytes(W);
    const long TD = split_tile_bytes(W) / 4, boards = 5 * 32;
    CHECK(split_tile_bytes(W) % 128 == 0 || n != 24, "n=24 tiles are whole 128-byte lines");
    CHECK(LFB % 16 == 0 && LFB >= 24 * W && RA == RP + 3, "n=%d region size", n);
    std::vector<int> owner((size_t)(boards / 32) * TD, 0);
    for (long b = 0; b < boards; b++) {
      for (int p = 0; p < twx::kNumPlanes; p++)
        for (int i = 0; i < W; i++) {
          const long wi = split_word_index(p, i, b, W);
          CHECK(wi >= 0 && wi / TD == b / 32, "n=%d: plane %d word %d of board %ld outside its tile", n, p, i, b);
          const long in = wi % TD;
          if (p < 2) CHECK(in / 128 == i / 2 && (in % 128) / 4 == b % 32 && in % 4 == (i & 1) * 2 + p, "n=%d peg row", n);
          else CHECK(in >= RA * 128 && (in - RA * 128) / (LFB / 4) == b % 32 && (in - RA * 128) % (LFB / 4) == 6 * i + p - 2, "n=%d region", n);
          owner[wi]++;
        }
      for (int j = 0; j < 4; j++) {
        const long wi = split_scalar_index(j, b, W);
        CHECK(wi / TD == b / 32 && (wi % TD) / 128 == RP && ((wi % TD) % 128) == (b % 32) * 4 + j, "n=%d scalars", n);
        owner[wi]++;
      }
      for (int s = 0; s < twx::kJournalSlots; s++) {
        const long wi = split_journal_index(s, b, W);
        CHECK(wi / TD == b / 32 && (wi % TD) == (long)(RP + 1) * 128 + s * 32 + b % 32, "n=%d journal [slot][board]", n);
        owner[wi]++;
      }
    }
    long claimed = 0;
    for (size_t i = 0; i < owner.size(); i++) { CHECK(owner[i] <= 1, "n=%d: dword %zu claimed %d times", n, i, owner[i]); claimed += owner[i]; }
    CHECK(claimed == boards * (8 * W + 4 + twx::kJournalSlots), "n=%d: %ld dwords claimed", n, claimed);
  }
  // ---- the window of a move (step_window1_kernel): for every size of the split layout, every board column and
  //      every set of candidate directions, the pieces win_range() names lie inside the window's NPC pieces
  //      and cover every word that holds a bit the move can touch: the link columns of the candidates'
  //      crossing masks (twixt_geom.h kCross, independent of win_range's own column rule), the flag columns
  //      of the candidates and the peg's own column ----
  for (int n = 17; n <= 24; n++) {
    const int W = (n * n + 31) / 32, LFP = split_lf_bytes(W) / 16, NPC = win_pieces(n, W);
    for (int x = 0; x < n; x++) {
      const int p0 = win_piece0(n, W, x);
      CHECK(p0 >= 0 && p0 + NPC <= LFP, "n=%d x=%d: window outside the region", n, x);
      for (uint32_t cand = 1; cand < 256; cand++) {
        const WinRange r = win_range(n, cand, x);
        CHECK(r.np >= 1 && r.p_lo >= p0 && r.p_lo + r.np <= p0 + NPC - 1, "n=%d x=%d cand=%02x: pieces %d..%d outside window %d..%d",
              n, x, cand, r.p_lo, r.p_lo + r.np, p0, p0 + NPC - 1);
        bool need[32] = {};
        need[x] = true;  // own flags, own link words
        for (int d = 0; d < 8; d++) {
          if (!((cand >> d) & 1u)) continue;
          const int cf = x + twx::kDX[d];
          if (cf >= 0 && cf < n) need[cf] = true;  // the candidate's flags (and its link word for a west link)
          for (int i = 0; i < twx::kWinCols; i++) {
            const int c = x - 3 + i;
            if (twx::kCross.m[d][i] != 0u && c >= 0 && c < n) need[c] = true;
          }
        }
        for (int c = 0; c < n; c++) {
          if (!need[c]) continue;
          for (int w = (c * n) >> 5; w <= ((c + 1) * n - 1) >> 5; w++)
            CHECK(24 * w >= 16 * r.p_lo && 24 * w + 24 <= 16 * (r.p_lo + r.np + 1), "n=%d x=%d cand=%02x: word %d of column %d not fetched",
                  n, x, cand, w, c);
        }
      }
    }
  }
  // ---- backward batch order: a bijection that reverses blocks of 8 and keeps i % 8 ----
  for (long nb : {1L, 7L, 8L, 9L, 16L, 37L, 64L, 1000L, 16384L, 32768L, 32771L}) {
    std::set<long> seen;
    const long q = nb & ~7L;
    for (long i = 0; i < nb; i++) {
      const long t = reversed_batch(i, nb);
      CHECK(t >= 0 && t < nb, "nbatch=%ld: %ld -> %ld", nb, i, t);
      CHECK(seen.insert(t).second, "nbatch=%ld: %ld taken twice", nb, t);
      CHECK(t % 8 == i % 8, "nbatch=%ld: batch %ld moves to another XCD label in the backward walk (%ld)", nb, t, i);
      if (i < q) CHECK(t / 8 == (q - 1 - i) / 8, "nbatch=%ld: blocks of 8 not reversed", nb);
      else CHECK(t == i, "nbatch=%ld: the ragged last block moved", nb);
    }
  }
  printf("OK\n");
  return 0;
}
