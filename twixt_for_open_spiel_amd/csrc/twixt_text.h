// twixt_text.h -- host-side text of a board: Board::ToString (twixtboard.cc:278-448)
// and TwixTState::ActionToString (twixt.cc:67-74), rendered from the reference-shaped
// Cell bytes that twx_export_cells produces ({colour, links, blocked, flags} per cell,
// cell index x*N+y).  Host only: text has no throughput role.
//
// The three text rows of a board row are each 3 character slots per cell; a slot shows
// the link glyphs of a fixed list of (neighbour cell, direction) probes, or one space.
#ifndef TWIXT_TEXT_H_
#define TWIXT_TEXT_H_

#include <stdint.h>

#include <string>

namespace twx {
namespace text {

enum Dir { NNE, ENE, ESE, SSE, SSW, WSW, WNW, NNW };

struct Probe {
  int dx, dy, dir;
  char glyph;
  bool only_if_empty;  // probe skipped when the slot already shows something
};

// twixtboard.cc:379-403 (row above the pegs), 405-422 (peg row), 424-448 (row below)
static const Probe kBefore[3][3] = {
    {{-1, 0, ENE, '/', false}, {-1, -1, NNE, '/', false}, {0, 0, WNW, '_', false}},
    {{0, 0, NNE, '|', false}, {0, 0, NNW, '|', true}, {0, 0, 0, 0, false}},
    {{+1, 0, WNW, '\\', false}, {+1, -1, NNW, '\\', false}, {0, 0, ENE, '_', false}}};
static const Probe kPegL[2] = {{-1, -1, NNE, '|', false}, {0, 0, WSW, '_', false}};
static const Probe kPegR[2] = {{+1, -1, NNW, '|', false}, {0, 0, ESE, '_', false}};
static const Probe kAfter[3][4] = {
    {{+1, -1, WNW, '\\', false}, {0, -1, NNW, '\\', false}, {0, 0, 0, 0, false}, {0, 0, 0, 0, false}},
    {{-1, -1, ENE, '_', false}, {+1, -1, WNW, '_', false}, {0, 0, SSW, '|', false}, {0, 0, SSE, '|', true}},
    {{-1, -1, ENE, '/', false}, {0, -1, NNE, '/', false}, {0, 0, 0, 0, false}, {0, 0, 0, 0, false}}};

class Renderer {
 public:
  Renderer(const uint8_t *cells, int n, bool ansi) : c_(cells), n_(n), ansi_(ansi) {}

  std::string Board(bool swapped, int result) {
    s_.clear();
    s_ += "     ";
    for (int i = 0; i < n_; i++) Colored(kRed, std::string(1, (char)('a' + i)) + "  ");  // tb.cc:281-289
    s_ += "\n";
    for (int y = n_ - 1; y >= 0; y--) {
      s_ += "    ";
      for (int x = 0; x < n_; x++)
        for (int k = 0; k < 3; k++) Slot(kBefore[k], 3, x, y);
      s_ += "\n";
      s_ += (n_ - y < 10) ? "  " : " ";  // tb.cc:300-301
      Colored(kBlue, std::to_string(n_ - y) + " ");
      for (int x = 0; x < n_; x++) {
        Slot(kPegL, 2, x, y);
        Peg(x, y);
        Slot(kPegR, 2, x, y);
      }
      s_ += "\n";
      s_ += "    ";
      for (int x = 0; x < n_; x++)
        for (int k = 0; k < 3; k++) Slot(kAfter[k], 4, x, y);
      s_ += "\n";
    }
    s_ += "\n";
    if (swapped) s_ += "[swapped]";  // tb.cc:316-332
    if (result == 1) s_ += "[x has won]";
    else if (result == 2) s_ += "[o has won]";
    else if (result == 3) s_ += "[draw]";
    return s_;
  }

 private:
  static constexpr const char *kRed = "\x1b[91m", *kBlue = "\x1b[94m", *kOff = "\x1b[0m";  // tb.cc:22-25
  const uint8_t *c_;
  int n_;
  bool ansi_;
  std::string s_;

  bool OnBoard(int x, int y) const {  // tb.cc:625-631
    return x >= 0 && x < n_ && y >= 0 && y < n_ && !((x == 0 || x == n_ - 1) && (y == 0 || y == n_ - 1));
  }
  void Colored(const char *col, const std::string &t) {  // tb.cc:350-355
    if (ansi_) s_ += col;
    s_ += t;
    if (ansi_) s_ += kOff;
  }
  void Slot(const Probe *p, int count, int x, int y) {
    const size_t len = s_.size();
    for (int i = 0; i < count; i++) {
      if (!p[i].glyph) continue;
      if (p[i].only_if_empty && s_.size() != len) continue;
      const int cx = x + p[i].dx, cy = y + p[i].dy;
      if (!OnBoard(cx, cy)) continue;  // tb.cc:337-348
      const uint8_t *cell = c_ + 4 * (cx * n_ + cy);
      if (!(cell[1] & (1 << p[i].dir))) continue;
      const std::string g(1, p[i].glyph);
      if (cell[0] == 0) Colored(kRed, g);
      else if (cell[0] == 1) Colored(kBlue, g);
      else s_ += g;
    }
    if (s_.size() == len) s_ += " ";
  }
  void Peg(int x, int y) {  // tb.cc:357-377
    const int colour = c_[4 * (x * n_ + y)];
    if (colour == 0) Colored(kRed, "x");
    else if (colour == 1) Colored(kBlue, "o");
    else if (!OnBoard(x, y)) s_ += " ";
    else if (x == 0 || x == n_ - 1) Colored(kBlue, ".");
    else if (y == 0 || y == n_ - 1) Colored(kRed, ".");
    else s_ += ".";
  }
};

inline std::string BoardToString(const uint8_t *cells, int n, bool ansi, bool swapped, int result) {
  return Renderer(cells, n, ansi).Board(swapped, result);
}

// twixt.cc:67-74: "x"/"o" + column letter + row number counted from the top
inline std::string ActionToString(int n, int player, long action) {
  const int x = (int)(action / n), y = (int)(action % n);
  return std::string(player == 0 ? "x" : "o") + (char)('a' + x) + std::to_string(n - y);
}

}  // namespace text
}  // namespace twx
#endif  // TWIXT_TEXT_H_
