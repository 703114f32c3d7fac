/*
 * oracle/twixt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * CPU restatement of the reference's algorithm; see twixt_oracle.h for the
 * pinning statement and the citation key (tb.cc / tb.h / tc.h / tw.h / tw.cc).
 */
#include "twixt_oracle.h"

#include <stdio.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* tc.h:58-68 Compass + tb.cc:38-144 target offsets: NNE ENE ESE SSE SSW WSW WNW NNW */
static const int kDx[8] = {1, 2, 2, 1, -1, -2, -2, -1};
static const int kDy[8] = {2, 1, -1, -2, -2, -1, 1, 2};

/* tb.cc:38-144 kLinkDescriptorTable, blocking links: for a link leaving a peg
 * in direction `dir`, nine {dx, dy, canonical direction 0..3} triples naming
 * the links (by their from-cell relative to the peg) that cross it. */
static const signed char kBlock[8][9][3] = {
    /* NNE */ {{0, 1, 1}, {-1, 0, 1}, {0, 2, 2}, {0, 1, 2}, {-1, 2, 2}, {-1, 1, 2}, {0, 1, 3}, {0, 2, 3}, {0, 3, 3}},
    /* ENE */ {{0, -1, 0}, {1, 0, 0}, {-1, 1, 2}, {0, 1, 2}, {1, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 1, 3}, {1, 2, 3}},
    /* ESE */ {{0, -1, 0}, {1, -1, 0}, {0, -2, 0}, {1, -2, 0}, {-1, -1, 1}, {0, -1, 1}, {1, -1, 1}, {0, 1, 3}, {1, 0, 3}},
    /* SSE */ {{0, -1, 0}, {0, -2, 0}, {0, -3, 0}, {-1, -1, 1}, {0, -1, 1}, {-1, -2, 1}, {0, -2, 1}, {-1, 0, 2}, {0, -1, 2}},
    /* SSW */ {{-1, -1, 1}, {-2, -2, 1}, {-2, 0, 2}, {-1, 0, 2}, {-2, -1, 2}, {-1, -1, 2}, {-1, 1, 3}, {-1, 0, 3}, {-1, -1, 3}},
    /* WSW */ {{-2, -2, 0}, {-1, -1, 0}, {-3, 0, 2}, {-2, 0, 2}, {-1, 0, 2}, {-2, 1, 3}, {-1, 1, 3}, {-2, 0, 3}, {-1, 0, 3}},
    /* WNW */ {{-2, 0, 0}, {-1, 0, 0}, {-2, -1, 0}, {-1, -1, 0}, {-3, 0, 1}, {-2, 0, 1}, {-1, 0, 1}, {-2, 2, 3}, {-1, 1, 3}},
    /* NNW */ {{-1, 1, 0}, {-1, 0, 0}, {-1, -1, 0}, {-2, 1, 1}, {-1, 1, 1}, {-2, 0, 1}, {-1, 0, 1}, {-2, 2, 2}, {-1, 1, 2}},
};

/* tb.cc:28-30 */
static int opp_dir(int d) { return (d + 4) % 8; }

int orc_size_ok(int n) { return n >= ORC_MIN_N && n <= ORC_MAX_N; } /* tw.cc:139 */

/* tb.cc:625-631: out of range or one of the four corners */
int orc_off_board(int n, int x, int y) {
  return (y < 0 || y > n - 1 || x < 0 || x > n - 1 ||
          ((x == 0 || x == n - 1) && (y == 0 || y == n - 1)));
}

/* tb.cc:615-623 */
int orc_on_border(int n, int player, int x, int y) {
  if (player == ORC_RED) return ((y == 0 || y == n - 1) && (x > 0 && x < n - 1));
  return ((x == 0 || x == n - 1) && (y > 0 && y < n - 1));
}

/* tb.cc:252-276 InitializeLegalActions: two ascending lists */
static void init_legal(orc_state *s) {
  int n = s->n;
  s->nlegal[0] = s->nlegal[1] = 0;
  for (int col = 0; col < n; col++) {
    for (int row = 0; row < n; row++) {
      int a = col * n + row;
      if (orc_off_board(n, col, row)) continue;
      if (orc_on_border(n, ORC_RED, col, row)) {
        s->legal[0][s->nlegal[0]++] = a;
      } else if (orc_on_border(n, ORC_BLUE, col, row)) {
        s->legal[1][s->nlegal[1]++] = a;
      } else {
        s->legal[0][s->nlegal[0]++] = a;
        s->legal[1][s->nlegal[1]++] = a;
      }
    }
  }
}

/* tb.cc:168-174 Board ctor -> tb.cc:209-236 InitializeCells (the BlockerMap
 * of tb.cc:176-190 is replaced by probing kBlock directly, see is_blocked) */
void orc_init(orc_state *s, int n) {
  memset(s, 0, sizeof(*s));
  s->n = n;
  s->result = ORC_OPEN;
  s->current_player = ORC_RED; /* tw.h:107 */
  for (int x = 0; x < n; x++) {
    for (int y = 0; y < n; y++) {
      orc_cell *c = &s->cell[x][y];
      if (orc_off_board(n, x, y)) {
        c->color = ORC_OFFBOARD;
      } else {
        c->color = ORC_EMPTY;
        if (x == 0) c->ltb[ORC_BLUE][ORC_START] = 1;          /* tb.cc:223-224 */
        else if (x == n - 1) c->ltb[ORC_BLUE][ORC_END] = 1;   /* tb.cc:225-226 */
        else if (y == 0) c->ltb[ORC_RED][ORC_START] = 1;      /* tb.cc:227-228 */
        else if (y == n - 1) c->ltb[ORC_RED][ORC_END] = 1;    /* tb.cc:229-230 */
      }
    }
  }
  init_legal(s);
}

/* tb.cc:633-640 RemoveLegalAction: find + erase, keeps order */
static void remove_legal(orc_state *s, int player, int action) {
  int *la = s->legal[player];
  int cnt = s->nlegal[player];
  for (int i = 0; i < cnt; i++) {
    if (la[i] == action) {
      memmove(la + i, la + i + 1, (size_t)(cnt - i - 1) * sizeof(int));
      s->nlegal[player] = cnt - 1;
      return;
    }
  }
}

/* tb.cc:519-527 with the blocker set of tb.cc:176-190: a blocker entry exists
 * only if both of its end cells are on the board; both stored orientations of
 * a crossing link are probed (tb.cc:185-186). */
static int is_blocked(const orc_state *s, int x, int y, int dir) {
  int n = s->n;
  for (int i = 0; i < 9; i++) {
    int fx = x + kBlock[dir][i][0], fy = y + kBlock[dir][i][1], d = kBlock[dir][i][2];
    if (orc_off_board(n, fx, fy)) continue;
    int tx = fx + kDx[d], ty = fy + kDy[d];
    if (orc_off_board(n, tx, ty)) continue;
    if (s->cell[fx][fy].links & (1 << d)) return 1;
    if (s->cell[tx][ty].links & (1 << opp_dir(d))) return 1;
  }
  return 0;
}

/* tb.cc:573-588 ExploreLocalGraph.  The by-value `visited` set of the
 * reference never changes the outcome: a cell enters it only right after its
 * flag was set, and flagged cells are rejected by the flag test anyway. */
static void explore(orc_state *s, int player, int x, int y, int border) {
  for (int dir = 0; dir < 8; dir++) {
    if (s->cell[x][y].links & (1 << dir)) {
      int tx = x + kDx[dir], ty = y + kDy[dir];
      orc_cell *t = &s->cell[tx][ty];
      if (!t->ltb[player][border]) {
        t->ltb[player][border] = 1;
        explore(s, player, tx, ty, border);
      }
    }
  }
}

/* tb.cc:501-571 SetPegAndLinks */
static void set_peg_and_links(orc_state *s, int player, int x, int y) {
  int n = s->n;
  int linked_to_neutral = 0, new_links = 0;
  orc_cell *cell = &s->cell[x][y];
  cell->color = player;
  for (int dir = 0; dir < 8; dir++) {
    int tx = x + kDx[dir], ty = y + kDy[dir];
    if (orc_off_board(n, tx, ty)) continue;
    orc_cell *t = &s->cell[tx][ty];
    if (t->color != cell->color) continue;
    if (!is_blocked(s, x, y, dir)) {
      cell->links |= 1 << dir;           /* tb.cc:532 */
      t->links |= 1 << opp_dir(dir);     /* tb.cc:533 */
      new_links = 1;
      if (t->ltb[player][ORC_START]) {   /* tb.cc:538-546 */
        cell->ltb[player][ORC_START] = 1;
      } else if (t->ltb[player][ORC_END]) {
        cell->ltb[player][ORC_END] = 1;
      } else {
        linked_to_neutral = 1;
      }
    } else {
      cell->blocked |= 1 << dir;         /* tb.cc:550-551 */
      t->blocked |= 1 << opp_dir(dir);
    }
  }
  if (new_links) { /* tb.cc:558-570 */
    if (cell->ltb[player][ORC_START] && linked_to_neutral) explore(s, player, x, y, ORC_START);
    if (cell->ltb[player][ORC_END] && linked_to_neutral) explore(s, player, x, y, ORC_END);
  }
}

/* tb.cc:192-207 UpdateResult */
static void update_result(orc_state *s, int player, int x, int y) {
  const orc_cell *c = &s->cell[x][y];
  if (c->ltb[player][ORC_START] && c->ltb[player][ORC_END]) {
    s->result = (player == ORC_RED) ? ORC_RED_WIN : ORC_BLUE_WIN;
    return;
  }
  if (s->nlegal[1 - player] == 0) s->result = ORC_DRAW;
}

/* tb.cc:457-499 Board::ApplyAction (no validation, like the reference) */
void orc_board_apply(orc_state *s, int player, int action) {
  int n = s->n;
  int x = action / n, y = action % n; /* tb.cc:599-601 */
  if (s->move_counter == 1) {
    if (x == s->m1x && y == s->m1y) {
      s->swapped = 1;                                  /* tb.cc:464 */
      s->cell[s->m1x][s->m1y].color = ORC_EMPTY;       /* tb.cc:450-455 UndoFirstMove */
      init_legal(s);
      int nx = y, ny = n - x - 1;                      /* tb.cc:471-473 */
      x = nx;
      y = ny;
    } else {
      remove_legal(s, ORC_RED, s->m1x * n + s->m1y);   /* tb.cc:478-479 */
      remove_legal(s, ORC_BLUE, s->m1x * n + s->m1y);
    }
  }
  set_peg_and_links(s, player, x, y);
  if (s->move_counter == 0) {
    s->m1x = x;                                        /* tb.cc:488 */
    s->m1y = y;
  } else {
    remove_legal(s, ORC_RED, x * n + y);               /* tb.cc:491-492 */
    remove_legal(s, ORC_BLUE, x * n + y);
  }
  s->move_counter++;
  update_result(s, player, x, y);
}

int orc_is_terminal(const orc_state *s) { return s->result != ORC_OPEN; } /* tw.h:45-48 */

/* tw.h:86-90 */
int orc_legal_actions(const orc_state *s, int *out) {
  if (orc_is_terminal(s)) return 0;
  int p = s->current_player;
  if (out) memcpy(out, s->legal[p], (size_t)s->nlegal[p] * sizeof(int));
  return s->nlegal[p];
}

/* tw.h:93-104 DoApplyAction */
int orc_apply(orc_state *s, int action) {
  if (orc_is_terminal(s)) return -1;
  int p = s->current_player, found = 0;
  for (int i = 0; i < s->nlegal[p]; i++)
    if (s->legal[p][i] == action) { found = 1; break; }
  if (!found) return -1; /* "Not a legal action: <a>" */
  orc_board_apply(s, p, action);
  s->current_player = (s->result == ORC_OPEN) ? 1 - p : ORC_TERMINAL_PLAYER;
  return 0;
}

/* tw.h:50-63 */
void orc_returns(const orc_state *s, double out[2]) {
  out[0] = out[1] = 0.0;
  if (s->result == ORC_RED_WIN) { out[0] = 1.0; out[1] = -1.0; }
  else if (s->result == ORC_BLUE_WIN) { out[0] = -1.0; out[1] = 1.0; }
}

/* tw.cc:76-132 ObservationTensor, tb.cc:590-597 GetTensorPosition */
void orc_observation(const orc_state *s, float *out) {
  int n = s->n;
  memset(out, 0, sizeof(float) * 12 * (size_t)n * (size_t)(n - 2));
  for (int c = 0; c < n; c++) {
    for (int r = 0; r < n; r++) {
      const orc_cell *cell = &s->cell[c][r];
      int offset, tx, ty;
      if (cell->color == ORC_RED) { offset = 0; tx = n - r - 1; ty = c - 1; }
      else if (cell->color == ORC_BLUE) { offset = 6; tx = n - c - 1; ty = n - r - 2; }
      else continue;
      if (cell->links > 0) {
        for (int dir = 0; dir < 4; dir++)
          if (cell->links & (1 << dir)) out[((offset + 1 + dir) * n + tx) * (n - 2) + ty] = 1.0f;
      } else {
        out[((offset + 0) * n + tx) * (n - 2) + ty] = 1.0f;
      }
      if (cell->blocked & 15) out[((offset + 5) * n + tx) * (n - 2) + ty] = 1.0f;
    }
  }
}

/* ---- text (tb.cc:22-25, 278-448) ---- */
typedef struct { char *b; int len, cap, ansi; } sbuf;
static void sb_put(sbuf *s, const char *t) {
  while (*t) { if (s->len + 1 < s->cap) s->b[s->len] = *t; s->len++; t++; }
}
static void sb_color(sbuf *s, const char *col, const char *t) { /* tb.cc:350-355 */
  if (s->ansi) sb_put(s, col);
  sb_put(s, t);
  if (s->ansi) sb_put(s, "\x1b[0m");
}
#define ANSI_RED "\x1b[91m"
#define ANSI_BLUE "\x1b[94m"
static void link_char(const orc_state *st, sbuf *s, int x, int y, int dir, const char *ch) { /* tb.cc:337-348 */
  if (orc_off_board(st->n, x, y)) return;
  const orc_cell *c = &st->cell[x][y];
  if (!(c->links & (1 << dir))) return;
  if (c->color == ORC_RED) sb_color(s, ANSI_RED, ch);
  else if (c->color == ORC_BLUE) sb_color(s, ANSI_BLUE, ch);
  else sb_put(s, ch);
}
static void peg_char(const orc_state *st, sbuf *s, int x, int y) { /* tb.cc:357-377 */
  int n = st->n, col = st->cell[x][y].color;
  if (col == ORC_RED) sb_color(s, ANSI_RED, "x");
  else if (col == ORC_BLUE) sb_color(s, ANSI_BLUE, "o");
  else if (orc_off_board(n, x, y)) sb_put(s, " ");
  else if (x == 0 || x == n - 1) sb_color(s, ANSI_BLUE, ".");
  else if (y == 0 || y == n - 1) sb_color(s, ANSI_RED, ".");
  else sb_put(s, ".");
}
enum { NNE, ENE, ESE, SSE, SSW, WSW, WNW, NNW };
static void before_row(const orc_state *st, sbuf *s, int x, int y) { /* tb.cc:379-403 */
  int len = s->len;
  link_char(st, s, x - 1, y, ENE, "/");
  link_char(st, s, x - 1, y - 1, NNE, "/");
  link_char(st, s, x, y, WNW, "_");
  if (len == s->len) sb_put(s, " ");
  len = s->len;
  link_char(st, s, x, y, NNE, "|");
  if (len == s->len) link_char(st, s, x, y, NNW, "|");
  if (len == s->len) sb_put(s, " ");
  len = s->len;
  link_char(st, s, x + 1, y, WNW, "\\");
  link_char(st, s, x + 1, y - 1, NNW, "\\");
  link_char(st, s, x, y, ENE, "_");
  if (len == s->len) sb_put(s, " ");
}
static void peg_row(const orc_state *st, sbuf *s, int x, int y) { /* tb.cc:405-422 */
  int len = s->len;
  link_char(st, s, x - 1, y - 1, NNE, "|");
  link_char(st, s, x, y, WSW, "_");
  if (len == s->len) sb_put(s, " ");
  peg_char(st, s, x, y);
  len = s->len;
  link_char(st, s, x + 1, y - 1, NNW, "|");
  link_char(st, s, x, y, ESE, "_");
  if (len == s->len) sb_put(s, " ");
}
static void after_row(const orc_state *st, sbuf *s, int x, int y) { /* tb.cc:424-448 */
  int len = s->len;
  link_char(st, s, x + 1, y - 1, WNW, "\\");
  link_char(st, s, x, y - 1, NNW, "\\");
  if (len == s->len) sb_put(s, " ");
  len = s->len;
  link_char(st, s, x - 1, y - 1, ENE, "_");
  link_char(st, s, x + 1, y - 1, WNW, "_");
  link_char(st, s, x, y, SSW, "|");
  if (len == s->len) link_char(st, s, x, y, SSE, "|");
  if (len == s->len) sb_put(s, " ");
  len = s->len;
  link_char(st, s, x - 1, y - 1, ENE, "/");
  link_char(st, s, x, y - 1, NNE, "/");
  if (len == s->len) sb_put(s, " ");
}
/* tb.cc:278-335; returns the full length (like snprintf) */
int orc_to_string(const orc_state *st, int ansi, char *buf, int cap) {
  sbuf s = {buf, 0, cap, ansi};
  int n = st->n;
  char t[16];
  sb_put(&s, "     ");
  for (int y = 0; y < n; y++) {
    t[0] = (char)('a' + y); t[1] = ' '; t[2] = ' '; t[3] = 0;
    sb_color(&s, ANSI_RED, t);
  }
  sb_put(&s, "\n");
  for (int y = n - 1; y >= 0; y--) {
    sb_put(&s, "    ");
    for (int x = 0; x < n; x++) before_row(st, &s, x, y);
    sb_put(&s, "\n");
    sb_put(&s, (n - y < 10) ? "  " : " ");
    snprintf(t, sizeof t, "%d ", n - y);
    sb_color(&s, ANSI_BLUE, t);
    for (int x = 0; x < n; x++) peg_row(st, &s, x, y);
    sb_put(&s, "\n");
    sb_put(&s, "    ");
    for (int x = 0; x < n; x++) after_row(st, &s, x, y);
    sb_put(&s, "\n");
  }
  sb_put(&s, "\n");
  if (st->swapped) sb_put(&s, "[swapped]");
  if (st->result == ORC_RED_WIN) sb_put(&s, "[x has won]");
  else if (st->result == ORC_BLUE_WIN) sb_put(&s, "[o has won]");
  else if (st->result == ORC_DRAW) sb_put(&s, "[draw]");
  if (cap > 0) buf[s.len < cap ? s.len : cap - 1] = 0;
  return s.len;
}

/* tw.cc:67-74 */
int orc_action_to_string(const orc_state *s, int player, int action, char *buf) {
  int x = action / s->n, y = action % s->n;
  return sprintf(buf, "%c%c%d", player == ORC_RED ? 'x' : 'o', 'a' + x, s->n - y);
}

/* ---- parity views ---- */
void orc_export_cells(const orc_state *s, uint8_t *out) {
  int n = s->n;
  for (int x = 0; x < n; x++)
    for (int y = 0; y < n; y++) {
      const orc_cell *c = &s->cell[x][y];
      uint8_t *o = out + 4 * (x * n + y);
      o[0] = (uint8_t)c->color;
      o[1] = (uint8_t)(c->links & 0xff);
      o[2] = (uint8_t)(c->blocked & 0xff);
      o[3] = (uint8_t)(c->ltb[0][0] | c->ltb[0][1] << 1 | c->ltb[1][0] << 2 | c->ltb[1][1] << 3);
    }
}

int orc_export_planes(const orc_state *s, uint32_t *out) {
  int n = s->n, W = (n * n + 31) / 32;
  memset(out, 0, sizeof(uint32_t) * (size_t)(8 * W + 4));
  for (int x = 0; x < n; x++)
    for (int y = 0; y < n; y++) {
      const orc_cell *c = &s->cell[x][y];
      int i = x * n + y;
      uint32_t bit = 1u << (i & 31);
      int w = i >> 5;
      if (c->color == ORC_RED) out[0 * W + w] |= bit;
      if (c->color == ORC_BLUE) out[1 * W + w] |= bit;
      for (int d = 0; d < 4; d++)
        if (c->links & (1 << d)) out[(2 + d) * W + w] |= bit;
      /* flag planes: the flag of the colour that owns / can own the cell */
      int fs, fe;
      if (c->color == ORC_RED || c->color == ORC_BLUE) {
        fs = c->ltb[c->color][ORC_START];
        fe = c->ltb[c->color][ORC_END];
      } else {
        fs = c->ltb[0][ORC_START] | c->ltb[1][ORC_START];
        fe = c->ltb[0][ORC_END] | c->ltb[1][ORC_END];
      }
      if (fs) out[6 * W + w] |= bit;
      if (fe) out[7 * W + w] |= bit;
    }
  int m1 = (s->move_counter > 0) ? s->m1x * n + s->m1y : 0;
  out[8 * W + 0] = (uint32_t)s->move_counter | ((uint32_t)m1 << 16);
  out[8 * W + 1] = (uint32_t)s->result | ((uint32_t)s->swapped << 8);
  return 8 * W + 4;
}

/* ---- SURVEY.md 8(c) KATs ---- */
#define FNV_OFF 0xcbf29ce484222325ULL
#define FNV_PRIME 0x100000001b3ULL
static uint64_t fnv_byte(uint64_t h, uint8_t b) { return (h ^ b) * FNV_PRIME; }

void orc_kat_game(int n, uint64_t seed, int forced_swap, orc_kat *out) {
  static _Thread_local orc_state st;
  orc_state *s = &st;
  orc_init(s, n);
  uint64_t rng = seed, hh = FNV_OFF, hs = FNV_OFF, hf = FNV_OFF;
  int first = -1;
  int la[ORC_MAX_LEGAL];
  while (!orc_is_terminal(s)) {
    int cnt = orc_legal_actions(s, la);
    rng = rng * 6364136223846793005ULL + 1442695040888963407ULL;
    int a = la[(rng >> 33) % (uint64_t)cnt];
    if (forced_swap && s->move_counter == 1) {
      for (int i = 0; i < cnt; i++) if (la[i] == first) a = first;
    }
    if (s->move_counter == 0) first = a;
    orc_apply(s, a);
    hh = fnv_byte(fnv_byte(hh, (uint8_t)(a & 0xff)), (uint8_t)(a >> 8));
    int term = orc_is_terminal(s);
    int nc = term ? 0 : orc_legal_actions(s, NULL);
    hs = fnv_byte(hs, term ? 0xfc : (uint8_t)s->current_player);
    hs = fnv_byte(hs, (uint8_t)s->result);
    hs = fnv_byte(hs, (uint8_t)(nc & 0xff));
    hs = fnv_byte(hs, (uint8_t)(nc >> 8));
  }
  int links = 0;
  for (int x = 0; x < n; x++)
    for (int y = 0; y < n; y++) {
      const orc_cell *c = &s->cell[x][y];
      hf = fnv_byte(hf, (uint8_t)c->color);
      hf = fnv_byte(hf, (uint8_t)(c->links & 0xff));
      hf = fnv_byte(hf, (uint8_t)((c->blocked & 15) ? 1 : 0));
      for (int d = 0; d < 4; d++) links += (c->links >> d) & 1;
    }
  out->moves = s->move_counter;
  out->result = s->result;
  out->swapped = s->swapped;
  out->links = links;
  out->hist_fnv = hh;
  out->step_fnv = hs;
  out->final_fnv = hf;
}

/* ---- counter-based policy (specification shared with the HIP engine) ---- */
uint32_t orc_policy_hash(uint32_t seed, uint32_t board, uint32_t episode, uint32_t move) {
  uint32_t h = seed ^ (board * 0x9E3779B1u) ^ (episode * 0xC2B2AE3Du) ^ (move * 0x85EBCA77u);
  h ^= h >> 16; h *= 0x7feb352du;
  h ^= h >> 15; h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}

int64_t orc_rollout(orc_state *s, uint32_t seed, uint32_t board, uint32_t *episode,
                    int nsteps, int auto_reset, int32_t *actions_out, uint8_t *status_out) {
  int64_t played = 0;
  int n = s->n;
  for (int t = 0; t < nsteps; t++) {
    int was_reset = 0;
    if (orc_is_terminal(s)) {
      if (!auto_reset) {
        if (actions_out) actions_out[t] = -1;
        if (status_out) status_out[t] = (uint8_t)(s->result | (s->swapped << 5) | (1 << 6));
        continue;
      }
      orc_init(s, n);
      (*episode)++;
      was_reset = 1;
    }
    int p = s->current_player;
    int cnt = s->nlegal[p];
    uint32_t h = orc_policy_hash(seed, board, *episode, (uint32_t)s->move_counter);
    int k = (int)(((uint64_t)h * (uint64_t)cnt) >> 32);
    int a = s->legal[p][k];
    orc_apply(s, a);
    played++;
    if (actions_out) actions_out[t] = a;
    if (status_out) status_out[t] = (uint8_t)(s->result | (was_reset << 4) | (s->swapped << 5));
  }
  return played;
}

int64_t orc_rollout_batch(int n, uint32_t seed, uint32_t first_board, int nboards, int nsteps,
                          int auto_reset, int threads, uint32_t *planes_out, uint8_t *cells_out,
                          int32_t *actions_out, uint8_t *status_out) {
  int64_t total = 0;
  const int SW = 8 * ((n * n + 31) / 32) + 4;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel reduction(+ : total)
#endif
  {
    orc_state st;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (int b = 0; b < nboards; b++) {
      uint32_t ep = 0;
      orc_init(&st, n);
      total += orc_rollout(&st, seed, first_board + (uint32_t)b, &ep, nsteps, auto_reset,
                           actions_out ? actions_out + (size_t)b * nsteps : NULL,
                           status_out ? status_out + (size_t)b * nsteps : NULL);
      if (planes_out) {
        orc_export_planes(&st, planes_out + (size_t)b * SW);
        planes_out[(size_t)b * SW + SW - 2] = ep;
      }
      if (cells_out) orc_export_cells(&st, cells_out + (size_t)b * n * n * 4);
    }
  }
  return total;
}

void orc_block_entry(int dir, int i, int out[3]) {
  out[0] = kBlock[dir][i][0];
  out[1] = kBlock[dir][i][1];
  out[2] = kBlock[dir][i][2];
}

int64_t orc_bench(int n, uint32_t seed, int nboards, int nsteps, int threads, uint64_t *digest) {
  int64_t total = 0;
  uint64_t dg = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel reduction(+ : total) reduction(^ : dg)
#endif
  {
    orc_state st;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int b = 0; b < nboards; b++) {
      uint32_t ep = 0;
      orc_init(&st, n);
      total += orc_rollout(&st, seed, (uint32_t)b, &ep, nsteps, 1, NULL, NULL);
      uint64_t h = FNV_OFF;
      h = fnv_byte(h, (uint8_t)st.result);
      h = fnv_byte(h, (uint8_t)(st.move_counter & 0xff));
      h = fnv_byte(h, (uint8_t)(st.move_counter >> 8));
      dg ^= h * (uint64_t)(2 * b + 1);
    }
  }
  if (digest) *digest = dg;
  return total;
}
