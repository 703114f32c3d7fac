// twixt_engine.hip -- libtwixt_hip.so: batched TwixT board engine for MI355X
// (gfx950 / CDNA4).  Implements include/twixt_hip.h.
//
// Hot path (reference: twixtboard.cc:457-588 Board::ApplyAction /
// SetPegAndLinks / ExploreLocalGraph / UpdateResult, twixt.h:86-104
// LegalActions / DoApplyAction) = step_kernel<N>:
//   * one wavefront = 64 boards, one board per lane;
//   * the wave streams its 64 boards (CH 16-byte chunks each, layout
//     state[chunk][board]) into LDS with coalesced 1 KiB loads, works on them
//     there (data-dependent word access is a single ds_read), and streams them
//     back with 1 KiB stores -> HBM traffic is exactly 2 x state bytes / step;
//   * legal mask + k-th-legal-action pick are whole-plane popcount scans;
//   * the 8-direction link scan and the 72 crossing probes are constant-mask
//     tests on a 5x7 window of the link planes (twixt_geom.h kCross);
//   * the rare border-flag flood fill (5.8 % of steps, twixtboard.cc:573-588)
//     is done by the whole wave cooperating on one board at a time
//     (lane = board column, frontier expansion with shuffles + ballot), so a
//     long fill never leaves 63 lanes idle behind one.
// Integer / bitwise work only: no MFMA, the bound is HBM bandwidth.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/twixt_hip.h"
#include "twixt_geom.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(TWX_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// LDS addressing: wave region of CH chunks; chunk c of lane l at dword
// c*256 + l*4 (+0..3).  Word w of a board lives in chunk w>>2, slot w&3.
// A per-lane word index therefore always maps lane l to banks 4l..4l+3 (mod 32).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lds_idx(int w, int lane) {
  return ((w >> 2) << 8) + (lane << 2) + (w & 3);
}

__device__ __forceinline__ uint32_t funnel(uint32_t lo, uint32_t hi, int sh) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> sh);
}

// N-bit column `c` (bits c*N .. c*N+N-1) of the plane starting at word `base`
// of the board held by `lane`; 0 when c is off the board.
template <int N>
__device__ __forceinline__ uint32_t lds_column(const uint32_t *lds, int lane, int base, int c) {
  const bool ok = (unsigned)c < (unsigned)N;
  const int o = (ok ? c : 0) * N;
  const int w0 = base + (o >> 5);
  const uint32_t lo = lds[lds_idx(w0, lane)];
  const uint32_t hi = lds[lds_idx(w0 + 1, lane)];  // at worst the first scalar word: in bounds
  const uint32_t v = funnel(lo, hi, o & 31) & ((1u << N) - 1u);
  return ok ? v : 0u;
}

// position of the k-th (0-based) set bit of a non-zero-enough word
__device__ __forceinline__ int select_bit(uint32_t t, int k) {
  int pos = 0;
  int c = __popc(t & 0xffffu);
  if (k >= c) { k -= c; pos += 16; t >>= 16; }
  c = __popc(t & 0xffu);
  if (k >= c) { k -= c; pos += 8; t >>= 8; }
  c = __popc(t & 0xfu);
  if (k >= c) { k -= c; pos += 4; t >>= 4; }
  c = __popc(t & 0x3u);
  if (k >= c) { k -= c; pos += 2; t >>= 2; }
  c = (int)(t & 1u);
  if (k >= c) { pos += 1; }
  return pos;
}

constexpr int kFlagRandom = 1, kFlagAutoReset = 2;

template <int N>
__global__ __launch_bounds__(kWave) void step_kernel(uint4 *__restrict__ state, long bpad, long batch,
                                                     const int32_t *__restrict__ actions,
                                                     int32_t *__restrict__ actions_out,
                                                     uint8_t *__restrict__ status, uint32_t seed,
                                                     uint32_t first_id, int flags) {
  using G = twx::Geo<N>;
  constexpr int W = G::W, CH = G::CH, SW = G::SW;
  constexpr int PR = twx::kPlaneRed * W, PB = twx::kPlaneBlue * W, PL = twx::kPlaneLink0 * W,
                PS = twx::kPlaneStart * W, PE = twx::kPlaneEnd * W, SC = twx::kNumPlanes * W;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint4 *lds4 = reinterpret_cast<uint4 *>(lds);

  const int lane = threadIdx.x;
  const long board = (long)blockIdx.x * kWave + lane;
  const bool inb = board < batch;

  // ---- stream the 64 boards in (allocation is padded to a multiple of 64) ----
#pragma unroll
  for (int c = 0; c < CH; c++) lds4[c * kWave + lane] = state[(long)c * bpad + board];
  __syncthreads();

  auto L = [&](int w) -> uint32_t & { return lds[lds_idx(w, lane)]; };

  uint32_t counter = L(SC) & 0xffffu, move_one = L(SC) >> 16;
  uint32_t result = L(SC + 1) & 0xffu, swapped = (L(SC + 1) >> 8) & 1u;
  uint32_t episode = L(SC + 2);
  const bool random = (flags & kFlagRandom) != 0;
  bool was_reset = false;

  // ---- auto-reset: Board::Board (tb.cc:168-174, 209-276) in place ----
  if (random && (flags & kFlagAutoReset) && result != 0u && inb) {
#pragma unroll
    for (int w = 0; w < SC; w++) L(w) = G::tab.init[w];
    counter = 0; move_one = 0; result = 0; swapped = 0; episode += 1;
    was_reset = true;
  }

  const int player = (int)(counter & 1u);
  const bool open = (result == 0u);

  // ---- legal mask of the player to move (tw.h:86-90) and the
  //      opponent's free-cell count (for the draw test of tb.cc:202-206) ----
  uint32_t m[W];
  int cnt = 0, cnt_opp = 0;
#pragma unroll
  for (int i = 0; i < W; i++) {
    const uint32_t r = L(PR + i), b = L(PB + i);
    const uint32_t reg_cur = player ? G::tab.region[1][i] : G::tab.region[0][i];
    const uint32_t reg_opp = player ? G::tab.region[0][i] : G::tab.region[1][i];
    const uint32_t occ = b | (counter == 1u ? 0u : r);  // red's first peg stays legal for blue: the swap
    m[i] = reg_cur & ~occ;
    cnt += __popc(m[i]);
    cnt_opp += __popc(reg_opp & ~(r | b));
  }

  // ---- choose / validate the action ----
  int a = -1;
  bool act = inb && open;
  bool illegal = false;
  if (random) {
    act = act && cnt > 0;
    const uint32_t h = twx::policy_hash(seed, first_id + (uint32_t)board, episode, counter);
    int k = (int)__umulhi(h, (uint32_t)cnt);
    int cum = 0, wsel = 0, krem = 0;
    uint32_t msel = m[0];
#pragma unroll
    for (int i = 0; i < W; i++) {
      const int c = __popc(m[i]);
      if (k >= cum && k < cum + c) { wsel = i; msel = m[i]; krem = k - cum; }
      cum += c;
    }
    a = wsel * 32 + select_bit(msel, krem);
    if (!act) a = -1;
  } else {
    a = inb ? actions[board] : -1;
    bool legal = false;
    if (a >= 0 && a < N * N) {
      const int ax = a / N, ay = a - ax * N;
      const bool inreg = player ? (ay >= 1 && ay <= N - 2) : (ax >= 1 && ax <= N - 2);
      const uint32_t bit = 1u << (a & 31);
      const uint32_t r = L(PR + (a >> 5)) & bit, b = L(PB + (a >> 5)) & bit;
      const bool occ = (b != 0u) || (counter != 1u && r != 0u);
      legal = inreg && !occ;
    }
    illegal = inb && !(open && legal);  // tw.h:95-97 (terminal: LegalActions() is empty)
    act = act && legal;
  }

  int pa = 0;         // cell the peg finally lands on
  uint32_t pend = 0;  // bit0: flood START from pa, bit1: flood END from pa
  if (act) {
    // ---- Board::ApplyAction (tb.cc:457-499) ----
    pa = a;
    const bool swapnow = (counter == 1u) && ((uint32_t)a == move_one);
    if (swapnow) {  // tb.cc:462-473: undo red's first peg, rotate 90 degrees
      swapped = 1u;
      L(PR + (a >> 5)) &= ~(1u << (a & 31));
      const int ax = a / N, ay = a - ax * N;
      pa = ay * N + (N - 1 - ax);
    }
    const int x = pa / N, y = pa - x * N;
    const int own = player ? PB : PR;
    L(own + (pa >> 5)) |= 1u << (pa & 31);  // tb.cc:507-508 set peg

    // ---- SetPegAndLinks (tb.cc:501-571), all 8 directions at once ----
    // 5x5 windows (columns x-2..x+2, rows y-2..y+2) of own pegs / START / END
    uint32_t wo[5], ws[5], we[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      wo[i] = ((lds_column<N>(lds, lane, own, x + i - 2) << 2) >> y) & 0x1fu;
      ws[i] = ((lds_column<N>(lds, lane, PS, x + i - 2) << 2) >> y) & 0x1fu;
      we[i] = ((lds_column<N>(lds, lane, PE, x + i - 2) << 2) >> y) & 0x1fu;
    }
    uint32_t cand = 0, ts = 0, te = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      const int ci = twx::kDX[d] + 2, rb = twx::kDY[d] + 2;
      cand |= ((wo[ci] >> rb) & 1u) << d;  // same-colour peg a knight's move away (tb.cc:517)
      ts |= ((ws[ci] >> rb) & 1u) << d;
      te |= ((we[ci] >> rb) & 1u) << d;
    }
    uint32_t blocked = 0;
    if (cand) {
      // crossing test (tb.cc:519-527): 5 columns x 7 rows x 4 link planes, then
      // constant masks (twixt_geom.h)
      uint32_t win[twx::kWinCols];
#pragma unroll
      for (int i = 0; i < twx::kWinCols; i++) {
        uint32_t v = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t col = lds_column<N>(lds, lane, PL + e * W, x + i - 3);
          v |= (((col << 3) >> y) & 0x7fu) << (8 * e);
        }
        win[i] = v;
      }
#pragma unroll
      for (int d = 0; d < 8; d++) {
        uint32_t hit = 0;
#pragma unroll
        for (int i = 0; i < twx::kWinCols; i++) hit |= win[i] & twx::kCross.m[d][i];
        blocked |= (hit != 0u ? 1u : 0u) << d;
      }
    }
    const uint32_t linkm = cand & ~blocked;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      if (linkm & (1u << d)) {  // tb.cc:532-533: one stored bit per physical link, at the west end
        const int cell = (d < 4) ? pa : pa + twx::kDX[d] * N + twx::kDY[d];
        L(PL + (d & 3) * W + (cell >> 5)) |= 1u << (cell & 31);
      }
    }
    // tb.cc:538-546: inherit START, else END, else remember a neutral neighbour
    const uint32_t gain_s = linkm & ts, gain_e = linkm & ~ts & te, neutral = linkm & ~ts & ~te;
    const bool p_s = ((ws[2] >> 2) & 1u) || gain_s;
    const bool p_e = ((we[2] >> 2) & 1u) || gain_e;
    if (gain_s) L(PS + (pa >> 5)) |= 1u << (pa & 31);
    if (gain_e) L(PE + (pa >> 5)) |= 1u << (pa & 31);
    if (linkm && neutral) pend = (p_s ? 1u : 0u) | (p_e ? 2u : 0u);  // tb.cc:558-570

    if (counter == 0u) move_one = (uint32_t)pa;  // tb.cc:485-488
    counter += 1u;
    // ---- UpdateResult (tb.cc:192-207) ----
    if (p_s && p_e) {
      result = player ? 2u : 1u;
    } else if (counter > 2u) {
      const bool in_opp = player ? (x >= 1 && x <= N - 2) : (y >= 1 && y <= N - 2);
      if (cnt_opp - (in_opp ? 1 : 0) == 0) result = 3u;
    }
  }
  L(SC) = counter | (move_one << 16);
  L(SC + 1) = result | (swapped << 8);
  L(SC + 2) = episode;
  L(SC + 3) = 0u;
  __syncthreads();

  // ---- ExploreLocalGraph (tb.cc:573-588), wave-cooperative: for each board
  //      that needs it, lane c holds column c; frontier expansion over the four
  //      link planes until no new cell is reached ----
  uint64_t todo = __ballot(pend != 0u);
  while (todo) {
    const int src = (int)__ffsll((unsigned long long)todo) - 1;
    todo &= todo - 1;
    const int spa = __builtin_amdgcn_readlane(pa, src);
    const uint32_t spend = (uint32_t)__builtin_amdgcn_readlane((int)pend, src);
    const int sx = spa / N, sy = spa - sx * N;
    const int c = lane;
    uint32_t lk[4];
#pragma unroll
    for (int e = 0; e < 4; e++) lk[e] = lds_column<N>(lds, src, PL + e * W, c);
#pragma unroll
    for (int f = 0; f < 2; f++) {
      if (!(spend & (1u << f))) continue;  // wave-uniform
      const int base = f ? PE : PS;
      const uint32_t flag0 = lds_column<N>(lds, src, base, c);
      uint32_t flag = flag0;
      uint32_t front = (c == sx) ? (1u << sy) : 0u;
      for (;;) {
        uint32_t nb = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int dx = twx::kDX[e], dy = twx::kDY[e];
          // forward: frontier cell with link e reaches (c+dx, y+dy)
          uint32_t s = __shfl(front & lk[e], c - dx);
          if (c - dx < 0) s = 0u;
          nb |= (dy > 0) ? (s << dy) : (s >> (-dy));
          // backward: cell (c,y) whose link e ends in the frontier
          uint32_t t = __shfl(front, c + dx);
          if (c + dx >= N) t = 0u;
          t = (dy > 0) ? (t >> dy) : (t << (-dy));
          nb |= t & lk[e];
        }
        const uint32_t fresh = (c < N) ? (nb & ~flag & ((1u << N) - 1u)) : 0u;
        flag |= fresh;
        front = fresh;
        if (__ballot(fresh != 0u) == 0ull) break;
      }
      const uint32_t add = flag & ~flag0;
      if (c < N && add) {
        const int o = c * N, w0 = base + (o >> 5), sh = o & 31;
        atomicOr(&lds[lds_idx(w0, src)], add << sh);
        if (sh + N > 32) atomicOr(&lds[lds_idx(w0 + 1, src)], add >> (32 - sh));
      }
      __syncthreads();
    }
  }
  __syncthreads();

  // ---- stream the boards back ----
#pragma unroll
  for (int c = 0; c < CH; c++) state[(long)c * bpad + board] = lds4[c * kWave + lane];
  if (inb) {
    if (status) {
      uint8_t st = (uint8_t)(result | (swapped << 5));
      if (illegal) st |= TWX_ST_ILLEGAL;
      if (was_reset) st |= TWX_ST_RESET;
      if (random && !act) st |= TWX_ST_NOOP;
      status[board] = st;
    }
    if (actions_out) actions_out[board] = a;
  }
  (void)SW;
}

// ---------------------------------------------------------------------------
// Layout-agnostic utility kernels (runtime N): reset / export / import / copy /
// queries.  One thread per (board, chunk) or per board.
// ---------------------------------------------------------------------------
struct Dims {
  int n, W, SW, CH;
  long batch, bpad;
};

__device__ __forceinline__ uint32_t comp(const uint4 &v, int j) {
  return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
}

__device__ __forceinline__ uint32_t state_word(const uint4 *state, const Dims &d, long b, int w) {
  return comp(state[(long)(w >> 2) * d.bpad + b], w & 3);
}

__global__ void reset_kernel(uint4 *state, Dims d, const uint32_t *init, const int32_t *ids, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t % (ids ? n : d.bpad);
  const int c = (int)(t / (ids ? n : d.bpad));
  if (c >= d.CH) return;
  long b = j;
  if (ids) {
    b = ids[j];
    if (b < 0 || b >= d.batch) return;
  }
  uint4 v;
  v.x = init[4 * c + 0]; v.y = init[4 * c + 1]; v.z = init[4 * c + 2]; v.w = init[4 * c + 3];
  state[(long)c * d.bpad + b] = v;
}

__global__ void export_kernel(const uint4 *state, Dims d, const int32_t *ids, long n, uint32_t *out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t / d.CH;
  const int c = (int)(t % d.CH);
  if (j >= n) return;
  const long b = ids ? ids[j] : j;
  if (b < 0 || b >= d.batch) return;
  const uint4 v = state[(long)c * d.bpad + b];
  uint32_t *o = out + j * d.SW + 4 * c;
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

__global__ void import_kernel(uint4 *state, Dims d, const int32_t *ids, long n, const uint32_t *in) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t / d.CH;
  const int c = (int)(t % d.CH);
  if (j >= n) return;
  const long b = ids ? ids[j] : j;
  if (b < 0 || b >= d.batch) return;
  const uint32_t *i = in + j * d.SW + 4 * c;
  uint4 v;
  v.x = i[0]; v.y = i[1]; v.z = i[2]; v.w = i[3];
  state[(long)c * d.bpad + b] = v;
}

__global__ void copy_kernel(uint4 *dst, Dims dd, const int32_t *dst_ids, const uint4 *src, Dims sd,
                            const int32_t *src_ids, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t % n;
  const int c = (int)(t / n);
  if (c >= dd.CH) return;
  const long db = dst_ids[j], sb = src_ids[j];
  if (db < 0 || db >= dd.batch || sb < 0 || sb >= sd.batch) return;
  dst[(long)c * dd.bpad + db] = src[(long)c * sd.bpad + sb];
}

// tw.h:86-90 as a mask: one thread per (board, mask word)
__global__ void legal_mask_kernel(const uint4 *state, Dims d, const uint32_t *region, uint32_t *mask,
                                  int32_t *count) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long b = t % d.batch;
  const int i = (int)(t / d.batch);
  if (i >= d.W) return;
  const int SC = twx::kNumPlanes * d.W;
  const uint32_t s0 = state_word(state, d, b, SC), s1 = state_word(state, d, b, SC + 1);
  const uint32_t counter = s0 & 0xffffu, result = s1 & 0xffu;
  const int player = (int)(counter & 1u);
  uint32_t mword = 0;
  if (result == 0u) {
    const uint32_t r = state_word(state, d, b, i), bl = state_word(state, d, b, d.W + i);
    const uint32_t occ = bl | (counter == 1u ? 0u : r);
    mword = region[player * d.W + i] & ~occ;
  }
  mask[b * d.W + i] = mword;
  if (count) {
    const int pc = __popc(mword);
    if (pc) atomicAdd(&count[b], pc);
  }
}

__global__ void status_kernel(const uint4 *state, Dims d, uint8_t *result, int32_t *move_counter,
                              int8_t *current_player) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.batch) return;
  const uint4 s = state[(long)(d.CH - 1) * d.bpad + b];
  const uint32_t counter = s.x & 0xffffu, res = s.y & 0xffu;
  if (result) result[b] = (uint8_t)res;
  if (move_counter) move_counter[b] = (int32_t)counter;
  if (current_player) current_player[b] = res ? (int8_t)-4 : (int8_t)(counter & 1u);  // tw.h:99-103
}

__device__ __forceinline__ int plane_bit(const uint4 *state, const Dims &d, long b, int plane, int cell) {
  return (state_word(state, d, b, plane * d.W + (cell >> 5)) >> (cell & 31)) & 1;
}

// reference-shaped Cell view (tc.h:98-108): one thread per (board, cell)
__global__ void cells_kernel(const uint4 *state, Dims d, const int32_t *ids, long n, uint8_t *out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = d.n * d.n;
  const long j = t / nn;
  const int cell = (int)(t % nn);
  if (j >= n) return;
  const long b = ids ? ids[j] : j;
  if (b < 0 || b >= d.batch) return;
  const int N = d.n, x = cell / N, y = cell % N;
  uint8_t *o = out + (j * nn + cell) * 4;
  if (!twx::on_board(N, x, y)) {
    o[0] = 3; o[1] = 0; o[2] = 0; o[3] = 0;
    return;
  }
  const int red = plane_bit(state, d, b, twx::kPlaneRed, cell);
  const int blue = plane_bit(state, d, b, twx::kPlaneBlue, cell);
  const int colour = red ? 0 : blue ? 1 : 2;
  int links = 0, blk = 0;
  if (colour != 2) {
    for (int dir = 0; dir < 8; dir++) {
      const int tx = x + twx::kDX[dir], ty = y + twx::kDY[dir];
      if (!twx::on_board(N, tx, ty)) continue;
      const int tc = tx * N + ty;
      const int same = plane_bit(state, d, b, colour, tc);
      const int lk = (dir < 4) ? plane_bit(state, d, b, twx::kPlaneLink0 + dir, cell)
                               : plane_bit(state, d, b, twx::kPlaneLink0 + dir - 4, tc);
      if (lk) links |= 1 << dir;
      // blocked_neighbors_ (tb.cc:550-551) is derived: two same-colour pegs a
      // knight's move apart are either linked or blocked
      if (same && !lk) blk |= 1 << dir;
    }
  }
  const int fs = plane_bit(state, d, b, twx::kPlaneStart, cell);
  const int fe = plane_bit(state, d, b, twx::kPlaneEnd, cell);
  int flags;
  if (colour == 0) flags = fs | fe << 1;
  else if (colour == 1) flags = fs << 2 | fe << 3;
  else flags = (x == 0 ? 4 : 0) | (x == N - 1 ? 8 : 0) | ((y == 0 && x > 0 && x < N - 1) ? 1 : 0) |
               ((y == N - 1 && x > 0 && x < N - 1) ? 2 : 0);
  o[0] = (uint8_t)colour; o[1] = (uint8_t)links; o[2] = (uint8_t)blk; o[3] = (uint8_t)flags;
}

// tw.cc:76-132 ObservationTensor: one thread per (board, output element)
__global__ void observation_kernel(const uint4 *state, Dims d, float *out) {
  const int N = d.n, per = 12 * N * (N - 2);
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long b = t / per;
  if (b >= d.batch) return;
  int r = (int)(t % per);
  const int plane = r / (N * (N - 2));
  r -= plane * N * (N - 2);
  const int tx = r / (N - 2), ty = r % (N - 2);
  const int colour = plane >= 6 ? 1 : 0, k = plane - 6 * colour;
  // invert tb.cc:590-597 GetTensorPosition
  int x, y;
  if (colour == 0) { y = N - 1 - tx; x = ty + 1; }
  else { x = N - 1 - tx; y = N - 2 - ty; }
  const int cell = x * N + y;
  float v = 0.0f;
  if (twx::on_board(N, x, y) && plane_bit(state, d, b, colour, cell)) {
    int links = 0, blk_east = 0;
    for (int dir = 0; dir < 8; dir++) {
      const int tx2 = x + twx::kDX[dir], ty2 = y + twx::kDY[dir];
      if (!twx::on_board(N, tx2, ty2)) continue;
      const int tc = tx2 * N + ty2;
      const int lk = (dir < 4) ? plane_bit(state, d, b, twx::kPlaneLink0 + dir, cell)
                               : plane_bit(state, d, b, twx::kPlaneLink0 + dir - 4, tc);
      if (lk) links |= 1 << dir;
      if (dir < 4 && !lk && plane_bit(state, d, b, colour, tc)) blk_east = 1;
    }
    if (k == 0) v = links ? 0.0f : 1.0f;                 // tw.cc:90-93
    else if (k <= 4) v = (links >> (k - 1)) & 1 ? 1.0f : 0.0f;  // tw.cc:83-89
    else v = blk_east ? 1.0f : 0.0f;                     // tw.cc:96-98
  }
  out[t] = v;
}

template <int N>
hipError_t launch_step(uint4 *state, long bpad, long batch, const int32_t *actions, int32_t *actions_out,
                       uint8_t *status, uint32_t seed, uint32_t first_id, int flags, hipStream_t s) {
  constexpr size_t lds_bytes = (size_t)twx::Geo<N>::CH * kWave * 16;  // 37 KiB at N=24: 4 waves per CU
  hipLaunchKernelGGL(step_kernel<N>, dim3((unsigned)(bpad / kWave)), dim3(kWave), lds_bytes, s, state, bpad,
                     batch, actions, actions_out, status, seed, first_id, flags);
  return hipGetLastError();
}

typedef hipError_t (*step_fn)(uint4 *, long, long, const int32_t *, int32_t *, uint8_t *, uint32_t, uint32_t,
                              int, hipStream_t);

template <int N>
struct StepTable {
  static void fill(step_fn *t) {
    t[N] = &launch_step<N>;
    StepTable<N - 1>::fill(t);
  }
};
template <>
struct StepTable<TWX_MIN_BOARD_SIZE - 1> {
  static void fill(step_fn *) {}
};

step_fn step_for(int n) {
  static step_fn table[TWX_MAX_BOARD_SIZE + 1];
  static bool init = false;
  if (!init) {
    StepTable<TWX_MAX_BOARD_SIZE>::fill(table);
    init = true;
  }
  return table[n];
}

inline unsigned blocks_for(long threads, int bs) { return (unsigned)((threads + bs - 1) / bs); }

}  // namespace

struct twx_env {
  Dims d;
  int device;
  uint64_t first_id;
  uint4 *state;
  uint32_t *tables;  // region[2][W] then init[SW]
};

extern "C" {

const char *twx_last_error(void) { return g_err.c_str(); }

int twx_state_words(int n) {
  if (n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE) return TWX_E_SIZE;
  return twx::state_words(n);
}

int twx_mask_words(int n) {
  if (n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE) return TWX_E_SIZE;
  return twx::plane_words(n);
}

int twx_create(int n, int64_t batch, int device, uint64_t first_board_id, twx_env **out) {
  if (!out) return fail(TWX_E_ARG, "twx_create: out is NULL");
  *out = nullptr;
  if (n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE)  // tw.cc:139-144
    return fail(TWX_E_SIZE, "board_size out of range [5..24]: " + std::to_string(n));
  if (batch <= 0) return fail(TWX_E_ARG, "twx_create: batch must be positive");
  HIP_TRY(hipSetDevice(device));
  twx_env *e = new twx_env();
  e->d.n = n;
  e->d.W = twx::plane_words(n);
  e->d.SW = twx::state_words(n);
  e->d.CH = twx::state_chunks(n);
  e->d.batch = batch;
  e->d.bpad = (batch + kWave - 1) / kWave * kWave;
  e->device = device;
  e->first_id = first_board_id;
  e->state = nullptr;
  e->tables = nullptr;
  const size_t bytes = (size_t)e->d.CH * (size_t)e->d.bpad * sizeof(uint4);
  if (hipMalloc(&e->state, bytes) != hipSuccess) {
    delete e;
    return fail(TWX_E_NOMEM, "twx_create: hipMalloc of " + std::to_string(bytes) + " bytes failed");
  }
  std::vector<uint32_t> tab(2 * e->d.W + e->d.SW);
  twx::fill_tables(n, tab.data(), tab.data() + 2 * e->d.W);
  if (hipMalloc(&e->tables, tab.size() * 4) != hipSuccess ||
      hipMemcpy(e->tables, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(e->state);
    if (e->tables) (void)hipFree(e->tables);
    delete e;
    return fail(TWX_E_NOMEM, "twx_create: table upload failed");
  }
  *out = e;
  int rc = twx_reset(e, nullptr);
  if (rc != TWX_OK) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  return TWX_OK;
}

int twx_destroy(twx_env *e) {
  if (!e) return TWX_OK;
  (void)hipSetDevice(e->device);
  (void)hipFree(e->state);
  (void)hipFree(e->tables);
  delete e;
  return TWX_OK;
}

int twx_board_size(const twx_env *e) { return e ? e->d.n : TWX_E_ARG; }
int64_t twx_batch(const twx_env *e) { return e ? e->d.batch : TWX_E_ARG; }

int twx_reset(twx_env *e, void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_reset: env is NULL");
  const long threads = (long)e->d.CH * e->d.bpad;
  hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, e->tables + 2 * e->d.W, (const int32_t *)nullptr, 0L);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_reset_boards(twx_env *e, const int32_t *ids, int64_t n, void *stream) {
  if (!e || !ids || n < 0) return fail(TWX_E_ARG, "twx_reset_boards: bad argument");
  if (n == 0) return TWX_OK;
  const long threads = (long)e->d.CH * n;
  hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, e->tables + 2 * e->d.W, ids, (long)n);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_step(twx_env *e, const int32_t *actions, uint8_t *status, void *stream) {
  if (!e || !actions) return fail(TWX_E_ARG, "twx_step: env or actions is NULL");
  HIP_TRY(step_for(e->d.n)(e->state, e->d.bpad, e->d.batch, actions, nullptr, status, 0u,
                           (uint32_t)e->first_id, 0, (hipStream_t)stream));
  return TWX_OK;
}

int twx_step_random(twx_env *e, uint32_t seed, int auto_reset, int32_t *actions_out, uint8_t *status,
                    void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_step_random: env is NULL");
  const int flags = kFlagRandom | (auto_reset ? kFlagAutoReset : 0);
  HIP_TRY(step_for(e->d.n)(e->state, e->d.bpad, e->d.batch, nullptr, actions_out, status, seed,
                           (uint32_t)e->first_id, flags, (hipStream_t)stream));
  return TWX_OK;
}

int twx_legal_mask(const twx_env *e, uint32_t *mask, int32_t *count, void *stream) {
  if (!e || !mask) return fail(TWX_E_ARG, "twx_legal_mask: env or mask is NULL");
  hipStream_t s = (hipStream_t)stream;
  if (count) HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t) * e->d.batch, s));
  const long threads = (long)e->d.W * e->d.batch;
  hipLaunchKernelGGL(legal_mask_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, s, e->state, e->d,
                     e->tables, mask, count);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_status(const twx_env *e, uint8_t *result, int32_t *move_counter, int8_t *current_player,
               void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_status: env is NULL");
  hipLaunchKernelGGL(status_kernel, dim3(blocks_for(e->d.batch, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, result, move_counter, current_player);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_observation(const twx_env *e, float *out, void *stream) {
  if (!e || !out) return fail(TWX_E_ARG, "twx_observation: env or out is NULL");
  const long threads = (long)12 * e->d.n * (e->d.n - 2) * e->d.batch;
  hipLaunchKernelGGL(observation_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, out);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_copy_boards(twx_env *dst, const int32_t *dst_ids, const twx_env *src, const int32_t *src_ids,
                    int64_t n, void *stream) {
  if (!dst || !src || !dst_ids || !src_ids || n < 0) return fail(TWX_E_ARG, "twx_copy_boards: bad argument");
  if (dst->d.n != src->d.n) return fail(TWX_E_ARG, "twx_copy_boards: board sizes differ");
  if (n == 0) return TWX_OK;
  const long threads = (long)dst->d.CH * n;
  hipLaunchKernelGGL(copy_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     dst->state, dst->d, dst_ids, src->state, src->d, src_ids, (long)n);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_export_state(const twx_env *e, const int32_t *ids, int64_t n, uint32_t *out, void *stream) {
  if (!e || !out || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_export_state: bad argument");
  if (n == 0) return TWX_OK;
  const long threads = (long)e->d.CH * n;
  hipLaunchKernelGGL(export_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, ids, (long)n, out);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_import_state(twx_env *e, const int32_t *ids, int64_t n, const uint32_t *in, void *stream) {
  if (!e || !in || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_import_state: bad argument");
  if (n == 0) return TWX_OK;
  const long threads = (long)e->d.CH * n;
  hipLaunchKernelGGL(import_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, ids, (long)n, in);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_export_cells(const twx_env *e, const int32_t *ids, int64_t n, uint8_t *out, void *stream) {
  if (!e || !out || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_export_cells: bad argument");
  if (n == 0) return TWX_OK;
  const long threads = (long)e->d.n * e->d.n * n;
  hipLaunchKernelGGL(cells_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, ids, (long)n, out);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

}  // extern "C"
