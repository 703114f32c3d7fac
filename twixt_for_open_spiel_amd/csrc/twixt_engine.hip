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
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/twixt_hip.h"
#include "twixt_geom.h"
#include "twixt_text.h"

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
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // native 16-byte vector: stays in VGPRs

// ---------------------------------------------------------------------------
// Internal layout (HBM and LDS alike).  A board is CHT = 2W+3 chunks of 16 B:
//   chunk 2i   = { red_i, blue_i, linkNNE_i, linkENE_i }      i = 0..W-1
//   chunk 2i+1 = { linkESE_i, linkSSE_i, start_i, end_i }
//   chunk 2W   = scalars { s0, s1, s2, s3 }
//   chunk 2W+1, 2W+2 = redo journal, 8 x 32-bit records
// i.e. the 8 planes are interleaved word by word: everything about the 32 cells
// 32i..32i+31 sits in two adjacent chunks.  One address computation per cell /
// column then reaches every plane through immediate offsets.
// HBM: state[chunk][board] (uint4); LDS image of a wave: [chunk][lane] (uint4),
// so chunk c of lane l starts at dword c*256 + l*4.
// ---------------------------------------------------------------------------

__device__ __forceinline__ void lds_or(uint32_t *p, uint32_t v) {
  __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // ds_or_b32
}
__device__ __forceinline__ void lds_and(uint32_t *p, uint32_t v) {
  __hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // ds_and_b32
}

// A workgroup is ONE wavefront and the LDS unit executes a wave's DS instructions in
// issue order, so cross-lane hand-offs through LDS need no s_barrier: only the compiler
// must not reorder the accesses.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// DPP wavefront shifts: lane i takes the value of lane i-1 / i+1, the end lane gets 0
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

__device__ __forceinline__ uint32_t funnel(uint32_t lo, uint32_t hi, int sh) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> sh);
}

// position of the k-th (0-based) set bit of a word that has more than k bits set
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

// In-kernel phase stamps (diagnostic build -DTWX_STAMPS only; never in the shipped
// library): cycles per phase summed over all waves, read with twx_debug_stamps().
#ifdef TWX_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime(); unsigned long long st_acc[16] = {0}
#define STAMP(i)                                                     \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               \
    const unsigned long long st_now = __builtin_amdgcn_s_memtime();  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               \
    st_acc[i] += st_now - st_prev;                                   \
    st_prev = st_now;                                                \
    __builtin_amdgcn_sched_barrier(0);                               \
  } while (0)
#define STAMP_FLUSH                                                                      \
  do {                                                                                   \
    if (threadIdx.x == 0)                                                                \
      for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_stamps[i_], st_acc[i_]);              \
  } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

constexpr int kFlagRandom = 1, kFlagAutoReset = 2, kFlagCompact = 4, kFlagFlush = 8, kFlagReverse = 16,
              kFlagRolling = 32;  // + bits 8..11: journal period, bits 12..27: batch stride of a wave

// One launch = one env step on every board (or, with kFlagFlush, just the
// materialisation of the redo journal).
//
// Log-structured state: between two compactions a launch does not
// rewrite the planes; it appends one 32-bit record of the move to the board's
// journal (slot `jpos`, the same for every board) and writes back only the
// scalars chunk and that journal chunk.  The next launches replay slots
// 0..jpos-1 on top of the planes they stream in.  Every `period`-th launch
// (kFlagCompact) writes everything and clears the journal.  Lanes that
// auto-reset write their whole board; lanes whose flood fill changed flag
// words write those chunks.
template <int N, bool SUB, int BPW>
__device__ __forceinline__ void step_body(uint4 *__restrict__ state, long bpad, long batch,
                                                     const int32_t *__restrict__ actions,
                                                     int32_t *__restrict__ actions_out,
                                                     uint8_t *__restrict__ status, uint32_t seed,
                                                     uint32_t first_id, int flags, int jpos_arg,
                                                     const int32_t *__restrict__ ids_arg, long nsub,
                                                     int rsteps) {
  using G = twx::Geo<N>;
  const int32_t *const ids = SUB ? ids_arg : nullptr;  // compile-time null in the whole-batch kernel
  constexpr int W = G::W, CS = 2 * W, CJ = 2 * W + 1, CHT = G::CHT;
  // BPW boards per wave (64: one per lane; 32: both 32-lane halves run the same per-board
  // logic on the same board -- every LDS update is an idempotent OR / AND / equal-value
  // store -- while fills, prefetches and stores are split by half: the LDS image is half
  // as big, so 8 waves fit a CU instead of 4).  RS = dwords per chunk row of the image.
  static_assert(BPW == 64 || BPW == 32, "boards per wave");
  constexpr int RS = BPW * 4, GS = 2 * RS;  // chunk-row / word-group stride in dwords
  constexpr int NJ = (BPW == 64) ? CHT : (CHT + 1) / 2;  // 16-byte registers a lane prefetches
#define plane_off(p) ((((p) >> 2) * RS) + ((p) & 3))
  constexpr int oR = plane_off(twx::kPlaneRed), oL0 = plane_off(twx::kPlaneLink0),
                oS = plane_off(twx::kPlaneStart), oE = plane_off(twx::kPlaneEnd);
  constexpr uint32_t kColMask = (1u << N) - 1u;
  static_assert(CHT == 2 * W + 3 && W <= 32, "layout");
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint4 *lds4 = reinterpret_cast<uint4 *>(lds);

  const int lane = threadIdx.x;
  const int half = (BPW == 64) ? 0 : (lane >> 5);  // which chunk parity this lane moves
  const int bl = lane & (BPW - 1);                 // board slot of this lane inside the batch
  auto chunk_of = [&](int j) -> int { return (BPW == 64) ? j : 2 * j + half; };
  const bool compact_all = (flags & kFlagCompact) != 0;
  const bool flush = (flags & kFlagFlush) != 0;
  const bool random = (flags & kFlagRandom) != 0;
  // Rows: the boards this launch works on.  Whole batch: row j is board j (the allocation
  // is padded to a multiple of 64, padded boards are real storage).  Subset (`ids`): row j
  // is board ids[j]; lanes past nsub load board 0 and store nothing.
  const long nbatch = ids ? (nsub + BPW - 1) / BPW : bpad / BPW;

  // Persistent wave: walks batches idx = blockIdx.x, +gridDim.x, ...  While batch k is
  // processed out of LDS, batch k+1 is already streaming into VGPRs (CHT x 16 B per
  // lane), so HBM latency hides behind the integer work even at one wave per SIMD.
  // kFlagReverse walks the batches backwards: the boards a launch touched last are the
  // ones the next launch touches first (Infinity Cache reuse).
  auto batch_row = [&](long idx) -> long {
    return ((flags & kFlagReverse) ? (nbatch - 1 - idx) : idx) * BPW + bl;
  };
  // (an id outside [0, batch) makes its row a padding lane: nothing is stored, status untouched)
  auto row_board = [&](long j) -> long { return ids ? (j < nsub ? (long)ids[j] : -1L) : j; };
  auto batch_board = [&](long idx) -> long { return row_board(batch_row(idx)); };
  STAMP_DECL;
  u32x4 nxt[NJ];
  const u32x4 *__restrict__ gstate = reinterpret_cast<const u32x4 *>(state);
  u32x4 *ldsv = reinterpret_cast<u32x4 *>(lds);
  long idx = blockIdx.x;  // the launcher guarantees gridDim.x <= nbatch
  {
    long b0 = batch_board(idx);
    if (ids && (unsigned long)b0 >= (unsigned long)batch) b0 = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) nxt[j] = gstate[(long)(chunk_of(j) < CHT ? chunk_of(j) : CHT - 1) * bpad + b0];
#pragma unroll
    for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(nxt[j]));  // first batch: nothing to overlap with
  }
  for (; idx < nbatch; idx += gridDim.x) {
  STAMP(0);  // loop overhead / previous stores issue
  // Rolling compaction (TWX_ROLLING=1, off by default): batch b is at journal position
  // (phase + b) mod period, so every launch compacts one batch in `period` instead of every
  // period-th launch compacting all of them.  Same traffic, spread evenly -- and 10 % slower
  // on MI355X than the burst (0.173 vs 0.158 ms per launch): kept as a tested option only.
  int jpos = jpos_arg;
  bool compact = compact_all;
  if (flags & kFlagRolling) {
    const int period = (flags >> 8) & 15;
    const long tb = (flags & kFlagReverse) ? (nbatch - 1 - idx) : idx;
    // (+ tb / stride: a wave's successive batches tb, tb+stride, ... get different
    // positions, so every wave compacts its share in every launch; `stride` is a per-handle
    // constant, bits 12..27 of flags, so a batch's position is the same in every launch)
    const long stride = (flags >> 12) & 0xffff;
    jpos = (int)((jpos_arg + tb + tb / stride) % period);
    compact = compact_all || (jpos == period - 1);
  }
  const long row = batch_row(idx);
  const long board_raw = row_board(row);
  const bool inb = (unsigned long)board_raw < (unsigned long)batch;
  const long board = (ids && !inb) ? 0L : board_raw;  // padding lanes of a subset load board 0
  const bool wr = !ids || inb;  // subset mode: padding lanes must not store
  const int a_ext = (!random && !flush && inb) ? actions[row] : -1;
#pragma unroll
  for (int j = 0; j < NJ; j++) ldsv[j * kWave + lane] = nxt[j];  // row chunk_of(j), slot bl
  wave_sync();
  // prefetch of the next batch, issued in three slices between the phases below so
  // that the wave never sits behind a 39-deep burst in its memory-issue queue (the
  // last iteration harmlessly re-reads its own batch)
  long b1 = batch_board((idx + gridDim.x < nbatch) ? idx + gridDim.x : idx);
  if (ids && (unsigned long)b1 >= (unsigned long)batch) b1 = 0;
  auto prefetch = [&](int c0, int c1) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = c0; j < c1; j++) nxt[j] = gstate[(long)(chunk_of(j) < CHT ? chunk_of(j) : CHT - 1) * bpad + b1];
    __builtin_amdgcn_sched_barrier(0);
  };
#ifndef TWX_PF_A
#define TWX_PF_A 4
#define TWX_PF_B 8
#endif
  constexpr int kPf1 = NJ * TWX_PF_A / 12, kPf2 = NJ * TWX_PF_B / 12;  // slice boundaries
  prefetch(0, kPf1);

  STAMP(1);  // LDS fill + prefetch issue
#ifdef TWX_SKELETON
  prefetch(kPf1, NJ);
  const bool was_reset = false; const uint32_t dirty = 0u; uint32_t result = 0, swapped = 0; bool illegal = false, act = true; int a = 0;
  { uint32_t *const sk = lds + (bl << 2) + 2 * G::W * RS; sk[0] += 1u; }
#else
  uint32_t *const me = lds + (bl << 2);  // this board's slot inside every chunk row
  // word group i (cells 32i..32i+31) of this board: me + i * GS [+ plane_off(p)]
  uint32_t *const sc = me + CS * RS;     // scalars
  uint32_t *const jn = me + CJ * RS;     // journal slot s at jn[(s >> 2) * RS + (s & 3)]

  const uint4 sv = *reinterpret_cast<const uint4 *>(sc);
  uint32_t counter = sv.x & 0xffffu, move_one = sv.x >> 16;
  uint32_t result = sv.y & 0xffu, swapped = (sv.y >> 8) & 1u;
  uint32_t episode = sv.z;
  bool was_reset = false;

  if (random && (flags & kFlagAutoReset) && result != 0u && inb) {
    // ---- auto-reset: Board::Board (tb.cc:168-174, 209-276) in place ----
#pragma unroll
    for (int i = 0; i < W; i++) {
      *reinterpret_cast<uint4 *>(me + i * GS) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4 *>(me + i * GS + RS) =
          make_uint4(0u, 0u, G::tab.init[twx::kPlaneStart * W + i], G::tab.init[twx::kPlaneEnd * W + i]);
    }
    *reinterpret_cast<uint4 *>(jn) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4 *>(jn + RS) = make_uint4(0u, 0u, 0u, 0u);
    counter = 0; move_one = 0; result = 0; swapped = 0; episode += 1;
    was_reset = true;
  } else {
    // ---- replay the redo journal: slots 0..jpos-1 in order, branch-free ORs ----
#pragma unroll
    for (int s = 0; s < twx::kJournalSlots - 1; s++) {
      if (s < jpos) {  // wave-uniform
        const uint32_t e = jn[(s >> 2) * RS + (s & 3)];
        const int cell = (int)(e & 1023u);
        const uint32_t bit = (e & twx::kJValid) ? (1u << (cell & 31)) : 0u;  // empty record: every OR is a no-op
        uint32_t *const g = me + (cell >> 5) * GS;
        if (e & (1u << 21)) lds_and(me + (move_one >> 5) * GS + oR, ~(1u << (move_one & 31)));  // swap: tb.cc:450-455
        lds_or(g + oR + ((e >> 10) & 1u), bit);  // peg: red / blue words are adjacent
#pragma unroll
        for (int d = 0; d < 4; d++)  // links whose west end is this peg
          lds_or(g + oL0 + plane_off(twx::kPlaneLink0 + d) - oL0, (e & (1u << (11 + d))) ? bit : 0u);
#pragma unroll
        for (int d = 4; d < 8; d++) {  // links whose west end is the other peg
          const bool on = (e & (1u << (11 + d))) != 0u;
          const int tc = on ? cell + twx::kDX[d] * N + twx::kDY[d] : cell;
          lds_or(me + (tc >> 5) * GS + plane_off(twx::kPlaneLink0 + d - 4), on ? (1u << (tc & 31)) : 0u);
        }
        lds_or(g + oS, (e & (1u << 19)) ? bit : 0u);
        lds_or(g + oE, (e & (1u << 20)) ? bit : 0u);
      }
    }
  }
  wave_sync();

  STAMP(2);  // reset / journal replay
  int a = -1;
  bool act = false, illegal = false;
  uint32_t dirty = 0u;
  // rsteps == 1: one env step.  rsteps > 1 (twx_random_rollout): up to rsteps moves per
  // board with the state staying in LDS, until every board of the batch is terminal.
  for (int rs = 0; rs < rsteps; rs++) {
  if (rs == 0) prefetch(kPf1, kPf2);
  const int player = (int)(counter & 1u);
  const bool open = (result == 0u);

  // ---- legal mask of the player to move (tw.h:86-90) and the
  //      opponent's free-cell count (for the draw test of tb.cc:202-206) ----
  uint32_t m[W];
  int cnt_r = 0, cnt_b = 0;
#pragma unroll
  for (int i = 0; i < W; i++) {
    const uint2 rb = *reinterpret_cast<const uint2 *>(me + i * GS);
    const uint32_t occ = rb.x | rb.y;
    uint32_t mr = G::tab.region[0][i] & ~occ;
    uint32_t mb = G::tab.region[1][i] & ~occ;
    cnt_r += __popc(mr);
    cnt_b += __popc(mb);
    if (counter == 1u) mb = G::tab.region[1][i] & ~rb.y;  // red's first peg stays legal for blue: the swap
    m[i] = player ? mb : mr;
  }
  const int cnt_opp = player ? cnt_r : cnt_b;
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < W; i++) cnt += __popc(m[i]);

  STAMP(3);  // legal mask + counts
  // ---- choose / validate the action ----
  a = -1;
  act = inb && open && !flush;
  illegal = false;
  if (random) {
    act = act && cnt > 0;
    const uint32_t h = twx::policy_hash(seed, first_id + (uint32_t)board, episode, counter);
    const int k = (int)__umulhi(h, (uint32_t)cnt);
    int cum = 0, wsel = 0, base = 0;
    uint32_t msel = m[0];
#pragma unroll
    for (int i = 0; i < W; i++) {  // last word whose prefix count is <= k
      const bool ge = k >= cum;
      wsel = ge ? i : wsel;
      msel = ge ? m[i] : msel;
      base = ge ? cum : base;
      cum += __popc(m[i]);
    }
    a = wsel * 32 + select_bit(msel, k - base);
    if (!act) a = -1;
  } else if (!flush) {
    a = a_ext;
    bool legal = false;
    if (a >= 0 && a < N * N) {
      const int ax = a / N, ay = a - ax * N;
      const bool inreg = player ? (ay >= 1 && ay <= N - 2) : (ax >= 1 && ax <= N - 2);
      const uint32_t bit = 1u << (a & 31);
      const uint2 rb = *reinterpret_cast<const uint2 *>(me + (a >> 5) * GS);
      const bool occ = ((rb.y & bit) != 0u) || (counter != 1u && (rb.x & bit) != 0u);
      legal = inreg && !occ;
    }
    illegal = inb && !(open && legal);  // tw.h:95-97 (terminal: LegalActions() is empty)
    act = act && legal;
  }

  STAMP(4);  // action select
  if (rs == 0) prefetch(kPf2, NJ);
  int pa = 0;          // cell the peg finally lands on
  uint32_t pend = 0;   // bit0: flood START from pa, bit1: flood END from pa
  uint32_t entry = 0;  // journal record of this move
  uint32_t fast_dirty = 0;  // flag chunk touched by the single-cell fill shortcut
  if (act) {
    // ---- Board::ApplyAction (tb.cc:457-499) ----
    pa = a;
    const bool swapnow = (counter == 1u) && ((uint32_t)a == move_one);
    if (swapnow) {  // tb.cc:462-473: undo red's first peg, rotate 90 degrees
      swapped = 1u;
      lds_and(me + (a >> 5) * GS + oR, ~(1u << (a & 31)));
      const int ax = a / N, ay = a - ax * N;
      pa = ay * N + (N - 1 - ax);
    }
    const int x = pa / N, y = pa - x * N;
    lds_or(me + (pa >> 5) * GS + oR + player, 1u << (pa & 31));  // tb.cc:507-508 set peg

    // ---- SetPegAndLinks (tb.cc:501-571), all 8 directions at once.  Six board
    //      columns x-3..x+2; per column one address, every plane by immediate
    //      offset.  wo/ws/we: rows y-2..y+2 of own pegs / START / END (columns
    //      x-2..x+2); win: rows y-3..y+3 of the 4 link planes (columns x-3..x+1) ----
    uint32_t wo[5], ws[5], we[5], win[twx::kWinCols];
#pragma unroll
    for (int ci = -3; ci <= 2; ci++) {
      const int c = x + ci;
      const bool ok = (unsigned)c < (unsigned)N;
      const int o = (ok ? c : 0) * N, sh = o & 31;
      const uint32_t *const g = me + (o >> 5) * GS;  // next word group: +GS dwords (in bounds: scalars/journal follow)
      auto col = [&](int off) -> uint32_t {
        const uint32_t v = funnel(g[off], g[off + GS], sh) & kColMask;
        return ok ? v : 0u;
      };
      if (ci >= -2) {
        wo[ci + 2] = ((col(oR + player) << 2) >> y) & 0x1fu;
        ws[ci + 2] = ((col(oS) << 2) >> y) & 0x1fu;
        we[ci + 2] = ((col(oE) << 2) >> y) & 0x1fu;
      }
      if (ci <= 1) {
        uint32_t v = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) v |= (((col(plane_off(twx::kPlaneLink0 + e)) << 3) >> y) & 0x7fu) << (8 * e);
        win[ci + 3] = v;
      }
    }
    uint32_t cand = 0, ts = 0, te = 0, blocked = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      const int ci = twx::kDX[d] + 2, rb = twx::kDY[d] + 2;
      cand |= ((wo[ci] >> rb) & 1u) << d;  // same-colour peg a knight's move away (tb.cc:517)
      ts |= ((ws[ci] >> rb) & 1u) << d;
      te |= ((we[ci] >> rb) & 1u) << d;
      uint32_t hit = 0;  // crossing test (tb.cc:519-527) as constant masks (twixt_geom.h)
#pragma unroll
      for (int i = 0; i < twx::kWinCols; i++) hit |= win[i] & twx::kCross.m[d][i];
      blocked |= (hit != 0u ? 1u : 0u) << d;
    }
    const uint32_t linkm = cand & ~blocked;
    // tb.cc:538-546: inherit START, else END, else remember a neutral neighbour
    const uint32_t gain_s = linkm & ts, gain_e = linkm & ~ts & te, neutral = linkm & ~ts & ~te;
    const bool p_s = ((ws[2] >> 2) & 1u) || gain_s;
    const bool p_e = ((we[2] >> 2) & 1u) || gain_e;
    uint32_t want = (linkm && neutral) ? ((p_s ? 1u : 0u) | (p_e ? 2u : 0u)) : 0u;  // tb.cc:558-570
    // Half of all fills flag exactly one cell: the new peg's only unflagged neighbour is a
    // peg without any other link.  Detect that here, on the pre-move planes, and set the
    // flag directly; only the other fills go through the wave-cooperative pass below.
    if (want && !(neutral & (neutral - 1u)) && (!(want & 1u) || (linkm & ~ts) == neutral) &&
        (!(want & 2u) || (linkm & ~te) == neutral)) {
      const int d = __ffs((int)neutral) - 1;
      constexpr uint32_t kDxPack = 0x10013443u, kDyPack = 0x43100134u;  // (offset + 2) per direction, 4 bits each
      const int tx = x + (int)((kDxPack >> (4 * d)) & 15u) - 2, ty = y + (int)((kDyPack >> (4 * d)) & 15u) - 2;
      const int t = tx * N + ty;
      uint32_t *const gt = me + (t >> 5) * GS;
      uint32_t other = (gt[plane_off(2)] | gt[plane_off(3)] | gt[plane_off(4)] | gt[plane_off(5)]) >> (t & 31);
#pragma unroll
      for (int e = 0; e < 4; e++) {  // links that END at t: stored at t - off[e]
        const int sx = tx - twx::kDX[e], sy = ty - twx::kDY[e];
        const bool valid = sx >= 0 && sy >= 0 && sy < N;
        const int sc2 = valid ? sx * N + sy : 0;
        const uint32_t w = me[(sc2 >> 5) * GS + plane_off(twx::kPlaneLink0 + e)];
        other |= valid ? (w >> (sc2 & 31)) : 0u;
      }
      if (!(other & 1u)) {
        const uint32_t tbit = 1u << (t & 31);
        lds_or(gt + oS, (want & 1u) ? tbit : 0u);
        lds_or(gt + oE, (want & 2u) ? tbit : 0u);
        fast_dirty = 1u << (t >> 5);
        want = 0u;
      }
    }
    pend = want;
    {  // tb.cc:532-533: one stored bit per physical link, at its west end
      uint32_t *const g = me + (pa >> 5) * GS;
      const uint32_t bit = 1u << (pa & 31);
#pragma unroll
      for (int d = 0; d < 4; d++) lds_or(g + plane_off(twx::kPlaneLink0 + d), (linkm & (1u << d)) ? bit : 0u);
#pragma unroll
      for (int d = 4; d < 8; d++) {
        const bool on = (linkm & (1u << d)) != 0u;
        const int tc = on ? pa + twx::kDX[d] * N + twx::kDY[d] : pa;
        lds_or(me + (tc >> 5) * GS + plane_off(twx::kPlaneLink0 + d - 4), on ? (1u << (tc & 31)) : 0u);
      }
      lds_or(g + oS, gain_s ? bit : 0u);
      lds_or(g + oE, gain_e ? bit : 0u);
      entry = twx::kJValid | (uint32_t)pa | ((uint32_t)player << 10) | (linkm << 11) |
              (gain_s ? 1u << 19 : 0u) | (gain_e ? 1u << 20 : 0u) | (swapnow ? 1u << 21 : 0u);

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
  }
  STAMP(5);  // apply: windows, crossing test, links, result
  // s3 is used below as the board's dirty mask (bit i: flag words of group i changed)
  *reinterpret_cast<uint4 *>(sc) = make_uint4(counter | (move_one << 16), result | (swapped << 8), episode, 0u);
  if (compact) {
    *reinterpret_cast<uint4 *>(jn) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4 *>(jn + RS) = make_uint4(0u, 0u, 0u, 0u);
  } else {
    jn[(jpos >> 2) * RS + (jpos & 3)] = entry;
  }
  wave_sync();

  // ---- ExploreLocalGraph (tb.cc:573-588), wave-cooperative, two boards per pass:
  //      lanes 0..31 work on one board that needs a fill, lanes 32..63 on another;
  //      lane c (within its half) holds column c of the four link planes and of the
  //      flag plane.  Frontier expansion: columns move by DPP wave shifts (1 VALU
  //      op each), rows by bit shifts; stop when no half found a new cell. ----
  uint64_t todo = __ballot(pend != 0u);
  if (BPW == 32) todo &= 0xffffffffull;  // the upper half mirrors the lower one
  while (todo) {
    const int s0 = (int)__ffsll((unsigned long long)todo) - 1;
    todo &= todo - 1;
    const bool two = todo != 0ull;
    const int s1 = two ? (int)__ffsll((unsigned long long)todo) - 1 : s0;
    if (two) todo &= todo - 1;
    const int fh = lane >> 5, c = lane & 31;  // flood half: which of the two boards this lane works on
    const int src = fh ? s1 : s0;
    const int spa = fh ? __builtin_amdgcn_readlane(pa, s1) : __builtin_amdgcn_readlane(pa, s0);
    const uint32_t spend = (fh && !two) ? 0u
                           : (uint32_t)(fh ? __builtin_amdgcn_readlane((int)pend, s1)
                                             : __builtin_amdgcn_readlane((int)pend, s0));
    const int sx = spa / N, sy = spa - sx * N;
    const bool ok = c < N;
    const int o = (ok ? c : 0) * N, sh = o & 31, grp = o >> 5;
    uint32_t *const g = lds + (src << 2) + grp * GS;  // board `src`, this lane's column
    auto col = [&](int off) -> uint32_t {
      const uint32_t v = funnel(g[off], g[off + GS], sh) & kColMask;
      return ok ? v : 0u;
    };
    uint32_t lk[4];
#pragma unroll
    for (int e = 0; e < 4; e++) lk[e] = col(plane_off(twx::kPlaneLink0 + e));
    STAMP(12);  // flood: pass setup
#pragma unroll
    for (int f = 0; f < 2; f++) {
      const bool en = (spend >> f) & 1u;  // uniform per half
      if (!__ballot(en)) continue;
      const int off = f ? oE : oS;
      const uint32_t flag0 = col(off);
      uint32_t flag = flag0;
      uint32_t front = (en && c == sx) ? (1u << sy) : 0u;
      for (;;) {
        // forward: a frontier cell with link e reaches (c+dx, y+dy).  dx = 1 for NNE/SSE,
        // 2 for ENE/ESE: merge the row-shifted pairs, then shift lanes right by dx
        const uint32_t a1 = ((front & lk[0]) << 2) | ((front & lk[3]) >> 2);
        const uint32_t a2 = ((front & lk[1]) << 1) | ((front & lk[2]) >> 1);
        uint32_t nb = wave_shr1(a1) | wave_shr1(wave_shr1(a2));
        // backward: cell (c,y) whose link e ends in the frontier at (c+dx, y+dy)
        const uint32_t f1 = wave_shl1(front), f2 = wave_shl1(f1);
        nb |= ((f1 >> 2) & lk[0]) | ((f2 >> 1) & lk[1]) | ((f2 << 1) & lk[2]) | ((f1 << 2) & lk[3]);
        const uint32_t fresh = ok ? (nb & ~flag & kColMask) : 0u;
        flag |= fresh;
        front = fresh;
        if (__ballot(fresh != 0u) == 0ull) break;
      }
      STAMP(13);  // flood: iterations
      const uint32_t add = flag & ~flag0;
      if (ok && add) {
        uint32_t dirty = 1u << grp;
        lds_or(g + off, add << sh);
        if (sh + N > 32) {
          lds_or(g + off + GS, add >> (32 - sh));
          dirty |= 2u << grp;
        }
        lds_or(lds + (src << 2) + CS * RS + 3, dirty);
      }
      wave_sync();
      STAMP(14);  // flood: write-back
    }
  }
  wave_sync();
  STAMP(6);  // flood fill
  dirty |= sc[3] | fast_dirty;
  sc[3] = 0u;
  wave_sync();
  if (rs + 1 < rsteps && __ballot(inb && result == 0u) == 0ull) break;  // everyone finished
  }  // rollout steps

#endif  // TWX_SKELETON
  // The prefetched batch has had the whole step to arrive: retire its loads HERE, before
  // this batch's stores are issued, so that the wait at the top of the next iteration
  // does not also wait for those stores (loads and stores share one in-order counter).
#pragma unroll
  for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(nxt[j]));

  STAMP(7);  // wait for the prefetched batch
  // ---- stream back ----
  // (lane -> LDS row j*64+lane holds chunk chunk_of(j) of board bl: the fill's mapping)
  if (compact) {
#pragma unroll
    for (int j = 0; j < NJ; j++)
      if (wr && chunk_of(j) < CHT) state[(long)chunk_of(j) * bpad + board] = lds4[j * kWave + lane];
  } else {
    const int jc = CJ + (jpos >> 2);  // the journal chunk of slot jpos
    if (BPW == 64) {
      if (wr) {
        state[(long)CS * bpad + board] = lds4[CS * kWave + lane];  // scalars
        state[(long)jc * bpad + board] = lds4[jc * kWave + lane];
      }
    } else {  // one store: lower half the scalars chunk, upper half the journal chunk
      const int c = half ? jc : CS;
      if (wr) state[(long)c * bpad + board] = lds4[c * BPW + bl];
    }
    STAMP(9);  // fixed stores (scalars + journal chunk)
    // flag chunks a flood fill changed: every dirty board walks its own mask, all dirty
    // boards in parallel (per-lane chunk index), so the loop runs max-popcount times
    uint32_t dd = (was_reset || !wr || half) ? 0u : dirty;
    while (__ballot(dd != 0u)) {
      if (dd) {
        const int c = 2 * (__ffs((int)dd) - 1) + 1;  // chunk 2i+1 holds the flag words of group i
        dd &= dd - 1;
        state[(long)c * bpad + board] = lds4[c * BPW + bl];
      }
    }
    STAMP(10);  // dirty flag chunks
    if (__ballot(was_reset)) {  // auto-reset boards are rewritten whole
#pragma unroll
      for (int j = 0; j < NJ; j++)
        if (was_reset && chunk_of(j) < CHT) state[(long)chunk_of(j) * bpad + board] = lds4[j * kWave + lane];
    }
  }
  STAMP(11);  // reset lanes / compaction
  if (inb && !flush && half == 0) {
    if (status) {
      uint8_t st = (uint8_t)(result | (swapped << 5));
      if (illegal) st |= TWX_ST_ILLEGAL;
      if (was_reset) st |= TWX_ST_RESET;
      if (random && !act) st |= TWX_ST_NOOP;
      status[row] = st;
    }
    if (actions_out) actions_out[row] = a;
  }
  wave_sync();
  STAMP(8);  // stream back + status
  }  // batch loop
  STAMP_FLUSH;
}
#undef plane_off

// 64 boards per wave: 39 KiB LDS image, 4 waves per CU, all 512 registers of the SIMD
template <int N, bool SUB>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 1))) void step_kernel64(
    uint4 *__restrict__ state, long bpad, long batch, const int32_t *__restrict__ actions,
    int32_t *__restrict__ actions_out, uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
    int jpos, const int32_t *__restrict__ ids, long nsub, int rsteps) {
  step_body<N, SUB, 64>(state, bpad, batch, actions, actions_out, status, seed, first_id, flags, jpos, ids, nsub,
                        rsteps);
}

#ifdef TWX_ENABLE_WINDOW
// ---------------------------------------------------------------------------
// Windowed step kernel: the non-compaction launches of the whole batch on large boards.
//
// A move only looks at (a) the peg words of the whole board -- legal mask -- and (b) the
// planes around the chosen cell: columns x-4..x+2, at most WG word groups.  So the wave
// streams only the EVEN chunks {red, blue, linkNNE, linkENE} + scalars + journal of its 64
// boards (coalesced, prefetched one batch ahead as in step_body), picks the move, and then
// gathers just the WG odd chunks {linkESE, linkSSE, start, end} of each board's window
// (per-lane chunk index).  HBM reads drop from 39 to 21 + WG(=7) chunks per board, and the
// LDS image from 39 to 28 rows, so 5 waves fit a CU instead of 4.  The journal replay is
// restricted to what is resident.  EXPERIMENT (-DTWX_ENABLE_WINDOW, TWX_WINDOW=1): bit-exact
// (the whole GPU parity suite passes with it), 40 % less HBM traffic per launch, but 5 %
// SLOWER on MI355X: the gather depends on the chosen move and the on-demand flood loads add
// a second dependent round trip, and five waves per CU do not hide them.  Not built by
// default.  The rare cooperative flood fill (tb.cc:573-588) can
// reach any part of a board: for those boards the wave loads all odd chunks of the board
// into a small scratch area on demand, replays the journal onto it and runs the same
// two-boards-per-pass fill as step_body.  Compaction, flush, subset and roll-out launches
// keep using step_body.
// ---------------------------------------------------------------------------
template <int N>
struct WinGeo {
  static constexpr int W = twx::Geo<N>::W;
  static constexpr int WG = ((7 * N - 2) >> 5) + 2;  // groups that 7 consecutive columns can span
  static constexpr bool kUsable = WG < W;
  static constexpr int ROW_O = W, ROW_SC = W + WG, ROW_J = ROW_SC + 1, ROWS = ROW_J + twx::kJournalChunks;
  static constexpr int NJ = W + 1 + twx::kJournalChunks;  // coalesced chunks per board
  static constexpr int SCR = ROWS * 256;                  // dword offset of the flood scratch
  static constexpr int SCR_ROWS = W + 1;                  // odd chunks of one board + pad
  static constexpr size_t lds_bytes = (size_t)(ROWS * 256 + 2 * SCR_ROWS * 4) * 4;
  static constexpr int hbm_chunk(int j) { return j < W ? 2 * j : 2 * W + (j - W); }
  static constexpr int lds_row(int j) { return j < W ? j : ROW_SC + (j - W); }
};

template <int N>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 2))) void step_window_kernel(
    uint4 *__restrict__ state, long bpad, long batch, const int32_t *__restrict__ actions,
    int32_t *__restrict__ actions_out, uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
    int jpos, const uint4 *__restrict__ init_chunks) {
  using G = twx::Geo<N>;
  using WGEO = WinGeo<N>;
  constexpr int W = G::W, CS = 2 * W, CJ = 2 * W + 1, CHT = G::CHT, RS = 256;
  constexpr int WG = WGEO::WG, ROW_O = WGEO::ROW_O, ROW_SC = WGEO::ROW_SC, ROW_J = WGEO::ROW_J, NJ = WGEO::NJ;
  constexpr int SCR = WGEO::SCR, SCR_ROWS = WGEO::SCR_ROWS;
  constexpr uint32_t kColMask = (1u << N) - 1u;
  // even words: plane 0..3 = red, blue, NNE, ENE at me + i*RS + p
  // odd words : q = 0..3 = ESE, SSE, start, end at ob + i*RS + q, valid for g0 <= i < g0 + WG
  constexpr int eR = 0, eL0 = 2, eL1 = 3, oL2 = 0, oL3 = 1, oS = 2, oE = 3;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint4 *lds4 = reinterpret_cast<uint4 *>(lds);
  u32x4 *ldsv = reinterpret_cast<u32x4 *>(lds);
  const u32x4 *__restrict__ gstate = reinterpret_cast<const u32x4 *>(state);

  const int lane = threadIdx.x;
  const bool random = (flags & kFlagRandom) != 0;
  const long nbatch = bpad / kWave;
  auto batch_board = [&](long idx) -> long {
    return ((flags & kFlagReverse) ? (nbatch - 1 - idx) : idx) * kWave + lane;
  };
  u32x4 nxt[NJ];
  long idx = blockIdx.x;
  {
    const long b0 = batch_board(idx);
#pragma unroll
    for (int j = 0; j < NJ; j++) nxt[j] = gstate[(long)WGEO::hbm_chunk(j) * bpad + b0];
#pragma unroll
    for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(nxt[j]));
  }
  uint32_t *const me = lds + (lane << 2);
  uint32_t *const sc = me + ROW_SC * RS;
  uint32_t *const jn = me + ROW_J * RS;  // slot s at jn[(s >> 2) * RS + (s & 3)]
  for (; idx < nbatch; idx += gridDim.x) {
    const long board = batch_board(idx);
    const bool inb = board < batch;
    const int a_ext = (!random && inb) ? actions[board] : -1;
#pragma unroll
    for (int j = 0; j < NJ; j++) ldsv[WGEO::lds_row(j) * kWave + lane] = nxt[j];
    wave_sync();
    const long b1 = batch_board((idx + gridDim.x < nbatch) ? idx + gridDim.x : idx);
    auto prefetch = [&](int c0, int c1) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = c0; j < c1; j++) nxt[j] = gstate[(long)WGEO::hbm_chunk(j) * bpad + b1];
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int kPf1 = NJ / 3, kPf2 = 2 * NJ / 3;
    prefetch(0, kPf1);

    const uint4 sv = *reinterpret_cast<const uint4 *>(sc);
    uint32_t counter = sv.x & 0xffffu, move_one = sv.x >> 16;
    uint32_t result = sv.y & 0xffu, swapped = (sv.y >> 8) & 1u;
    uint32_t episode = sv.z;
    bool was_reset = false;

    // one journal record, applied to the resident words.  EVEN: peg, swap, NNE/ENE links;
    // ODD (needs the window base `ob`): ESE/SSE links, flags -- only inside the window
    auto replay_even = [&](uint32_t e) {
      const int cell = (int)(e & 1023u);
      const uint32_t bit = (e & twx::kJValid) ? (1u << (cell & 31)) : 0u;
      uint32_t *const g = me + (cell >> 5) * RS;
      if (e & (1u << 21)) lds_and(me + (move_one >> 5) * RS + eR, ~(1u << (move_one & 31)));  // swap: tb.cc:450-455
      lds_or(g + eR + ((e >> 10) & 1u), bit);
      lds_or(g + eL0, (e & (1u << 11)) ? bit : 0u);
      lds_or(g + eL1, (e & (1u << 12)) ? bit : 0u);
#pragma unroll
      for (int d = 4; d < 6; d++) {  // SSW / WSW: an NNE / ENE link whose west end is the other peg
        const bool on = (e & (1u << (11 + d))) != 0u;
        const int tc = on ? cell + twx::kDX[d] * N + twx::kDY[d] : cell;
        lds_or(me + (tc >> 5) * RS + eL0 + (d - 4), on ? (1u << (tc & 31)) : 0u);
      }
    };

    if (random && (flags & kFlagAutoReset) && result != 0u && inb) {
      // ---- auto-reset: Board::Board (tb.cc:168-174, 209-276); odd window rows follow below ----
#pragma unroll
      for (int i = 0; i < W; i++) *reinterpret_cast<uint4 *>(me + i * RS) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int k = 0; k < twx::kJournalChunks; k++) *reinterpret_cast<uint4 *>(jn + k * RS) = make_uint4(0u, 0u, 0u, 0u);
      counter = 0; move_one = 0; result = 0; swapped = 0; episode += 1;
      was_reset = true;
    } else {
#pragma unroll
      for (int s = 0; s < twx::kJournalSlots - 1; s++)
        if (s < jpos) replay_even(jn[(s >> 2) * RS + (s & 3)]);
    }
    wave_sync();
    prefetch(kPf1, kPf2);

    const int player = (int)(counter & 1u);
    const bool open = (result == 0u);
    // ---- legal mask (tw.h:86-90) and the opponent's free-cell count (tb.cc:202-206) ----
    uint32_t m[W];
    int cnt_r = 0, cnt_b = 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
      const uint2 rb = *reinterpret_cast<const uint2 *>(me + i * RS);
      const uint32_t occ = rb.x | rb.y;
      uint32_t mr = G::tab.region[0][i] & ~occ;
      uint32_t mb = G::tab.region[1][i] & ~occ;
      cnt_r += __popc(mr);
      cnt_b += __popc(mb);
      if (counter == 1u) mb = G::tab.region[1][i] & ~rb.y;
      m[i] = player ? mb : mr;
    }
    const int cnt_opp = player ? cnt_r : cnt_b;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < W; i++) cnt += __popc(m[i]);

    int a = -1;
    bool act = inb && open, illegal = false;
    if (random) {
      act = act && cnt > 0;
      const uint32_t h = twx::policy_hash(seed, first_id + (uint32_t)board, episode, counter);
      const int k = (int)__umulhi(h, (uint32_t)cnt);
      int cum = 0, wsel = 0, base = 0;
      uint32_t msel = m[0];
#pragma unroll
      for (int i = 0; i < W; i++) {
        const bool ge = k >= cum;
        wsel = ge ? i : wsel;
        msel = ge ? m[i] : msel;
        base = ge ? cum : base;
        cum += __popc(m[i]);
      }
      a = wsel * 32 + select_bit(msel, k - base);
      if (!act) a = -1;
    } else {
      a = a_ext;
      bool legal = false;
      if (a >= 0 && a < N * N) {
        const int ax = a / N, ay = a - ax * N;
        const bool inreg = player ? (ay >= 1 && ay <= N - 2) : (ax >= 1 && ax <= N - 2);
        const uint32_t bit = 1u << (a & 31);
        const uint2 rb = *reinterpret_cast<const uint2 *>(me + (a >> 5) * RS);
        const bool occ = ((rb.y & bit) != 0u) || (counter != 1u && (rb.x & bit) != 0u);
        legal = inreg && !occ;
      }
      illegal = inb && !(open && legal);
      act = act && legal;
    }

    // ---- where the peg lands (swap: tb.cc:462-473), and the window around it ----
    const bool swapnow = act && (counter == 1u) && ((uint32_t)a == move_one);
    int pa = act ? a : 0;
    if (swapnow) {
      const int ax = a / N, ay = a - ax * N;
      pa = ay * N + (N - 1 - ax);
    }
    const int x = pa / N, y = pa - x * N;
    int g0 = ((x > 4 ? x - 4 : 0) * N) >> 5;
    if (g0 > W - WG) g0 = W - WG;
    uint32_t *const ob = me + (ROW_O - g0) * RS;  // odd words of group i: ob + i*RS + q
    auto inwin = [&](int i) -> bool { return (unsigned)(i - g0) < (unsigned)WG; };

    // gather the odd chunks of the window (a reset board: the initial position's)
    {
      u32x4 gw[WG];
      const u32x4 *__restrict__ initv = reinterpret_cast<const u32x4 *>(init_chunks);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < WG; k++)
        gw[k] = was_reset ? initv[2 * (g0 + k) + 1] : gstate[(long)(2 * (g0 + k) + 1) * bpad + board];
      __builtin_amdgcn_sched_barrier(0);
      prefetch(kPf2, NJ);
#pragma unroll
      for (int k = 0; k < WG; k++) ldsv[(ROW_O + k) * kWave + lane] = gw[k];
    }
    wave_sync();
    auto replay_odd = [&](uint32_t e, uint32_t *base, int stride, bool all) {
      // ESE / SSE links and flags of one record onto odd words at base + i*stride + q
      const int cell = (int)(e & 1023u);
      const bool v = (e & twx::kJValid) != 0u;
      const int gi = cell >> 5;
      const bool in0 = v && (all || inwin(gi));
      const uint32_t bit = in0 ? (1u << (cell & 31)) : 0u;
      uint32_t *const g = base + (in0 ? gi : g0) * stride;
      lds_or(g + oL2, (e & (1u << 13)) ? bit : 0u);
      lds_or(g + oL3, (e & (1u << 14)) ? bit : 0u);
      lds_or(g + oS, (e & (1u << 19)) ? bit : 0u);
      lds_or(g + oE, (e & (1u << 20)) ? bit : 0u);
#pragma unroll
      for (int d = 6; d < 8; d++) {  // WNW / NNW: an ESE / SSE link whose west end is the other peg
        const bool on = v && (e & (1u << (11 + d))) != 0u;
        const int tc = on ? cell + twx::kDX[d] * N + twx::kDY[d] : cell;
        const bool in1 = on && (all || inwin(tc >> 5));
        lds_or(base + (in1 ? (tc >> 5) : g0) * stride + oL2 + (d - 6), in1 ? (1u << (tc & 31)) : 0u);
      }
    };
    if (!was_reset) {
#pragma unroll
      for (int s = 0; s < twx::kJournalSlots - 1; s++)
        if (s < jpos) replay_odd(jn[(s >> 2) * RS + (s & 3)], ob, RS, false);
    }
    wave_sync();

    uint32_t pend = 0, entry = 0, fast_dirty = 0;
    if (act) {
      // ---- Board::ApplyAction / SetPegAndLinks (tb.cc:457-571); everything read or written
      //      lies in columns x-4..x+2, i.e. inside the window ----
      if (swapnow) {
        swapped = 1u;
        lds_and(me + (a >> 5) * RS + eR, ~(1u << (a & 31)));
      }
      lds_or(me + (pa >> 5) * RS + eR + player, 1u << (pa & 31));  // tb.cc:507-508 set peg
      uint32_t wo[5], ws[5], we[5], win[twx::kWinCols];
#pragma unroll
      for (int ci = -3; ci <= 2; ci++) {
        const int c = x + ci;
        const bool ok = (unsigned)c < (unsigned)N;
        const int o = (ok ? c : x) * N, sh = o & 31, gi = o >> 5;
        const uint32_t *const ge = me + gi * RS, *const go = ob + gi * RS;
        auto cole = [&](int p) -> uint32_t { const uint32_t v = funnel(ge[p], ge[p + RS], sh) & kColMask; return ok ? v : 0u; };
        auto colo = [&](int q) -> uint32_t { const uint32_t v = funnel(go[q], go[q + RS], sh) & kColMask; return ok ? v : 0u; };
        if (ci >= -2) {
          wo[ci + 2] = ((cole(eR + player) << 2) >> y) & 0x1fu;
          ws[ci + 2] = ((colo(oS) << 2) >> y) & 0x1fu;
          we[ci + 2] = ((colo(oE) << 2) >> y) & 0x1fu;
        }
        if (ci <= 1) {
          win[ci + 3] = (((cole(eL0) << 3) >> y) & 0x7fu) | ((((cole(eL1) << 3) >> y) & 0x7fu) << 8) |
                        ((((colo(oL2) << 3) >> y) & 0x7fu) << 16) | ((((colo(oL3) << 3) >> y) & 0x7fu) << 24);
        }
      }
      uint32_t cand = 0, ts = 0, te = 0, blocked = 0;
#pragma unroll
      for (int d = 0; d < 8; d++) {
        const int ci = twx::kDX[d] + 2, rb = twx::kDY[d] + 2;
        cand |= ((wo[ci] >> rb) & 1u) << d;
        ts |= ((ws[ci] >> rb) & 1u) << d;
        te |= ((we[ci] >> rb) & 1u) << d;
        uint32_t hit = 0;
#pragma unroll
        for (int i = 0; i < twx::kWinCols; i++) hit |= win[i] & twx::kCross.m[d][i];
        blocked |= (hit != 0u ? 1u : 0u) << d;
      }
      const uint32_t linkm = cand & ~blocked;
      const uint32_t gain_s = linkm & ts, gain_e = linkm & ~ts & te, neutral = linkm & ~ts & ~te;
      const bool p_s = ((ws[2] >> 2) & 1u) || gain_s;
      const bool p_e = ((we[2] >> 2) & 1u) || gain_e;
      uint32_t want = (linkm && neutral) ? ((p_s ? 1u : 0u) | (p_e ? 2u : 0u)) : 0u;
      if (want && !(neutral & (neutral - 1u)) && (!(want & 1u) || (linkm & ~ts) == neutral) &&
          (!(want & 2u) || (linkm & ~te) == neutral)) {  // single-cell fill shortcut (see step_body)
        const int d = __ffs((int)neutral) - 1;
        constexpr uint32_t kDxPack = 0x10013443u, kDyPack = 0x43100134u;
        const int tx = x + (int)((kDxPack >> (4 * d)) & 15u) - 2, ty = y + (int)((kDyPack >> (4 * d)) & 15u) - 2;
        const int t = tx * N + ty;
        const int gt = t >> 5;
        uint32_t other = (me[gt * RS + eL0] | me[gt * RS + eL1] | ob[gt * RS + oL2] | ob[gt * RS + oL3]) >> (t & 31);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int sx = tx - twx::kDX[e], sy = ty - twx::kDY[e];
          const bool valid = sx >= 0 && sy >= 0 && sy < N;
          const int sc2 = valid ? sx * N + sy : t;
          const uint32_t w = (e < 2) ? me[(sc2 >> 5) * RS + eL0 + e] : ob[(sc2 >> 5) * RS + oL2 + (e - 2)];
          other |= valid ? (w >> (sc2 & 31)) : 0u;
        }
        if (!(other & 1u)) {
          const uint32_t tbit = 1u << (t & 31);
          lds_or(ob + gt * RS + oS, (want & 1u) ? tbit : 0u);
          lds_or(ob + gt * RS + oE, (want & 2u) ? tbit : 0u);
          fast_dirty = 1u << gt;
          want = 0u;
        }
      }
      pend = want;
      entry = twx::kJValid | (uint32_t)pa | ((uint32_t)player << 10) | (linkm << 11) |
              (gain_s ? 1u << 19 : 0u) | (gain_e ? 1u << 20 : 0u) | (swapnow ? 1u << 21 : 0u);
      replay_even(entry & ~(1u << 21));          // the move's own even-word updates (peg already set: idempotent)
      replay_odd(entry, ob, RS, false);          // ... and odd-word updates, all inside the window
      if (counter == 0u) move_one = (uint32_t)pa;
      counter += 1u;
      if (p_s && p_e) {
        result = player ? 2u : 1u;
      } else if (counter > 2u) {
        const bool in_opp = player ? (x >= 1 && x <= N - 2) : (y >= 1 && y <= N - 2);
        if (cnt_opp - (in_opp ? 1 : 0) == 0) result = 3u;
      }
    }
    *reinterpret_cast<uint4 *>(sc) = make_uint4(counter | (move_one << 16), result | (swapped << 8), episode, 0u);
    jn[(jpos >> 2) * RS + (jpos & 3)] = entry;
    wave_sync();

    // ---- flood fills that the shortcut could not settle: on-demand full boards ----
    uint64_t todo = __ballot(pend != 0u);
    while (todo) {
      const int s0 = (int)__ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const bool two = todo != 0ull;
      const int s1 = two ? (int)__ffsll((unsigned long long)todo) - 1 : s0;
      if (two) todo &= todo - 1;
      const int fh = lane >> 5, c = lane & 31;
      const int src = fh ? s1 : s0;
      const bool live = !fh || two;
      const long sboard = fh ? (((long)__builtin_amdgcn_readlane((int)(board >> 32), s1) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)board, s1))
                             : (((long)__builtin_amdgcn_readlane((int)(board >> 32), s0) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)board, s0));
      const int spa = fh ? __builtin_amdgcn_readlane(pa, s1) : __builtin_amdgcn_readlane(pa, s0);
      const uint32_t spend = !live ? 0u
                             : (uint32_t)(fh ? __builtin_amdgcn_readlane((int)pend, s1)
                                             : __builtin_amdgcn_readlane((int)pend, s0));
      uint32_t *const scr = lds + SCR + fh * SCR_ROWS * 4;  // odd chunk of group i at scr + 4*i
      // all odd chunks of the board (a flood never follows a reset: move 0 makes no link)
      if (c < W) *reinterpret_cast<u32x4 *>(scr + 4 * c) = gstate[(long)(2 * c + 1) * bpad + sboard];
      if (c == W) *reinterpret_cast<uint4 *>(scr + 4 * c) = make_uint4(0u, 0u, 0u, 0u);  // pad row: dirty mask in .x
      wave_sync();
      // journal slots 0..jpos (the new record included) onto the scratch: lane c takes slot c
      if (c <= jpos && c < twx::kJournalSlots) {
        const uint32_t e = lds[(ROW_J + (c >> 2)) * RS + (src << 2) + (c & 3)];
        // (same as replay_odd with all = true, on the scratch layout: stride 4)
        const int cell = (int)(e & 1023u);
        const bool v = (e & twx::kJValid) != 0u;
        const uint32_t bit = v ? (1u << (cell & 31)) : 0u;
        uint32_t *const g = scr + 4 * (cell >> 5);
        lds_or(g + oL2, (e & (1u << 13)) ? bit : 0u);
        lds_or(g + oL3, (e & (1u << 14)) ? bit : 0u);
        lds_or(g + oS, (e & (1u << 19)) ? bit : 0u);
        lds_or(g + oE, (e & (1u << 20)) ? bit : 0u);
#pragma unroll
        for (int d = 6; d < 8; d++) {
          const bool on = v && (e & (1u << (11 + d))) != 0u;
          const int tc = on ? cell + twx::kDX[d] * N + twx::kDY[d] : cell;
          lds_or(scr + 4 * (tc >> 5) + oL2 + (d - 6), on ? (1u << (tc & 31)) : 0u);
        }
      }
      wave_sync();
      const int sx = spa / N, sy = spa - sx * N;
      const bool ok = c < N;
      const int o = (ok ? c : 0) * N, sh = o & 31, grp = o >> 5;
      const uint32_t *const ge = lds + (src << 2) + grp * RS;
      uint32_t *const go = scr + 4 * grp;
      auto cole = [&](int p) -> uint32_t { const uint32_t v = funnel(ge[p], ge[p + RS], sh) & kColMask; return ok ? v : 0u; };
      auto colo = [&](int q) -> uint32_t { const uint32_t v = funnel(go[q], go[q + 4], sh) & kColMask; return ok ? v : 0u; };
      uint32_t lk[4];
      lk[0] = cole(eL0); lk[1] = cole(eL1); lk[2] = colo(oL2); lk[3] = colo(oL3);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const bool en = (spend >> f) & 1u;
        if (!__ballot(en)) continue;
        const int q = f ? oE : oS;
        const uint32_t flag0 = colo(q);
        uint32_t flag = flag0;
        uint32_t front = (en && c == sx) ? (1u << sy) : 0u;
        for (;;) {
          const uint32_t a1 = ((front & lk[0]) << 2) | ((front & lk[3]) >> 2);
          const uint32_t a2 = ((front & lk[1]) << 1) | ((front & lk[2]) >> 1);
          uint32_t nb = wave_shr1(a1) | wave_shr1(wave_shr1(a2));
          const uint32_t f1 = wave_shl1(front), f2 = wave_shl1(f1);
          nb |= ((f1 >> 2) & lk[0]) | ((f2 >> 1) & lk[1]) | ((f2 << 1) & lk[2]) | ((f1 << 2) & lk[3]);
          const uint32_t fresh = ok ? (nb & ~flag & kColMask) : 0u;
          flag |= fresh;
          front = fresh;
          if (__ballot(fresh != 0u) == 0ull) break;
        }
        const uint32_t add = flag & ~flag0;
        if (ok && add) {
          uint32_t dm = 1u << grp;
          lds_or(go + q, add << sh);
          if (sh + N > 32) {
            lds_or(go + q + 4, add >> (32 - sh));
            dm |= 2u << grp;
          }
          lds_or(scr + 4 * W, dm);
        }
        wave_sync();
      }
      // the odd chunks the fill changed go straight back to HBM
      if (live && c < W && ((scr[4 * W] >> c) & 1u))
        state[(long)(2 * c + 1) * bpad + sboard] = *reinterpret_cast<const uint4 *>(scr + 4 * c);
      wave_sync();
    }

    // retire the prefetched batch before this batch's stores (one in-order counter)
#pragma unroll
    for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(nxt[j]));

    // ---- stream back: scalars + this launch's journal chunk; the shortcut's flag chunk;
    //      whole boards for auto-reset lanes ----
    state[(long)CS * bpad + board] = lds4[ROW_SC * kWave + lane];
    state[(long)(CJ + (jpos >> 2)) * bpad + board] = lds4[(ROW_J + (jpos >> 2)) * kWave + lane];
    if (__ballot(fast_dirty != 0u && !was_reset)) {
      if (fast_dirty && !was_reset) {
        const int gi = __ffs((int)fast_dirty) - 1;
        state[(long)(2 * gi + 1) * bpad + board] = lds4[(ROW_O + gi - g0) * kWave + lane];
      }
    }
    if (__ballot(was_reset)) {
      if (was_reset) {
        const uint4 *__restrict__ initc = init_chunks;
#pragma unroll
        for (int i = 0; i < W; i++) {
          state[(long)(2 * i) * bpad + board] = lds4[i * kWave + lane];
          state[(long)(2 * i + 1) * bpad + board] = inwin(i) ? lds4[(ROW_O + i - g0) * kWave + lane] : initc[2 * i + 1];
        }
#pragma unroll
        for (int k = 0; k < twx::kJournalChunks; k++)
          state[(long)(CJ + k) * bpad + board] = lds4[(ROW_J + k) * kWave + lane];
      }
    }
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
    wave_sync();
    (void)CHT;
  }
}

#endif  // TWX_ENABLE_WINDOW

// small boards (LDS image <= 20 KiB, N <= 16): several waves per SIMD fit, so no
// one-wave-per-SIMD register budget; the compiler keeps it within 256 registers
template <int N, bool SUB>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 8))) void step_kernel64_small(
    uint4 *__restrict__ state, long bpad, long batch, const int32_t *__restrict__ actions,
    int32_t *__restrict__ actions_out, uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
    int jpos, const int32_t *__restrict__ ids, long nsub, int rsteps) {
  step_body<N, SUB, 64>(state, bpad, batch, actions, actions_out, status, seed, first_id, flags, jpos, ids, nsub,
                        rsteps);
}

// 32 boards per wave: 20 KiB LDS image, 8 waves per CU (two per SIMD, 256 registers each).
// Experiment (-DTWX_ENABLE_BPW32, TWX_BOARDS_PER_WAVE=32): bit-exact, but 15 % slower than
// the 64-board kernel on MI355X (the duplicated per-board work doubles the issue demand
// and the 256-register budget spills), so it is not built by default.
#ifdef TWX_ENABLE_BPW32
template <int N, bool SUB>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void step_kernel32(
    uint4 *__restrict__ state, long bpad, long batch, const int32_t *__restrict__ actions,
    int32_t *__restrict__ actions_out, uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
    int jpos, const int32_t *__restrict__ ids, long nsub, int rsteps) {
  step_body<N, SUB, 32>(state, bpad, batch, actions, actions_out, status, seed, first_id, flags, jpos, ids, nsub,
                        rsteps);
}
#endif

// ---------------------------------------------------------------------------
// Layout-agnostic utility kernels (runtime N): reset / export / import / copy /
// queries.  One thread per (board, chunk) or per board.
// ---------------------------------------------------------------------------
struct Dims {
  int n, W, SW, CH, CHT;  // CH: state chunks (planes + scalars); CHT: + journal chunks
  long batch, bpad;
};

__device__ __forceinline__ uint32_t comp(const uint4 &v, int j) {
  return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
}

// word i of plane p / scalar j of board b in the interleaved internal layout
__device__ __forceinline__ uint32_t plane_word(const uint4 *state, const Dims &d, long b, int p, int i) {
  return comp(state[(long)(2 * i + (p >> 2)) * d.bpad + b], p & 3);
}
__device__ __forceinline__ uint32_t scalar_word(const uint4 *state, const Dims &d, long b, int j) {
  return comp(state[(long)(2 * d.W) * d.bpad + b], j);
}

__global__ void reset_kernel(uint4 *state, Dims d, const uint32_t *init, const int32_t *ids, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t % (ids ? n : d.bpad);
  const int c = (int)(t / (ids ? n : d.bpad));
  if (c >= d.CHT) return;
  long b = j;
  if (ids) {
    b = ids[j];
    if (b < 0 || b >= d.batch) return;
  }
  uint4 v;  // init is in internal chunk order; scalars and journal chunks are zero
  v.x = init[4 * c + 0]; v.y = init[4 * c + 1]; v.z = init[4 * c + 2]; v.w = init[4 * c + 3];
  state[(long)c * d.bpad + b] = v;
}

// canonical (include/twixt_hip.h) order out: one thread per (board, word)
__global__ void export_kernel(const uint4 *state, Dims d, const int32_t *ids, long n, uint32_t *out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t / d.SW;
  const int w = (int)(t % d.SW);
  if (j >= n) return;
  const long b = ids ? ids[j] : j;
  if (b < 0 || b >= d.batch) return;
  const int np = twx::kNumPlanes * d.W;
  out[j * d.SW + w] = (w < np) ? plane_word(state, d, b, w / d.W, w % d.W) : scalar_word(state, d, b, w - np);
}

// canonical order in: one thread per (board, chunk 0..2W)
__global__ void import_kernel(uint4 *state, Dims d, const int32_t *ids, long n, const uint32_t *in) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t / d.CH;
  const int c = (int)(t % d.CH);
  if (j >= n) return;
  const long b = ids ? ids[j] : j;
  if (b < 0 || b >= d.batch) return;
  const uint32_t *src = in + j * d.SW;
  uint4 v;
  if (c < 2 * d.W) {
    const int i = c >> 1, p0 = (c & 1) * 4;
    v.x = src[(p0 + 0) * d.W + i]; v.y = src[(p0 + 1) * d.W + i];
    v.z = src[(p0 + 2) * d.W + i]; v.w = src[(p0 + 3) * d.W + i];
  } else {
    const uint32_t *sc = src + twx::kNumPlanes * d.W;
    v.x = sc[0]; v.y = sc[1]; v.z = sc[2]; v.w = sc[3];
  }
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
__global__ void legal_mask_kernel(const uint4 *state, Dims d, const uint32_t *region, const int32_t *ids,
                                  long n, uint32_t *mask, int32_t *count) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t % n;  // output row
  const int i = (int)(t / n);
  if (i >= d.W) return;
  const long b = ids ? ids[j] : j;
  if (b < 0 || b >= d.batch) return;
  const uint32_t s0 = scalar_word(state, d, b, 0), s1 = scalar_word(state, d, b, 1);
  const uint32_t counter = s0 & 0xffffu, result = s1 & 0xffu;
  const int player = (int)(counter & 1u);
  uint32_t mword = 0;
  if (result == 0u) {
    const uint32_t r = plane_word(state, d, b, twx::kPlaneRed, i), bl = plane_word(state, d, b, twx::kPlaneBlue, i);
    const uint32_t occ = bl | (counter == 1u ? 0u : r);
    mword = region[player * d.W + i] & ~occ;
  }
  mask[j * d.W + i] = mword;
  if (count) {
    const int pc = __popc(mword);
    if (pc) atomicAdd(&count[j], pc);
  }
}

__global__ void status_kernel(const uint4 *state, Dims d, const int32_t *ids, long n, uint8_t *result,
                              int32_t *move_counter, int8_t *current_player) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long b = ids ? ids[j] : j;
  if (b < 0 || b >= d.batch) return;
  const uint4 s = state[(long)(d.CH - 1) * d.bpad + b];
  const uint32_t counter = s.x & 0xffffu, res = s.y & 0xffu;
  if (result) result[j] = (uint8_t)res;
  if (move_counter) move_counter[j] = (int32_t)counter;
  if (current_player) current_player[j] = res ? (int8_t)-4 : (int8_t)(counter & 1u);  // tw.h:99-103
}

__device__ __forceinline__ int plane_bit(const uint4 *state, const Dims &d, long b, int plane, int cell) {
  return (plane_word(state, d, b, plane, cell >> 5) >> (cell & 31)) & 1;
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

// tw.cc:76-132 ObservationTensor x batch.  Write-bound: 12*N*(N-2) floats per board
// (25,344 B at N=24) against 592 B of state.  One wavefront per board:
//   1. the board's 2W+1 state chunks -> LDS (board-contiguous);
//   2. every cell with a peg -> one 6-bit feature {no-link peg, NNE, ENE, ESE, SSE link,
//      blocked east} (tw.cc:83-98) stored at its tensor position (tb.cc:590-597: red
//      (N-1-y, x-1), blue (N-1-x, N-2-y)) in a red and a blue feature plane in LDS;
//   3. the wave streams the 12 planes out as coalesced 16-byte stores.
constexpr int kObsMaxPos = TWX_MAX_BOARD_SIZE * (TWX_MAX_BOARD_SIZE - 2);
__global__ __launch_bounds__(kWave) void observation_kernel(const uint4 *__restrict__ state, Dims d,
                                                            const int32_t *__restrict__ ids,
                                                            float *__restrict__ out) {
  __shared__ uint4 sst[2 * 18 + 1];
  __shared__ uint8_t feat[2][kObsMaxPos];
  const int lane = threadIdx.x, N = d.n, W = d.W, npos = N * (N - 2);
  const long j = blockIdx.x;  // output row
  long b = ids ? ids[j] : j;
  if (b < 0 || b >= d.batch) b = 0;  // host validated; keep the block's barriers uniform
  for (int c = lane; c < 2 * W + 1; c += kWave) sst[c] = state[(long)c * d.bpad + b];
  for (int i = lane; i < npos; i += kWave) { feat[0][i] = 0; feat[1][i] = 0; }
  __syncthreads();
  const uint32_t *w = reinterpret_cast<const uint32_t *>(sst);
  auto bit = [&](int p, int cell) -> uint32_t {  // plane p, cell: word 8*(cell>>5)+p of the interleaved layout
    return (w[8 * (cell >> 5) + p] >> (cell & 31)) & 1u;
  };
  for (int cell = lane; cell < N * N; cell += kWave) {
    const int x = cell / N, y = cell - x * N;
    const uint32_t red = bit(twx::kPlaneRed, cell), blue = bit(twx::kPlaneBlue, cell);
    if (!(red | blue)) continue;
    const int colour = blue ? 1 : 0;
    uint32_t west = 0, east = 0, blk = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t lk = bit(twx::kPlaneLink0 + e, cell);  // link leaving this peg eastwards
      west |= lk << e;
      const int tx = x + twx::kDX[e], ty = y + twx::kDY[e];
      if (twx::on_board(N, tx, ty) && !lk && bit(colour, tx * N + ty)) blk = 1u;  // tb.cc:550-551, derived
      const int sx = x - twx::kDX[e], sy = y - twx::kDY[e];   // link arriving from the west
      if (twx::on_board(N, sx, sy)) east |= bit(twx::kPlaneLink0 + e, sx * N + sy);
    }
    const uint32_t f = ((west | east) ? (west << 1) : 1u) | (blk << 5);  // tw.cc:83-98
    const int pos = colour ? (N - 1 - x) * (N - 2) + (N - 2 - y) : (N - 1 - y) * (N - 2) + (x - 1);
    feat[colour][pos] = (uint8_t)f;
  }
  __syncthreads();
  float4 *o = reinterpret_cast<float4 *>(out + j * (long)(12 * npos));
  const uint32_t magic = (1u << 24) / (uint32_t)npos + 1u;  // e / npos == (e * magic) >> 24 for e < 12*npos <= 6336
  for (int j = lane; j < 3 * npos; j += kWave) {  // 12*npos floats = 3*npos float4
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int e = 4 * j + q;
      const int k = (int)(((uint32_t)e * magic) >> 24), pos = e - k * npos;
      const int colour = k >= 6 ? 1 : 0;
      v[q] = ((feat[colour][pos] >> (k - 6 * colour)) & 1u) ? 1.0f : 0.0f;
    }
    o[j] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int N>
hipError_t launch_step(uint4 *state, long bpad, long batch, const int32_t *actions, int32_t *actions_out,
                       uint8_t *status, uint32_t seed, uint32_t first_id, int flags, int jpos, int waves,
                       const int32_t *ids, long nsub, int rsteps, int bpw, hipStream_t s) {
  const long nbatch = ids ? (nsub + bpw - 1) / bpw : bpad / bpw;
  const long grid = nbatch < (long)waves ? nbatch : (long)waves;
  const size_t rows = (bpw == 64) ? twx::Geo<N>::CHT : 2 * ((twx::Geo<N>::CHT + 1) / 2);
  const size_t lds_bytes = rows * bpw * 16;  // 39 KiB (64 boards) or 20 KiB (32 boards) at N=24
#define TWX_LAUNCH(K)                                                                                      \
  hipLaunchKernelGGL((K), dim3((unsigned)grid), dim3(kWave), lds_bytes, s, state, bpad, batch, actions, \
                     actions_out, status, seed, first_id, flags, jpos, ids, nsub, rsteps)
  if (bpw == 64) {
    if constexpr (twx::Geo<N>::CHT <= 20) {
      if (ids) TWX_LAUNCH((step_kernel64_small<N, true>)); else TWX_LAUNCH((step_kernel64_small<N, false>));
    } else {
      if (ids) TWX_LAUNCH((step_kernel64<N, true>)); else TWX_LAUNCH((step_kernel64<N, false>));
    }
  } else {
#ifdef TWX_ENABLE_BPW32
    if (ids) TWX_LAUNCH((step_kernel32<N, true>)); else TWX_LAUNCH((step_kernel32<N, false>));
#else
    return hipErrorInvalidConfiguration;
#endif
  }
#undef TWX_LAUNCH
  return hipGetLastError();
}

// resident waves per CU of the whole-batch kernel for this N (LDS image + registers)
template <int N>
int step_occupancy(int bpw) {
  const size_t rows = (bpw == 64) ? twx::Geo<N>::CHT : 2 * ((twx::Geo<N>::CHT + 1) / 2);
  const size_t lds_bytes = rows * bpw * 16;
  int blocks = 0;
  hipError_t e;
  if (bpw == 64) {
    if constexpr (twx::Geo<N>::CHT <= 20)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, step_kernel64_small<N, false>, kWave, lds_bytes);
    else
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, step_kernel64<N, false>, kWave, lds_bytes);
  } else {
#ifdef TWX_ENABLE_BPW32
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, step_kernel32<N, false>, kWave, lds_bytes);
#else
    e = hipErrorInvalidConfiguration;
#endif
  }
  return (e == hipSuccess && blocks > 0) ? blocks : 0;
}

// windowed kernel: launch + occupancy (0 waves = this board size has no windowed variant)
template <int N>
hipError_t launch_window(uint4 *state, long bpad, long batch, const int32_t *actions, int32_t *actions_out,
                         uint8_t *status, uint32_t seed, uint32_t first_id, int flags, int jpos, int waves,
                         const uint4 *init_chunks, hipStream_t s) {
#ifdef TWX_ENABLE_WINDOW
  if constexpr (WinGeo<N>::kUsable && twx::Geo<N>::CHT > 20) {
    const long nbatch = bpad / kWave;
    const long grid = nbatch < (long)waves ? nbatch : (long)waves;
    hipLaunchKernelGGL((step_window_kernel<N>), dim3((unsigned)grid), dim3(kWave), WinGeo<N>::lds_bytes, s, state,
                       bpad, batch, actions, actions_out, status, seed, first_id, flags, jpos, init_chunks);
    return hipGetLastError();
  }
#endif
  return hipErrorInvalidConfiguration;
}
template <int N>
int window_occupancy() {
#ifdef TWX_ENABLE_WINDOW
  if constexpr (WinGeo<N>::kUsable && twx::Geo<N>::CHT > 20) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, step_window_kernel<N>, kWave, WinGeo<N>::lds_bytes) !=
        hipSuccess)
      return 0;
    return blocks;
  }
#endif
  return 0;
}
typedef hipError_t (*win_fn)(uint4 *, long, long, const int32_t *, int32_t *, uint8_t *, uint32_t, uint32_t, int, int,
                             int, const uint4 *, hipStream_t);
typedef int (*wocc_fn)();

typedef int (*occ_fn)(int);
typedef hipError_t (*step_fn)(uint4 *, long, long, const int32_t *, int32_t *, uint8_t *, uint32_t, uint32_t,
                              int, int, int, const int32_t *, long, int, int, hipStream_t);

template <int N>
struct StepTable {
  static void fill(step_fn *t, occ_fn *o, win_fn *w, wocc_fn *wo) {
    t[N] = &launch_step<N>;
    o[N] = &step_occupancy<N>;
    w[N] = &launch_window<N>;
    wo[N] = &window_occupancy<N>;
    StepTable<N - 1>::fill(t, o, w, wo);
  }
};
template <>
struct StepTable<TWX_MIN_BOARD_SIZE - 1> {
  static void fill(step_fn *, occ_fn *, win_fn *, wocc_fn *) {}
};

struct StepTables {
  step_fn step[TWX_MAX_BOARD_SIZE + 1];
  occ_fn occ[TWX_MAX_BOARD_SIZE + 1];
  win_fn win[TWX_MAX_BOARD_SIZE + 1];
  wocc_fn wocc[TWX_MAX_BOARD_SIZE + 1];
  StepTables() { StepTable<TWX_MAX_BOARD_SIZE>::fill(step, occ, win, wocc); }
};
const StepTables &step_tables() {
  static const StepTables t;  // thread-safe static init
  return t;
}
step_fn step_for(int n) { return step_tables().step[n]; }

inline unsigned blocks_for(long threads, int bs) { return (unsigned)((threads + bs - 1) / bs); }

}  // namespace

struct twx_env {
  Dims d;
  int device;
  uint64_t first_id;
  uint4 *state;
  uint32_t *tables;  // region[2][W] then the initial board in internal chunk order
  // log-structured state: launches since the last compaction (= journal slots in
  // use, identical for every board), compaction period (1..kJournalSlots), and
  // the launch counter that alternates the block order (Infinity Cache reuse)
  int jpos, period;   // journal position (rolling: phase) of the next whole-batch launch
  bool rolling;       // rolling compaction (TWX_ROLLING)
  bool journal_live;  // some journal may hold records
  bool pingpong;
  uint64_t launches;
  int waves;  // persistent grid: resident waves = CUs x waves the LDS image admits per CU
  int bpw;    // boards per wave: 64 or 32 (TWX_BOARDS_PER_WAVE)
  int init_off;    // word offset of the initial-position chunks inside `tables`
  bool window;     // use step_window_kernel for the non-compaction whole-batch launches
  int waves_win;   // its persistent grid
};

namespace {

int env_int(const char *name, int dflt, int lo, int hi) {
  const char *v = getenv(name);
  if (!v || !*v) return dflt;
  const int x = atoi(v);
  return x < lo ? lo : x > hi ? hi : x;
}

// a handle's work always goes to the handle's device, whatever the caller's current one is
#define TWX_USE(e) HIP_TRY(hipSetDevice((e)->device))

// one env step (or a flush) on every board, or on the `nsub` boards listed in `ids`;
// maintains the journal position
int run_step(twx_env *e, const int32_t *actions, int32_t *actions_out, uint8_t *status, uint32_t seed,
             int flags, const int32_t *ids, long nsub, hipStream_t s, int rsteps = 1) {
  TWX_USE(e);
  const bool flush = (flags & kFlagFlush) != 0;
  const bool whole_step = !ids && rsteps == 1 && !flush;
  if (!whole_step && !flush && e->journal_live) {
    // journals belong to the whole-batch step launches: subset and roll-out launches work
    // on materialised planes and rewrite their boards directly
    if (int rc = run_step(e, nullptr, nullptr, nullptr, 0u, kFlagFlush, nullptr, 0, s)) return rc;
  }
  int jarg = e->jpos;
  if (flush) {
    flags |= kFlagCompact;  // replay what is there, write everything, clear the journals
  } else if (!whole_step || e->period == 1) {
    flags |= kFlagCompact;  // no journaling: planes rewritten in place
    jarg = 0;
  } else if (!e->rolling && e->jpos == e->period - 1) {
    flags |= kFlagCompact;
  }
  if ((whole_step || flush) && e->rolling && e->period > 1)
    flags |= kFlagRolling | (e->period << 8) | ((e->waves > 0xffff ? 0xffff : e->waves) << 12);
  if (e->pingpong && !ids && (e->launches & 1)) flags |= kFlagReverse;
  if (e->window && whole_step && !(flags & (kFlagCompact | kFlagRolling))) {
    HIP_TRY(step_tables().win[e->d.n](e->state, e->d.bpad, e->d.batch, actions, actions_out, status, seed,
                                      (uint32_t)e->first_id, flags, jarg, e->waves_win,
                                      reinterpret_cast<const uint4 *>(e->tables + e->init_off), s));
  } else {
    HIP_TRY(step_for(e->d.n)(e->state, e->d.bpad, e->d.batch, actions, actions_out, status, seed,
                             (uint32_t)e->first_id, flags, jarg, e->waves, ids, nsub, rsteps, e->bpw, s));
  }
  if (flush) {
    e->journal_live = false;
    if (!e->rolling) e->jpos = 0;  // (rolling: the phase just keeps running over empty journals)
  } else if (whole_step && e->period > 1) {
    if (e->rolling) {
      e->jpos = (e->jpos + 1) % e->period;
      e->journal_live = true;
    } else {
      e->jpos = (flags & kFlagCompact) ? 0 : e->jpos + 1;
      e->journal_live = e->jpos != 0;
    }
  }
  e->launches++;
  return TWX_OK;
}

// every entry point other than the step functions sees materialised planes
int materialise(const twx_env *ce, hipStream_t s) {
  twx_env *e = const_cast<twx_env *>(ce);
  if (!e->journal_live) return TWX_OK;
  return run_step(e, nullptr, nullptr, nullptr, 0u, kFlagFlush, nullptr, 0, s);
}

}  // namespace

#ifdef TWX_STAMPS
extern "C" int twx_debug_stamps(unsigned long long *out, int reset) {
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16));
  if (reset) {
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z));
  }
  return TWX_OK;
}
#endif

extern "C" {

const char *twx_last_error(void) { return g_err.c_str(); }

// Host-only text helpers (no device work): Board::ToString / ActionToString
int twx_cells_to_string(const uint8_t *cells_host, int n, int ansi, int swapped, int result, char *buf,
                        int cap) {
  if (!cells_host || n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE || (cap > 0 && !buf))
    return fail(TWX_E_ARG, "twx_cells_to_string: bad argument");
  const std::string s = twx::text::BoardToString(cells_host, n, ansi != 0, swapped != 0, result);
  if (cap > 0) {
    const size_t k = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
    memcpy(buf, s.data(), k);
    buf[k] = 0;
  }
  return (int)s.size();
}

int twx_action_to_string(int n, int player, int64_t action, char *buf, int cap) {
  if (!buf || cap < 8 || n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE || action < 0 || action >= n * n)
    return fail(TWX_E_ARG, "twx_action_to_string: bad argument");
  const std::string s = twx::text::ActionToString(n, player, (long)action);
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

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
  int cus = 0;
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  twx_env *e = new twx_env();
  e->d.n = n;
  e->d.W = twx::plane_words(n);
  e->d.SW = twx::state_words(n);
  e->d.CH = twx::state_chunks(n);
  e->d.CHT = twx::total_chunks(n);
  e->jpos = 0;
  e->period = env_int("TWX_JOURNAL_PERIOD", twx::kJournalSlots, 1, twx::kJournalSlots);
  e->pingpong = env_int("TWX_PINGPONG", 1, 0, 1) != 0;
  e->rolling = env_int("TWX_ROLLING", 0, 0, 1) != 0;  // measured 10 % slower than compacting all at once
  e->journal_live = false;
  e->launches = 0;
  {
#ifdef TWX_ENABLE_BPW32
    e->bpw = env_int("TWX_BOARDS_PER_WAVE", 64, 32, 64) >= 48 ? 64 : 32;
#else
    e->bpw = 64;
#endif
    // as many waves as are resident at once (LDS image and registers; 4 per CU at N=24,
    // 8 at N=12, 12 at N=8): each walks its share of the batches
    int per_cu = step_tables().occ[n](e->bpw);
    if (per_cu < 1) per_cu = 4;
    if (per_cu > 16) per_cu = 16;
    e->waves = (cus > 0 ? cus : 256) * env_int("TWX_WAVES_PER_CU", per_cu, 1, 32);
    const int wocc = step_tables().wocc[n]();
    e->window = wocc > 0 && e->bpw == 64 && e->period > 1 && !e->rolling && env_int("TWX_WINDOW", 0, 0, 1) != 0;
    e->waves_win = (cus > 0 ? cus : 256) * env_int("TWX_WINDOW_WAVES_PER_CU", wocc > 0 ? wocc : 1, 1, 32);
  }
  e->d.batch = batch;
  e->d.bpad = (batch + kWave - 1) / kWave * kWave;
  e->device = device;
  e->first_id = first_board_id;
  e->state = nullptr;
  e->tables = nullptr;
  const size_t bytes = (size_t)e->d.CHT * (size_t)e->d.bpad * sizeof(uint4);
  if (hipMalloc(&e->state, bytes) != hipSuccess) {
    delete e;
    return fail(TWX_E_NOMEM, "twx_create: hipMalloc of " + std::to_string(bytes) + " bytes failed");
  }
  // tables: region[2][W] (padded to a multiple of 4 words), then the initial position in
  // internal chunk order (CHT uint4)
  e->init_off = (2 * e->d.W + 3) & ~3;
  std::vector<uint32_t> canon(2 * e->d.W + e->d.SW), tab(e->init_off + 4 * e->d.CHT, 0u);
  twx::fill_tables(n, canon.data(), canon.data() + 2 * e->d.W);
  for (int i = 0; i < 2 * e->d.W; i++) tab[i] = canon[i];
  for (int p = 0; p < twx::kNumPlanes; p++)
    for (int i = 0; i < e->d.W; i++)
      tab[e->init_off + 4 * (2 * i + (p >> 2)) + (p & 3)] = canon[2 * e->d.W + p * e->d.W + i];
  if (hipMalloc(&e->tables, tab.size() * 4) != hipSuccess ||
      hipMemcpy(e->tables, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(e->state);
    if (e->tables) (void)hipFree(e->tables);
    delete e;
    return fail(TWX_E_NOMEM, "twx_create: table upload failed");
  }
  if (twx_reset(e, nullptr) != TWX_OK || hipStreamSynchronize(nullptr) != hipSuccess) {
    const std::string why = g_err;
    twx_destroy(e);
    return fail(TWX_E_DEVICE, "twx_create: initial reset failed: " + why);
  }
  *out = e;
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
  TWX_USE(e);
  e->jpos = 0;  // every journal is cleared below
  e->journal_live = false;
  const long threads = (long)e->d.CHT * e->d.bpad;
  hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, e->tables + e->init_off, (const int32_t *)nullptr, 0L);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_reset_boards(twx_env *e, const int32_t *ids, int64_t n, void *stream) {
  if (!e || !ids || n < 0) return fail(TWX_E_ARG, "twx_reset_boards: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  const long threads = (long)e->d.CHT * n;
  hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, e->tables + e->init_off, ids, (long)n);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_step(twx_env *e, const int32_t *actions, uint8_t *status, void *stream) {
  if (!e || !actions) return fail(TWX_E_ARG, "twx_step: env or actions is NULL");
  return run_step(e, actions, nullptr, status, 0u, 0, nullptr, 0, (hipStream_t)stream);
}

int twx_step_boards(twx_env *e, const int32_t *ids, int64_t n, const int32_t *actions, uint8_t *status,
                    void *stream) {
  if (!e || !ids || !actions || n < 0) return fail(TWX_E_ARG, "twx_step_boards: bad argument");
  if (n == 0) return TWX_OK;
  return run_step(e, actions, nullptr, status, 0u, 0, ids, (long)n, (hipStream_t)stream);
}

int twx_step_random_boards(twx_env *e, const int32_t *ids, int64_t n, uint32_t seed, int auto_reset,
                           int32_t *actions_out, uint8_t *status, void *stream) {
  if (!e || !ids || n < 0) return fail(TWX_E_ARG, "twx_step_random_boards: bad argument");
  if (n == 0) return TWX_OK;
  const int flags = kFlagRandom | (auto_reset ? kFlagAutoReset : 0);
  return run_step(e, nullptr, actions_out, status, seed, flags, ids, (long)n, (hipStream_t)stream);
}

int twx_random_rollout(twx_env *e, const int32_t *ids, int64_t n, uint32_t seed, int max_steps, uint8_t *status,
                       void *stream) {
  if (!e || n < 0 || max_steps < 1 || (!ids && n != 0 && n != e->d.batch))
    return fail(TWX_E_ARG, "twx_random_rollout: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (int rc = materialise(e, s)) return rc;  // the rollout works on materialised planes
  if (ids && n == 0) return TWX_OK;
  // one launch: every board plays on, out of LDS, until it is terminal or max_steps moves
  // were made; written back as a compaction (journal stays empty)
  return run_step(e, nullptr, nullptr, status, seed, kFlagRandom | kFlagCompact, ids, ids ? (long)n : 0, s, max_steps);
}

int twx_step_random(twx_env *e, uint32_t seed, int auto_reset, int32_t *actions_out, uint8_t *status,
                    void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_step_random: env is NULL");
  const int flags = kFlagRandom | (auto_reset ? kFlagAutoReset : 0);
  return run_step(e, nullptr, actions_out, status, seed, flags, nullptr, 0, (hipStream_t)stream);
}

int twx_legal_mask_boards(const twx_env *e, const int32_t *ids, int64_t n, uint32_t *mask, int32_t *count,
                          void *stream) {
  if (!e || !mask || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_legal_mask: bad argument");
  if (n == 0) return TWX_OK;
  hipStream_t s = (hipStream_t)stream;
  TWX_USE(e);
  if (int rc = materialise(e, s)) return rc;
  if (count) HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t) * n, s));
  const long threads = (long)e->d.W * n;
  hipLaunchKernelGGL(legal_mask_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, s, e->state, e->d,
                     e->tables, ids, (long)n, mask, count);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_legal_mask(const twx_env *e, uint32_t *mask, int32_t *count, void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_legal_mask: env is NULL");
  return twx_legal_mask_boards(e, nullptr, e->d.batch, mask, count, stream);
}

int twx_status_boards(const twx_env *e, const int32_t *ids, int64_t n, uint8_t *result, int32_t *move_counter,
                      int8_t *current_player, void *stream) {
  if (!e || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_status: bad argument");
  if (n == 0) return TWX_OK;
  // (scalars are written by every launch: no materialisation needed)
  TWX_USE(e);
  hipLaunchKernelGGL(status_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, e->state, e->d,
                     ids, (long)n, result, move_counter, current_player);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_status(const twx_env *e, uint8_t *result, int32_t *move_counter, int8_t *current_player,
               void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_status: env is NULL");
  return twx_status_boards(e, nullptr, e->d.batch, result, move_counter, current_player, stream);
}

int twx_observation_boards(const twx_env *e, const int32_t *ids, int64_t n, float *out, void *stream) {
  if (!e || !out || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_observation: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  hipLaunchKernelGGL(observation_kernel, dim3((unsigned)n), dim3(kWave), 0, (hipStream_t)stream, e->state, e->d,
                     ids, out);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_observation(const twx_env *e, float *out, void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_observation: env is NULL");
  return twx_observation_boards(e, nullptr, e->d.batch, out, stream);
}

int twx_copy_boards(twx_env *dst, const int32_t *dst_ids, const twx_env *src, const int32_t *src_ids,
                    int64_t n, void *stream) {
  if (!dst || !src || !dst_ids || !src_ids || n < 0) return fail(TWX_E_ARG, "twx_copy_boards: bad argument");
  if (dst->d.n != src->d.n) return fail(TWX_E_ARG, "twx_copy_boards: board sizes differ");
  if (n == 0) return TWX_OK;
  if (dst->device != src->device) return fail(TWX_E_ARG, "twx_copy_boards: handles live on different devices");
  TWX_USE(dst);
  if (int rc = materialise(dst, (hipStream_t)stream)) return rc;
  if (int rc = materialise(src, (hipStream_t)stream)) return rc;
  const long threads = (long)dst->d.CH * n;
  hipLaunchKernelGGL(copy_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     dst->state, dst->d, dst_ids, src->state, src->d, src_ids, (long)n);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_export_state(const twx_env *e, const int32_t *ids, int64_t n, uint32_t *out, void *stream) {
  if (!e || !out || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_export_state: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  const long threads = (long)e->d.SW * n;
  hipLaunchKernelGGL(export_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, ids, (long)n, out);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_import_state(twx_env *e, const int32_t *ids, int64_t n, const uint32_t *in, void *stream) {
  if (!e || !in || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_import_state: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  const long threads = (long)e->d.CH * n;
  hipLaunchKernelGGL(import_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, ids, (long)n, in);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_export_cells(const twx_env *e, const int32_t *ids, int64_t n, uint8_t *out, void *stream) {
  if (!e || !out || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_export_cells: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  const long threads = (long)e->d.n * e->d.n * n;
  hipLaunchKernelGGL(cells_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, ids, (long)n, out);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

}  // extern "C"
