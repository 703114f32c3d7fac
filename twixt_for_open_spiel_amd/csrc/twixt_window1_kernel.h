// twixt_window1_kernel.h -- step_window1_kernel<N>: the whole-batch step of the big boards (N >= 17) on the SPLIT
// layout (journal launches, compactions and flushes), ONE LANE PER BOARD, 64 boards per wave, software-pipelined.
// Part of libtwixt_hip.so; included by twixt_engine.hip only (device code, anonymous namespace).
#ifndef TWIXT_WINDOW1_KERNEL_H_
#define TWIXT_WINDOW1_KERNEL_H_

// Why a third mapping.  With the windowed fetch (twixt_window_kernel.h) a journal launch moves a third fewer
// 128-byte lines, and the two-lanes-per-board kernel stopped being bound by the CU's memory pipe: its waves
// spend their time ISSUING -- 1,360 vector instructions per 32 boards, the SIMDs' vector issue 58-64 % busy
// with two waves each, every phase of the step about equally guilty (profiles/r03_*: SQ counters, phase
// stamps).  Two lanes per board bought occupancy when the whole board had to sit in LDS (20 KiB per 32
// boards); it costs instructions, because everything that is not split is computed twice.  A windowed step
// keeps only the pegs (144 B) and the window (160 B) of a board in LDS: 64 boards fit in the same 20 KiB, so
// the occupancy stays (eight waves per CU, two per SIMD) and every wave instruction serves twice the boards.
//
// A lane owns board l of the batch (tile l >> 5 of the batch's two tiles, slot l & 31):
//   * its 16 bytes of every peg row and of the scalars row are ITS board's, and its journal records are
//     single dwords of the [slot][board] rows: everything is loaded straight into the lane that needs it
//     (scalars and records never pass through LDS; the pegs do, for the data-dependent accesses);
//   * its window is up to 10 x 16 B fetched by the lane itself out of its board's links + flags region -- the
//     pieces of the board columns its move's candidate links can touch, none at all for a move without a
//     neighbour to link to (the memory side counts 128-byte lines, not requests: tools/window_probe3.hip);
//   * LDS per wave: 64 windows x 176 B (the eleventh piece of a window is padding: the column extraction
//     reads one word past the last column) + 9 peg rows x 1 KiB = 20,480 B at N=24.
// The pipeline is step_window_kernel's: per iteration a wave moves the pegs of batch k+1 and the windows of
// batch k into LDS, requests the rows of batch k+2, chooses the moves of batch k+1 and requests their
// windows, then applies the moves of batch k.  Multi-cell flood fills (tb.cc:573-588) wait, two boards to a
// pass (one per half-wave, lane = board column), in two pass slots whose regions are requested an iteration
// ahead; only a fifth and sixth pending fill of one iteration is flooded on the spot.
// A board's links + flags region is kept CURRENT in HBM: a move's changed region words are stored by the launch
// that makes it (through L2, where the window's lines still are), so no launch replays records on a region and a
// compaction (every eighth launch) or flush is this same kernel storing the peg rows and scalars as well.
// (Measured without any journal -- peg chunk and scalars stored every step: 0.1327 against 0.1236 ms per step;
// two scattered 16-byte stores per board cost more than one record in a full line and a compaction in eight.)
// Batches too small to fill the chip rewrite whole tiles instead: step_window_full_kernel.
//
// Reference lines: Board::ApplyAction tb.cc:457-499, SetPegAndLinks tb.cc:501-571, ExploreLocalGraph
// tb.cc:573-588, UpdateResult tb.cc:192-207, LegalActions / DoApplyAction tw.h:86-104, Board::Board tb.cc:168-276.

template <int N>
struct Win1Geo {
  using WG = WinGeo<N>;
  static constexpr int L_WIN = 64 * WG::WSTR * 16;  // windows first: an over-read behind the last one lands in the pegs
  static constexpr int L_PEG = WG::RP * 1024;       // peg rows of the 64 boards: row r, lane l at r * 1024 + l * 16
  static constexpr int lds_bytes = L_WIN + L_PEG;
  static_assert(2 * WG::LFB + 64 <= L_PEG, "the flood pass keeps its two regions over the peg rows");
};

// the links + flags region of the empty board (tb.cc:223-231: border flags), in 16-byte pieces
template <int N>
struct EmptyRegion {
  uint4 piece[WinGeo<N>::LFP];
};
template <int N>
constexpr EmptyRegion<N> make_empty_region() {
  EmptyRegion<N> r{};
  uint32_t w[WinGeo<N>::LFP * 4] = {};
  for (int i = 0; i < WinGeo<N>::W; i++) {
    w[6 * i + 4] = twx::Geo<N>::tab.init[twx::kPlaneStart * WinGeo<N>::W + i];
    w[6 * i + 5] = twx::Geo<N>::tab.init[twx::kPlaneEnd * WinGeo<N>::W + i];
  }
  for (int p = 0; p < WinGeo<N>::LFP; p++) r.piece[p] = uint4{w[4 * p], w[4 * p + 1], w[4 * p + 2], w[4 * p + 3]};
  return r;
}
template <int N>
__device__ const EmptyRegion<N> kEmptyRegion = make_empty_region<N>();

template <int IMM>
__device__ __forceinline__ void gload4iv(uint32_t &dst, uint32_t voff, const void *base) {
  asm volatile(TWX_SGPR_HAZARD_NOP "global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}

template <int N>
__device__ __forceinline__ void step_window1_body(char *__restrict__ state, long bpad, long batch,
                                                  const int32_t *__restrict__ actions, int32_t *__restrict__ actions_out,
                                                  uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
                                                  int jpos, uint32_t *__restrict__ ticket, TicketBases tbase, int nstatic) {
  using G = twx::Geo<N>;
  using WG = WinGeo<N>;
  using W1 = Win1Geo<N>;
  constexpr int W = WG::W, RP = WG::RP, RSC = WG::RSC, RJ = WG::RJ, LFB = WG::LFB, LFP = WG::LFP;
  constexpr int NPC = WG::NPC, WSTR = WG::WSTR;
  constexpr int BPW = 64;
  constexpr int NS = twx::kJournalSlots - 1;  // records a journal launch may have to replay (on the pegs and scalars)
  constexpr uint32_t kColMask = (1u << N) - 1u;
  static_assert(twx::kJournalChunks == 2 && W <= 32 && LFP <= 32 && WSTR >= NPC && (WSTR & 1) == 1, "layout");
  static_assert(RSC * 512 - 2048 <= 4095 && RJ * 512 + 7 * 128 - 2048 <= 4095, "row offsets fit the load immediates");
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_u32 *)lds);

  const int lane = threadIdx.x;
  const int th = lane >> 5, q = lane & 31;  // tile of the batch, board slot of the tile
  const bool random = (flags & kFlagRandom) != 0;
  const bool compact = (flags & kFlagCompact) != 0;  // this launch also writes peg rows and scalars back (journal base 0)
  const bool flush = (flags & kFlagFlush) != 0;      // ... and makes no move
  const long nbatch = bpad / BPW;

  const bool dyn = (flags & kFlagDynamic) != 0;
  const int tgroup = blockIdx.x & (kTicketGroups - 1);
  auto draw = [&]() -> uint32_t {
    uint32_t t = 0u;
    if (lane == 0)
      t = __hip_atomic_fetch_add(ticket + tgroup * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  auto ticket_batch = [&](uint32_t tv) -> long {
    const uint32_t k = __builtin_amdgcn_readfirstlane(tv) - tbase.b[tgroup];
    return uniform_long((long)nstatic * gridDim.x + (long)k * kTicketGroups + tgroup);
  };
  auto batch_of = [&](long i) -> long { return (flags & kFlagReverse) ? reversed_batch(i, nbatch) : i; };
  // a batch = two consecutive tiles
  auto tile_of = [&](long i) -> char * { return uniform_ptr(state + batch_of(i) * (2 * WG::TILE)); };
  const uint32_t my_tile = (uint32_t)th * (uint32_t)WG::TILE;  // this lane's tile inside the batch

  // ---- LDS ----
  uint32_t *const winme = lds + lane * (WSTR * 4);  // this board's window: piece p at winme + 4 p; padding at kTrash
  uint32_t *const pegme = lds + W1::L_WIN / 4 + (lane << 2);  // this board's 16 bytes of peg row r: pegme + 256 r
  auto pegw = [&](int i) -> uint32_t * { return pegme + (i >> 1) * 256 + (i & 1) * 2; };  // word i: { red, blue }
  const uint32_t lds_peg = lds0 + (uint32_t)W1::L_WIN + (uint32_t)lane * 16u;
  const uint32_t lds_win = lds0 + (uint32_t)lane * (uint32_t)(WSTR * 16);
  uint32_t *const fbase = lds + W1::L_WIN / 4;  // flood pass: region of half-wave h at fbase + h * LFB / 4

  // ---- the rows of a batch, straight into the lanes that own them ----
  u32x4 pegv[RP], scv;
  uint32_t jrv[NS];
  uint32_t voff_rows = my_tile + (uint32_t)q * 16u + 2048u;
  uint32_t voff_j = my_tile + (uint32_t)q * 4u + 2048u;
  asm volatile("" : "+v"(voff_rows), "+v"(voff_j));
  auto load_rows = [&](const char *tile) {
    static_for<0, RP>([&](auto rt) {
      constexpr int R = decltype(rt)::value;
      gload16iv<R * 512 - 2048>(pegv[R], voff_rows, tile);
    });
    gload16iv<RSC * 512 - 2048>(scv, voff_rows, tile);
    static_for<0, NS>([&](auto st) {  // record s of this board: one dword of the [slot][board] rows (a line per tile)
      constexpr int S = decltype(st)::value;
      if (S < jpos) gload4iv<RJ * 512 + S * 128 - 2048>(jrv[S], voff_j, tile);
    });
  };
  u32x4 wv[NPC];  // the window under way (a lane's loads leave the pieces it does not need as they are)
#pragma unroll
  for (int p = 0; p < NPC; p++) wv[p] = u32x4{0u, 0u, 0u, 0u};
  auto window_piece0 = [&](int x) -> int { return win_piece0(N, W, x); };  // first piece of the window of a move in column x

  long idx = blockIdx.x;  // the launcher guarantees gridDim.x <= nbatch
  long round = 0;
  uint32_t tv = (dyn && nstatic == 1) ? draw() : 0u;
  STAMP_DECL;
  load_rows(tile_of(idx));
  vm_wait_all();
  long idx_next = (dyn && nstatic == 1) ? ticket_batch(tv) : idx + gridDim.x;
  bool drawn = dyn && round + 2 >= nstatic;
  if (drawn) tv = draw();

  // ---- what a batch carries from its select phase to its apply phase ----
  bool c_valid = false;
  long c_idx = 0;
  int c_a = -1, c_pa = 0;
  uint32_t c_counter = 0, c_move_one = 0, c_result = 0, c_swapped = 0, c_cand = 0, c_episode = 0;
  uint32_t c_wrange = 0;  // the bytes of its region the batch's lane fetched: offset | (length - 24) << 16
  uint32_t c_jr[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) c_jr[s] = 0u;
  uint32_t c_bits = 0;  // 1 act, 2 illegal, 4 was_reset, 8 swapnow, 16 inb, 32 the move fills the opponent's last cell
  // ---- flood passes in waiting: two slots of two boards each (lane ids of the batch they came from) ----
  struct Pass {
    int fill;          // boards in the slot: 0, 1 (half-wave 0) or 2
    int s0, s1;        // their lanes
    char *t0, *t1;     // their batches' first tiles
    u32x4 pv;          // region pieces under way: one per lane, LFP lanes per half-wave
    uint32_t notes;    // lane j < 8 of half h: record j of its board (slot jpos: the move itself), lane 8 cell, lane 9 flags
  };
  Pass pass[2];
#pragma unroll
  for (int k = 0; k < 2; k++) { pass[k].fill = 0; pass[k].s0 = pass[k].s1 = 0; pass[k].t0 = pass[k].t1 = state; pass[k].notes = 0u; }

  while (idx < nbatch || c_valid || pass[0].fill != 0 || pass[1].fill != 0) {
    STAMP(0);  // loop end of the previous iteration
    const bool have = idx < nbatch;  // (wave-uniform) is there a batch to choose moves for?
    const long row = batch_of(have ? idx : 0) * BPW + lane;  // = board: whole-batch launches only
    const bool inb = have && row < batch;
    int a_ext = -1;
    if (!random && !flush && have) {  // external actions (twx_step): loaded, and waited for, before anything untracked is issued
      if (inb) a_ext = actions[row];
      asm volatile("" : "+v"(a_ext));
    }
    // ---- 1. what arrived: pegs, scalars and records of the batch to choose for; windows of the batch to apply ----
    uint32_t counter = 0, move_one = 0, result = 0, swapped = 0, episode = 0, jbase = 0;
    uint32_t jr[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) jr[s] = 0u;
    if (have) {
      static_for<0, RP>([&](auto rt) {
        constexpr int R = decltype(rt)::value;
        lwrite16iv<R * 1024>(lds_peg, pegv[R]);
      });
      // (out of the prefetch registers before the next prefetch is issued into them.  ALL FOUR registers of the
      // scalars must count as read here: the compiler does not know they were in flight, and the one nobody
      // reads -- s3 -- was handed to another value while the load was still to write it)
      asm volatile("" ::"v"(scv));
      counter = scv.x & 0xffffu; move_one = scv.x >> 16;
      result = scv.y & 0xffu; swapped = (scv.y >> 8) & 1u;
      episode = scv.z;
      jbase = (scv.y >> twx::kJBaseShift) & 15u;  // first journal slot that belongs to this board's game
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const uint32_t e = (s < jpos) ? jrv[s] : 0u;  // (wave-uniform; slots from jpos on are stale and were not loaded)
        jr[s] = ((uint32_t)s >= jbase) ? e : 0u;
      }
    }
    if (c_valid) {
      static_for<0, NPC>([&](auto pt) {
        constexpr int P = decltype(pt)::value;
        lwrite16iv<P * 16>(lds_win, wv[P]);
      });
    }
    wave_sync();
    STAMP(1);  // pegs and windows -> LDS

    // ---- 2. choose the moves of batch idx ----
    int n_a = -1, n_pa = 0;
    uint32_t n_counter = 0, n_move_one = 0, n_result = 0, n_swapped = 0, n_cand = 0, n_bits = 0, n_episode = 0, n_wrange = 0;
    if (have) {
      const char *const tile_cur = tile_of(idx);
      __builtin_amdgcn_sched_barrier(0);
      load_rows(tile_of((idx_next < nbatch) ? idx_next : idx));  // (they have two phases to arrive)
      __builtin_amdgcn_sched_barrier(0);
      // the stored scalars are as old as the planes: advance them by the journal records first
#pragma unroll
      for (int s = 0; s < NS; s++)
        if (s < jpos) twx::journal_scalars(jr[s], counter, move_one, result, swapped);
      const bool was_reset = random && (flags & kFlagAutoReset) && result != 0u && inb;
      if (__ballot(was_reset)) {  // rare: one board in ~520 per step
        if (was_reset) {
          // ---- auto-reset (tb.cc:168-174, 209-276): empty pegs here, and the whole empty board to HBM -- rows and
          //      scalars (with the journal base: the records before this launch's slot belong to the finished
          //      game, the first move of the new one is in this launch's record) and the region of the empty
          //      board.  Its window will still show the finished game: see `fresh` below ----
#pragma unroll
          for (int s = 0; s < NS; s++) jr[s] = 0u;
          jbase = (uint32_t)jpos;
          counter = 0; move_one = 0; result = 0; swapped = 0; episode += 1;
          char *const tw = const_cast<char *>(tile_cur) + my_tile;
#pragma unroll
          for (int r = 0; r < RP; r++) {
            *reinterpret_cast<uint4 *>(pegme + r * 256) = make_uint4(0u, 0u, 0u, 0u);
            tile_store<true>(reinterpret_cast<uint4 *>(tw + r * 512 + q * 16), make_uint4(0u, 0u, 0u, 0u));
          }
          tile_store<true>(reinterpret_cast<uint4 *>(tw + RSC * 512 + q * 16), make_uint4(0u, jbase << twx::kJBaseShift, episode, 0u));
          // (a rolled loop over a table in memory: unrolled, the constants of the empty board sat in 28 registers
          // of a kernel that has none to spare; nothing untracked is under way at this point of the iteration)
          uint4 *const dst = reinterpret_cast<uint4 *>(tw + WG::L_ROWS + q * LFB);
#pragma unroll 1
          for (int p = 0; p < LFP; p++) tile_store<true>(dst + p, kEmptyRegion<N>.piece[p]);
        }
        wave_sync();
      }
      // the records' peg bits; move_one is final already (move 0 sets it, a swap is move 1)
#pragma unroll
      for (int s = 0; s < NS; s++) {
        if (s < jpos) {  // wave-uniform
          const uint32_t e = jr[s];
          const int cell = (int)(e & 1023u);
          const uint32_t bit = (1u << (cell & 31)) & (uint32_t)((int32_t)e >> 31);  // empty record: the OR is a no-op
          if (__ballot((e & twx::kJSwap) != 0u))  // rare: undo red's first peg on a swap (tb.cc:450-455)
            lds_and(pegw((int)(move_one >> 5)), (e & twx::kJSwap) ? ~(1u << (move_one & 31)) : 0xffffffffu);
          lds_or(pegw(cell >> 5) + ((e >> 10) & 1u), bit);
        }
      }
      wave_sync();
      STAMP(2);  // row prefetch issue, scalars, reset, peg replay

      const int player = (int)(counter & 1u);
      const bool open = (result == 0u);
      // ---- legal mask of the player to move (tw.h:86-90), two words per peg row; the opponent's
      //      free-cell count for the draw test (tb.cc:202-206) only when some board of the wave is nearly full ----
      uint32_t m[2 * RP];
      const bool count_opp = __ballot(counter + 1u >= (uint32_t)(N * (N - 2))) != 0ull;
      int cnt = 0, cnt_opp = 1 << 20;  // "plenty": no draw
      if (count_opp) cnt_opp = 0;
      const uint32_t not_swap = (counter == 1u) ? 0u : 0xffffffffu;
#pragma unroll
      for (int r = 0; r < RP; r++) {
        const uint4 rb = *reinterpret_cast<const uint4 *>(pegme + r * 256);
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int w = 2 * r + k;
          const uint32_t reg_r = w < W ? G::tab.region[0][w < W ? w : 0] : 0u, reg_b = w < W ? G::tab.region[1][w < W ? w : 0] : 0u;
          const uint32_t red = k ? rb.z : rb.x, blue = k ? rb.w : rb.y;
          // (move 1, blue to move: red's first peg stays legal for blue -- the swap; blue has no peg yet)
          m[w] = (player ? reg_b : reg_r) & ~((red & not_swap) | blue);
          cnt += __popc(m[w]);
          if (count_opp) cnt_opp += __popc((player ? reg_r : reg_b) & ~(red | blue));
        }
      }
      STAMP(3);  // legal mask

      int a = -1;
      bool act = inb && open && !flush;
      bool illegal = false;
      if (random) {
        act = act && cnt > 0;
        const uint32_t hsh = twx::policy_hash(seed, first_id + (uint32_t)row, episode, counter);
        const int k = (int)__umulhi(hsh, (uint32_t)cnt);
        int cum = __popc(m[0]), wsel = 0, base = 0;
        uint32_t msel = m[0];
#pragma unroll
        for (int i = 1; i < 2 * RP; i++) {  // last word whose prefix count is <= k
          // (one compare in vcc per word, consumed on the spot: left to itself the scheduler hoists all the
          // compares and parks their masks in eighteen SGPR pairs it has to spill for)
          const bool ge = k >= cum;
          wsel += ge ? 1 : 0;
          msel = ge ? m[i] : msel;
          base = ge ? cum : base;
          cum += __popc(m[i]);
          __builtin_amdgcn_sched_barrier(0);
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
          const uint2 rb = *reinterpret_cast<const uint2 *>(pegw(a >> 5));
          const bool occ = ((rb.y & bit) != 0u) || (counter != 1u && (rb.x & bit) != 0u);
          legal = inreg && !occ;
        }
        illegal = inb && !(open && legal);  // tw.h:95-97 (terminal: LegalActions() is empty)
        act = act && legal;
      }
      if (kAblate & 8) act = false;
      // where the peg lands (tb.cc:462-473: a swap rotates it)
      int pa = act ? a : 0;
      const bool swapnow = act && (counter == 1u) && ((uint32_t)a == move_one);
      if (swapnow) {
        const int ax = a / N, ay = a - ax * N;
        pa = ay * N + (N - 1 - ax);
      }
      const int x = pa / N, y = pa - x * N;
      // knight neighbours of the mover's colour (tb.cc:517), from the pegs: the apply phase will not have them
      uint32_t wo[5];
#pragma unroll
      for (int ci = -2; ci <= 2; ci++) {
        if (ci == 0) { wo[2] = 0u; continue; }
        const int c = x + ci;
        const bool ok = (unsigned)c < (unsigned)N;
        const int o = (ok ? c : x) * N, sh = o & 31, wi = o >> 5;
        const uint32_t own = ok ? (funnel(pegw(wi)[player], pegw(wi + 1 < W ? wi + 1 : wi)[player], sh) & kColMask) : 0u;
        wo[ci + 2] = ((own << 2) >> y) & 0x1fu;  // bit 2 + dy <-> row y + dy
      }
      uint32_t cand = 0;
#pragma unroll
      for (int d = 0; d < 8; d++) cand |= ((wo[twx::kDX[d] + 2] >> (twx::kDY[d] + 2)) & 1u) << d;
      // would this move take the opponent's last free cell (tb.cc:202-206)?
      const bool in_opp = player ? (x >= 1 && x <= N - 2) : (y >= 1 && y <= N - 2);
      const bool wdraw = counter + 1u > 2u && cnt_opp - (in_opp ? 1 : 0) == 0;
      if (compact) {
        // ---- a compaction: the pegs as they stand after this move go back to their rows (the scalars follow from
        //      the apply phase, which knows the result).  Regions are always current in HBM: nothing else to write ----
        if (swapnow) lds_and(pegw(a >> 5), ~(1u << (a & 31)));            // tb.cc:450-455: undo red's first peg
        if (act) lds_or(pegw(pa >> 5) + player, 1u << (pa & 31));         // tb.cc:507-508
        wave_sync();
        char *const tw = const_cast<char *>(tile_cur) + my_tile + q * 16;
#pragma unroll
        for (int r = 0; r < RP; r++)
          tile_store<true>(reinterpret_cast<uint4 *>(tw + r * 512), *reinterpret_cast<const uint4 *>(pegme + r * 256));
      }
      // ---- the window: the pieces of this board's region that the move can touch, fetched by the lane itself.
      //      A peg without a neighbour of its colour a knight's move away (3 moves in 10 of a random game) links
      //      nothing, gains no flag and changes no region word (tb.cc:517: the scan finds nothing): its lane
      //      fetches nothing and the apply phase ignores what its window holds.  Otherwise the candidates say which
      //      columns count: a link to x+1 can only be crossed by links whose west ends lie in columns x-1, x; to
      //      x+2: x-1..x+1; to x-1: x-2, x-1; to x-2: x-3..x-1 (twixt_geom.h kCross: the other columns' masks are
      //      zero), the flags are those of the candidates' columns and of the peg's own.  The words of that column
      //      range, as 16-byte pieces: 1.7 lines of 128 bytes on average instead of 2.1 for all six columns ----
      const int piece0 = window_piece0(x);
      const uint32_t voff_w = my_tile + (uint32_t)(WG::L_ROWS + q * LFB + 16 * piece0);
      uint32_t wmask = 0u;
      {
        const WinRange wr = win_range(N, cand, x);  // (twixt_device_common.h; coverage proved in tests/layout_check.hip)
        if (act && cand != 0u) wmask = ((2u << wr.np) - 1u) << (wr.p_lo - piece0);
        n_wrange = (uint32_t)(16 * wr.p_lo) | ((uint32_t)(16 * wr.np - 8) << 16);  // byte offset; bytes - 24: see win_has
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NPC>([&](auto pt) {
        constexpr int P = decltype(pt)::value;
        gload16iv_if<P * 16>(wv[P], voff_w, tile_cur, wmask & (1u << P));
      });
      __builtin_amdgcn_sched_barrier(0);
      n_a = a; n_pa = pa;
      n_counter = counter; n_move_one = move_one; n_result = result; n_swapped = swapped; n_cand = cand; n_episode = episode;
      n_bits = (act ? 1u : 0u) | (illegal ? 2u : 0u) | (was_reset ? 4u : 0u) | (swapnow ? 8u : 0u) | (inb ? 16u : 0u) |
               (wdraw ? 32u : 0u);
    }
    STAMP(4);  // move, neighbours, window fetch issue

    // ---- 3. apply the moves of the carried batch on its windows ----
    uint32_t entry = 0, c_pend = 0;  // this move's record; bit0 / bit1: flood START / END from the new peg
    uint32_t o_result = c_result, o_swapped = c_swapped, o_counter = c_counter, o_move_one = c_move_one;
    if (c_valid) {
      // (fresh: the window does not show this board -- reset in its select phase, or not fetched at all)
      const bool act = (c_bits & 1u) != 0u, fresh = (c_bits & 4u) != 0u || c_cand == 0u, swapnow = (c_bits & 8u) != 0u;
      const int pa = c_pa;
      const int x = pa / N, y = pa - x * N;
      const int player = (int)(c_counter & 1u);
      const int piece0 = window_piece0(x);
      // is word w of the region among the pieces this lane fetched?
      auto win_has = [&](int w) -> bool { return (uint32_t)(24 * w) - (c_wrange & 0xffffu) <= (c_wrange >> 16); };
      STAMP(5);  // (no replay here: the regions in HBM are always current)
      uint32_t pend = 0, fill1 = 0;
      uint32_t counter2 = c_counter, move_one2 = c_move_one, result2 = c_result, swapped2 = c_swapped;
      if (act) {
        // ---- Board::ApplyAction (tb.cc:457-499), SetPegAndLinks (tb.cc:501-571), all 8 directions at once.
        //      Six board columns x-3..x+2 of the window: ws / we rows y-2..y+2 of START / END (columns x-2..x+2),
        //      win rows y-3..y+3 of the four link planes (columns x-3..x+1) ----
        uint32_t *const wq = winme - 4 * piece0;  // word w of the region: wq + 6 w (inside the window)
        const uint32_t pbit = 1u << (pa & 31);
        if (swapnow) swapped2 = 1u;
        uint32_t ws[5], we[5], win[twx::kWinCols];
#pragma unroll
        for (int ci = -3; ci <= 2; ci++) {
          const int c = x + ci;
          const bool ok = (unsigned)c < (unsigned)N;
          const int o = (ok ? c : x) * N, sh = o & 31;
          const uint32_t *const g = wq + 6 * (o >> 5);  // next word: + 6 dwords (behind the window: padding / next area)
          if (ci >= -2) {
            const uint2 f0 = *reinterpret_cast<const uint2 *>(g + 4), f1 = *reinterpret_cast<const uint2 *>(g + 10);
            const uint32_t fs = ok ? (funnel(f0.x, f1.x, sh) & kColMask) : 0u, fe = ok ? (funnel(f0.y, f1.y, sh) & kColMask) : 0u;
            ws[ci + 2] = ((fs << 2) >> y) & 0x1fu;
            we[ci + 2] = ((fe << 2) >> y) & 0x1fu;
          }
          if (ci <= 1) {
            const uint2 a0 = *reinterpret_cast<const uint2 *>(g), b0 = *reinterpret_cast<const uint2 *>(g + 2);
            const uint2 a1 = *reinterpret_cast<const uint2 *>(g + 6), b1 = *reinterpret_cast<const uint2 *>(g + 8);
            const uint32_t p0 = funnel(a0.x, a1.x, sh) & kColMask, p1 = funnel(a0.y, a1.y, sh) & kColMask;
            const uint32_t p2 = funnel(b0.x, b1.x, sh) & kColMask, p3 = funnel(b0.y, b1.y, sh) & kColMask;
            const uint32_t v = (((p0 << 3) >> y) & 0x7fu) | ((((p1 << 3) >> y) & 0x7fu) << 8) |
                               ((((p2 << 3) >> y) & 0x7fu) << 16) | ((((p3 << 3) >> y) & 0x7fu) << 24);
            win[ci + 3] = ok ? v : 0u;
          }
        }
        uint32_t ts = 0, te = 0, blocked = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) {
          const int ci = twx::kDX[d] + 2, rb = twx::kDY[d] + 2;
          ts |= ((ws[ci] >> rb) & 1u) << d;
          te |= ((we[ci] >> rb) & 1u) << d;
          uint32_t hit = 0;  // crossing test (tb.cc:519-527) as constant masks (twixt_geom.h)
#pragma unroll
          for (int i = 0; i < twx::kWinCols; i++) hit |= win[i] & twx::kCross.m[d][i];
          blocked |= (hit != 0u ? 1u : 0u) << d;
        }
        const uint32_t linkm = c_cand & ~blocked;
        // tb.cc:538-546: inherit START, else END, else remember a neutral neighbour
        const uint32_t gain_s = linkm & ts, gain_e = linkm & ~ts & te, neutral = linkm & ~ts & ~te;
        // border flags the empty cell already carries (tb.cc:223-231).  (The window of a board that was reset in its
        // select phase still shows the finished game.  Its first peg has no neighbour of its colour, so nothing of
        // the link scan counts; only these two flags would -- and no cell of the empty board carries both)
        const bool own_s = !fresh && ((ws[2] >> 2) & 1u), own_e = !fresh && ((we[2] >> 2) & 1u);
        const bool p_s = own_s || gain_s;
        const bool p_e = own_e || gain_e;
        uint32_t want = (linkm && neutral) ? ((p_s ? 1u : 0u) | (p_e ? 2u : 0u)) : 0u;  // tb.cc:558-570
        // the single-cell fill (half of all fills), settled on the pre-move planes when all of the neighbour's links
        // lie inside the window; otherwise the long way
        if (want && !(neutral & (neutral - 1u)) && (!(want & 1u) || (linkm & ~ts) == neutral) &&
            (!(want & 2u) || (linkm & ~te) == neutral)) {
          const int d = __ffs((int)neutral) - 1;
          const int tx = x + (int)((twx::kDxPack >> (4 * d)) & 15u) - 2, ty = y + (int)((twx::kDyPack >> (4 * d)) & 15u) - 2;
          const int t = tx * N + ty;
          bool seen = win_has(t >> 5);
          const uint32_t *const gt = wq + 6 * ((seen ? t : pa) >> 5);  // links that START at t
          uint32_t other = (gt[0] | gt[1] | gt[2] | gt[3]) >> (t & 31);
#pragma unroll
          for (int e = 0; e < 4; e++) {  // links that END at t: stored at t - off[e]
            const int sx = tx - twx::kDX[e], sy = ty - twx::kDY[e];
            const bool valid = sx >= 0 && sy >= 0 && sy < N;
            const int sc2 = valid ? sx * N + sy : pa;
            const bool here = win_has(sc2 >> 5);
            seen = seen && here;
            const uint32_t w = wq[6 * ((here ? sc2 : pa) >> 5) + e];
            other |= valid ? (w >> (sc2 & 31)) : 0u;
          }
          if (seen && !(other & 1u)) {
            fill1 = ((uint32_t)d << twx::kJFillDirShift) | ((want & 1u) ? twx::kJFillStart : 0u) |
                    ((want & 2u) ? twx::kJFillEnd : 0u);
            want = 0u;
          }
        }
        pend = want;
        if (counter2 == 0u) move_one2 = (uint32_t)pa;  // tb.cc:485-488
        counter2 += 1u;
        // ---- UpdateResult (tb.cc:192-207) ----
        if (p_s && p_e) result2 = player ? 2u : 1u;
        else if (c_bits & 32u) result2 = 3u;
        entry = twx::kJValid | (uint32_t)pa | ((uint32_t)player << 10) | (linkm << 11) |
                (gain_s ? twx::kJGainStart : 0u) | (gain_e ? twx::kJGainEnd : 0u) | (swapnow ? twx::kJSwap : 0u) |
                (result2 << twx::kJResultShift) | fill1;
        // ---- the region words this move changes (tb.cc:532-546): ORed into the window, then stored from there --
        //      8-byte pairs / single dwords through L2, where the window's lines still are (a board's region in
        //      HBM is current after every step; the record above keeps carrying the same bits for the kernels that
        //      replay whole boards).  One stored bit per physical link, at its west end. ----
        uint32_t *const gp = wq + 6 * (pa >> 5);
        const uint32_t b01 = linkm & 3u, b23 = linkm & 12u, bse = (gain_s ? 1u : 0u) | (gain_e ? 2u : 0u);
        lds_or2(gp + 0, (linkm & 1u) ? pbit : 0u, (linkm & 2u) ? pbit : 0u);
        lds_or2(gp + 2, (linkm & 4u) ? pbit : 0u, (linkm & 8u) ? pbit : 0u);
        lds_or2(gp + 4, gain_s ? pbit : 0u, gain_e ? pbit : 0u);
        int tcell[4];
#pragma unroll
        for (int d = 4; d < 8; d++) {  // links whose west end is the other peg
          const bool on = (linkm >> d) & 1u;
          tcell[d - 4] = on ? pa + twx::kDX[d] * N + twx::kDY[d] : pa;
          lds_or(wq + 6 * (tcell[d - 4] >> 5) + (d - 4), on ? (1u << (tcell[d - 4] & 31)) : 0u);
        }
        int tfill = pa;
        if (fill1) {  // the single neighbour the shortcut flagged
          const int fd = (int)((fill1 >> twx::kJFillDirShift) & 7u);
          tfill = pa + ((int)((twx::kDxPack >> (4 * fd)) & 15u) - 2) * N + (int)((twx::kDyPack >> (4 * fd)) & 15u) - 2;
          lds_or2(wq + 6 * (tfill >> 5) + 4, (fill1 & twx::kJFillStart) ? (1u << (tfill & 31)) : 0u,
                  (fill1 & twx::kJFillEnd) ? (1u << (tfill & 31)) : 0u);
        }
        wave_sync();
        char *const rb = tile_of(c_idx) + my_tile + (WG::L_ROWS + q * LFB);  // this board's region in HBM
        if (b01 | b23) {  // (the four link words of the cell's word as one 16-byte piece)
          const uint2 la = *reinterpret_cast<const uint2 *>(gp + 0), lb = *reinterpret_cast<const uint2 *>(gp + 2);
          uint32_t *const d4 = reinterpret_cast<uint32_t *>(rb + 24 * (pa >> 5));
          d4[0] = la.x; d4[1] = la.y; d4[2] = lb.x; d4[3] = lb.y;
        }
        if (bse) *reinterpret_cast<uint2 *>(rb + 24 * (pa >> 5) + 16) = *reinterpret_cast<const uint2 *>(gp + 4);
#pragma unroll
        for (int d = 4; d < 8; d++)
          if ((linkm >> d) & 1u)
            *reinterpret_cast<uint32_t *>(rb + 24 * (tcell[d - 4] >> 5) + 4 * (d - 4)) = wq[6 * (tcell[d - 4] >> 5) + (d - 4)];
        if (fill1)
          *reinterpret_cast<uint2 *>(rb + 24 * (tfill >> 5) + 16) = *reinterpret_cast<const uint2 *>(wq + 6 * (tfill >> 5) + 4);
      }
      o_result = result2; o_swapped = swapped2; o_counter = counter2; o_move_one = move_one2;
      c_pend = pend;
    }
    STAMP(6);  // link scan, result, record

    // What was requested in this iteration has had the apply phase to arrive: retire it here, before the
    // stores are issued (loads and stores share one in-order counter).
    vm_wait_all();
    long idx_after = 0;
    STAMP(7);  // wait for the rows and windows under way
    if (have) {
      idx_after = drawn ? ticket_batch(tv) : (long)blockIdx.x + (round + 2) * gridDim.x;
      drawn = dyn && round + 3 >= nstatic && idx_next < nbatch;
      if (drawn) tv = draw();
    }
    // ---- ExploreLocalGraph (tb.cc:573-588) for the fills that flag more than one cell (2.8 % of the moves: 1.8
    //      boards of a batch).  Such a fill needs the board's WHOLE region, and a pass floods two boards as cheaply
    //      as one (a half-wave each, lane = board column).  So the boards wait in two pass slots: a slot that has
    //      its two boards requests their regions and is flooded after the NEXT wait (nobody reads these boards
    //      again in this launch, and the result of a move never depends on its fill); a lone board is flooded
    //      when the batches run out.  Region pieces (current in HBM, this move included) -> the flood buffers (over
    //      the peg rows, dead by now), the flood, the flag words that changed stored as 8-byte pieces. ----
    auto region_of = [&](int s) -> uint32_t { return (uint32_t)(s >> 5) * (uint32_t)WG::TILE + (uint32_t)(WG::L_ROWS + (s & 31) * LFB); };
    auto flood_notes = [&](uint32_t &notes, auto half, int sq) {  // cell and pending flags of the board in lane sq -> half-wave `half`
      constexpr int Q = decltype(half)::value;
      set_lane<32 * Q + 8>(notes, __builtin_amdgcn_readlane(c_pa, sq));
      set_lane<32 * Q + 9>(notes, __builtin_amdgcn_readlane((int)c_pend, sq));
    };
    using Half0 = std::integral_constant<int, 0>;
    using Half1 = std::integral_constant<int, 1>;
    const bool up = th != 0;
    auto flood_run = [&](uint32_t notes, int s0, int s1, bool two, char *tile0, char *tile1) {
      const int c = q;
      const int src = up ? s1 : s0;
      char *const tile = up ? tile1 : tile0;
      const bool idle = up && !two;  // one board only: the upper half has nothing to do
      const int spa = up ? __builtin_amdgcn_readlane((int)notes, 40) : __builtin_amdgcn_readlane((int)notes, 8);
      const uint32_t spend = idle ? 0u : (uint32_t)(up ? __builtin_amdgcn_readlane((int)notes, 41) : __builtin_amdgcn_readlane((int)notes, 9));
      uint32_t *const fb = fbase + th * (LFB / 4);  // (the region as it is in HBM: current, this move included)
      const int sx = spa / N, sy = spa - sx * N;
      const bool ok = c < N;
      const int o = (ok ? c : 0) * N, sh = o & 31, grp = o >> 5;
      uint32_t *const g = fb + 6 * grp;
      auto col = [&](int off) -> uint32_t {
        const uint32_t v = funnel(g[off], g[off + 6], sh) & kColMask;
        return ok ? v : 0u;
      };
      uint32_t lk[4];
#pragma unroll
      for (int e = 0; e < 4; e++) lk[e] = col(e);
      bool changed = false;
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const bool en = (spend >> f) & 1u;  // uniform per half
        if (!__ballot(en)) continue;
        const int off = 4 + f;
        const uint32_t flag0 = col(off);
        uint32_t flag = flag0;
        uint32_t front = (en && c == sx) ? (1u << sy) : 0u;
        for (;;) {
          uint32_t fresh_bits = 0u;
#pragma unroll
          for (int rep = 0; rep < 2; rep++) {  // two expansions per exit test
            // forward: a frontier cell with link e reaches (c+dx, y+dy); backward: cell (c,y) whose link e ends in
            // the frontier.  Columns move by DPP wave shifts, rows by bit shifts.  (Lanes 31 / 32 are neighbours
            // for the shifts but column 31 does not exist: nothing crosses between the halves.)
            const uint32_t a1 = ((front & lk[0]) << 2) | ((front & lk[3]) >> 2);
            const uint32_t a2 = ((front & lk[1]) << 1) | ((front & lk[2]) >> 1);
            uint32_t nb = wave_shr1(a1) | wave_shr1(wave_shr1(a2));
            const uint32_t f1 = wave_shl1(front), f2 = wave_shl1(f1);
            nb |= ((f1 >> 2) & lk[0]) | ((f2 >> 1) & lk[1]) | ((f2 << 1) & lk[2]) | ((f1 << 2) & lk[3]);
            fresh_bits = ok ? (nb & ~flag & kColMask) : 0u;
            flag |= fresh_bits;
            front = fresh_bits;
          }
          if (__ballot(fresh_bits != 0u) == 0ull) break;
        }
        const uint32_t add = flag & ~flag0;
        if (ok && add && en) {
          changed = true;
          lds_or(g + off, add << sh);
          if (sh + N > 32) lds_or(g + off + 6, add >> (32 - sh));
        }
      }
      wave_sync();
      // the flag words this lane's column changed go back as 8-byte pieces { start_i, end_i } (two columns that
      // share a word store the same bytes twice)
      if (changed && !(kAblate & 2)) {
        uint32_t *const dst = reinterpret_cast<uint32_t *>(tile + region_of(src) + 24 * grp + 16);
        const uint2 fv = *reinterpret_cast<const uint2 *>(g + 4);
        __builtin_nontemporal_store(fv.x, dst);
        __builtin_nontemporal_store(fv.y, dst + 1);
        if (sh + N > 32) {
          const uint2 fw = *reinterpret_cast<const uint2 *>(g + 10);
          __builtin_nontemporal_store(fw.x, dst + 6);
          __builtin_nontemporal_store(fw.y, dst + 7);
        }
      }
      wave_sync();
    };
    auto region_loads = [&](Pass &p) {  // both boards of a slot (a waiting board is simply read again)
      const char *const a = (up ? p.t1 : p.t0) + (region_of(up ? p.s1 : p.s0) + (uint32_t)((q < LFP ? q : 0) * 16));
      __builtin_amdgcn_sched_barrier(0);
      gload16p(p.pv, a);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto run_pass = [&](Pass &p) {
      const bool two = p.fill == 2;
      if (q < LFP && !(up && !two)) lwrite16v(lds0 + (uint32_t)(W1::L_WIN + th * LFB + q * 16), p.pv);
      wave_sync();
      flood_run(p.notes, p.s0, two ? p.s1 : p.s0, two, p.t0, p.t1);
      p.fill = 0;
    };
    const bool draining = !have && !c_valid;
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (pass[k].fill == 2 || (pass[k].fill == 1 && draining)) run_pass(pass[k]);  // requested an iteration ago or earlier
    STAMP(8);  // ticket; the flood passes that were waiting
    if (c_valid) {
      uint64_t todo = __ballot(c_pend != 0u);
      if (kAblate & 1) todo = 0ull;
      if (todo) {
        char *const tile_c = tile_of(c_idx);
#pragma unroll
        for (int k = 0; k < 2; k++) {
          Pass &p = pass[k];
          if (todo && p.fill == 0) {
            p.s0 = p.s1 = (int)__ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            p.t0 = p.t1 = tile_c;
            flood_notes(p.notes, Half0{}, p.s0);
            p.fill = 1;
            if (!todo) region_loads(p);
          }
          if (todo && p.fill == 1) {
            p.s1 = (int)__ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            p.t1 = tile_c;
            flood_notes(p.notes, Half1{}, p.s1);
            p.fill = 2;
            region_loads(p);
          }
        }
        while (todo) {  // more fills than the waiting passes take (rare): on the spot
          const int s0 = (int)__ffsll((unsigned long long)todo) - 1;
          todo &= todo - 1;
          const bool two = todo != 0ull;
          const int s1 = two ? (int)__ffsll((unsigned long long)todo) - 1 : s0;
          if (two) todo &= todo - 1;
          Pass now;
          now.fill = two ? 2 : 1; now.s0 = s0; now.s1 = s1; now.t0 = now.t1 = tile_c; now.notes = 0u;
          flood_notes(now.notes, Half0{}, s0);
          flood_notes(now.notes, Half1{}, s1);
          region_loads(now);
          vm_wait_all();
          run_pass(now);
        }
      }
    }
    STAMP(9);  // new fills: notes, region requests, passes on the spot
    if (c_valid) {
      // this move's record: all a step normally writes -- one dword per board, one 128-byte line per tile
      char *const tile_c = tile_of(c_idx);
      const long crow = batch_of(c_idx) * BPW + lane;
      if (compact) {  // the scalars as they stand after the move (the rows went back in the select phase); base 0
        tile_store<true>(reinterpret_cast<uint4 *>(tile_c + my_tile + RSC * 512 + q * 16),
                         make_uint4(o_counter | (o_move_one << 16), o_result | (o_swapped << 8), c_episode, 0u));
      } else if (!(kAblate & 16)) {
        tile_store_word<true>(reinterpret_cast<uint32_t *>(tile_c + my_tile + RJ * 512 + jpos * 128) + q, entry);
      }
      if ((c_bits & 16u) && !flush) {
        if (status) {
          uint8_t st = (uint8_t)(o_result | (o_swapped << 5));
          if (c_bits & 2u) st |= TWX_ST_ILLEGAL;
          if (c_bits & 4u) st |= TWX_ST_RESET;
          if (random && !(c_bits & 1u)) st |= TWX_ST_NOOP;
          status[crow] = st;
        }
        if (actions_out) actions_out[crow] = c_a;
      }
    }
    wave_sync();
    STAMP(10);  // record and status stores
    // the batch just chosen for becomes the carried one
    c_valid = have;
    c_idx = uniform_long(idx);
    c_a = n_a; c_pa = n_pa; c_counter = n_counter; c_move_one = n_move_one; c_result = n_result; c_swapped = n_swapped;
    c_cand = n_cand; c_bits = n_bits; c_episode = n_episode; c_wrange = n_wrange;
#pragma unroll
    for (int s = 0; s < NS; s++) c_jr[s] = jr[s];
    if (have) {  // (kept in scalar registers: the loop and its phases branch on them)
      idx = uniform_long(idx_next);
      idx_next = uniform_long(idx_after);
      round++;
    }
  }  // batch loop
  STAMP_FLUSH;
}

// 20,480 B of LDS per wave at N=24: eight waves per CU, two per SIMD
template <int N>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void step_window1_kernel(
    char *__restrict__ state, long bpad, long batch, const int32_t *__restrict__ actions,
    int32_t *__restrict__ actions_out, uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
    int jpos, uint32_t *__restrict__ ticket, TicketBases tbase, int nstatic) {
  step_window1_body<N>(state, bpad, batch, actions, actions_out, status, seed, first_id, flags, jpos, ticket, tbase, nstatic);
}

#endif  // TWIXT_WINDOW1_KERNEL_H_
