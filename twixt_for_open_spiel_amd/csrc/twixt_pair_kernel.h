// twixt_pair_kernel.h -- step_pair_kernel<N>: the whole-batch env step for big boards, TWO LANES PER BOARD
// Part of libtwixt_hip.so; included by twixt_engine.hip only (device code, anonymous namespace).
#ifndef TWIXT_PAIR_KERNEL_H_
#define TWIXT_PAIR_KERNEL_H_

// Why a second mapping.  step_kernel64 gives every lane a board; at N >= 19 the 64-board LDS image
// (39 KiB at N=24) admits four waves per CU, one per SIMD, and a lone wave issues at most one
// vector instruction every four cycles and sits through every LDS round trip and every flood
// fill by itself: measured by leaving phases out (tools/README.md, TWX_ABLATE), the step logic
// hides completely behind the board stream only when nothing else stalls the wave.  Here a wave
// holds 32 boards (20 KiB image, eight waves per CU, two per SIMD) and the two lanes l and l+32
// SHARE board l:
//   * memory: lane l moves the even chunks of its board, lane l+32 the odd ones (loads, LDS fill,
//     stores); the state array is tiled by 32 boards for these sizes, so every instruction still
//     covers one contiguous KiB;
//   * legal mask / k-th legal action: each lane scans half of the word groups; counts and the
//     chosen cell are combined with one v_permlane32_swap each;
//   * link scan (tb.cc:501-555): the lower lane reads link planes NNE/ENE and the knight
//     neighbours to the west (directions 4..7), the upper lane planes ESE/SSE and the neighbours
//     to the east (0..3); one swap merges {candidates, START, END, blocked} of all 8 directions;
//   * every LDS update of the move and of the journal replay is issued once, by one of the two.
// Everything per board that is a handful of scalar operations (counters, coordinates, result) is
// simply computed by both lanes.  The flood fill (tb.cc:573-588) is the wave-cooperative pass of
// step_kernel64, two boards at a time.  Same state layout, journal, batch order and results as
// step_kernel64: the engine may switch between the two kernels from one launch to the next.
//
// Reference lines: Board::ApplyAction tb.cc:457-499, SetPegAndLinks tb.cc:501-571, UpdateResult
// tb.cc:192-207, LegalActions / DoApplyAction tw.h:86-104, Board::Board tb.cc:168-276.

// both lanes of a pair get the value of the lower (x) and of the upper (y) lane
struct PairVal { uint32_t lo, hi; };
__device__ __forceinline__ PairVal pair_swap(uint32_t v) {
  // v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return PairVal{r[0], r[1]};
}

// bit `pos` of v as a mask: 0 or -1 (v_bfe_i32)
__device__ __forceinline__ int32_t sbit(uint32_t v, int pos) { return ((int32_t)(v << (31 - pos))) >> 31; }

template <int N>
__device__ __forceinline__ void step_pair_body(uint4 *__restrict__ state, long bpad, long batch,
                                               const int32_t *__restrict__ actions, int32_t *__restrict__ actions_out,
                                               uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
                                               int jpos, uint32_t *__restrict__ ticket, TicketBases tbase, int nstatic) {
  using G = twx::Geo<N>;
  constexpr int W = G::W, WH = (W + 1) / 2, CS = 2 * W, CJ = 2 * W + 1, CHT = G::CHT;
  constexpr int BPW = 32;                   // boards per wave
  constexpr int RS = BPW * 4, GS = 2 * RS;  // chunk-row / word-group stride of the LDS image, in dwords
  constexpr int NJ = (CHT + 1) / 2;         // chunks a lane moves: 2j + half
  constexpr int NG = (NJ + 7) / 8;          // per-lane offset registers (8 chunk pairs = 8 KiB each)
  constexpr bool kOddTail = (CHT & 1) != 0; // the upper lane's last chunk does not exist
#define plane_off(p) ((((p) >> 2) * RS) + ((p) & 3))
  constexpr int oR = plane_off(twx::kPlaneRed), oS = plane_off(twx::kPlaneStart), oE = plane_off(twx::kPlaneEnd);
  constexpr uint32_t kColMask = (1u << N) - 1u;
  static_assert(CHT == 2 * W + 1 + twx::kJournalChunks && twx::kJournalChunks == 2 && W <= 32, "layout");
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint4 *lds4 = reinterpret_cast<uint4 *>(lds);

  const int lane = threadIdx.x;
  const int h = lane >> 5;   // which lane of the pair: 0 lower, 1 upper
  const int bl = lane & 31;  // board slot inside the batch
  const bool up = h != 0;
  const bool compact = (flags & kFlagCompact) != 0;
  const bool flush = (flags & kFlagFlush) != 0;
  const bool random = (flags & kFlagRandom) != 0;
  const long nbatch = bpad / BPW;  // batches of 32 boards = tiles of the state array

  // ---- batch order: as step_kernel64 (static rounds, then tickets) ----
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
    return (long)nstatic * gridDim.x + (long)k * kTicketGroups + tgroup;
  };
  // batch i -> its 32 boards = one tile of the state array (tile_shift 5: CHT rows of 512 B)
  static_assert(tile_shift(CHT) == 5, "step_pair_kernel works on 32-board tiles");
  auto batch_of = [&](long i) -> long { return (flags & kFlagReverse) ? reversed_batch(i, nbatch) : i; };
  auto tile_of = [&](long i) -> char * { return reinterpret_cast<char *>(state) + batch_of(i) * (long)(CHT * 512); };

  // a journal row (slots 0..3: chunk row CJ, the upper lane's half of pair W; slots 4..7: the odd
  // tail) is read only when a slot in it holds a record of this period; nobody ever clears the
  // rows: slots from jpos on hold stale records of older periods and are never looked at
  const bool need_j1 = jpos > 0, need_j2 = jpos > 4;
  // ---- tile I/O (twixt_device_common.h): lane moves chunks 2j + h of board bl ----
  u32x4 nxt[NJ];
  uint32_t voff[NG];
#pragma unroll
  for (int g = 0; g < NG; g++) {
    voff[g] = (uint32_t)bl * 16u + (uint32_t)h * 512u + (uint32_t)g * 8192u + 4096u;
    asm volatile("" : "+v"(voff[g]));
  }
  // the last chunk of an odd count belongs to the lower lane (the upper one's would be chunk CHT)
  uint32_t voff_tail = (uint32_t)bl * 16u + (uint32_t)((NJ - 1) / 8) * 8192u + 4096u;
  asm volatile("" : "+v"(voff_tail));
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const uint32_t lds_lane = (uint32_t)(uintptr_t)((lds_u32 *)lds) + (uint32_t)h * 512u + (uint32_t)bl * 16u;
  auto load_slice = [&](auto c0t, auto c1t, const char *tile) {
    constexpr int c0 = decltype(c0t)::value, c1 = decltype(c1t)::value;
    static_assert(kOddTail && NJ == W + 2, "pairs 0..W-1: planes; pair W: scalars | journal row 0; tail: journal row 1");
    constexpr int c1r = (c1 == NJ) ? NJ - 2 : c1;  // plane pairs of this slice
    static_assert(c0 <= c1r, "the journal pairs belong to the last slice");
    pair_tile_load<c0, c1r>(nxt, voff, tile);
    if constexpr (c1 == NJ) {
      // (a journal row is four 128-byte lines, one per slot; lanes 8s..8s+7 of a half move slot s:
      // only the slots that hold records of this period are fetched)
      if (!up || (bl >> 3) < jpos) pair_tile_load_chunk<NJ - 2>(nxt[NJ - 2], voff[(NJ - 2) / 8], tile);
      if (!up && (bl >> 3) < jpos - 4) pair_tile_load_chunk<NJ - 1>(nxt[NJ - 1], voff_tail, tile);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using INJ = std::integral_constant<int, NJ>;
  constexpr int kPf1 = NJ * TWX_PF_A / 12, kPf2 = NJ * TWX_PF_B / 12;
  using IP1 = std::integral_constant<int, kPf1>;
  using IP2 = std::integral_constant<int, kPf2>;

  long idx = blockIdx.x;  // the launcher guarantees gridDim.x <= nbatch
  long round = 0;
  uint32_t tv = (dyn && nstatic == 1) ? draw() : 0u;
  load_slice(I0{}, INJ{}, tile_of(idx));
  vm_wait_all();  // first batch: nothing to overlap with
  long idx_next = (dyn && nstatic == 1) ? ticket_batch(tv) : idx + gridDim.x;
  // NOTHING the compiler tracks in its vmcnt bookkeeping (a load, an atomic with a result) may be
  // consumed between the issue of the prefetch and vm_wait_all(): the compiler does not see the
  // prefetch loads, so its own `s_waitcnt vmcnt(0)` for such a result would wait for the whole
  // prefetch in the middle of the step (it did, for the ticket and for the external action, until
  // round 2: the step then ran at the memory LATENCY, not its bandwidth).  So the ticket of a
  // batch is drawn right after the wait of the batch before and read after the next wait, and
  // the external action is loaded and waited for before the prefetch is issued.
  bool drawn = dyn && round + 2 >= nstatic;  // is the batch after next a ticket?
  if (drawn) tv = draw();

  // loop-invariant per-lane constants of the split
  const uint32_t hm = up ? 0xffffffffu : 0u;
  const int link_lo = up ? RS : 2;        // dword offset of this lane's two link planes inside a word group
  const int wsh = up ? 16 : 0;            // their byte positions in a crossing window
  const int sgn = up ? 1 : -1;            // side of the knight neighbours this lane looks at
  const int dflip = up ? 0 : 7;           // direction numbering: lower lane d = 7 - (upper lane's d)
  // links of a move this lane writes: d = 2h, 2h+1 (west end = the new peg) and 4+2h, 5+2h (west end =
  // the target, dlt cells away); their planes are the two at link_lo
  const int own_sh = 11 + 2 * h;          // record bit of link 2h
  const int dlt0 = up ? twx::kDX[6] * N + twx::kDY[6] : twx::kDX[4] * N + twx::kDY[4];
  const int dlt1 = up ? twx::kDX[7] * N + twx::kDY[7] : twx::kDX[5] * N + twx::kDY[5];
  const int offA = up ? oS : oR;          // first flag / peg word of a record's own cell
  const uint32_t peg_sel = up ? 0u : 1u;  // (the lower lane adds the player: red / blue words are adjacent)
  uint32_t *const me = lds + (bl << 2);   // this board's slot inside every chunk row
  uint32_t *const meh = me + (up ? WH * GS : 0);  // first word group of this lane's half
  uint32_t *const sc = me + CS * RS;      // scalars
  const uint32_t *const jrow = lds + CJ * RS + bl;  // journal record s of this board: jrow[s * BPW] ([slot][board])

  STAMP_DECL;  // (diagnostic -DTWX_STAMPS build only: cycles per phase, tools/stamps.py)
  while (idx < nbatch) {
    STAMP(0);  // loop end / store issue of the previous batch
    const long row = batch_of(idx) * BPW + bl;  // = board: whole-batch launches only
    const bool inb = row < batch;
    int a_ext = -1;
    if (!random && !flush) {  // wave-uniform: external actions (twx_step)
      if (inb) a_ext = actions[row];
      asm volatile("" : "+v"(a_ext));  // the compiler's wait for this load belongs HERE, before the prefetch
    }
    pair_tile_fill<0, NJ - 2>(lds_lane, nxt);
    if (!up || need_j1) pair_tile_fill<NJ - 2, NJ - 1>(lds_lane, nxt);  // scalars | journal row 0
    if (need_j2) pair_tile_fill<NJ - 1, NJ>(lds_lane, nxt);             // journal row 1 (the upper lanes' slot is spare)
    wave_sync();
    const char *const tile_next = tile_of((idx_next < nbatch) ? idx_next : idx);
    __builtin_amdgcn_sched_barrier(0);
    load_slice(I0{}, IP1{}, tile_next);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);  // LDS fill + first prefetch slice

    if constexpr ((kAblate & 2048) != 0) {  // diagnostic 2048: nothing but the board stream
      load_slice(IP1{}, IP2{}, tile_next);
      load_slice(IP2{}, INJ{}, tile_next);
      vm_wait_all();
      const long idx_after0 = (long)blockIdx.x + (round + 2) * gridDim.x;
      idx = idx_next; idx_next = idx_after0; round++;
      continue;
    }
    const uint4 sv = *reinterpret_cast<const uint4 *>(sc);
    uint32_t counter = sv.x & 0xffffu, move_one = sv.x >> 16;
    uint32_t result = sv.y & 0xffu, swapped = (sv.y >> 8) & 1u;
    uint32_t episode = sv.z;
    // The stored scalars are as old as the planes: advance them by the journal records first
    // (slots 0..jpos-1, in order) -- the reset decision below needs the TRUE result.
    uint32_t jbase = (sv.y >> twx::kJBaseShift) & 15u;  // first journal slot that belongs to this board's game
    uint32_t jr[twx::kJournalSlots];
#pragma unroll
    for (int s = 0; s < twx::kJournalSlots; s++) {
      const uint32_t e = (s < jpos) ? jrow[s * BPW] : 0u;  // (wave-uniform; rows that were not loaded are not read)
      jr[s] = ((uint32_t)s >= jbase) ? e : 0u;
    }
#pragma unroll
    for (int s = 0; s < twx::kJournalSlots - 1; s++)
      if (s < jpos) twx::journal_scalars(jr[s], counter, move_one, result, swapped);
    const bool was_reset = random && (flags & kFlagAutoReset) && result != 0u && inb;
    if (__ballot(was_reset)) {  // rare: one board in ~520 per step
      if (was_reset) {
        // ---- auto-reset: Board::Board (tb.cc:168-174, 209-276) in place; each lane its half ----
#pragma unroll
        for (int ii = 0; ii < WH; ii++) {
          const bool has_u = WH + ii < W;  // (W odd: the upper lane has one word group less)
          const int iu = has_u ? WH + ii : 0;
          const uint32_t s_l = G::tab.init[twx::kPlaneStart * W + ii], s_u = G::tab.init[twx::kPlaneStart * W + iu];
          const uint32_t e_l = G::tab.init[twx::kPlaneEnd * W + ii], e_u = G::tab.init[twx::kPlaneEnd * W + iu];
          if (has_u || !up) {
            *reinterpret_cast<uint4 *>(meh + ii * GS) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(meh + ii * GS + RS) =
                make_uint4(0u, 0u, s_l ^ ((s_l ^ s_u) & hm), e_l ^ ((e_l ^ e_u) & hm));
          }
        }
#pragma unroll
        for (int s = 0; s < twx::kJournalSlots; s++) jr[s] = 0u;  // nothing to replay on the empty board
        jbase = (uint32_t)jpos;  // the records before this launch's slot belong to the finished game
        counter = 0; move_one = 0; result = 0; swapped = 0; episode += 1;
      }
      wave_sync();
    }
    // ---- replay the redo journal: slots 0..jpos-1 in order.  Work of one record, split over the
    //      pair: lower lane the peg, links 0, 1 (west end = this cell) and 4, 5 (west end = the
    //      target), the single-cell fill's START; upper lane START / END gained, links 2, 3 and
    //      6, 7, the fill's END.  Both lanes' link planes are the two at link_lo, and every
    //      condition is a sign-extended record bit ANDed into the value: no selects.
    //      (a reset board reads empty records) ----
#pragma unroll
    for (int s = 0; s < twx::kJournalSlots - 1; s++) {
      if (s < jpos && !(kAblate & 4)) {  // wave-uniform
        const uint32_t e = jr[s];  // (move_one is final already: it is set by move 0 and a swap is move 1)
        const int cell = (int)(e & 1023u);
        const uint32_t bit = (1u << (cell & 31)) & (uint32_t)((int32_t)e >> 31);  // empty record: every OR is a no-op
        uint32_t *const g = me + (cell >> 5) * GS;
        if (__ballot((e & twx::kJSwap) != 0u))  // rare: undo red's first peg on a swap (tb.cc:450-455)
          lds_and(me + (move_one >> 5) * GS + oR, (e & twx::kJSwap) ? ~(1u << (move_one & 31)) : 0xffffffffu);
        // lower: the peg (red / blue words are adjacent); upper: START gained
        lds_or(g + offA + (int)((e >> 10) & peg_sel), bit & ((uint32_t)sbit(e, 19) | ~hm));
        lds_or(g + oE, bit & (uint32_t)sbit(e, 20) & hm);  // upper: END gained
        const uint32_t eo = e >> own_sh;  // bits 0, 1: this lane's links at the cell; bits 4, 5: at the targets
        lds_or2(g + link_lo, bit & (uint32_t)sbit(eo, 0), bit & (uint32_t)sbit(eo, 1));
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const uint32_t on = (uint32_t)sbit(eo, 4 + j);
          const int tc = cell + (int)((uint32_t)(j ? dlt1 : dlt0) & on);
          lds_or(me + (tc >> 5) * GS + link_lo + j, (1u << (tc & 31)) & on);
        }
        if (__ballot((e & (twx::kJFillStart | twx::kJFillEnd)) != 0u)) {
          // the single neighbour a fill flagged: lower lane START, upper lane END
          const int fd = (int)((e >> twx::kJFillDirShift) & 7u);
          const int t = cell + ((int)((twx::kDxPack >> (4 * fd)) & 15u) - 2) * N + (int)((twx::kDyPack >> (4 * fd)) & 15u) - 2;
          const uint32_t on = (uint32_t)sbit(e >> 27, h);
          const int tt = on ? t : cell;  // (no fill: t is meaningless and may lie outside the board)
          lds_or(me + (tt >> 5) * GS + (up ? oE : oS), (1u << (tt & 31)) & on);
        }
      }
    }
    wave_sync();
    STAMP(2);  // scalars, reset, journal replay
    __builtin_amdgcn_sched_barrier(0);
    load_slice(IP1{}, IP2{}, tile_next);
    __builtin_amdgcn_sched_barrier(0);

    const int player = (int)(counter & 1u);
    const bool open = (result == 0u);

    // ---- legal mask of the player to move (tw.h:86-90), this lane's half of the
    //      word groups, and the opponent's free-cell count for the draw test (tb.cc:202-206).  The
    //      opponent cannot be out of cells while fewer pegs are on the board than its region has
    //      cells: the count is taken only when some board of the wave is that full (wave-uniform) ----
    uint32_t m[WH];
    const bool count_opp = __ballot(counter + 1u >= (uint32_t)(N * (N - 2))) != 0ull;
    int n_opp = 0, n_mine = 0;
#pragma unroll
    for (int ii = 0; ii < WH; ii++) {
      const uint2 rb = *reinterpret_cast<const uint2 *>(meh + ii * GS);
      const uint32_t occ = rb.x | rb.y;
      // region words of this lane's half (a constant pair: two operations where they differ;
      // W odd: the upper lane's last group does not exist and reads as empty)
      const bool has_u = WH + ii < W;
      const uint32_t r0l = G::tab.region[0][ii], r0u = has_u ? G::tab.region[0][has_u ? WH + ii : 0] : 0u;
      const uint32_t r1l = G::tab.region[1][ii], r1u = has_u ? G::tab.region[1][has_u ? WH + ii : 0] : 0u;
      const uint32_t reg_r = r0l ^ ((r0l ^ r0u) & hm), reg_b = r1l ^ ((r1l ^ r1u) & hm);
      // (move 1, blue to move: red's first peg stays legal for blue -- the swap)
      m[ii] = (player ? reg_b : reg_r) & ~((counter == 1u) ? rb.y : occ);
      n_mine += __popc(m[ii]);
      if (count_opp) n_opp += __popc((player ? reg_r : reg_b) & ~occ);
    }
    const PairVal pc = pair_swap((uint32_t)n_mine | ((uint32_t)n_opp << 10));
    const int cnt_lo = (int)(pc.lo & 1023u), cnt = cnt_lo + (int)(pc.hi & 1023u);
    const int cnt_opp = count_opp ? (int)((pc.lo >> 10) + (pc.hi >> 10)) : (1 << 20);  // "plenty": no draw

    // ---- choose / validate the action ----
    int a = -1;
    bool act = inb && open && !flush;
    bool illegal = false;
    if (random) {
      act = act && cnt > 0;
      const uint32_t hsh = twx::policy_hash(seed, first_id + (uint32_t)row, episode, counter);
      const int k = (int)__umulhi(hsh, (uint32_t)cnt);
      const bool lower_has_it = k < cnt_lo;
      const int kk = up ? k - cnt_lo : k;  // rank inside this lane's half (meaningful in the lane that has it)
      int cum = 0, wsel = 0, base = 0;
      uint32_t msel = m[0];
#pragma unroll
      for (int ii = 0; ii < WH; ii++) {  // last word whose prefix count is <= kk
        const bool ge = kk >= cum;
        wsel = ge ? ii : wsel;
        msel = ge ? m[ii] : msel;
        base = ge ? cum : base;
        cum += __popc(m[ii]);
      }
      const int a_half = ((up ? WH : 0) + wsel) * 32 + select_bit(msel, kk - base);
      const PairVal pa2 = pair_swap((uint32_t)a_half);
      a = (int)(lower_has_it ? pa2.lo : pa2.hi);
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
    STAMP(3);  // legal mask + select
    __builtin_amdgcn_sched_barrier(0);
    load_slice(IP2{}, INJ{}, tile_next);
    __builtin_amdgcn_sched_barrier(0);
    if (kAblate & 8) act = false;

    int pa = 0;          // cell the peg finally lands on
    uint32_t pend = 0;   // bit0: flood START from pa, bit1: flood END from pa
    uint32_t entry = 0;  // journal record of this move
    uint32_t fill1 = 0;  // single-cell fill of the shortcut below, as record bits
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
      uint32_t *const gp = me + (pa >> 5) * GS;
      const uint32_t pbit = 1u << (pa & 31);
      lds_or(gp + oR + player, pbit);  // tb.cc:507-508 set peg (both lanes: the same bit)
      // border flags the empty cell already carries (tb.cc:223-231)
      const uint2 own_se = *reinterpret_cast<const uint2 *>(gp + oS);
      const bool own_s = (own_se.x & pbit) != 0u, own_e = (own_se.y & pbit) != 0u;

      // ---- SetPegAndLinks (tb.cc:501-571).  Crossing window: board columns x-3..x+1, rows
      //      y-3..y+3 of this lane's two link planes, at their byte positions ----
      uint32_t win[twx::kWinCols];
#pragma unroll
      for (int cc = 0; cc < twx::kWinCols; cc++) {
        const int c = x + cc - 3;
        const bool ok = (unsigned)c < (unsigned)N;
        const int o = (ok ? c : 0) * N, sh = o & 31;
        const uint32_t *const g = me + (o >> 5) * GS + link_lo;  // next word group: +GS (in bounds: scalars / journal follow)
        const uint2 w0 = *reinterpret_cast<const uint2 *>(g), w1 = *reinterpret_cast<const uint2 *>(g + GS);
        const uint32_t p0 = funnel(w0.x, w1.x, sh) & kColMask, p1 = funnel(w0.y, w1.y, sh) & kColMask;
        const uint32_t v = (((p0 << 3) >> y) & 0x7fu) | ((((p1 << 3) >> y) & 0x7fu) << 8);
        win[cc] = ok ? (v << wsh) : 0u;
      }
      // knight neighbours on this lane's side: column x +- 1 carries rows y +- 2, column x +- 2 rows
      // y +- 1.  Bits in the UPPER lane's direction numbers (0 NNE, 1 ENE, 2 ESE, 3 SSE); the
      // lower lane's neighbours are the mirror images 7 - d (7 NNW, 6 WNW, 5 WSW, 4 SSW)
      uint32_t cand = 0, ts = 0, te = 0;
#pragma unroll
      for (int cj = 0; cj < 2; cj++) {
        const int c = x + sgn * (cj + 1);
        const bool ok = (unsigned)c < (unsigned)N;
        const int o = (ok ? c : 0) * N, sh = o & 31;
        const uint32_t *const g = me + (o >> 5) * GS;
        const uint32_t own = ok ? (funnel(g[oR + player], g[oR + player + GS], sh) & kColMask) : 0u;
        const uint2 f0 = *reinterpret_cast<const uint2 *>(g + oS), f1 = *reinterpret_cast<const uint2 *>(g + oS + GS);
        const uint32_t fs = ok ? (funnel(f0.x, f1.x, sh) & kColMask) : 0u, fe = ok ? (funnel(f0.y, f1.y, sh) & kColMask) : 0u;
        const int dy = 2 - cj;                       // rows y + dy and y - dy
        const int dup = cj == 0 ? 0 : 1, ddn = cj == 0 ? 3 : 2;  // their directions (upper lane's numbers)
        const uint32_t wo = (own << 2) >> y, wsf = (fs << 2) >> y, wef = (fe << 2) >> y;  // bit 2 + dy <-> row y + dy
        cand |= (((wo >> (2 + dy)) & 1u) << (dup ^ dflip)) | (((wo >> (2 - dy)) & 1u) << (ddn ^ dflip));
        ts |= (((wsf >> (2 + dy)) & 1u) << (dup ^ dflip)) | (((wsf >> (2 - dy)) & 1u) << (ddn ^ dflip));
        te |= (((wef >> (2 + dy)) & 1u) << (dup ^ dflip)) | (((wef >> (2 - dy)) & 1u) << (ddn ^ dflip));
      }
      uint32_t blocked = 0;
#pragma unroll
      for (int d = 0; d < 8; d++) {  // crossing test (tb.cc:519-527) on this lane's planes: constant masks (twixt_geom.h)
        uint32_t hit = 0;
#pragma unroll
        for (int i = 0; i < twx::kWinCols; i++) hit |= win[i] & twx::kCross.m[d][i];
        blocked |= (hit != 0u ? 1u : 0u) << d;
      }
      const PairVal pq = pair_swap(cand | (ts << 8) | (te << 16) | (blocked << 24));
      const uint32_t all = pq.lo | pq.hi;
      cand = all & 0xffu; ts = (all >> 8) & 0xffu; te = (all >> 16) & 0xffu; blocked = all >> 24;

      const uint32_t linkm = cand & ~blocked;
      // tb.cc:538-546: inherit START, else END, else remember a neutral neighbour
      const uint32_t gain_s = linkm & ts, gain_e = linkm & ~ts & te, neutral = linkm & ~ts & ~te;
      const bool p_s = own_s || gain_s;
      const bool p_e = own_e || gain_e;
      uint32_t want = (linkm && neutral) ? ((p_s ? 1u : 0u) | (p_e ? 2u : 0u)) : 0u;  // tb.cc:558-570
      // Half of all fills flag exactly one cell: the new peg's only unflagged neighbour is a peg
      // without any other link.  Settled here on the pre-move planes (both lanes alike).
      if (want && !(neutral & (neutral - 1u)) && (!(want & 1u) || (linkm & ~ts) == neutral) &&
          (!(want & 2u) || (linkm & ~te) == neutral)) {
        const int d = __ffs((int)neutral) - 1;
        const int tx = x + (int)((twx::kDxPack >> (4 * d)) & 15u) - 2, ty = y + (int)((twx::kDyPack >> (4 * d)) & 15u) - 2;
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
          lds_or(gt + (up ? oE : oS), (want & (up ? 2u : 1u)) ? tbit : 0u);  // lower: START, upper: END
          // persisted by this move's journal record (or by the whole-board write of this launch)
          fill1 = ((uint32_t)d << twx::kJFillDirShift) | ((want & 1u) ? twx::kJFillStart : 0u) |
                  ((want & 2u) ? twx::kJFillEnd : 0u);
          want = 0u;
        }
      }
      pend = want;
      {  // tb.cc:532-533: one stored bit per physical link, at its west end; lower lane links 0, 1
         // (west end = the new peg) and 4, 5 (west end = the target), upper lane 2, 3 and 6, 7
        const uint32_t lo = linkm >> (2 * h);
        lds_or2(gp + link_lo, pbit & (uint32_t)sbit(lo, 0), pbit & (uint32_t)sbit(lo, 1));
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const uint32_t on = (uint32_t)sbit(lo, 4 + j);
          const int tc = pa + (int)((uint32_t)(j ? dlt1 : dlt0) & on);
          lds_or(me + (tc >> 5) * GS + link_lo + j, (1u << (tc & 31)) & on);
        }
        lds_or(gp + (up ? oE : oS), (up ? gain_e : gain_s) ? pbit : 0u);
        if (counter == 0u) move_one = (uint32_t)pa;  // tb.cc:485-488
        counter += 1u;
        // ---- UpdateResult (tb.cc:192-207) ----
        if (p_s && p_e) {
          result = player ? 2u : 1u;
        } else if (counter > 2u) {
          const bool in_opp = player ? (x >= 1 && x <= N - 2) : (y >= 1 && y <= N - 2);
          if (cnt_opp - (in_opp ? 1 : 0) == 0) result = 3u;
        }
        entry = twx::kJValid | (uint32_t)pa | ((uint32_t)player << 10) | (linkm << 11) |
                (gain_s ? twx::kJGainStart : 0u) | (gain_e ? twx::kJGainEnd : 0u) | (swapnow ? twx::kJSwap : 0u) |
                (result << twx::kJResultShift) | fill1;
      }
    }
    STAMP(5);  // apply
    // scalars (s3 = the board's dirty mask during the flood below) and this move's record
    // (a board that was reset in a journalling launch is written whole with the scalars of the
    // EMPTY board and its journal base: its first move is in the record, and stored scalars +
    // records from the base on = the truth; a launch that rewrites everything stores base 0)
    *reinterpret_cast<uint4 *>(sc) =
        (was_reset && !compact)
            ? make_uint4(0u, jbase << twx::kJBaseShift, episode, 0u)
            : make_uint4(counter | (move_one << 16), result | (swapped << 8) | (compact ? 0u : jbase << twx::kJBaseShift),
                         episode, 0u);
    wave_sync();

    // ---- ExploreLocalGraph (tb.cc:573-588), wave-cooperative, two boards per pass: lanes 0..31 on
    //      one board that needs a fill, lanes 32..63 on another; lane c holds column c of the four
    //      link planes and of the flag plane (step_kernel64's pass) ----
    uint64_t todo = __ballot(pend != 0u) & 0xffffffffull;  // the upper lanes mirror the lower ones
    if (kAblate & 1) todo = 0ull;
    while (todo) {
      const int s0 = (int)__ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const bool two = todo != 0ull;
      const int s1 = two ? (int)__ffsll((unsigned long long)todo) - 1 : s0;
      if (two) todo &= todo - 1;
      const int c = bl;  // column this lane holds
      const int src = up ? s1 : s0;
      const int spa = up ? __builtin_amdgcn_readlane(pa, s1) : __builtin_amdgcn_readlane(pa, s0);
      const uint32_t spend = (up && !two) ? 0u
                             : (uint32_t)(up ? __builtin_amdgcn_readlane((int)pend, s1)
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
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const bool en = (spend >> f) & 1u;  // uniform per half
        if (!__ballot(en)) continue;
        const int off = f ? oE : oS;
        const uint32_t flag0 = col(off);
        uint32_t flag = flag0;
        uint32_t front = (en && c == sx) ? (1u << sy) : 0u;
        for (;;) {
          uint32_t fresh = 0u;
#pragma unroll
          for (int rep = 0; rep < 2; rep++) {  // two expansions per exit test (1..4 measured alike)
            // forward: a frontier cell with link e reaches (c+dx, y+dy); backward: cell (c,y) whose
            // link e ends in the frontier.  Columns move by DPP wave shifts, rows by bit shifts.
            const uint32_t a1 = ((front & lk[0]) << 2) | ((front & lk[3]) >> 2);
            const uint32_t a2 = ((front & lk[1]) << 1) | ((front & lk[2]) >> 1);
            uint32_t nb = wave_shr1(a1) | wave_shr1(wave_shr1(a2));
            const uint32_t f1 = wave_shl1(front), f2 = wave_shl1(f1);
            nb |= ((f1 >> 2) & lk[0]) | ((f2 >> 1) & lk[1]) | ((f2 << 1) & lk[2]) | ((f1 << 2) & lk[3]);
            // (lane 31 / 32 are neighbours for the DPP shifts but belong to different boards: column
            // 31 does not exist (N <= 24 < 31), so nothing crosses between the halves)
            fresh = ok ? (nb & ~flag & kColMask) : 0u;
            flag |= fresh;
            front = fresh;
          }
          if (__ballot(fresh != 0u) == 0ull) break;
        }
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
      }
    }
    wave_sync();
    const uint32_t dirty = sc[3];
    sc[3] = 0u;
    wave_sync();

    STAMP(6);  // flood fill
    // The prefetched batch has had the whole step to arrive: retire its loads here, before this
    // batch's stores are issued (loads and stores share one in-order counter).
    vm_wait_all();
    STAMP(7);  // wait for the prefetched batch
    const long idx_after = drawn ? ticket_batch(tv) : (long)blockIdx.x + (round + 2) * gridDim.x;
    // the ticket of the batch after the next one, if the next batch exists (the host counts on it)
    drawn = dyn && round + 3 >= nstatic && idx_next < nbatch;
    if (drawn) tv = draw();

    // ---- stream back: lane moves chunks 2j + h ----
    char *const tile_cur = tile_of(idx);
    auto chunk_ptr = [&](int c) -> uint4 * {
      return reinterpret_cast<uint4 *>(tile_cur + ((uint32_t)bl * 16u + (uint32_t)c * 512u));
    };
    const bool wr = !(kAblate & 16);
    if (compact) {  // planes and scalars from LDS; the journal rows stay as they are (stale from now on)
#pragma unroll
      for (int j = 0; j < NJ; j++)
        if (wr && 2 * j + h < CJ) tile_store<true>(chunk_ptr(2 * j + h), lds4[(2 * j + h) * BPW + bl]);
    } else {
      // this move's record: all a step normally writes -- one dword per board, one 128-byte line
      // per tile (the record carries the scalars)
      if (wr && !up)
        tile_store_word<true>(reinterpret_cast<uint32_t *>(tile_cur + CJ * 512 + jpos * (BPW * 4)) + bl, entry);
      STAMP(9);  // ticket + journal record store
      // flag chunks a flood fill changed (taking the first one along in the idle upper lanes of the
      // journal store -- then a 16-byte chunk per board -- measured 1.9 % SLOWER: scattered lanes make
      // that instruction expensive)
      uint32_t dd = (was_reset || !wr || (kAblate & 2)) ? 0u : dirty;
      // ... every dirty board walks its own mask, two chunks per trip (lower lane the lowest dirty
      // group, upper lane the next one)
      while (__ballot(dd != 0u)) {
        const uint32_t d1 = dd & (dd - 1u);            // without the lowest
        const uint32_t mine = up ? d1 : dd;
        if (mine) {
          const int c = 2 * (__ffs((int)mine) - 1) + 1;  // chunk 2i+1 holds the flag words of group i
          if (!(kAblate & 64)) tile_store<true>(chunk_ptr(c), lds4[c * BPW + bl]);
          else { const uint32_t keep = lds[(c * BPW + bl) * 4]; asm volatile("" ::"v"(keep)); }
        }
        dd = d1 & (d1 - 1u);
      }
      STAMP(10);  // dirty flag chunks
      if (__ballot(was_reset)) {  // auto-reset boards are rewritten whole
#pragma unroll
        for (int j = 0; j < NJ; j++)
          if (was_reset && wr && 2 * j + h < CJ) tile_store<true>(chunk_ptr(2 * j + h), lds4[(2 * j + h) * BPW + bl]);
      }
    }
    STAMP(11);  // reset boards / compaction
    if (inb && !flush && !up) {
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
    idx = idx_next;
    idx_next = idx_after;
    round++;
  }  // batch loop
  STAMP_FLUSH;
}
#undef plane_off

template <int N>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void step_pair_kernel(
    uint4 *__restrict__ state, long bpad, long batch, const int32_t *__restrict__ actions,
    int32_t *__restrict__ actions_out, uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
    int jpos, uint32_t *__restrict__ ticket, TicketBases tbase, int nstatic) {
  step_pair_body<N>(state, bpad, batch, actions, actions_out, status, seed, first_id, flags, jpos, ticket, tbase, nstatic);
}

#endif  // TWIXT_PAIR_KERNEL_H_
