// twixt_step_kernel.h -- step_body<N,SUB> and its __global__ wrappers: the fused env step, one lane per board
// Part of libtwixt_hip.so; included by twixt_engine.hip only (device code, anonymous namespace).
#ifndef TWIXT_STEP_KERNEL_H_
#define TWIXT_STEP_KERNEL_H_

// One launch = one env step on every board (or, with kFlagFlush, just the
// materialisation of the redo journal).
//
// Log-structured state: between two compactions a launch does not
// rewrite the planes; it appends one 32-bit record of the move to the board's
// journal (slot `jpos`, the same for every board) and writes back only that
// record: it carries the scalars and the single-cell fills (twixt_geom.h).  The
// next launches replay slots 0..jpos-1 on top of the planes they stream in.  Every
// `period`-th launch (kFlagCompact) writes planes and scalars and starts a new
// period.  Lanes that auto-reset write their whole board; lanes whose multi-cell
// flood fill changed flag words write those chunks.
// Diagnostic builds only (-DTWX_ABLATE=mask, tools/README.md): leave a phase out to measure what
// it really costs inside the overlapped loop (results are then wrong by construction).
//   1 flood fill   2 dirty-chunk stores   4 journal replay (the planes' part; the scalars still advance)
//   8 apply   16 every store   32 LDS fill   64 (pair kernel) dirty-chunk loop runs but stores nothing
//   2048 (pair kernel) nothing but the board stream: loads and the wait for them
#ifndef TWX_ABLATE
#define TWX_ABLATE 0
#endif
constexpr int kAblate = TWX_ABLATE;

// SPLIT: the handle's state array is in the split layout (twixt_device_common.h; big boards whose
// whole-batch step is step_window_kernel): this kernel then serves their listed-boards and play-out
// launches only -- always on materialised planes, always rewriting its boards -- and gathers / scatters
// a board's logical chunks piecewise (8-byte pieces) into the same LDS image.
template <int W>
__device__ __forceinline__ u32x4 split_load_chunk(const uint4 *state, long b, int c) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(state);
  if (c >= 2 * W) return *reinterpret_cast<const u32x4 *>(w + split_scalar_index(0, b, W));
  const int i = c >> 1;
  const uint2 lo = *reinterpret_cast<const uint2 *>(w + split_word_index((c & 1) ? 4 : 0, i, b, W));
  const uint2 hi = *reinterpret_cast<const uint2 *>(w + split_word_index((c & 1) ? 6 : 2, i, b, W));
  return u32x4{lo.x, lo.y, hi.x, hi.y};
}
template <int W>
__device__ __forceinline__ void split_store_chunk(uint4 *state, long b, int c, uint4 v) {
  uint32_t *w = reinterpret_cast<uint32_t *>(state);
  if (c >= 2 * W) {
    *reinterpret_cast<uint4 *>(w + split_scalar_index(0, b, W)) = v;
    return;
  }
  const int i = c >> 1;
  *reinterpret_cast<uint2 *>(w + split_word_index((c & 1) ? 4 : 0, i, b, W)) = make_uint2(v.x, v.y);
  *reinterpret_cast<uint2 *>(w + split_word_index((c & 1) ? 6 : 2, i, b, W)) = make_uint2(v.z, v.w);
}

template <int N, bool SUB, bool SPLIT = false>
__device__ __forceinline__ void step_body(uint4 *__restrict__ state, long bpad, long batch,
                                                     const int32_t *__restrict__ actions,
                                                     int32_t *__restrict__ actions_out,
                                                     uint8_t *__restrict__ status, uint32_t seed,
                                                     uint32_t first_id, int flags, int jpos_arg,
                                                     const int32_t *__restrict__ ids_arg, long nsub,
                                                     int rsteps, int rpw, uint32_t *__restrict__ ticket,
                                                     TicketBases tbase, int nstatic) {
  using G = twx::Geo<N>;
  const int32_t *const ids = SUB ? ids_arg : nullptr;  // compile-time null in the whole-batch kernel
  constexpr int W = G::W, CS = 2 * W, CJ = 2 * W + 1, CHT = G::CHT;
  constexpr int BPW = kWave;                // boards per wave: one per lane
  constexpr int RS = BPW * 4, GS = 2 * RS;  // chunk-row / word-group stride of the LDS image in dwords
  constexpr int NJ = CHT;                   // 16-byte registers a lane prefetches
#define plane_off(p) ((((p) >> 2) * RS) + ((p) & 3))
  constexpr int oR = plane_off(twx::kPlaneRed), oS = plane_off(twx::kPlaneStart), oE = plane_off(twx::kPlaneEnd);
  constexpr uint32_t kColMask = (1u << N) - 1u;
  static_assert(oE == oS + 1 && (oS & 1) == 0 && (plane_off(twx::kPlaneLink0) & 1) == 0 && (plane_off(twx::kPlaneLink0 + 2) & 1) == 0,
                "lds_or2 pairs: adjacent, 8-byte aligned planes");
  static_assert(CHT == 2 * W + 1 + twx::kJournalChunks && W <= 32, "layout");
  // whole-batch launches on the tiled layout address a batch as one contiguous tile and keep
  // the prefetched batch in AGPRs (big boards: one wave per SIMD owns all 512 registers)
  constexpr int TS = tile_shift(CHT);         // boards per tile of the state array: 64, or 32 for N >= 19
  constexpr int RB = 16 << TS;                // bytes of a chunk row of a tile
  constexpr bool FAST = !SUB && !SPLIT;
  constexpr bool AG = FAST && CHT > TWX_SMALL_CHT;
  constexpr bool NT = CHT > TWX_PLAIN_STORE_CHT;  // non-temporal write-back where the board stream is the bound
  constexpr int NG = (CHT * RB + 8191) / 8192;  // per-lane offset registers: one per 8 KiB of a tile
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint4 *lds4 = reinterpret_cast<uint4 *>(lds);

  const int lane = threadIdx.x;
  const int bl = lane;  // board slot of this lane inside the batch
  const bool compact_all = (flags & kFlagCompact) != 0;
  const bool flush = (flags & kFlagFlush) != 0;
  const bool random = (flags & kFlagRandom) != 0;
  // Rows: the boards this launch works on.  Whole batch: row j is board j (the allocation
  // is padded to a multiple of 64, padded boards are real storage).  Subset (`ids`): row j
  // is board ids[j]; lanes past nsub load board 0 and store nothing.
  // A listed-boards launch gives a wave only RPW rows (a power of two, the host's choice): a short
  // list spread over many waves finishes sooner than a few full ones, because the flood fills of a
  // wave's boards run one after the other (2,048 play-outs at N=12: 0.44 ms with 64 boards per
  // wave, 0.28 ms with 8).  Lanes RPW..63 are padding lanes then.
  const int RPW = SUB ? rpw : BPW;
  const long nbatch = ids ? (nsub + RPW - 1) / RPW : bpad / BPW;

  // Persistent wave: walks batches idx = blockIdx.x, +gridDim.x, ...  While batch k is
  // processed out of LDS, batch k+1 is already streaming into VGPRs (CHT x 16 B per
  // lane), so HBM latency hides behind the integer work even at one wave per SIMD.
  // kFlagReverse walks the batches backwards: the boards a launch touched last are the
  // ones the next launch touches first (Infinity Cache reuse).
  auto batch_row = [&](long idx) -> long {
    return ((flags & kFlagReverse) ? reversed_batch(idx, nbatch) : idx) * RPW + bl;
  };
  // (an id outside [0, batch) makes its row a padding lane: nothing is stored, status untouched)
  auto row_board = [&](long j) -> long { return ids ? ((bl < RPW && j < nsub) ? (long)ids[j] : -1L) : j; };
  auto batch_board = [&](long idx) -> long { return row_board(batch_row(idx)); };
  STAMP_DECL;
  u32x4 nxt[NJ];
  const u32x4 *__restrict__ gstate = reinterpret_cast<const u32x4 *>(state);
  u32x4 *ldsv = reinterpret_cast<u32x4 *>(lds);
  long idx = blockIdx.x;  // the launcher guarantees gridDim.x <= nbatch
  // Batch order.  Static: idx, idx + gridDim.x, ...  Dynamic (kFlagDynamic): every wave takes
  // its first `nstatic` batches in that static order and then draws tickets from a device
  // counter, so waves whose boards needed long flood fills take fewer batches and the launch
  // does not end on its slowest wave (-4 % per launch at N=24, B=2^20).
  //   * One counter for the whole chip is too slow: same-address device-scope atomics retire
  //     at ~75 M/s on MI355X (0.229 ms per launch instead of 0.16 with a ticket per batch).
  //     So only the tail of a launch is dynamic, and the waves form kTicketGroups groups by
  //     blockIdx.x % kTicketGroups, each with its own counter: group g owns the dynamic
  //     batches with idx % kTicketGroups == g.
  //   * Counters are never reset.  Per launch, group g draws exactly (its dynamic batches +
  //     2 per wave) tickets -- the two draws a wave has in flight when it runs past the end --
  //     the host advances that group's base by as much, and ticket t is the group's dynamic
  //     batch number t - base.
  //   * A ticket is drawn two batches ahead of its use, together with the prefetch of the
  //     batch before it, so its latency hides behind the step like the loads'.
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
  long round = 0;  // batches this wave has begun
  uint32_t tv = (dyn && nstatic == 1) ? draw() : 0u;
  // FAST: byte offset of this lane's 16 bytes inside its tile, + 8 KiB * g + 4 KiB (the loads'
  // immediates reach -4 KiB .. +4 KiB), and the first tile of batch idx (scalar; a batch is one
  // 64-board tile or two consecutive 32-board tiles: CHT KiB either way).  lds_lane: LDS byte
  // address of this lane's slot in chunk row 0.
  const uint32_t lane_off = (uint32_t)(lane >> TS) * (uint32_t)(CHT * RB) + (uint32_t)(lane & ((1 << TS) - 1)) * 16u;
  uint32_t voff[NG];
#pragma unroll
  for (int g = 0; g < NG; g++) {
    voff[g] = lane_off + (uint32_t)g * 8192u + 4096u;
    asm volatile("" : "+v"(voff[g]));  // stays a register: not re-derived per load
  }
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const uint32_t lds_lane = (uint32_t)(uintptr_t)((lds_u32 *)lds) + (uint32_t)lane * 16u;
  auto tile_of = [&](long i) -> char * {
    const long t = (flags & kFlagReverse) ? reversed_batch(i, nbatch) : i;
    return reinterpret_cast<char *>(state) + t * (long)(CHT * 1024);
  };
  // A journal row (chunk row CJ: slots 0..3, CJ + 1: slots 4..7) is read only when a slot in it
  // may hold a record of the running period (under rolling compaction a batch's position may be
  // anything).  Outside rolling compaction nobody clears the rows: the slots from jpos on hold stale
  // records of older periods and are never looked at.  Listed-boards launches (jpos = 0) skip them.
  const bool rolling = (flags & kFlagRolling) != 0;
  const bool need_j1 = rolling || jpos_arg > 0, need_j2 = rolling || jpos_arg > 4;
  // (a journal row of a tile is four lines, one per slot; this lane's 16 bytes of a row belong to slot
  // jslot of row 0 and 4 + jslot of row 1: only the slots that hold records of the period are fetched)
  const int jslot = (lane & ((1 << TS) - 1)) >> (TS - 2);
  const bool load_j1 = rolling || jslot < jpos_arg, load_j2 = rolling || jslot < jpos_arg - 4;
  static_assert(twx::kJournalChunks == 2, "two journal rows");
  constexpr int NL = (SUB || SPLIT) ? CJ : NJ;  // chunks a listed-boards (or split-layout) launch moves
  if constexpr (FAST) {
    tile_load<0, CJ, AG, TS>(nxt, voff, tile_of(idx));
    if (load_j1) tile_load<CJ, CJ + 1, AG, TS>(nxt, voff, tile_of(idx));
    if (load_j2) tile_load<CJ + 1, CHT, AG, TS>(nxt, voff, tile_of(idx));
    vm_wait_all();  // first batch: nothing to overlap with
  } else {
    long b0 = batch_board(idx);
    if (ids && (unsigned long)b0 >= (unsigned long)batch) b0 = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      if constexpr (SPLIT) nxt[j] = split_load_chunk<W>(state, b0, j);
      else nxt[j] = gstate[chunk_index(j, b0, TS, CHT)];
    }
#pragma unroll
    for (int j = 0; j < NL; j++) asm volatile("" ::"v"(nxt[j]));  // first batch: nothing to overlap with
  }
  long idx_next = (dyn && nstatic == 1) ? ticket_batch(tv) : idx + gridDim.x;
  // Nothing the compiler tracks in its vmcnt bookkeeping may be consumed between the issue of the
  // (asm, untracked) prefetch and the wait for it: see step_pair_body.  The ticket of a batch is
  // drawn right after the wait of the batch before and read after the next wait; the external
  // action is loaded and waited for before the prefetch is issued.
  bool drawn = dyn && round + 2 >= nstatic;  // is the batch after next a ticket?
  if (drawn) tv = draw();
  while (idx < nbatch) {
  STAMP(0);  // loop overhead / previous stores issue
  // Rolling compaction (TWX_ROLLING=1, off by default): batch b is at journal position
  // (phase + b) mod period, so every launch compacts one batch in `period` instead of every
  // period-th launch compacting all of them.  Same traffic, spread evenly -- and 10 % slower
  // on MI355X than the burst (0.173 vs 0.158 ms per launch): kept as a tested option only.
  int jpos = jpos_arg;
  bool compact = compact_all;
  if (rolling) {
    const int period = (flags >> 8) & 15;
    const long tb = (flags & kFlagReverse) ? reversed_batch(idx, nbatch) : idx;
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
  const bool wr = (!ids || inb) && !(kAblate & 16);  // subset mode: padding lanes must not store
  int a_ext = -1;
  if (!random && !flush) {  // wave-uniform: external actions (twx_step, twx_step_boards)
    if (inb) a_ext = actions[row];
    asm volatile("" : "+v"(a_ext));  // the compiler's wait for this load belongs HERE, before the prefetch
  }
  if constexpr ((kAblate & 32) != 0) {
  } else if constexpr (FAST) {
    tile_fill<0, CJ, AG>(lds_lane, nxt);
    if (need_j1) tile_fill<CJ, CJ + 1, AG>(lds_lane, nxt);
    if (need_j2) tile_fill<CJ + 1, CHT, AG>(lds_lane, nxt);
  } else {
#pragma unroll
    for (int j = 0; j < NL; j++) ldsv[j * kWave + lane] = nxt[j];  // row j, slot bl
  }
  wave_sync();
  // prefetch of the next batch, issued in three slices between the phases below so
  // that the wave never sits behind a 39-deep burst in its memory-issue queue (the
  // last iteration harmlessly re-reads its own batch)
  long b1 = batch_board((idx_next < nbatch) ? idx_next : idx);
  if (ids && (unsigned long)b1 >= (unsigned long)batch) b1 = 0;
  const char *const tile_next = tile_of((idx_next < nbatch) ? idx_next : idx);
#ifndef TWX_PF_A
#define TWX_PF_A 4
#define TWX_PF_B 8
#endif
  constexpr int kPf1 = NJ * TWX_PF_A / 12, kPf2 = NJ * TWX_PF_B / 12;  // slice boundaries
  // slice 0: chunks [0, kPf1), slice 1: [kPf1, kPf2), slice 2: [kPf2, NJ)
  auto prefetch = [&](auto slice) {
    constexpr int sl = decltype(slice)::value;
    constexpr int c0 = sl == 0 ? 0 : sl == 1 ? kPf1 : kPf2, c1 = sl == 0 ? kPf1 : sl == 1 ? kPf2 : NJ;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FAST) {
      if constexpr (c1 == NJ) {
        static_assert(c0 <= CJ, "the journal rows belong to the last slice");
        tile_load<c0, CJ, AG, TS>(nxt, voff, tile_next);
        if (load_j1) tile_load<CJ, CJ + 1, AG, TS>(nxt, voff, tile_next);
        if (load_j2) tile_load<CJ + 1, CHT, AG, TS>(nxt, voff, tile_next);
      } else {
        tile_load<c0, c1, AG, TS>(nxt, voff, tile_next);
      }
    } else {
#pragma unroll
      for (int j = c0; j < (c1 < NL ? c1 : NL); j++) {
        if constexpr (SPLIT) nxt[j] = split_load_chunk<W>(state, b1, j);
        else nxt[j] = gstate[chunk_index(j, b1, TS, CHT)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using Slice0 = std::integral_constant<int, 0>;
  using Slice1 = std::integral_constant<int, 1>;
  using Slice2 = std::integral_constant<int, 2>;
  prefetch(Slice0{});

  STAMP(1);  // LDS fill + prefetch issue
#ifdef TWX_SKELETON
  prefetch(Slice1{});
  prefetch(Slice2{});
  const bool was_reset = false, terminal_on_entry = false; const uint32_t dirty = 0u, rec = 0u; uint32_t result = 0, swapped = 0; bool illegal = false, act = true; int a = 0;
  { uint32_t *const sk = lds + (bl << 2) + 2 * G::W * RS; sk[0] += 1u; }
#else
  uint32_t *const me = lds + (bl << 2);  // this board's slot inside every chunk row
  // word group i (cells 32i..32i+31) of this board: me + i * GS [+ plane_off(p)]
  uint32_t *const sc = me + CS * RS;     // scalars
  // journal record s of this board: [slot][board] dword rows inside the tile's journal area (a
  // 64-board batch of 32-board tiles has two areas); subset launches (jpos = 0) never read it
  // (the LDS image keeps the tiles of a batch side by side in every chunk row: slot s of the board in
  // tile slot q sits at dword (s >> 2) * RS + tile * 4 TB + (s & 3) * TB + q of the journal rows)
  constexpr int TB = 1 << TS;  // boards per tile
  const uint32_t *const jrow = lds + CJ * RS + (bl >> TS) * (4 * TB) + (bl & (TB - 1));
  auto jrec = [&](int s) -> uint32_t { return jrow[(s >> 2) * RS + (s & 3) * TB]; };

  const uint4 sv = *reinterpret_cast<const uint4 *>(sc);
  uint32_t counter = sv.x & 0xffffu, move_one = sv.x >> 16;
  uint32_t result = sv.y & 0xffu, swapped = (sv.y >> 8) & 1u;
  uint32_t episode = sv.z;
  bool was_reset = false;
  // The stored scalars are as old as the planes: advance them by the journal records first
  // (slots 0..jpos-1, in order) -- the reset decision below needs the TRUE result.
  uint32_t jbase = (sv.y >> twx::kJBaseShift) & 15u;  // first journal slot that belongs to this board's game
  uint32_t jr[twx::kJournalSlots];
#pragma unroll
  for (int s = 0; s < twx::kJournalSlots; s++) {
    const uint32_t e = (s < jpos) ? jrec(s) : 0u;  // (wave-uniform guard: slots from jpos on are never replayed)
    jr[s] = ((uint32_t)s >= jbase) ? e : 0u;
  }
#pragma unroll
  for (int s = 0; s < twx::kJournalSlots - 1; s++)
    if (s < jpos) twx::journal_scalars(jr[s], counter, move_one, result, swapped);

  if (random && (flags & kFlagAutoReset) && result != 0u && inb) {
    // ---- auto-reset: Board::Board (tb.cc:168-174, 209-276) in place ----
#pragma unroll
    for (int i = 0; i < W; i++) {
      *reinterpret_cast<uint4 *>(me + i * GS) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4 *>(me + i * GS + RS) =
          make_uint4(0u, 0u, G::tab.init[twx::kPlaneStart * W + i], G::tab.init[twx::kPlaneEnd * W + i]);
    }
    jbase = (uint32_t)jpos;  // the records before this launch's slot belong to the finished game
    counter = 0; move_one = 0; result = 0; swapped = 0; episode += 1;
    was_reset = true;
  } else {
    // ---- replay the redo journal: slots 0..jpos-1 in order, branch-free ORs ----
#pragma unroll
    for (int s = 0; s < twx::kJournalSlots - 1; s++) {
      if (s < jpos && !(kAblate & 4)) {  // wave-uniform
        const uint32_t e = jr[s];  // (move_one is final already: it is set by move 0 and a swap is move 1)
        const int cell = (int)(e & 1023u);
        const uint32_t bit = (e & twx::kJValid) ? (1u << (cell & 31)) : 0u;  // empty record: every OR is a no-op
        uint32_t *const g = me + (cell >> 5) * GS;
        if (e & (1u << 21)) lds_and(me + (move_one >> 5) * GS + oR, ~(1u << (move_one & 31)));  // swap: tb.cc:450-455
        lds_or(g + oR + ((e >> 10) & 1u), bit);  // peg: red / blue words are adjacent
        // links whose west end is this peg: planes 2, 3 and 4, 5 are adjacent dword pairs
        lds_or2(g + plane_off(twx::kPlaneLink0), (e & (1u << 11)) ? bit : 0u, (e & (1u << 12)) ? bit : 0u);
        lds_or2(g + plane_off(twx::kPlaneLink0 + 2), (e & (1u << 13)) ? bit : 0u, (e & (1u << 14)) ? bit : 0u);
#pragma unroll
        for (int d = 4; d < 8; d++) {  // links whose west end is the other peg
          const bool on = (e & (1u << (11 + d))) != 0u;
          const int tc = on ? cell + twx::kDX[d] * N + twx::kDY[d] : cell;
          lds_or(me + (tc >> 5) * GS + plane_off(twx::kPlaneLink0 + d - 4), on ? (1u << (tc & 31)) : 0u);
        }
        lds_or2(g + oS, (e & (1u << 19)) ? bit : 0u, (e & (1u << 20)) ? bit : 0u);  // START | END gained
        {  // the single neighbour a fill flagged (the record carries its direction)
          const int fd = (int)((e >> twx::kJFillDirShift) & 7u);
          const bool on = (e & (twx::kJFillStart | twx::kJFillEnd)) != 0u;
          const int t = on ? cell + ((int)((twx::kDxPack >> (4 * fd)) & 15u) - 2) * N + (int)((twx::kDyPack >> (4 * fd)) & 15u) - 2 : cell;
          uint32_t *const gt = me + (t >> 5) * GS;
          lds_or2(gt + oS, (e & twx::kJFillStart) ? (1u << (t & 31)) : 0u, (e & twx::kJFillEnd) ? (1u << (t & 31)) : 0u);
        }
      }
    }
  }
  wave_sync();

  STAMP(2);  // reset / journal replay
  const bool terminal_on_entry = result != 0u;  // (after the auto-reset, if any)
  int a = -1;
  bool act = false, illegal = false;
  uint32_t dirty = 0u;
  uint32_t rec = 0u;  // journal record of this launch's move (single-step launches journal)
  // rsteps == 1: one env step.  rsteps > 1 (twx_random_rollout): up to rsteps moves per
  // board with the state staying in LDS, until every board of the batch is terminal.
  for (int rs = 0; rs < rsteps; rs++) {
  if (rs == 0) prefetch(Slice1{});
  const int player = (int)(counter & 1u);
  const bool open = (result == 0u);

  // ---- legal mask of the player to move (tw.h:86-90) and the opponent's free-cell
  //      count for the draw test of tb.cc:202-206.  The opponent cannot be out of cells while fewer
  //      pegs are on the board than its region has cells, so the count is taken only when some board
  //      of the wave is that full (wave-uniform; never at N >= 9 in random play). ----
  uint32_t m[W];
  const bool count_opp = __ballot(counter + 1u >= (uint32_t)(N * (N - 2))) != 0ull;
  int cnt_opp = 1 << 20;  // "plenty": the draw test below cannot fire
  if (count_opp) {
    cnt_opp = 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
      const uint2 rb = *reinterpret_cast<const uint2 *>(me + i * GS);
      cnt_opp += __popc((player ? G::tab.region[0][i] : G::tab.region[1][i]) & ~(rb.x | rb.y));
    }
  }
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < W; i++) {
    const uint2 rb = *reinterpret_cast<const uint2 *>(me + i * GS);
    // (move 1, blue to move: red's first peg stays legal for blue -- the swap)
    const uint32_t occ = (counter == 1u) ? rb.y : (rb.x | rb.y);
    m[i] = (player ? G::tab.region[1][i] : G::tab.region[0][i]) & ~occ;
    cnt += __popc(m[i]);
  }

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
  if (rs == 0) prefetch(Slice2{});
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
        lds_or2(gt + oS, (want & 1u) ? tbit : 0u, (want & 2u) ? tbit : 0u);
        // persisted by this move's journal record (or by the whole-board write of this launch)
        fill1 = ((uint32_t)d << twx::kJFillDirShift) | ((want & 1u) ? twx::kJFillStart : 0u) |
                ((want & 2u) ? twx::kJFillEnd : 0u);
        want = 0u;
      }
    }
    pend = want;
    {  // tb.cc:532-533: one stored bit per physical link, at its west end
      uint32_t *const g = me + (pa >> 5) * GS;
      const uint32_t bit = 1u << (pa & 31);
      lds_or2(g + plane_off(twx::kPlaneLink0), (linkm & 1u) ? bit : 0u, (linkm & 2u) ? bit : 0u);
      lds_or2(g + plane_off(twx::kPlaneLink0 + 2), (linkm & 4u) ? bit : 0u, (linkm & 8u) ? bit : 0u);
#pragma unroll
      for (int d = 4; d < 8; d++) {
        const bool on = (linkm & (1u << d)) != 0u;
        const int tc = on ? pa + twx::kDX[d] * N + twx::kDY[d] : pa;
        lds_or(me + (tc >> 5) * GS + plane_off(twx::kPlaneLink0 + d - 4), on ? (1u << (tc & 31)) : 0u);
      }
      lds_or2(g + oS, gain_s ? bit : 0u, gain_e ? bit : 0u);
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
  STAMP(5);  // apply: windows, crossing test, links, result
  // s3 is used below as the board's dirty mask (bit i: flag words of group i changed)
  // (a board that was reset in a journalling launch is written whole with the scalars of the
  // EMPTY board and its journal base: its first move is in the record, and stored scalars +
  // records from the base on = the truth; a launch that rewrites everything stores base 0)
  *reinterpret_cast<uint4 *>(sc) =
      (was_reset && !compact)
          ? make_uint4(0u, jbase << twx::kJBaseShift, episode, 0u)
          : make_uint4(counter | (move_one << 16), result | (swapped << 8) | (compact ? 0u : jbase << twx::kJBaseShift),
                       episode, 0u);
  rec = entry;
  wave_sync();

  // ---- ExploreLocalGraph (tb.cc:573-588), wave-cooperative, two boards per pass:
  //      lanes 0..31 work on one board that needs a fill, lanes 32..63 on another;
  //      lane c (within its half) holds column c of the four link planes and of the
  //      flag plane.  Frontier expansion: columns move by DPP wave shifts (1 VALU
  //      op each), rows by bit shifts; stop when no half found a new cell. ----
  uint64_t todo = __ballot(pend != 0u);
  if (kAblate & 1) todo = 0ull;
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
        uint32_t touched = 1u << grp;  // flag words this column changed
        lds_or(g + off, add << sh);
        if (sh + N > 32) {
          lds_or(g + off + GS, add >> (32 - sh));
          touched |= 2u << grp;
        }
        lds_or(lds + (src << 2) + CS * RS + 3, touched);
      }
      wave_sync();
      STAMP(14);  // flood: write-back
    }
  }
  wave_sync();
  STAMP(6);  // flood fill
  dirty |= sc[3];
  sc[3] = 0u;
  wave_sync();
  if (rs + 1 < rsteps && __ballot(inb && result == 0u) == 0ull) break;  // everyone finished
  }  // rollout steps

#endif  // TWX_SKELETON
  // The prefetched batch has had the whole step to arrive: retire its loads HERE, before
  // this batch's stores are issued, so that the wait at the top of the next iteration
  // does not also wait for those stores (loads and stores share one in-order counter).
  if constexpr (FAST) {
    vm_wait_all();
  } else {
#pragma unroll
    for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(nxt[j]));
  }
  // (the draw is older than the prefetch, so it has returned as well)
  const long idx_after = drawn ? ticket_batch(tv) : (long)blockIdx.x + (round + 2) * gridDim.x;
  // the ticket of the batch after the next one, if the next batch exists (the host counts on it)
  drawn = dyn && round + 3 >= nstatic && idx_next < nbatch;
  if (drawn) tv = draw();

  STAMP(7);  // wait for the prefetched batch
  // ---- stream back ----
  // (LDS row j, slot lane holds chunk j of board bl: the fill's mapping)
  char *const tile_cur = tile_of(idx);
  auto store_c = [&](int c, uint4 v) {  // chunk c of this lane's board
    if constexpr (SPLIT)
      split_store_chunk<W>(state, board, c, v);
    else if constexpr (FAST)
      tile_store<NT>(reinterpret_cast<uint4 *>(tile_cur + (lane_off + (uint32_t)c * (uint32_t)RB)), v);
    else
      tile_store<NT>(state + chunk_index(c, board, TS, CHT), v);
  };
  if (compact) {
    // planes and scalars from LDS.  The journal rows stay as they are (stale from now on); only
    // rolling compaction, where a reader cannot know a batch's position, clears them
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      if (wr && j < CJ) store_c(j, lds4[j * kWave + lane]);
      if (FAST && rolling && wr && j >= CJ) store_c(j, make_uint4(0u, 0u, 0u, 0u));
    }
  } else {
    // this move's record: all a step normally writes -- one dword per board, a contiguous line per
    // tile (the record carries the scalars).  Only whole-batch launches journal.
    if constexpr (FAST) {
      if (wr)
        tile_store_word<NT>(reinterpret_cast<uint32_t *>(tile_cur + (uint32_t)(lane >> TS) * (uint32_t)(CHT * RB) +
                                                         (uint32_t)(CJ * RB) + (uint32_t)jpos * (4u << TS)) + (lane & ((1 << TS) - 1)),
                            rec);
    }
    STAMP(9);  // fixed store (journal record)
    // flag chunks a flood fill changed: every dirty board walks its own mask, all dirty
    // boards in parallel (per-lane chunk index), so the loop runs max-popcount times
    uint32_t dd = (was_reset || !wr || (kAblate & 2)) ? 0u : dirty;
    while (__ballot(dd != 0u)) {
      if (dd) {
        const int c = 2 * (__ffs((int)dd) - 1) + 1;  // chunk 2i+1 holds the flag words of group i
        dd &= dd - 1;
        store_c(c, lds4[c * BPW + bl]);
      }
    }
    STAMP(10);  // dirty flag chunks
    if (__ballot(was_reset)) {  // auto-reset boards are rewritten whole
#pragma unroll
      for (int j = 0; j < NJ; j++)
        if (was_reset && j < CJ) store_c(j, lds4[j * kWave + lane]);
    }
  }
  STAMP(11);  // reset lanes / compaction
  if (inb && !flush) {
    if (status) {
      uint8_t st = (uint8_t)(result | (swapped << 5));
      if (illegal) st |= TWX_ST_ILLEGAL;
      if (was_reset) st |= TWX_ST_RESET;
      // a play-out (rsteps > 1) reports NOOP only for a board that was terminal when it began:
      // the byte must not depend on when the OTHER boards of the wave finished
      if (random && (rsteps > 1 ? terminal_on_entry : !act)) st |= TWX_ST_NOOP;
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

// 64 boards per wave, big boards (N >= 19: listed-boards / play-out launches, and whole-batch steps
// under TWX_PAIR=0): 39 KiB LDS image at N=24, 4 waves per CU, all 512 registers of the SIMD
template <int N, bool SUB, bool SPLIT = false>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 1))) void step_kernel64(
    uint4 *__restrict__ state, long bpad, long batch, const int32_t *__restrict__ actions,
    int32_t *__restrict__ actions_out, uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
    int jpos, const int32_t *__restrict__ ids, long nsub, int rsteps, int rpw, uint32_t *__restrict__ ticket,
    TicketBases tbase, int nstatic) {
  step_body<N, SUB, SPLIT>(state, bpad, batch, actions, actions_out, status, seed, first_id, flags, jpos, ids, nsub,
                           rsteps, rpw, ticket, tbase, nstatic);
}


// small boards (LDS image <= 25 KiB, N <= 18; TWX_SMALL_CHT): several waves per SIMD fit, so no
// one-wave-per-SIMD register budget; the compiler keeps it within 256 registers
#ifndef TWX_SMALL_MIN_WAVES
#define TWX_SMALL_MIN_WAVES 2
#endif
// ... except that the smallest boards (N <= 8: at most two words per plane) are asked to leave room
// for FOUR waves per SIMD (128 registers): their kernels sit right at that line -- 126..131 registers
// by themselves -- and a register more or less was worth 6 % (N=6: 0.0432 ms per 2^20-board step with
// four waves, 0.0461 with three).
constexpr int small_min_waves(int n) { return (n * n + 31) / 32 <= 2 && TWX_SMALL_MIN_WAVES < 4 ? 4 : TWX_SMALL_MIN_WAVES; }
template <int N, bool SUB, bool SPLIT = false>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(small_min_waves(N), 8))) void step_kernel64_small(
    uint4 *__restrict__ state, long bpad, long batch, const int32_t *__restrict__ actions,
    int32_t *__restrict__ actions_out, uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
    int jpos, const int32_t *__restrict__ ids, long nsub, int rsteps, int rpw, uint32_t *__restrict__ ticket,
    TicketBases tbase, int nstatic) {
  step_body<N, SUB, SPLIT>(state, bpad, batch, actions, actions_out, status, seed, first_id, flags, jpos, ids, nsub,
                           rsteps, rpw, ticket, tbase, nstatic);
}

#endif  // TWIXT_STEP_KERNEL_H_
