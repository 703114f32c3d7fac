// twixt_window_kernel.h -- the big boards (N >= 17) on the SPLIT layout (twixt_device_common.h): geometry of the
// window of a move (WinGeo), the inline-asm load / LDS-write helpers of the windowed kernels, and
// step_window_full_kernel<N>: whole tiles rewritten in place (the steps of batches too small to fill the chip).
// Part of libtwixt_hip.so; included by twixt_engine.hip only (device code, anonymous namespace).
#ifndef TWIXT_WINDOW_KERNEL_H_
#define TWIXT_WINDOW_KERNEL_H_

// The windowed step.  A step reads every peg of its board (legal mask and k-th legal action, tw.h:86-90) but of
// the links and border flags -- six of the eight planes, 432 of 592 bytes at N=24 -- only the crossing window
// (board columns x-3..x+1, tb.cc:519-527) and the knight neighbours' flags (columns x-2..x+2, tb.cc:538-546).
// On the split layout a launch streams the rows of a tile (pegs, scalars, journal), chooses the move and only
// then fetches the links + flags of the six board columns around it: one contiguous run of the board's
// board-major region, about 2.1 lines of 128 B, instead of the region's 3.4 (tools/window_probe3.hip: the bare
// memory skeleton runs in 0.086 ms per 2^20 boards against 0.113 ms for whole tiles; rounds 1 and 2 had measured
// windowed steps as no faster -- their windows were gathers of 16-byte pieces out of [chunk][board] rows, a
// line per piece, and the probe that was to settle it read half a tile by accident:
// profiles/r03_window_probe3.txt).
//   * journal launches (7 of 8 at the default period): step_window1_kernel, twixt_window1_kernel.h;
//   * launches that rewrite the boards: step_window_full_kernel below -- the whole tile in LDS (rows and the 32
//     regions: 19.5 KiB at N=24, eight waves per CU), two lanes per board as step_pair_kernel (the pair splits
//     the legal-mask scan, the link planes -- lower lane NNE / ENE and the neighbours to the west, upper lane
//     ESE / SSE and the neighbours to the east -- and the journal replay), records replayed everywhere, the move
//     applied in LDS, flood fills run in place, pegs, scalars and regions written back.
// Same journal, records, batch order (static rounds + tickets, XCD-preserving ping-pong) and results as
// step_pair_kernel on the chunk layout.
//
// Reference lines: Board::ApplyAction tb.cc:457-499, SetPegAndLinks tb.cc:501-571, ExploreLocalGraph
// tb.cc:573-588, UpdateResult tb.cc:192-207, LegalActions / DoApplyAction tw.h:86-104, Board::Board tb.cc:168-276.

template <int N>
struct WinGeo {
  using G = twx::Geo<N>;
  static constexpr int W = G::W;
  static constexpr int RP = split_peg_rows(W);     // peg rows: row r = words 2r, 2r+1 of red and blue
  static constexpr int RSC = RP;                   // the scalars row
  static constexpr int RJ = RP + 1;                // journal rows RJ (slots 0..3), RJ + 1 (slots 4..7)
  static constexpr int RA = split_rows(W);         // rows of a tile
  static constexpr int LFB = split_lf_bytes(W);    // bytes of a board's links + flags region
  static constexpr int LFP = LFB / 16;             // ... in 16-byte pieces
  static constexpr long TILE = split_tile_bytes(W);
  // window: board columns x-3 .. x+2 = 6N bits from a bit offset < 32: NW words of every plane
  static constexpr int NW = win_words(N, W);
  // ... fetched as NPC 16-byte pieces from piece (24 w0) >> 4 on (twixt_device_common.h win_piece0 / win_range)
  static constexpr int NPC = win_pieces(N, W);
  static constexpr int WSTR = NPC | 1;             // LDS stride of a board's window, pieces (odd: banks; the extra piece is padding)
  // LDS of step_window_full_kernel (bytes): the rows, then the 32 whole regions as in the tile
  static constexpr int L_ROWS = RA * 512;
  static constexpr int L_ALL = 32 * LFB;
  static constexpr int L_SLACK = 32;               // (a column read may touch the word after the last one)
  static constexpr int lds_bytes_full = L_ROWS + L_ALL + L_SLACK;
  static_assert(LFB % 16 == 0 && NW >= 1 && NPC <= LFP, "window shape");
};

// ---- 16 bytes: global -> registers -> LDS, as inline asm (twixt_device_common.h "Tile I/O": the compiler does
//      not see these loads; the kernels wait for them explicitly, and every load with a scalar base carries the
//      wait states of the VALU-writes-SGPR hazard: TWX_SGPR_HAZARD_NOP) ----
// step_window_full_kernel keeps its prefetched tile in AGPRs ("a") ...
__device__ __forceinline__ void gload16(u32x4 &dst, uint32_t voff, const void *base) {
  asm volatile(TWX_SGPR_HAZARD_NOP "global_load_dwordx4 %0, %1, %2" : "=a"(dst) : "v"(voff), "s"(base) : "memory");
}
template <int IMM>
__device__ __forceinline__ void gload16i(u32x4 &dst, uint32_t voff, const void *base) {
  asm volatile(TWX_SGPR_HAZARD_NOP "global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
__device__ __forceinline__ void lwrite16(uint32_t lds_addr, const u32x4 &v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "a"(v) : "memory");
}
template <int IMM>
__device__ __forceinline__ void lwrite16i(uint32_t lds_addr, const u32x4 &v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lds_addr), "a"(v), "n"(IMM) : "memory");
}
// ... step_window1_kernel has no AGPR operand at all, so that its whole budget of 256 registers is VGPRs ("v")
template <int IMM>
__device__ __forceinline__ void gload16iv(u32x4 &dst, uint32_t voff, const void *base) {
  asm volatile(TWX_SGPR_HAZARD_NOP "global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
// ... for the lanes whose `bit` is not zero only (the others keep what the register holds): exec is narrowed inside
// the statement, so the instruction is always issued and there is no branch around it
template <int IMM>
__device__ __forceinline__ void gload16iv_if(u32x4 &dst, uint32_t voff, const void *base, uint32_t bit) {
  uint64_t saved;
  asm volatile("v_cmp_ne_u32 vcc, 0, %3\n\t"
               "s_and_saveexec_b64 %1, vcc\n\t" TWX_SGPR_HAZARD_NOP "global_load_dwordx4 %0, %2, %4 offset:%5\n\t"
               "s_mov_b64 exec, %1"
               : "+v"(dst), "=&s"(saved)
               : "v"(voff), "v"(bit), "s"(base), "n"(IMM)
               : "memory", "vcc");
}
__device__ __forceinline__ void gload16p(u32x4 &dst, const void *addr) {  // (per-lane 64-bit address)
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lwrite16v(uint32_t lds_addr, const u32x4 &v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}
template <int IMM>
__device__ __forceinline__ void lwrite16iv(uint32_t lds_addr, const u32x4 &v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lds_addr), "v"(v), "n"(IMM) : "memory");
}
// a wave-uniform value the compiler could not prove uniform (a batch index that went around the pipelined
// loop): back into scalar registers, where the loop's branches and the loads' base want it
__device__ __forceinline__ long uniform_long(long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
  return (long)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ char *uniform_ptr(const char *p) {
  return reinterpret_cast<char *>((uintptr_t)uniform_long((long)(uintptr_t)p));
}
// v[LANE] = a wave-uniform value (v_writelane_b32)
template <int LANE>
__device__ __forceinline__ void set_lane(uint32_t &v, int uniform_value) {
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(uniform_value), "n"(LANE));
}

template <int N>
__device__ __forceinline__ void step_window_full_body(char *__restrict__ state, long bpad, long batch,
                                                 const int32_t *__restrict__ actions, int32_t *__restrict__ actions_out,
                                                 uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
                                                 uint32_t *__restrict__ ticket, TicketBases tbase, int nstatic) {
  using G = twx::Geo<N>;
  using WG = WinGeo<N>;
  constexpr int W = WG::W, RP = WG::RP, RSC = WG::RSC, LFB = WG::LFB, LFP = WG::LFP;
  constexpr int BPW = 32;               // boards per wave
  constexpr int NJA = (RSC + 2) / 2;    // row pairs: lane half h moves rows 2j + h of the pegs and scalars (rows 0..RSC)
  constexpr int RPH = (RP + 1) / 2;     // peg rows the lower lane scans (the upper one: the rest)
  constexpr int NLF = (32 * LFP + 63) / 64;  // instructions that move the tile's links + flags regions
  constexpr uint32_t kColMask = (1u << N) - 1u;
  static_assert(NJA <= 8 && twx::kJournalChunks == 2 && W <= 32, "layout");
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint4 *lds4 = reinterpret_cast<uint4 *>(lds);
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_u32 *)lds);  // LDS byte address of the image

  const int lane = threadIdx.x;
  const int h = lane >> 5;   // which lane of the pair: 0 lower, 1 upper
  const int bl = lane & 31;  // board slot inside the batch
  const bool up = h != 0;
  const bool flush = (flags & kFlagFlush) != 0;
  const bool random = (flags & kFlagRandom) != 0;
  const long nbatch = bpad / BPW;  // batches of 32 boards = tiles of the state array

  // ---- batch order: as step_pair_kernel (static rounds, then tickets; backward launches keep the XCD) ----
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
  auto batch_of = [&](long i) -> long { return (flags & kFlagReverse) ? reversed_batch(i, nbatch) : i; };
  auto tile_of = [&](long i) -> char * { return state + batch_of(i) * WG::TILE; };

  // ---- LDS map (dwords) ----
  constexpr int D_REG = WG::L_ROWS / 4;                      // the 32 whole regions, as in the tile
  uint32_t *const rowme = lds + (bl << 2);                   // this board's 16 bytes of row r: rowme + 128 r
  uint32_t *const sc = rowme + RSC * 128;                    // scalars
  // peg word i of player p: rowme[(i >> 1) * 128 + (i & 1) * 2 + p]
  auto pegw = [&](int i) -> uint32_t * { return rowme + (i >> 1) * 128 + (i & 1) * 2; };

  // ---- prefetch of a batch's rows: lane half h moves rows 2j + h (pegs and scalars; the journal rows are
  //      neither read nor written: these launches work on materialised boards) ----
  u32x4 rowv[NJA];
  uint32_t voff_rows = (uint32_t)bl * 16u + (uint32_t)h * 512u + 4096u;
  asm volatile("" : "+v"(voff_rows));
  auto load_rows = [&](const char *tile) {
    static_for<0, NJA>([&](auto jt) {
      constexpr int J = decltype(jt)::value;
      if (2 * J + h <= RSC) gload16i<J * 1024 - 4096>(rowv[J], voff_rows, tile);
    });
  };
  const uint32_t lds_rows = lds0 + (uint32_t)h * 512u + (uint32_t)bl * 16u;
  auto fill_rows = [&]() {
    static_for<0, NJA>([&](auto jt) {
      constexpr int J = decltype(jt)::value;
      if (2 * J + h <= RSC) lwrite16i<J * 1024>(lds_rows, rowv[J]);
    });
  };
  // ---- the tile's 32 links + flags regions, contiguous in HBM and in LDS alike ----
  u32x4 lfv[NLF];
  auto load_regions = [&](const char *tile) {
#pragma unroll
    for (int k = 0; k < NLF; k++)
      if (k * 64 + lane < 32 * LFP) gload16(lfv[k], (uint32_t)(WG::L_ROWS + (k * 64 + lane) * 16), tile);
  };
  auto fill_regions = [&]() {
#pragma unroll
    for (int k = 0; k < NLF; k++)
      if (k * 64 + lane < 32 * LFP) lwrite16(lds0 + (uint32_t)(WG::L_ROWS + (k * 64 + lane) * 16), lfv[k]);
  };

  long idx = blockIdx.x;  // the launcher guarantees gridDim.x <= nbatch
  long round = 0;
  uint32_t tv = (dyn && nstatic == 1) ? draw() : 0u;
  load_rows(tile_of(idx));
  load_regions(tile_of(idx));
  vm_wait_all();  // first batch: nothing to overlap with
  long idx_next = (dyn && nstatic == 1) ? ticket_batch(tv) : idx + gridDim.x;
  // (nothing the compiler tracks in vmcnt may be consumed between the issue of an untracked load and its
  // wait: see step_pair_body.  The ticket of a batch is drawn right after a full wait and read after the next.)
  bool drawn = dyn && round + 2 >= nstatic;  // is the batch after next a ticket?
  if (drawn) tv = draw();

  // loop-invariant per-lane constants of the split
  const uint32_t hm = up ? 0xffffffffu : 0u;
  const int link_lo = up ? 2 : 0;           // this lane's two link planes inside a word's six dwords
  const int wsh = up ? 16 : 0;              // their byte positions in a crossing window
  const int sgn = up ? 1 : -1;              // side of the knight neighbours this lane looks at
  const int dflip = up ? 0 : 7;             // direction numbering: lower lane d = 7 - (upper lane's d)
  const int dlt0 = up ? twx::kDX[6] * N + twx::kDY[6] : twx::kDX[4] * N + twx::kDY[4];
  const int dlt1 = up ? twx::kDX[7] * N + twx::kDY[7] : twx::kDX[5] * N + twx::kDY[5];

  while (idx < nbatch) {
    const long row = batch_of(idx) * BPW + bl;  // = board: whole-batch launches only
    const bool inb = row < batch;
    const char *const tile_cur_c = tile_of(idx);
    int a_ext = -1;
    if (!random && !flush) {  // wave-uniform: external actions (twx_step)
      if (inb) a_ext = actions[row];
      asm volatile("" : "+v"(a_ext));  // the compiler's wait for this load belongs HERE, before any untracked load
    }
    fill_rows();
    fill_regions();
    wave_sync();
    const char *const tile_next = tile_of((idx_next < nbatch) ? idx_next : idx);
    __builtin_amdgcn_sched_barrier(0);  // the whole next tile behind the step, as step_pair_kernel
    load_rows(tile_next);
    load_regions(tile_next);
    __builtin_amdgcn_sched_barrier(0);

    const uint4 sv = *reinterpret_cast<const uint4 *>(sc);
    uint32_t counter = sv.x & 0xffffu, move_one = sv.x >> 16;
    uint32_t result = sv.y & 0xffu, swapped = (sv.y >> 8) & 1u;
    uint32_t episode = sv.z;
    // (whole-tile launches work on materialised boards: the stored scalars are current, there is no record to
    // replay, and what they write back carries journal base zero)
    const bool was_reset = random && (flags & kFlagAutoReset) && result != 0u && inb;
    if (__ballot(was_reset)) {  // rare: one board in ~520 per step
      if (was_reset) {
        // ---- auto-reset: Board::Board (tb.cc:168-174, 209-276) in the LDS copy, each lane its half ----
#pragma unroll
        for (int r = 0; r < RP; r++)
          if ((r & 1) == h) *reinterpret_cast<uint4 *>(rowme + r * 128) = make_uint4(0u, 0u, 0u, 0u);
        uint32_t *const fbme = lds + D_REG + bl * (LFB / 4);
#pragma unroll
        for (int i = 0; i < W; i++)
          if ((i & 1) == h) {
            *reinterpret_cast<uint2 *>(fbme + 6 * i) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2 *>(fbme + 6 * i + 2) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2 *>(fbme + 6 * i + 4) =
                make_uint2(G::tab.init[twx::kPlaneStart * W + i], G::tab.init[twx::kPlaneEnd * W + i]);
          }
        counter = 0; move_one = 0; result = 0; swapped = 0; episode += 1;
      }
      wave_sync();
    }
    uint32_t *const wq = lds + D_REG + bl * (LFB / 4);  // word w of this board's links + flags: wq + 6 w

    const int player = (int)(counter & 1u);
    const bool open = (result == 0u);

    // ---- legal mask of the player to move (tw.h:86-90): the lower lane scans peg rows
    //      0..RPH-1, the upper one the rest, two words per row; and the opponent's free-cell count for the
    //      draw test (tb.cc:202-206), only when some board of the wave is nearly full (wave-uniform) ----
    uint32_t m[2 * RPH];
    const bool count_opp = __ballot(counter + 1u >= (uint32_t)(N * (N - 2))) != 0ull;
    int n_opp = 0, n_mine = 0;
#pragma unroll
    for (int rr = 0; rr < RPH; rr++) {
      const bool has_u = RPH + rr < RP;  // (the upper lane may have a row less)
      const uint4 rb = *reinterpret_cast<const uint4 *>(rowme + ((up && has_u) ? RPH + rr : rr) * 128);
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int wl = 2 * rr + k, wu = 2 * (RPH + rr) + k;  // word of the lower / upper lane
        const uint32_t r0l = wl < W ? G::tab.region[0][wl < W ? wl : 0] : 0u, r0u = (has_u && wu < W) ? G::tab.region[0][wu < W ? wu : 0] : 0u;
        const uint32_t r1l = wl < W ? G::tab.region[1][wl < W ? wl : 0] : 0u, r1u = (has_u && wu < W) ? G::tab.region[1][wu < W ? wu : 0] : 0u;
        const uint32_t reg_r = r0l ^ ((r0l ^ r0u) & hm), reg_b = r1l ^ ((r1l ^ r1u) & hm);
        const uint32_t red = k ? rb.z : rb.x, blue = k ? rb.w : rb.y, occ = red | blue;
        // (move 1, blue to move: red's first peg stays legal for blue -- the swap)
        m[2 * rr + k] = (player ? reg_b : reg_r) & ~((counter == 1u) ? blue : occ);
        n_mine += __popc(m[2 * rr + k]);
        if (count_opp) n_opp += __popc((player ? reg_r : reg_b) & ~occ);
      }
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
      for (int ii = 0; ii < 2 * RPH; ii++) {  // last word whose prefix count is <= kk
        const bool ge = kk >= cum;
        wsel = ge ? ii : wsel;
        msel = ge ? m[ii] : msel;
        base = ge ? cum : base;
        cum += __popc(m[ii]);
      }
      const int a_half = ((up ? 2 * RPH : 0) + wsel) * 32 + select_bit(msel, kk - base);
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
        const uint2 rb = *reinterpret_cast<const uint2 *>(pegw(a >> 5));
        const bool occ = ((rb.y & bit) != 0u) || (counter != 1u && (rb.x & bit) != 0u);
        legal = inreg && !occ;
      }
      illegal = inb && !(open && legal);  // tw.h:95-97 (terminal: LegalActions() is empty)
      act = act && legal;
    }

    // ---- where the peg lands (tb.cc:462-473: a swap rotates it) ----
    int pa = act ? a : 0;
    const bool swapnow = act && (counter == 1u) && ((uint32_t)a == move_one);
    if (swapnow) {
      const int ax = a / N, ay = a - ax * N;
      pa = ay * N + (N - 1 - ax);
    }
    const int x = pa / N, y = pa - x * N;
    if (kAblate & 8) act = false;

    uint32_t pend = 0;   // bit0: flood START from pa, bit1: flood END from pa
    if (act) {
      // ---- Board::ApplyAction (tb.cc:457-499) ----
      uint32_t *const gp = wq + 6 * (pa >> 5);
      const uint32_t pbit = 1u << (pa & 31);
      if (swapnow) swapped = 1u;
      if (swapnow) lds_and(pegw(a >> 5), ~(1u << (a & 31)));  // tb.cc:450-455: undo red's first peg
      if (!up) lds_or(pegw(pa >> 5) + player, pbit);          // tb.cc:507-508 set peg
      // border flags the empty cell already carries (tb.cc:223-231)
      const uint2 own_se = *reinterpret_cast<const uint2 *>(gp + 4);
      const bool own_s = (own_se.x & pbit) != 0u, own_e = (own_se.y & pbit) != 0u;

      // ---- SetPegAndLinks (tb.cc:501-571).  Crossing window: board columns x-3..x+1, rows
      //      y-3..y+3 of this lane's two link planes, at their byte positions ----
      uint32_t win[twx::kWinCols];
#pragma unroll
      for (int cc = 0; cc < twx::kWinCols; cc++) {
        const int c = x + cc - 3;
        const bool ok = (unsigned)c < (unsigned)N;
        const int o = (ok ? c : x) * N, sh = o & 31;
        const uint32_t *const g = wq + 6 * (o >> 5) + link_lo;  // next word: + 6 dwords (in bounds: slack follows)
        const uint2 w0v = *reinterpret_cast<const uint2 *>(g), w1v = *reinterpret_cast<const uint2 *>(g + 6);
        const uint32_t p0 = funnel(w0v.x, w1v.x, sh) & kColMask, p1 = funnel(w0v.y, w1v.y, sh) & kColMask;
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
        const int o = (ok ? c : x) * N, sh = o & 31, wi = o >> 5;
        const uint32_t own = ok ? (funnel(pegw(wi)[player], pegw(wi + 1 < W ? wi + 1 : wi)[player], sh) & kColMask) : 0u;
        const uint32_t *const g = wq + 6 * wi + 4;
        const uint2 f0 = *reinterpret_cast<const uint2 *>(g), f1 = *reinterpret_cast<const uint2 *>(g + 6);
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
        const uint32_t *const gt = wq + 6 * (t >> 5);  // links that START at t
        uint32_t other = (gt[0] | gt[1] | gt[2] | gt[3]) >> (t & 31);
#pragma unroll
        for (int e = 0; e < 4; e++) {  // links that END at t: stored at t - off[e]
          const int sx = tx - twx::kDX[e], sy = ty - twx::kDY[e];
          const bool valid = sx >= 0 && sy >= 0 && sy < N;
          const int sc2 = valid ? sx * N + sy : pa;
          const uint32_t w = wq[6 * (sc2 >> 5) + e];
          other |= valid ? (w >> (sc2 & 31)) : 0u;
        }
        if (!(other & 1u)) {
          lds_or(wq + 6 * (t >> 5) + (up ? 5 : 4), (want & (up ? 2u : 1u)) ? (1u << (t & 31)) : 0u);  // lower: START, upper: END
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
          lds_or(wq + 6 * (tc >> 5) + link_lo + j, (1u << (tc & 31)) & on);
        }
        lds_or(gp + (up ? 5 : 4), (up ? gain_e : gain_s) ? pbit : 0u);
      }
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
    // the scalars as they stand after the move (everything is rewritten: journal base 0)
    if (!up)
      *reinterpret_cast<uint4 *>(sc) = make_uint4(counter | (move_one << 16), result | (swapped << 8), episode, 0u);
    wave_sync();

    // ---- ExploreLocalGraph (tb.cc:573-588), wave-cooperative, two boards per pass: lanes 0..31 on one
    //      board that needs a fill, lanes 32..63 on another; lane c holds column c of the four link
    //      planes and of the flag plane ----
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
      uint32_t *const fb = lds + D_REG + src * (LFB / 4);  // the board's whole links + flags region, in place
      const int sx = spa / N, sy = spa - sx * N;
      const bool ok = c < N;
      const int o = (ok ? c : 0) * N, sh = o & 31, grp = o >> 5;
      uint32_t *const g = fb + 6 * grp;  // this lane's column starts in word grp
      auto col = [&](int off) -> uint32_t {
        const uint32_t v = funnel(g[off], g[off + 6], sh) & kColMask;
        return ok ? v : 0u;
      };
      uint32_t lk[4];
#pragma unroll
      for (int e = 0; e < 4; e++) lk[e] = col(e);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const bool en = (spend >> f) & 1u;  // uniform per half
        if (!__ballot(en)) continue;
        const int off = 4 + f;
        const uint32_t flag0 = col(off);
        uint32_t flag = flag0;
        uint32_t front = (en && c == sx) ? (1u << sy) : 0u;
        for (;;) {
          uint32_t fresh = 0u;
#pragma unroll
          for (int rep = 0; rep < 2; rep++) {  // two expansions per exit test
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
          lds_or(g + off, add << sh);
          if (sh + N > 32) lds_or(g + off + 6, add >> (32 - sh));
        }
        wave_sync();
      }
    }
    wave_sync();

    // The prefetched batch has had the rest of the step to arrive: retire its loads here, before this
    // batch's stores are issued (loads and stores share one in-order counter).
    vm_wait_all();
    const long idx_after = drawn ? ticket_batch(tv) : (long)blockIdx.x + (round + 2) * gridDim.x;
    // the ticket of the batch after the next one, if the next batch exists (the host counts on it)
    drawn = dyn && round + 3 >= nstatic && idx_next < nbatch;
    if (drawn) tv = draw();

    // ---- stream back ----
    char *const tile_cur = const_cast<char *>(tile_cur_c);
    const bool wr = !(kAblate & 16);
    // pegs and scalars (rows 0..RP) and every links + flags region, from LDS; the journal rows stay as
    // they are (stale from now on)
#pragma unroll
    for (int j = 0; j < NJA; j++)
      if (wr && 2 * j + h <= RSC)
        tile_store<true>(reinterpret_cast<uint4 *>(tile_cur + (2 * j + h) * 512 + bl * 16), lds4[(2 * j + h) * 32 + bl]);
#pragma unroll
    for (int k = 0; k < NLF; k++)
      if (wr && k * 64 + lane < 32 * LFP)
        tile_store<true>(reinterpret_cast<uint4 *>(tile_cur + WG::L_ROWS + (k * 64 + lane) * 16), lds4[D_REG / 4 + k * 64 + lane]);
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
    idx = idx_next;
    idx_next = idx_after;
    round++;
  }  // batch loop
}

// Whole tiles rewritten in place: the whole-batch step of batches too small to fill the chip (a search's leaves:
// twixt_engine.hip run_step).  The whole tile in LDS (19.5 KiB at N=24), eight waves per CU.  Such launches work
// on materialised boards (run_step flushes a live journal first; the regions in HBM are current at any time, so
// records must never be replayed on them): the journal position is zero by construction.
template <int N>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void step_window_full_kernel(
    char *__restrict__ state, long bpad, long batch, const int32_t *__restrict__ actions,
    int32_t *__restrict__ actions_out, uint8_t *__restrict__ status, uint32_t seed, uint32_t first_id, int flags,
    int jpos, uint32_t *__restrict__ ticket, TicketBases tbase, int nstatic) {
  (void)jpos;
  step_window_full_body<N>(state, bpad, batch, actions, actions_out, status, seed, first_id, flags, ticket, tbase, nstatic);
}

#endif  // TWIXT_WINDOW_KERNEL_H_
