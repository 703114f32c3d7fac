// twixt_window_kernel.h -- step_window_kernel<N>: the windowed-step experiment (-DTWX_ENABLE_WINDOW)
// Part of libtwixt_hip.so; included by twixt_engine.hip only (device code, anonymous namespace).
#ifndef TWIXT_WINDOW_KERNEL_H_
#define TWIXT_WINDOW_KERNEL_H_

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
    for (int j = 0; j < NJ; j++) nxt[j] = gstate[chunk_index(WGEO::hbm_chunk(j), b0, bpad, G::CHT)];
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
      for (int j = c0; j < c1; j++) nxt[j] = gstate[chunk_index(WGEO::hbm_chunk(j), b1, bpad, G::CHT)];
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
        gw[k] = was_reset ? initv[2 * (g0 + k) + 1] : gstate[chunk_index((2 * (g0 + k) + 1), board, bpad, G::CHT)];
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
      if (c < W) *reinterpret_cast<u32x4 *>(scr + 4 * c) = gstate[chunk_index((2 * c + 1), sboard, bpad, G::CHT)];
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
        state[chunk_index((2 * c + 1), sboard, bpad, G::CHT)] = *reinterpret_cast<const uint4 *>(scr + 4 * c);
      wave_sync();
    }

    // retire the prefetched batch before this batch's stores (one in-order counter)
#pragma unroll
    for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(nxt[j]));

    // ---- stream back: scalars + this launch's journal chunk; the shortcut's flag chunk;
    //      whole boards for auto-reset lanes ----
    state[chunk_index(CS, board, bpad, G::CHT)] = lds4[ROW_SC * kWave + lane];
    state[chunk_index((CJ + (jpos >> 2)), board, bpad, G::CHT)] = lds4[(ROW_J + (jpos >> 2)) * kWave + lane];
    if (__ballot(fast_dirty != 0u && !was_reset)) {
      if (fast_dirty && !was_reset) {
        const int gi = __ffs((int)fast_dirty) - 1;
        state[chunk_index((2 * gi + 1), board, bpad, G::CHT)] = lds4[(ROW_O + gi - g0) * kWave + lane];
      }
    }
    if (__ballot(was_reset)) {
      if (was_reset) {
        const uint4 *__restrict__ initc = init_chunks;
#pragma unroll
        for (int i = 0; i < W; i++) {
          state[chunk_index((2 * i), board, bpad, G::CHT)] = lds4[i * kWave + lane];
          state[chunk_index((2 * i + 1), board, bpad, G::CHT)] = inwin(i) ? lds4[(ROW_O + i - g0) * kWave + lane] : initc[2 * i + 1];
        }
#pragma unroll
        for (int k = 0; k < twx::kJournalChunks; k++)
          state[chunk_index((CJ + k), board, bpad, G::CHT)] = lds4[(ROW_J + k) * kWave + lane];
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

#endif  // TWIXT_WINDOW_KERNEL_H_
