// twixt_device_common.h -- wave size, launch flags, LDS / DPP / bit helpers, diagnostic stamps
// Part of libtwixt_hip.so; included by twixt_engine.hip only (device code, anonymous namespace).
#ifndef TWIXT_DEVICE_COMMON_H_
#define TWIXT_DEVICE_COMMON_H_

constexpr int kWave = 64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // native 16-byte vector: stays in VGPRs

// ---------------------------------------------------------------------------
// Internal layout (HBM and LDS alike).  A board is CHT = 2W+3 chunks of 16 B:
//   chunk 2i   = { red_i, blue_i, linkNNE_i, linkENE_i }      i = 0..W-1
//   chunk 2i+1 = { linkESE_i, linkSSE_i, start_i, end_i }
//   chunk 2W   = scalars { s0, s1, s2, s3 }
//   chunk rows 2W+1, 2W+2 of a tile = redo journal, 8 slots x boards of the tile x 32-bit records
//   ([slot][board], journal_word_index() below)
// i.e. the 8 planes are interleaved word by word: everything about the 32 cells
// 32i..32i+31 sits in two adjacent chunks.  One address computation per cell /
// column then reaches every plane through immediate offsets.
// HBM: tiles of 32 or 64 boards, state[tile][chunk][board of the tile] (uint4, chunk_index()
// below); LDS image of a wave: [chunk][board slot] (uint4).
// ---------------------------------------------------------------------------

__device__ __forceinline__ void lds_or(uint32_t *p, uint32_t v) {
  __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // ds_or_b32
}
// two adjacent dwords (p 8-byte aligned) in one LDS operation
__device__ __forceinline__ void lds_or2(uint32_t *p, uint32_t v0, uint32_t v1) {
  __hip_atomic_fetch_or(reinterpret_cast<uint64_t *>(p), (uint64_t)v0 | ((uint64_t)v1 << 32), __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_WAVEFRONT);  // ds_or_b64
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
#elif defined(TWX_MARK)  // phase boundaries as comments in the ISA (static instruction counts per phase)
#define STAMP_DECL
#define STAMP(i) asm volatile("; MARK P" #i)
#define STAMP_FLUSH
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// Where chunk c (16 bytes) of board b lives in the state array, in uint4 units: tiles of 2^ts
// boards, [tile][chunk][2^ts boards].  ts = 6 (64 boards, a wave of step_kernel64: one contiguous
// CHT KiB block per wave, whole DRAM pages) for the boards up to N = 18; ts = 5 for N >= 19, whose
// whole-batch kernel works on 32 boards per wave (twixt_pair_kernel.h): with 64-board tiles its
// waves read interleaved 512-byte halves of every row and the bare read stream ran 6 % slower.
#ifndef TWX_SMALL_CHT
#define TWX_SMALL_CHT 25  // (twixt_engine.hip sets it first and says how it was measured)
#endif
constexpr int tile_shift(int cht) { return cht > TWX_SMALL_CHT ? 5 : 6; }
__host__ __device__ __forceinline__ long chunk_index(long c, long b, int ts, int cht) {
  return (((b >> ts) * cht + c) << ts) + (b & ((1L << ts) - 1));
}

// The journal of a tile lives in its last kJournalChunks chunk rows as kJournalSlots rows of
// (1 << ts) dwords, [slot][board]: record s of board b is ONE dword, and the records a launch
// appends for the boards of a tile are one contiguous 128- or 256-byte line (round 2: a step used
// to write a 16-byte journal chunk per board).  Dword index into the state array:
__host__ __device__ __forceinline__ long journal_word_index(int s, long b, int ts, int cht) {
  return ((((b >> ts) * cht + (cht - twx::kJournalChunks)) << ts) << 2) + ((long)s << ts) + (b & ((1L << ts) - 1));
}
// ---------------------------------------------------------------------------
// SPLIT layout (round 3; the big boards, N >= 17 -- TWX_SPLIT_CHT --, whose whole-batch step is step_window1_kernel,
// twixt_window_kernel.h).  A step needs ALL of a board's pegs (legal mask, k-th legal action) but its
// links and border flags only in the six board columns around the move, so the two live apart:
//   tile of 32 boards = RA rows of 512 B, [row][board] x 16 B, streamed by every launch:
//       rows 0..RP-1   pegs     { red_2r, blue_2r, red_2r+1, blue_2r+1 }        RP = ceil(W / 2)
//       row  RP        scalars  { s0, s1, s2, s3 }
//       rows RP+1, +2  redo journal, 8 slots x 32 boards x 32-bit records, [slot][board]
//   then 32 x LFB bytes BOARD-MAJOR: links + flags of board q at q * LFB, word i (cells 32i..32i+31)
//       = 6 dwords { linkNNE_i, linkENE_i, linkESE_i, linkSSE_i, start_i, end_i }  (LFB = 24 W, rounded
//       up to a multiple of 16)
// so the window of a move -- board columns x-3..x+2 = at most NW consecutive words of every plane -- is
// ONE contiguous run of about 150 bytes (two 128-byte lines) that the wave fetches after it has
// chosen the move.  N=24: 12 rows (6 KiB) + 32 x 432 B = 19,968 B per tile, as many as the chunk layout.
// ---------------------------------------------------------------------------
constexpr int split_peg_rows(int W) { return (W + 1) / 2; }
constexpr int split_rows(int W) { return split_peg_rows(W) + 1 + twx::kJournalChunks; }  // RA
constexpr int split_lf_bytes(int W) { return (24 * W + 15) & ~15; }                      // LFB
constexpr long split_tile_bytes(int W) { return (long)split_rows(W) * 512 + 32L * split_lf_bytes(W); }
// dword index into the state array of: word i of plane p, scalar j, journal record s -- of board b
__host__ __device__ __forceinline__ long split_word_index(int p, int i, long b, int W) {
  const long tile = (b >> 5) * (split_tile_bytes(W) >> 2), q = b & 31;
  if (p < 2) return tile + (long)(i >> 1) * 128 + q * 4 + (i & 1) * 2 + p;
  return tile + (long)split_rows(W) * 128 + q * (split_lf_bytes(W) >> 2) + i * 6 + (p - 2);
}
__host__ __device__ __forceinline__ long split_scalar_index(int j, long b, int W) {
  return (b >> 5) * (split_tile_bytes(W) >> 2) + (long)split_peg_rows(W) * 128 + (b & 31) * 4 + j;
}
__host__ __device__ __forceinline__ long split_journal_index(int s, long b, int W) {
  return (b >> 5) * (split_tile_bytes(W) >> 2) + (long)(split_peg_rows(W) + 1) * 128 + (long)s * 32 + (b & 31);
}

// ---- the window of a move in board column x (twixt_window1_kernel.h): which 16-byte pieces of the board's region.
//      Host-checkable arithmetic (tests/layout_check.hip proves the coverage for every size, column and
//      candidate set from the crossing masks of twixt_geom.h). ----
constexpr int win_words(int n, int W) {  // columns x-3 .. x+2 = 6n bits from a bit offset < 32
  const int nw = (6 * n + 31 + 31) / 32;
  return nw < W ? nw : W;
}
constexpr int win_pieces(int n, int W) {  // (24 w0 is 8 bytes off a piece for odd w0)
  const int npc = (24 * win_words(n, W) + 8 + 15) / 16, lfp = split_lf_bytes(W) / 16;
  return npc < lfp ? npc : lfp;
}
__host__ __device__ constexpr int win_piece0(int n, int W, int x) {  // first piece of the full window
  const int c_lo = x - 3 < 0 ? 0 : x - 3;
  int w0 = (c_lo * n) >> 5;
  w0 = w0 > W - win_words(n, W) ? W - win_words(n, W) : w0;
  const int piece0 = (24 * w0) >> 4, last = split_lf_bytes(W) / 16 - win_pieces(n, W);
  return piece0 > last ? last : piece0;
}
constexpr uint32_t win_dir_mask(int dx) {  // the directions (twx::kDX) that reach column x + dx
  uint32_t m = 0;
  for (int d = 0; d < 8; d++)
    if (twx::kDX[d] == dx) m |= 1u << d;
  return m;
}
struct WinRange {
  int p_lo, np;  // first piece (of the region), pieces - 1
};
// The pieces a move with candidate links `cand` (one bit per direction, not zero) can touch: a link to column x+1
// is crossed only by links whose west ends lie in columns x-1, x; to x+2: x-1..x+1; to x-1: x-2, x-1; to x-2:
// x-3..x-1 (twixt_geom.h kCross); the flags of the candidates' columns and of the peg's own.
__host__ __device__ constexpr WinRange win_range(int n, uint32_t cand, int x) {
  const int lo = (cand & win_dir_mask(-2)) ? -3 : (cand & win_dir_mask(-1)) ? -2 : -1;
  const int hi = (cand & win_dir_mask(2)) ? 2 : (cand & win_dir_mask(1)) ? 1 : 0;
  int c0 = x + lo, c1 = x + hi + 1;
  c0 = c0 < 0 ? 0 : c0;
  c1 = c1 > n ? n : c1;
  const int w_lo = (c0 * n) >> 5, w_hi = (c1 * n - 1) >> 5;      // words of the column range
  const int p_lo = (3 * w_lo) >> 1, p_hi = (3 * w_hi + 2) >> 1;  // ... their 16-byte pieces (24 B a word)
  return WinRange{p_lo, p_hi - p_lo};
}

template <bool NT>
__device__ __forceinline__ void tile_store_word(uint32_t *p, uint32_t v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---------------------------------------------------------------------------
// Tile I/O of the whole-batch step kernel (64 boards per wave): the 64 boards of a batch are ONE
// contiguous block of CHT KiB (one 64-board tile, or two 32-board tiles), chunk c of a tile at byte
// c * RB + slot * 16 with RB = 16 << TS bytes per chunk row.  Written as
// instructions, not as C++ loads, for two reasons:
//   * addressing: one scalar tile base + a handful of constant per-lane offsets + the 13-bit
//     immediate (8 KiB per offset register: chunk c % (8192/RB) at c' * RB - 4096) instead of a 64-bit address per chunk
//     (the generic form cost ~80 SGPRs -- the kernel spilled 49 -- and ~250 scalar instructions
//     per batch);
//   * registers: the prefetched batch (CHT x 16 B per lane) lives in AGPRs for the whole step.
//     global_load writes AGPRs and ds_write reads them directly on gfx950 (one register file);
//     left to the compiler, the loads landed in VGPRs and were copied to AGPRs and back
//     (156 v_accvgpr moves per batch) behind an s_waitcnt in the MIDDLE of the step -- i.e. the
//     wave waited for its prefetch long before it needed it.
// The compiler does not see these loads: the kernel waits for them explicitly (vm_wait_all)
// before the LDS fill, and nothing may copy the registers in between (checked in the ISA: no
// v_accvgpr_* in the kernel).  Untracked younger loads only make the compiler's own counted
// waits stricter, never weaker (vmcnt counts in order).
// ---------------------------------------------------------------------------
// HAZARD (gfx9 / CDNA): a VMEM instruction must not read an SGPR within 5 wait states of a VALU instruction that
// wrote it (v_readlane_b32 / v_readfirstlane_b32: the compiler's SGPR spill reloads, uniform_ptr()).  The
// compiler pads its own instructions with s_nop, but it does not look into inline asm -- with the tile
// pointer reloaded from a spill lane right before an asm load, the load used a stale base and faulted
// (round 3: step_window1_kernel<22>, "Memory access fault ... 0xffff..."; whether it bites depends on the
// register allocation).  So every asm load with a scalar base carries its own wait states.
#define TWX_SGPR_HAZARD_NOP "s_nop 4\n\t"
template <int J, bool AG, int TS>
__device__ __forceinline__ void tile_load_chunk(u32x4 &dst, uint32_t voff, const void *tile) {
  constexpr int RB = 16 << TS, CPO = 8192 / RB;
  if constexpr (AG)
    asm volatile(TWX_SGPR_HAZARD_NOP "global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst) : "v"(voff), "s"(tile),
                 "n"((J % CPO) * RB - 4096) : "memory");
  else
    asm volatile(TWX_SGPR_HAZARD_NOP "global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(tile),
                 "n"((J % CPO) * RB - 4096) : "memory");
}
template <int C0, int C1, bool AG, int TS>
__device__ __forceinline__ void tile_load(u32x4 *nxt, const uint32_t *voff, const void *tile) {
  if constexpr (C0 < C1) {
    tile_load_chunk<C0, AG, TS>(nxt[C0], voff[C0 / (8192 / (16 << TS))], tile);
    tile_load<C0 + 1, C1, AG, TS>(nxt, voff, tile);
  }
}
// LDS image row c of this lane: byte lds_lane + c*1024 (16-bit immediate)
template <int C0, int C1, bool AG>
__device__ __forceinline__ void tile_fill(uint32_t lds_lane, const u32x4 *nxt) {
  if constexpr (C0 < C1) {
    if constexpr (AG)
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lds_lane), "a"(nxt[C0]), "n"(C0 * 1024) : "memory");
    else
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lds_lane), "v"(nxt[C0]), "n"(C0 * 1024) : "memory");
    tile_fill<C0 + 1, C1, AG>(lds_lane, nxt);
  }
}
// The same for step_pair_kernel (two lanes per board, 32 boards per wave, 32-board tiles): a lane
// moves chunks 2j + half of its board; chunk pair j is the KiB at byte j * 1024 of the tile and the
// per-lane offset registers already contain half * 512, so every instruction reads one contiguous
// KiB; one register reaches eight pairs (8 KiB); always AGPRs.  LDS image rows are 512 B too.
template <int J>
__device__ __forceinline__ void pair_tile_load_chunk(u32x4 &dst, uint32_t voff, const void *tile) {
  asm volatile(TWX_SGPR_HAZARD_NOP "global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst) : "v"(voff), "s"(tile),
               "n"((J % 8) * 1024 - 4096) : "memory");
}
template <int C0, int C1>
__device__ __forceinline__ void pair_tile_load(u32x4 *nxt, const uint32_t *voff, const void *tile) {
  if constexpr (C0 < C1) {
    pair_tile_load_chunk<C0>(nxt[C0], voff[C0 / 8], tile);
    pair_tile_load<C0 + 1, C1>(nxt, voff, tile);
  }
}
template <int C0, int C1>
__device__ __forceinline__ void pair_tile_fill(uint32_t lds_lane, const u32x4 *nxt) {
  if constexpr (C0 < C1) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lds_lane), "a"(nxt[C0]), "n"(C0 * 1024) : "memory");
    pair_tile_fill<C0 + 1, C1>(lds_lane, nxt);
  }
}
// Write-back of board chunks.  Where the step is bound by the board stream (N >= 17) non-temporal
// stores measured 2.2 % faster than plain ones on the whole step (write-through sc0 sc1: 0.8 %):
// the written lines are not read again before the next launch.  The small boards are bound by
// their integer work and ran 2-4 % faster with plain stores.
template <bool NT>
__device__ __forceinline__ void tile_store(uint4 *p, uint4 v) {
  if constexpr (NT) {
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
  } else {
    *p = v;
  }
}
// compile-time loop: f(std::integral_constant<int, I>{}) for I = B .. E-1 (instruction immediates)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
__device__ __forceinline__ void vm_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int kFlagRandom = 1, kFlagAutoReset = 2, kFlagCompact = 4, kFlagFlush = 8, kFlagReverse = 16,
              kFlagRolling = 32,  // + bits 8..11: journal period, bits 12..27: batch stride of a wave
              kFlagDynamic = 64,  // batches after a wave's first are drawn from the ticket counters
              kFlagWholeTile = 128;  // split layout: this rewriting launch takes step_window_full_kernel (small batches)

// dynamic batch order: one ticket counter per group of waves (blockIdx.x % kTicketGroups),
// counters kTicketStride words apart; the bases travel as a kernel argument
#ifndef TWX_TICKET_GROUPS
#define TWX_TICKET_GROUPS 8  // one per XCD label (reversed_batch below); 1 and 2 contend on the counter
#endif
constexpr int kTicketGroups = TWX_TICKET_GROUPS, kTicketStride = 64;
// Batch a wave works on when the launch walks the batches backwards (kFlagReverse).  Workgroups are
// dealt to the eight XCDs round-robin (blocks b and b + 8 share one) and the persistent grids are
// multiples of 8, so in a forward launch batch t is worked on by the XCD with label t % 8, statically
// (t = b + round * grid) and by ticket (eight ticket groups, group = b % 8).  The backward walk keeps
// that: blocks of 8 batches are reversed, the order inside a block stays (a ragged last block stays
// last).  A board's lines are then asked for by the same XCD in every launch: +1.2 % at N=24, +7..8 %
// at N=19..21, every size gains (profiles/r02_xcd_order_ab.txt; a memory-side effect: the L2 hit
// counts do not change).
__host__ __device__ __forceinline__ long reversed_batch(long i, long nbatch) {
  const long q = nbatch & ~7L;
  return i < q ? (((q - 1 - i) & ~7L) | (i & 7L)) : i;
}
struct TicketBases { uint32_t b[kTicketGroups]; };

#endif  // TWIXT_DEVICE_COMMON_H_
