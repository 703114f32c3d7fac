// twixt_util_kernels.h -- reset / export / import / copy / legal mask / status / cells / observation kernels
// Part of libtwixt_hip.so; included by twixt_engine.hip only (device code, anonymous namespace).
#ifndef TWIXT_UTIL_KERNELS_H_
#define TWIXT_UTIL_KERNELS_H_

// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Layout-agnostic utility kernels (runtime N): reset / export / import / copy /
// queries.  One thread per (board, chunk) or per board.
// ---------------------------------------------------------------------------
struct Dims {
  int n, W, SW, CH, CHT;  // CH: state chunks (planes + scalars); CHT: + the journal's chunk rows
  int ts;                 // tile_shift(CHT): boards per tile of the state array = 1 << ts
  int lay;                // 0: chunk layout (tiles of [chunk][board]); 1: split layout (twixt_device_common.h)
  long batch, bpad;
};

__device__ __forceinline__ uint32_t comp(const uint4 &v, int j) {
  return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
}

// Logical chunk c of board b, whatever the handle's layout: c = 2i -> { red_i, blue_i, linkNNE_i,
// linkENE_i }, c = 2i+1 -> { linkESE_i, linkSSE_i, start_i, end_i }, c = 2W -> the scalars.  In the
// chunk layout that is one 16-byte load; in the split layout the pieces are 8-byte aligned.
__device__ __forceinline__ uint4 load_chunk(const uint4 *state, const Dims &d, long b, int c) {
  if (!d.lay) return state[chunk_index(c, b, d.ts, d.CHT)];
  const uint32_t *w = reinterpret_cast<const uint32_t *>(state);
  if (c >= 2 * d.W) return *reinterpret_cast<const uint4 *>(w + split_scalar_index(0, b, d.W));
  const int i = c >> 1;
  const uint2 lo = *reinterpret_cast<const uint2 *>(w + ((c & 1) ? split_word_index(4, i, b, d.W) : split_word_index(0, i, b, d.W)));
  const uint2 hi = *reinterpret_cast<const uint2 *>(w + ((c & 1) ? split_word_index(6, i, b, d.W) : split_word_index(2, i, b, d.W)));
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// ... and the store; c in (2W, CHT) addresses the 16 bytes at board b's position of a journal row
// (whole-batch reset only: together the boards of a tile cover the rows)
__device__ __forceinline__ void store_chunk(uint4 *state, const Dims &d, long b, int c, uint4 v) {
  if (!d.lay) {
    state[chunk_index(c, b, d.ts, d.CHT)] = v;
    return;
  }
  uint32_t *w = reinterpret_cast<uint32_t *>(state);
  if (c >= 2 * d.W) {
    *reinterpret_cast<uint4 *>(w + split_scalar_index(0, b, d.W) + (long)(c - 2 * d.W) * 128) = v;
    return;
  }
  const int i = c >> 1;
  *reinterpret_cast<uint2 *>(w + ((c & 1) ? split_word_index(4, i, b, d.W) : split_word_index(0, i, b, d.W))) = make_uint2(v.x, v.y);
  *reinterpret_cast<uint2 *>(w + ((c & 1) ? split_word_index(6, i, b, d.W) : split_word_index(2, i, b, d.W))) = make_uint2(v.z, v.w);
}

// word i of plane p / scalar j of board b
__device__ __forceinline__ uint32_t plane_word(const uint4 *state, const Dims &d, long b, int p, int i) {
  if (d.lay) return reinterpret_cast<const uint32_t *>(state)[split_word_index(p, i, b, d.W)];
  return comp(state[chunk_index((2 * i + (p >> 2)), b, d.ts, d.CHT)], p & 3);
}
__device__ __forceinline__ uint32_t scalar_word(const uint4 *state, const Dims &d, long b, int j) {
  if (d.lay) return reinterpret_cast<const uint32_t *>(state)[split_scalar_index(j, b, d.W)];
  return comp(state[chunk_index((2 * d.W), b, d.ts, d.CHT)], j);
}

__global__ void reset_kernel(uint4 *state, Dims d, const uint32_t *init, const int32_t *ids, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t % (ids ? n : d.bpad);
  const int c = (int)(t / (ids ? n : d.bpad));
  // A whole-batch reset clears the journal rows too.  A listed-boards reset must NOT: "chunk" CH or
  // CH + 1 at board b's position of a [slot][board] journal row is one slot's records of FOUR boards
  // of the tile, not b's own journal; planes and scalars (journal base 0) are all a board owns.
  if (c >= (ids ? d.CH : d.CHT)) return;
  long b = j;
  if (ids) {
    b = ids[j];
    if (b < 0 || b >= d.batch) return;
  }
  uint4 v;  // init is in logical chunk order; scalars and journal rows are zero
  v.x = init[4 * c + 0]; v.y = init[4 * c + 1]; v.z = init[4 * c + 2]; v.w = init[4 * c + 3];
  store_chunk(state, d, b, c, v);
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
    v.x = sc[0]; v.y = sc[1] & ~(15u << twx::kJBaseShift); v.z = sc[2]; v.w = sc[3];  // (journal base: an internal field)
  }
  store_chunk(state, d, b, c, v);
}

__global__ void copy_kernel(uint4 *dst, Dims dd, const int32_t *dst_ids, const uint4 *src, Dims sd,
                            const int32_t *src_ids, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long j = t % n;
  const int c = (int)(t / n);
  if (c >= dd.CH) return;
  const long db = dst_ids[j], sb = src_ids[j];
  if (db < 0 || db >= dd.batch || sb < 0 || sb >= sd.batch) return;
  store_chunk(dst, dd, db, c, load_chunk(src, sd, sb, c));
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

__global__ void status_kernel(const uint4 *state, Dims d, int jpos, const int32_t *ids, long n, uint8_t *result,
                              int32_t *move_counter, int8_t *current_player) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long b = ids ? ids[j] : j;
  if (b < 0 || b >= d.batch) return;
  const uint4 s = load_chunk(state, d, b, d.CH - 1);
  uint32_t counter = s.x & 0xffffu, res = s.y & 0xffu, move_one = s.x >> 16, swapped = (s.y >> 8) & 1u;
  // the stored scalars are as old as the planes: advance them by the board's journal records
  // ([slot][board] rows) from the board's base slot up to the handle's journal position (the
  // slots above hold stale records; rolling compaction passes kJournalSlots: its empty slots are zero)
  const uint32_t *const words = reinterpret_cast<const uint32_t *>(state);
  const int jbase = (int)((s.y >> twx::kJBaseShift) & 15u);
  for (int sl = jbase; sl < jpos; sl++)
    twx::journal_scalars(words[d.lay ? split_journal_index(sl, b, d.W) : journal_word_index(sl, b, d.ts, d.CHT)], counter,
                         move_one, res, swapped);
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
#ifndef TWX_OBS_NT
#define TWX_OBS_NT 1
#endif
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
  for (int c = lane; c < 2 * W + 1; c += kWave) sst[c] = load_chunk(state, d, b, c);
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
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 *o = reinterpret_cast<f32x4 *>(out + j * (long)(12 * npos));
  const uint32_t magic = (1u << 24) / (uint32_t)npos + 1u;  // e / npos == (e * magic) >> 24 for e < 12*npos <= 6336
  for (int q = lane; q < 3 * npos; q += kWave) {  // 12*npos floats = 3*npos float4
    float v[4];
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      const int e = 4 * q + qq;
      const int k = (int)(((uint32_t)e * magic) >> 24), pos = e - k * npos;
      const int colour = k >= 6 ? 1 : 0;
      v[qq] = ((feat[colour][pos] >> (k - 6 * colour)) & 1u) ? 1.0f : 0.0f;
    }
    const f32x4 vv = {v[0], v[1], v[2], v[3]};
#if TWX_OBS_NT
    __builtin_nontemporal_store(vv, o + q);  // write-only output, 25 KB per board: keep it out of the caches
#else
    o[q] = vv;
#endif
  }
}

#endif  // TWIXT_UTIL_KERNELS_H_
