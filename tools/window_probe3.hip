// tools/window_probe3.hip -- round-3 probe: a windowed step whose window is fetched by the WAVE, not by the lane.
// (diagnostic, not product code.)  Rounds 1 and 2 measured windowed steps whose window was a per-lane
// gather of 16-byte pieces: every piece occupies a miss entry of the CU's memory pipe for a whole memory
// latency, so 2.2x fewer bytes bought nothing (profiles/r02_window_probe2.txt).  Here the window of a
// board is CONTIGUOUS (board-major links + flags region, 432 B per board, window = 10 pieces of 16 B at
// a data-dependent offset) and the wave fetches it cooperatively: 10 consecutive lanes per board, six
// boards per instruction -- about two 128-byte lines per board instead of ten requests.
// Per 32-board tile: part A = 12 coalesced rows of 512 B (pegs, scalars, journal), part B = 32 boards x
// 432 B board-major.  MODE 0: per-lane gather (probe 2), 1: stream the whole tile, 2: cooperative window.
// FLOOD: per-mille of the boards that additionally need their WHOLE part B (a multi-cell flood fill):
// fetched cooperatively, 27 lanes per board, two boards per instruction, as a second dependent load.
// WORK: synthetic dependent VALU operations per batch (the step logic's stand-in).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/window_probe3 tools/window_probe3.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int RA = 12, LFB = 432, TILE = RA * 512 + 32 * LFB;  // 19,968 B per 32 boards
constexpr int WP = 10;                                         // 16-byte pieces of a window
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int MODE>
__global__ __launch_bounds__(64) void probe(char *__restrict__ state, long ntiles, uint32_t iter, int flood_pm, int work) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  // LDS: rows 0..11 part A image (6 KiB); then window area 32 x 160 B (5 KiB); then 2 full boards (2 x 432 B); then offsets
  uint4 *const lwin = lds + RA * 32;
  uint4 *const lfull = lwin + 32 * WP;
  uint32_t *const loff = reinterpret_cast<uint32_t *>(lfull + 2 * 27);
  const int lane = threadIdx.x, h = lane >> 5, bl = lane & 31;
  const int q10 = lane / WP, p10 = lane - q10 * WP;  // cooperative window fetch: board of the instruction, piece
  const int q27 = lane >= 27 ? 1 : 0, p27 = lane - 27 * q27;  // cooperative whole-board fetch (lanes 54.. idle)
  u32x4 a[RA / 2];
  long t = blockIdx.x;
  auto loadA = [&](long tt) {
    const char *p = state + tt * (long)TILE + h * 512 + bl * 16;
#pragma unroll
    for (int j = 0; j < RA / 2; j++) a[j] = *reinterpret_cast<const u32x4 *>(p + j * 1024);
  };
  loadA(t);
  uint32_t sink = 0;
  while (t < ntiles) {
#pragma unroll
    for (int j = 0; j < RA / 2; j++) reinterpret_cast<u32x4 *>(lds)[(2 * j + h) * 32 + bl] = a[j];
    __builtin_amdgcn_wave_barrier();
    const long tn = t + gridDim.x;
    loadA(tn < ntiles ? tn : t);
    // "select": a data-dependent move per board from its peg words (both lanes alike): column x of a 24 x 24
    // board; the window is board columns x-3..x+2 = bits (x-3)*24 .. +144 of every plane = words w0..w0+5
    uint32_t hsh = iter;
#pragma unroll
    for (int r = 0; r < 9; r++) { const uint4 v = lds[r * 32 + bl]; hsh = mix(hsh ^ v.x ^ v.w); }
    const int x = (int)(hsh % 24u);
    int w0 = ((x - 3) * 24) >> 5;
    w0 = w0 < 0 ? 0 : w0 > 12 ? 12 : w0;
    int piece0 = (w0 * 24) >> 4;              // first 16-byte piece of the 432-byte region that holds word w0
    piece0 = piece0 > 27 - WP ? 27 - WP : piece0;
    const bool flood = (int)((hsh >> 8) % 1000u) < flood_pm;
    uint32_t acc = 0;
    const char *tb = state + t * (long)TILE + RA * 512;
    if (MODE == 0) {
      const char *pb = tb + bl * LFB + piece0 * 16 + h * (WP / 2) * 16;
      u32x4 w[WP / 2];
#pragma unroll
      for (int k = 0; k < WP / 2; k++) w[k] = *reinterpret_cast<const u32x4 *>(pb + k * 16);
#pragma unroll
      for (int k = 0; k < WP / 2; k++) acc ^= w[k].x + w[k].y + w[k].z + w[k].w;
    } else if (MODE == 1) {
#pragma unroll
      for (int it = 0; it < 14; it++) {
        const int q = it * 64 + lane;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(tb + (q < 864 ? q : 863) * 16);
        acc ^= v.x + v.y + v.z + v.w;
      }
    } else {
      if (!h) loff[bl] = (uint32_t)(bl * LFB + piece0 * 16);
      __builtin_amdgcn_wave_barrier();
      u32x4 w[6];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int b = 6 * k + q10;
        const bool on = lane < 60 && b < 32;
        const uint32_t o = loff[on ? b : 0];
        if (on) w[k] = *reinterpret_cast<const u32x4 *>(tb + o + p10 * 16);
      }
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int b = 6 * k + q10;
        if (lane < 60 && b < 32) reinterpret_cast<u32x4 *>(lwin)[b * WP + p10] = w[k];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < WP / 2; k++) { const uint4 v = lwin[bl * WP + h * (WP / 2) + k]; acc ^= v.x + v.y + v.z + v.w; }
    }
    // synthetic step logic
    uint32_t z = acc | 1u;
    for (int i = 0; i < work; i++) z = z * 0x9E3779B1u + (z >> 15);
    acc ^= z;
    if (MODE != 1) {
      // boards with a multi-cell flood: the whole 432 bytes, two boards per pass
      uint64_t todo = __ballot(flood) & 0xffffffffull;
      while (todo) {
        const int s0 = __ffsll((unsigned long long)todo) - 1; todo &= todo - 1;
        const int s1 = todo ? __ffsll((unsigned long long)todo) - 1 : s0; todo &= todo - 1;
        const int b = q27 ? s1 : s0;
        if (lane < 54) {
          const u32x4 v = *reinterpret_cast<const u32x4 *>(tb + b * LFB + p27 * 16);
          reinterpret_cast<u32x4 *>(lfull)[q27 * 27 + p27] = v;
        }
        __builtin_amdgcn_wave_barrier();
        const uint4 v = lfull[lane % 54];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
        if (lane < 2 && (acc & 0xffff) == 0x1234)  // a (rare) dirty chunk store, so the pass is not dead code
          __builtin_nontemporal_store(acc, reinterpret_cast<uint32_t *>(const_cast<char *>(tb) + b * LFB + 16));
      }
    }
    // the step's write: one dword per board, one 128-byte line per tile
    if (!h) __builtin_nontemporal_store(acc, reinterpret_cast<uint32_t *>(state + t * (long)TILE + 10 * 512) + bl);
    sink ^= acc;
    __builtin_amdgcn_wave_barrier();
    t = tn;
  }
  if (sink == 0x9e3779b9u) state[0] = 1;
}

template <int MODE>
float run(char *state, long ntiles, int waves_per_cu, int reps, int flood_pm, int work) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * waves_per_cu;
  const size_t ldsb = (RA * 32 + 32 * WP + 2 * 27) * 16 + 32 * 4;
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL((probe<MODE>), dim3(grid), dim3(64), ldsb, 0, state, ntiles, i, flood_pm, work);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((probe<MODE>), dim3(grid), dim3(64), ldsb, 0, state, ntiles, 100 + i, flood_pm, work);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

__global__ void fill_random(uint32_t *p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = mix((uint32_t)i * 2654435761u + 7u);
}

int main(int argc, char **argv) {
  const long nb = 1 << 20, ntiles = nb / 32;
  char *state;
  CK(hipMalloc(&state, (size_t)ntiles * TILE));
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t *>(state), ntiles * TILE / 4);
  CK(hipDeviceSynchronize());
  const int only = argc > 1 ? atoi(argv[1]) : -1;  // optional: run one mode only (for rocprofv3 --pmc)
  for (int work : {0, 600, 1500})
    for (int flood : {0, 28})
      for (int w : {8, 12, 16}) {
        printf("work %4d flood %2d/1000 waves/CU %2d:", work, flood, w);
        if (only < 0 || only == 0) printf(" lane-gather %.4f ms |", run<0>(state, ntiles, w, 40, flood, work));
        if (only < 0 || only == 2) printf(" wave-window %.4f ms |", run<2>(state, ntiles, w, 40, flood, work));
        if (only < 0 || only == 1) printf(" stream all %.4f ms", run<1>(state, ntiles, w, 40, flood, work));
        printf("\n");
        fflush(stdout);
      }
  return 0;
}
