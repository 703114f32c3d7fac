// tools/window_probe2.hip -- round-2 probe: what would a WINDOWED step cost on the memory side?
// (diagnostic, not product code.)  Per 32-board tile: part A = 12 coalesced rows of 512 B (pegs,
// scalars, journal), part B = 32 boards x 432 B board-major (links + flags, 9 units of 48 B).
// A persistent wave (two lanes per board, as step_pair_kernel) prefetches part A one batch ahead,
// derives a data-dependent window start per board, then gathers G units of 48 B per board from
// part B (a DEPENDENT load), and writes one 512-byte journal row.  MODE 1 streams part B whole
// instead (what the kernel does today).  Prints ms per 2^20 boards for several waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/window_probe2 tools/window_probe2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int RA = 12, UB = 9, TILE = RA * 512 + 32 * UB * 48;  // 19,968 B per 32 boards
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

// G: units gathered per board (4 = 192 B, 3 = 144 B); MODE 0 windowed, 1 stream all of part B
template <int MODE, int G>
__global__ __launch_bounds__(64) void probe(char *__restrict__ state, long ntiles, uint32_t iter) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x, h = lane >> 5, bl = lane & 31;
  u32x4 a[RA / 2];
  long t = blockIdx.x;
  auto loadA = [&](long tt) {
    const char *p = state + tt * (long)TILE + h * 512 + bl * 16;
#pragma unroll
    for (int j = 0; j < RA / 2; j++) a[j] = *reinterpret_cast<const u32x4 *>(p + j * 1024);
  };
  loadA(t);
  while (t < ntiles) {
    // "fill": the part A batch goes to LDS, the next one is requested
#pragma unroll
    for (int j = 0; j < RA / 2; j++) reinterpret_cast<u32x4 *>(lds)[(2 * j + h) * 32 + bl] = a[j];
    __builtin_amdgcn_wave_barrier();
    const long tn = t + gridDim.x;
    loadA(tn < ntiles ? tn : t);
    // "select": a data-dependent window start from the board's peg words (both lanes alike)
    uint32_t hsh = iter;
#pragma unroll
    for (int r = 0; r < 9; r++) { const uint4 v = lds[r * 32 + bl]; hsh = mix(hsh ^ v.x ^ v.w); }
    const int u0 = (int)(hsh % (uint32_t)(UB - G + 1));
    uint32_t acc = 0;
    const char *pb = state + t * (long)TILE + RA * 512 + bl * (UB * 48);
    if (MODE == 0) {
      // dependent gather: this lane's half of the G units (3 chunks of 16 B each)
      u32x4 w[3 * ((G + 1) / 2)];
#pragma unroll
      for (int k = 0; k < 3 * ((G + 1) / 2); k++) {
        const int c = 3 * (u0 + h * ((G + 1) / 2)) + k;
        w[k] = *reinterpret_cast<const u32x4 *>(pb + (c < 3 * UB ? c : 3 * UB - 1) * 16);
      }
#pragma unroll
      for (int k = 0; k < 3 * ((G + 1) / 2); k++) acc ^= w[k].x + w[k].y + w[k].z + w[k].w;
    } else {
      // stream the whole tile's part B, coalesced: 32 x 432 B = 864 pieces of 16 B
      const char *tb = state + t * (long)TILE + RA * 512;
#pragma unroll
      for (int it = 0; it < 14; it++) {
        const int q = it * 64 + lane;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(tb + (q < 864 ? q : 863) * 16);
        acc ^= v.x + v.y + v.z + v.w;
      }
    }
    // the step's write: one journal row (lower lanes, 512 B contiguous)
    if (!h) {
      u32x4 o = {acc, hsh, (uint32_t)u0, iter};
      __builtin_nontemporal_store(o.x, reinterpret_cast<uint32_t *>(state + t * (long)TILE + 10 * 512 + bl * 16));
      __builtin_nontemporal_store(o.y, reinterpret_cast<uint32_t *>(state + t * (long)TILE + 10 * 512 + bl * 16 + 4));
      __builtin_nontemporal_store(o.z, reinterpret_cast<uint32_t *>(state + t * (long)TILE + 10 * 512 + bl * 16 + 8));
      __builtin_nontemporal_store(o.w, reinterpret_cast<uint32_t *>(state + t * (long)TILE + 10 * 512 + bl * 16 + 12));
    }
    __builtin_amdgcn_wave_barrier();
    t = tn;
  }
}

template <int MODE, int G>
float run(char *state, long ntiles, int waves_per_cu, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * waves_per_cu;
  const size_t ldsb = RA * 512;
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL((probe<MODE, G>), dim3(grid), dim3(64), ldsb, 0, state, ntiles, i);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((probe<MODE, G>), dim3(grid), dim3(64), ldsb, 0, state, ntiles, 100 + i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

__global__ void fill_random(uint32_t *p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = mix((uint32_t)i * 2654435761u + 7u);
}

int main(int argc, char **argv) {
  const long nb = 1 << 20, ntiles = nb / 32;
  char *state;
  CK(hipMalloc(&state, (size_t)ntiles * TILE)); CK(hipMemset(state, 1, (size_t)ntiles * TILE));
  if (argc > 1) {  // any argument: random board bytes instead of a constant fill
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t *>(state), ntiles * TILE / 4);
    CK(hipDeviceSynchronize());
    printf("(random data)\n");
  }
  for (int w : {8, 12, 16, 24, 32}) {
    printf("waves/CU %2d: window 4 units (192 B) %.4f ms | 3 units (144 B) %.4f ms | stream all %.4f ms\n", w,
           run<0, 4>(state, ntiles, w, 40), run<0, 3>(state, ntiles, w, 40), run<1, 4>(state, ntiles, w, 40));
  }
  return 0;
}
