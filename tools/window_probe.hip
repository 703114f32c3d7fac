// tools/window_probe.hip -- would a "windowed" step pay?  Each board = 12 coalesced chunks
// (pegs + scalars + journal, layout [chunk][board]) + 27 chunks of link/flag words.  A step
// reads the 12, derives a data-dependent window start, then reads only 12 of the 27 chunks
// (a) from a board-major region (per-lane contiguous 192 B) or (b) from [chunk][board]
// rows, writes 2 chunks back.  Compared with streaming all 39 chunks.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/window_probe tools/window_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NA = 12, NB = 27, WIN = 12;
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

// MODE 0: board-major window; 1: [chunk][board] window; 2: stream everything (39 chunks)
template <int MODE>
__global__ __launch_bounds__(64) void probe(uint4 *__restrict__ A, uint4 *__restrict__ Bm, long nb, uint32_t iter) {
  __shared__ uint4 win[WIN * 64];
  const int lane = threadIdx.x;
  const long b = (long)blockIdx.x * 64 + lane;
  uint4 a[NA];
#pragma unroll
  for (int c = 0; c < NA; c++) a[c] = A[(long)c * nb + b];
  uint32_t h = iter;
#pragma unroll
  for (int c = 0; c < NA; c++) h = mix(h ^ a[c].x ^ a[c].w);
  const int c0 = (int)(h % (uint32_t)(NB - WIN + 1));
  uint32_t acc = 0;
  if (MODE == 2) {
#pragma unroll
    for (int j = 0; j < NB; j++) { uint4 v = Bm[(long)j * nb + b]; acc ^= v.x + v.y + v.z + v.w; }
  } else {
#pragma unroll
    for (int j = 0; j < WIN; j++) {
      uint4 v = (MODE == 0) ? Bm[b * NB + c0 + j] : Bm[(long)(c0 + j) * nb + b];
      win[j * 64 + lane] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WIN; j++) { uint4 v = win[j * 64 + lane]; acc ^= v.x + v.y + v.z + v.w; }
  }
  a[NA - 1].x = acc; a[NA - 2].y += 1;
  A[(long)(NA - 1) * nb + b] = a[NA - 1];
  A[(long)(NA - 2) * nb + b] = a[NA - 2];
}

template <int MODE>
float run(uint4 *A, uint4 *Bm, long nb, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(probe<MODE>, dim3(nb / 64), dim3(64), 0, 0, A, Bm, nb, i);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(probe<MODE>, dim3(nb / 64), dim3(64), 0, 0, A, Bm, nb, 100 + i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main() {
  const long nb = 1 << 20;
  uint4 *A, *Bm;
  CK(hipMalloc(&A, (size_t)NA * nb * 16)); CK(hipMalloc(&Bm, (size_t)NB * nb * 16));
  CK(hipMemset(A, 1, (size_t)NA * nb * 16)); CK(hipMemset(Bm, 2, (size_t)NB * nb * 16));
  printf("board-major window (12 of 27 chunks):   %.4f ms\n", run<0>(A, Bm, nb, 50));
  printf("[chunk][board] window (12 of 27):       %.4f ms\n", run<1>(A, Bm, nb, 50));
  printf("stream all 39 chunks (read) + 2 writes: %.4f ms\n", run<2>(A, Bm, nb, 50));
  return 0;
}
