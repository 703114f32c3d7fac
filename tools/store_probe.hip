// tools/store_probe.hip -- diagnostic (not product code): what does a scattered partial-sector store cost next to
// the window fetch it follows?  One lane per board, regions of 432 B board-major (as in the split layout); per board
// a data-dependent window of 160 B is read, then MODE 0: nothing is stored; 1: 16 B at an 8-byte-aligned offset
// inside the window (what step_window1_kernel does for a move's link words); 2: the 32-byte-aligned sector around
// it; 3: the whole 128-byte line around it.  Time per pass over 2^20 boards: 74.1 / 105.2 / 59.3 / 52.1 us -- the
// dword stores look expensive here, but the probe does not carry over: in step_window1_kernel the changed words
// stored as whole aligned 16-byte pieces (bit-exact) ran 0.1057 against 0.1040 ms per step, and the probe's own
// read-only leg is slower than its storing legs (20 passes over 452 MB back to back: the Infinity Cache decides).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/store_probe tools/store_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int LFB = 432;
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int MODE>
__global__ __launch_bounds__(64) void probe(char *__restrict__ state, long nboards, uint32_t iter, uint32_t *__restrict__ out) {
  uint32_t sink = 0;
  for (long b = (long)blockIdx.x * 64 + threadIdx.x; b < nboards; b += (long)gridDim.x * 64) {
    const uint32_t h = mix((uint32_t)b * 0x9e3779b1u ^ iter);
    const uint32_t o = (h % 34u) * 8u;               // 8-byte-aligned offset of the changed words, 0..264
    char *const reg = state + b * (long)LFB;
    const uint32_t w0 = o & ~15u;                      // window: 160 B from the piece that holds the offset
    uint4 v[10];
#pragma unroll
    for (int p = 0; p < 10; p++) v[p] = *reinterpret_cast<const uint4 *>(reg + w0 + 16 * p);
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < 10; p++) acc ^= v[p].x + v[p].y * 3u + v[p].z * 5u + v[p].w * 7u;
    sink ^= acc;
    const uint32_t keep = acc & 0u;                    // (the stored data are the loaded data: the contents never change)
    if (MODE == 1) {
      uint32_t *const d = reinterpret_cast<uint32_t *>(reg + o);
      const uint32_t *const s = reinterpret_cast<const uint32_t *>(&v[0]) + ((o - w0) >> 2);
      d[0] = s[0] | keep; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
    } else if (MODE == 2) {
      const uint32_t a = (uint32_t)((b * LFB + o) & ~31L) - (uint32_t)(b * LFB);  // sector start relative to the region (may be < w0)
      const int p = ((int)a - (int)w0) >> 4;          // piece index of the sector start: -1 .. 8
      if (p >= 0 && p + 1 < 10) {
        uint4 x0 = v[0], x1 = v[1];
#pragma unroll
        for (int k = 1; k < 9; k++) if (k == p) { x0 = v[k]; x1 = v[k + 1]; }
        x0.x |= keep;
        *reinterpret_cast<uint4 *>(reg + a) = x0;
        *reinterpret_cast<uint4 *>(reg + a + 16) = x1;
      }
    } else if (MODE == 3) {
      const long la = (b * LFB + o) & ~127L;
      char *const line = state + la;
      if (la >= b * LFB + w0 && la + 128 <= b * LFB + w0 + 160) {
        const int p = (int)(la - (b * LFB + w0)) >> 4;  // 0..2
#pragma unroll
        for (int k = 0; k < 8; k++) {
          uint4 x = v[k];
          if (p == 1) x = v[k + 1];
          if (p == 2) x = v[k + 2];
          x.x |= keep;
          *reinterpret_cast<uint4 *>(line + 16 * k) = x;
        }
      }
    }
  }
  if (sink == 0x12345u) out[0] = sink;
}

int main() {
  const long nb = 1 << 20;
  char *st; uint32_t *out;
  CK(hipMalloc(&st, nb * LFB + 4096)); CK(hipMemset(st, 1, nb * LFB + 4096)); CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[4] = {"window read only", "+ 16 B store at an 8-byte offset", "+ aligned 32-byte sector", "+ whole 128-byte line (when inside the window)"};
  for (int mode = 0; mode < 4; mode++) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; it++) {
        const uint32_t iter = rep * 100 + it;
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(2048), dim3(64), 0, 0, st, nb, iter, out);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(2048), dim3(64), 0, 0, st, nb, iter, out);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(2048), dim3(64), 0, 0, st, nb, iter, out);
        if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(2048), dim3(64), 0, 0, st, nb, iter, out);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms / 20 < best) best = ms / 20;
    }
    printf("%-50s %.1f us per 2^20 boards\n", names[mode], best * 1e3f);
  }
  return 0;
}
