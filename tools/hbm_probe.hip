// tools/hbm_probe.hip -- micro-benchmarks of the step kernel's memory skeleton on
// MI355X: what does the [chunk][board] stream cost with and without LDS staging,
// with partial (lane-masked) write-back, with LDS-DMA loads, with nt hints.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/hbm_probe tools/hbm_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int CH = 37;
constexpr int WAVE = 64;

__device__ __forceinline__ uint32_t mix(uint32_t h) {
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h;
}

// mode 0: copy through LDS, full write; 1: direct regs; 2: partial write (thr/256 of chunks per lane);
// 3: LDS-DMA loads; 4: nt loads + nt stores through LDS; 5: read only (write scalars chunk only)
template <int MODE>
__global__ __launch_bounds__(WAVE) void probe(uint4 *__restrict__ st, long bpad, uint32_t iter, uint32_t thr, int rev) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint4 *lds4 = reinterpret_cast<uint4 *>(lds);
  const int lane = threadIdx.x;
  const long blk = rev ? (long)(gridDim.x - 1 - blockIdx.x) : (long)blockIdx.x;
  const long board = blk * WAVE + lane;
  if constexpr (MODE == 1) {
    uint4 v[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) v[c] = st[(long)c * bpad + board];
#pragma unroll
    for (int c = 0; c < CH; c++) { v[c].x += 1; st[(long)c * bpad + board] = v[c]; }
    return;
  }
  if constexpr (MODE == 3) {
#pragma unroll
    for (int c = 0; c < CH; c++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)&st[(long)c * bpad + board],
                                       (__attribute__((address_space(3))) void *)&lds4[c * WAVE], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if constexpr (MODE == 4) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const uint4 *p = &st[(long)c * bpad + board];
      uint4 v;
      v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y);
      v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w);
      lds4[c * WAVE + lane] = v;
    }
  } else {
#pragma unroll
    for (int c = 0; c < CH; c++) lds4[c * WAVE + lane] = st[(long)c * bpad + board];
  }
  __syncthreads();
  lds[(36 * 64 + lane) * 4] += 1;  // touch the scalars chunk
  __syncthreads();
  if constexpr (MODE == 2) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const bool w = (c == CH - 1) || ((mix((uint32_t)board * 37u + c + iter * 0x9E3779B1u) & 255u) < thr);
      if (w) st[(long)c * bpad + board] = lds4[c * WAVE + lane];
    }
  } else if constexpr (MODE == 5) {
    st[(long)(CH - 1) * bpad + board] = lds4[(CH - 1) * WAVE + lane];
  } else if constexpr (MODE == 4) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      uint4 v = lds4[c * WAVE + lane];
      uint4 *p = &st[(long)c * bpad + board];
      __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
      __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CH; c++) st[(long)c * bpad + board] = lds4[c * WAVE + lane];
  }
}

// journal skeleton: read CH+2 chunks; every launch writes the scalars chunk and one journal
// chunk; launches with (iter % 8 == 7) write everything (compaction); ~6 % of lanes also
// write 3 scattered chunks (flood-fill dirt).  USE_DMA selects LDS-DMA loads.
constexpr int CHJ = CH + 2;
// TILED: the layout [batch of 64 boards][chunk][64] (one contiguous 39 KiB tile per wave)
// instead of [chunk][board] (39 rows of 1 KiB, 16 MiB apart).
template <bool USE_DMA, bool TILED = false>
__global__ __launch_bounds__(WAVE) void journal(uint4 *__restrict__ st_, long bpad, uint32_t iter, int rev) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint4 *lds4 = reinterpret_cast<uint4 *>(lds);
  const int lane = threadIdx.x;
  const long blk = rev ? (long)(gridDim.x - 1 - blockIdx.x) : (long)blockIdx.x;
  const long board = TILED ? (long)lane : blk * WAVE + lane;
  uint4 *__restrict__ st = TILED ? st_ + blk * (long)CHJ * WAVE : st_;
  if (TILED) bpad = WAVE;
  if constexpr (USE_DMA) {
#pragma unroll
    for (int c = 0; c < CHJ; c++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)&st[(long)c * bpad + board],
                                       (__attribute__((address_space(3))) void *)&lds4[c * WAVE], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
#pragma unroll
    for (int c = 0; c < CHJ; c++) lds4[c * WAVE + lane] = st[(long)c * bpad + board];
  }
  __syncthreads();
  lds[(36 * 64 + lane) * 4] += 1;
  lds[((37 + ((iter >> 2) & 1)) * 64 + lane) * 4 + (iter & 3)] = iter;
  __syncthreads();
  const int j = iter & 7;
  if (j == 7) {
#pragma unroll
    for (int c = 0; c < CHJ; c++) st[(long)c * bpad + board] = lds4[c * WAVE + lane];
  } else {
    st[(long)36 * bpad + board] = lds4[36 * WAVE + lane];
    const int jc = 37 + (j >> 2);
    st[(long)jc * bpad + board] = lds4[jc * WAVE + lane];
    const uint32_t h = mix((uint32_t)board + iter * 0x9E3779B1u);
    const bool dirty = (h & 255u) < 15u;  // 5.9 %
    if (__ballot(dirty)) {
      if (dirty) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int c = 18 + (int)((h >> (8 + 5 * k)) % 18u);
          st[(long)c * bpad + board] = lds4[c * WAVE + lane];
        }
      }
    }
  }
}

template <bool USE_DMA, bool TILED = false>
float run_journal(uint4 *st, long B, int reps, int pingpong) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t lds = CHJ * WAVE * 16;
  for (int i = 0; i < 8; i++) hipLaunchKernelGGL((journal<USE_DMA, TILED>), dim3(B / WAVE), dim3(WAVE), lds, 0, st, B, i, pingpong ? (i & 1) : 0);
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((journal<USE_DMA, TILED>), dim3(B / WAVE), dim3(WAVE), lds, 0, st, B, 8 + i, pingpong ? (i & 1) : 0);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

template <int MODE>
float run(uint4 *st, long B, uint32_t thr, int reps, int pingpong = 0) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t lds = (MODE == 1) ? 0 : CH * WAVE * 16;
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(probe<MODE>, dim3(B / WAVE), dim3(WAVE), lds, 0, st, B, i, thr, pingpong ? (i & 1) : 0);
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(probe<MODE>, dim3(B / WAVE), dim3(WAVE), lds, 0, st, B, 100 + i, thr, pingpong ? (i & 1) : 0);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const long B = 1 << 20;
  uint4 *st;
  CK(hipMalloc(&st, (size_t)CHJ * B * 16));
  CK(hipMemset(st, 1, (size_t)CHJ * B * 16));
  const double rd = (double)CH * B * 16;
  auto rep = [&](const char *name, float ms, double bytes) {
    printf("%-34s %8.4f ms  %8.1f GB/s moved  (%.2f G boards/s)\n", name, ms, bytes / ms / 1e6, B / ms / 1e6);
  };
  const int R = 50;
  rep("copy via LDS (full write)", run<0>(st, B, 0, R), 2 * rd);
  rep("copy via LDS, ping-pong order", run<0>(st, B, 0, R, 1), 2 * rd);
  rep("copy LDS-DMA, ping-pong order", run<3>(st, B, 0, R, 1), 2 * rd);
  rep("read only, ping-pong order", run<5>(st, B, 0, R, 1), rd + B * 16.0);
  {
    const double jb = (double)CHJ * B * 16 + B * 16.0 * (2 * 7 + CHJ) / 8.0 + 0.059 * 3 * 16 * B * 7 / 8.0;
    rep("journal K=8", run_journal<false>(st, B, 64, 0), jb);
    rep("journal K=8, ping-pong", run_journal<false>(st, B, 64, 1), jb);
    rep("journal K=8, tiled layout", (run_journal<false, true>(st, B, 64, 0)), jb);
    rep("journal K=8, tiled, ping-pong", (run_journal<false, true>(st, B, 64, 1)), jb);
    rep("journal K=8, LDS-DMA", run_journal<true>(st, B, 64, 0), jb);
    rep("journal K=8, LDS-DMA, ping-pong", run_journal<true>(st, B, 64, 1), jb);
  }
  rep("copy direct regs", run<1>(st, B, 0, R), 2 * rd);
  rep("copy LDS-DMA loads", run<3>(st, B, 0, R), 2 * rd);
  rep("copy nt loads+stores", run<4>(st, B, 0, R), 2 * rd);
  rep("read only (+1 chunk write)", run<5>(st, B, 0, R), rd + B * 16.0);
  for (uint32_t thr : {0u, 13u, 26u, 64u, 128u, 256u}) {
    char nm[64];
    snprintf(nm, sizeof nm, "partial write %3u/256 of chunks", thr);
    rep(nm, run<2>(st, B, thr, R), rd + B * 16.0 * (1 + 36 * thr / 256.0));
  }
  return 0;
}
