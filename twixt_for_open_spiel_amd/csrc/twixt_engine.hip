// twixt_engine.hip -- libtwixt_hip.so: batched TwixT board engine for MI355X
// (gfx950 / CDNA4).  Implements include/twixt_hip.h.
//
// Hot path (reference: twixtboard.cc:457-588 Board::ApplyAction /
// SetPegAndLinks / ExploreLocalGraph / UpdateResult, twixt.h:86-104
// LegalActions / DoApplyAction): one fused kernel launch = one env step on
// every board of a handle.
//   * step_window1_kernel<N> (twixt_window1_kernel.h, N >= 17, split layout): 64 boards per
//     wave, one lane per board; a persistent, software-pipelined wave streams its boards' pegs
//     and fetches of their links and flags only what the chosen move can touch;
//     step_window_full_kernel<N> (twixt_window_kernel.h): whole tiles, for small batches;
//   * step_kernel64[_small]<N,SUB[,SPLIT]> (twixt_step_kernel.h): 64 boards per wave,
//     one lane per board: N <= 16, listed-boards and fused play-out launches;
//   * step_pair_kernel<N> (twixt_pair_kernel.h, N >= 19, chunk layout, TWX_WINDOW=0): 32
//     boards per wave, two lanes per board -- round 2's kernel, kept for A/B runs.
//   The chunk-layout kernels: a persistent wave streams its boards (CHT 16-byte chunks each, tiles
//   of the state array, twixt_device_common.h) into LDS with coalesced 1 KiB
//   loads prefetched a batch ahead, works on them there, and writes back one
//   32-bit journal record per board -- the state is log-structured;
//   legal mask + k-th-legal-action pick are whole-plane popcount scans; the
//   8-direction link scan and the 72 crossing probes are constant-mask tests on
//   a 5x7 window of the link planes (twixt_geom.h kCross); the border-flag
//   flood fill (twixtboard.cc:573-588) is done by the wave cooperating on two
//   boards at a time (lane = board column, DPP shifts + ballot).
// Integer / bitwise work only: no MFMA, the bound is HBM bandwidth.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <type_traits>
#include <vector>

#include "../../include/twixt_hip.h"
#include "twixt_geom.h"
#include "twixt_text.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(TWX_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

// board sizes whose state is at most this many chunks use the several-waves-per-SIMD build
#ifndef TWX_SMALL_CHT
#define TWX_SMALL_CHT 25  // N <= 18: measured +19 % (N=17), +7 % (N=18); -3 % / -6 % at N=19 / 20
#endif
// board sizes whose state is MORE than this many chunks (N >= 17) keep it in the split layout and take the
// windowed whole-batch step (measured against the chunk layout's kernels at B = 2^20: +15 % at N=17, +14 % at
// N=18; -7..-24 % at N = 12..16, where the windowed kernel's fixed cost -- about 0.078 ms -- is above what
// streaming the whole board takes).  Their play-outs and listed-boards steps keep the small-board kernel.
#ifndef TWX_SPLIT_CHT
#define TWX_SPLIT_CHT 19
#endif
// board sizes whose state is at most this many chunks (N <= 16) write back with plain stores
#ifndef TWX_PLAIN_STORE_CHT
#define TWX_PLAIN_STORE_CHT 19
#endif
#include "twixt_device_common.h"
#include "twixt_step_kernel.h"
#include "twixt_pair_kernel.h"
#include "twixt_window_kernel.h"
#include "twixt_window1_kernel.h"
#include "twixt_util_kernels.h"

template <int N>
hipError_t launch_step(uint4 *state, long bpad, long batch, const int32_t *actions, int32_t *actions_out,
                       uint8_t *status, uint32_t seed, uint32_t first_id, int flags, int jpos, int waves,
                       const int32_t *ids, long nsub, int rsteps, int rpw, uint32_t *ticket, TicketBases ticket_base,
                       int nstatic, bool split, hipStream_t s) {
  const long nbatch = ids ? (nsub + rpw - 1) / rpw : bpad / kWave;  // rpw: rows of the list per wave
  const long grid = nbatch < (long)waves ? nbatch : (long)waves;
  const size_t lds_bytes = (size_t)twx::Geo<N>::CHT * kWave * 16;  // the wave's 64 boards: 39 KiB at N=24
#define TWX_LAUNCH(K)                                                                                      \
  hipLaunchKernelGGL((K), dim3((unsigned)grid), dim3(kWave), lds_bytes, s, state, bpad, batch, actions, \
                     actions_out, status, seed, first_id, flags, jpos, ids, nsub, rsteps, rpw, ticket, ticket_base, nstatic)
  if constexpr (twx::Geo<N>::CHT <= TWX_SMALL_CHT) {
    if constexpr (twx::Geo<N>::CHT > TWX_SPLIT_CHT) {
      if (split) {
        if (ids) TWX_LAUNCH((step_kernel64_small<N, true, true>)); else TWX_LAUNCH((step_kernel64_small<N, false, true>));
        return hipGetLastError();
      }
    }
    if (ids) TWX_LAUNCH((step_kernel64_small<N, true>)); else TWX_LAUNCH((step_kernel64_small<N, false>));
  } else if (split) {  // the handle's state is in the split layout (listed boards, play-outs)
    if (ids) TWX_LAUNCH((step_kernel64<N, true, true>)); else TWX_LAUNCH((step_kernel64<N, false, true>));
  } else {
    if (ids) TWX_LAUNCH((step_kernel64<N, true>)); else TWX_LAUNCH((step_kernel64<N, false>));
  }
#undef TWX_LAUNCH
  return hipGetLastError();
}

// resident waves per CU of the whole-batch one-lane-per-board kernel for this N (LDS image + registers)
template <int N>
int step_occupancy() {
  const size_t lds_bytes = (size_t)twx::Geo<N>::CHT * kWave * 16;
  int blocks = 0;
  hipError_t e;
  if constexpr (twx::Geo<N>::CHT <= TWX_SMALL_CHT)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, step_kernel64_small<N, false>, kWave, lds_bytes);
  else
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, step_kernel64<N, false>, kWave, lds_bytes);
  return (e == hipSuccess && blocks > 0) ? blocks : 0;
}

// two-lanes-per-board kernel (twixt_pair_kernel.h): whole-batch steps of the big boards
template <int N>
hipError_t launch_pair(uint4 *state, long bpad, long batch, const int32_t *actions, int32_t *actions_out,
                       uint8_t *status, uint32_t seed, uint32_t first_id, int flags, int jpos, int waves,
                       uint32_t *ticket, TicketBases ticket_base, int nstatic, hipStream_t s) {
  if constexpr (twx::Geo<N>::CHT > TWX_SMALL_CHT) {
    const long nbatch = bpad / 32;
    const long grid = nbatch < (long)waves ? nbatch : (long)waves;
    const size_t lds_bytes = (size_t)(twx::Geo<N>::CHT + 1) / 2 * 2 * 32 * 16;  // 20 KiB at N=24
    hipLaunchKernelGGL((step_pair_kernel<N>), dim3((unsigned)grid), dim3(kWave), lds_bytes, s, state, bpad, batch,
                       actions, actions_out, status, seed, first_id, flags, jpos, ticket, ticket_base, nstatic);
    return hipGetLastError();
  }
  return hipErrorInvalidConfiguration;
}
template <int N>
int pair_occupancy() {
  if constexpr (twx::Geo<N>::CHT > TWX_SMALL_CHT) {
    int blocks = 0;
    const size_t lds_bytes = (size_t)(twx::Geo<N>::CHT + 1) / 2 * 2 * 32 * 16;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, step_pair_kernel<N>, kWave, lds_bytes) != hipSuccess)
      return 0;
    return blocks;
  }
  return 0;
}
typedef hipError_t (*pair_fn)(uint4 *, long, long, const int32_t *, int32_t *, uint8_t *, uint32_t, uint32_t, int, int,
                              int, uint32_t *, TicketBases, int, hipStream_t);

// windowed step on the split layout (twixt_window_kernel.h): whole-batch steps of the big boards; the
// launches that rewrite everything (compactions, flushes) run the FULL instantiation
template <int N>
hipError_t launch_window(uint4 *state, long bpad, long batch, const int32_t *actions, int32_t *actions_out,
                         uint8_t *status, uint32_t seed, uint32_t first_id, int flags, int jpos, int waves,
                         uint32_t *ticket, TicketBases ticket_base, int nstatic, hipStream_t s) {
  if constexpr (twx::Geo<N>::CHT > TWX_SPLIT_CHT) {
    const long nbatch = bpad / 32;
    const long grid = nbatch < (long)waves ? nbatch : (long)waves;
    const bool full = (flags & kFlagWholeTile) != 0;
    const size_t lds_bytes = (size_t)WinGeo<N>::lds_bytes_full;
    char *st = reinterpret_cast<char *>(state);
    if (!full) {  // journal launches, compactions, flushes: one lane per board, 64 boards per wave (twixt_window1_kernel.h)
      const long nb64 = bpad / 64;
      const long grid1 = nb64 < (long)waves ? nb64 : (long)waves;
      hipLaunchKernelGGL((step_window1_kernel<N>), dim3((unsigned)grid1), dim3(kWave), (size_t)Win1Geo<N>::lds_bytes, s, st, bpad,
                         batch, actions, actions_out, status, seed, first_id, flags, jpos, ticket, ticket_base, nstatic);
    } else {  // small batches: every step rewrites whole tiles, two lanes per board (twixt_window_kernel.h)
      hipLaunchKernelGGL((step_window_full_kernel<N>), dim3((unsigned)grid), dim3(kWave), lds_bytes, s, st, bpad, batch,
                         actions, actions_out, status, seed, first_id, flags, jpos, ticket, ticket_base, nstatic);
    }
    return hipGetLastError();
  }
  return hipErrorInvalidConfiguration;
}
// resident waves per CU of the journal (0) / the rewriting (1) instantiation
template <int N>
int window_occupancy_1() {
  if constexpr (twx::Geo<N>::CHT > TWX_SPLIT_CHT) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, step_window1_kernel<N>, kWave, (size_t)Win1Geo<N>::lds_bytes) != hipSuccess)
      return 0;
    return blocks;
  }
  return 0;
}
template <int N>
int window_occupancy_f() {
  if constexpr (twx::Geo<N>::CHT > TWX_SPLIT_CHT) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, step_window_full_kernel<N>, kWave, (size_t)WinGeo<N>::lds_bytes_full) != hipSuccess)
      return 0;
    return blocks;
  }
  return 0;
}

typedef int (*occ_fn)();
typedef hipError_t (*step_fn)(uint4 *, long, long, const int32_t *, int32_t *, uint8_t *, uint32_t, uint32_t,
                              int, int, int, const int32_t *, long, int, int, uint32_t *, TicketBases, int, bool, hipStream_t);

template <int N>
struct StepTable {
  static void fill(step_fn *t, occ_fn *o, pair_fn *p, occ_fn *po, pair_fn *w, occ_fn *wf, occ_fn *w1) {
    t[N] = &launch_step<N>;
    o[N] = &step_occupancy<N>;
    p[N] = &launch_pair<N>;
    po[N] = &pair_occupancy<N>;
    w[N] = &launch_window<N>;
    wf[N] = &window_occupancy_f<N>;
    w1[N] = &window_occupancy_1<N>;
    StepTable<N - 1>::fill(t, o, p, po, w, wf, w1);
  }
};
template <>
struct StepTable<TWX_MIN_BOARD_SIZE - 1> {
  static void fill(step_fn *, occ_fn *, pair_fn *, occ_fn *, pair_fn *, occ_fn *, occ_fn *) {}
};

struct StepTables {
  step_fn step[TWX_MAX_BOARD_SIZE + 1];
  occ_fn occ[TWX_MAX_BOARD_SIZE + 1];
  pair_fn pair[TWX_MAX_BOARD_SIZE + 1];
  occ_fn pocc[TWX_MAX_BOARD_SIZE + 1];
  pair_fn window[TWX_MAX_BOARD_SIZE + 1];
  occ_fn wocc_f[TWX_MAX_BOARD_SIZE + 1], wocc_1[TWX_MAX_BOARD_SIZE + 1];
  StepTables() { StepTable<TWX_MAX_BOARD_SIZE>::fill(step, occ, pair, pocc, window, wocc_f, wocc_1); }
};
const StepTables &step_tables() {
  static const StepTables t;  // thread-safe static init
  return t;
}
step_fn step_for(int n) { return step_tables().step[n]; }

inline unsigned blocks_for(long threads, int bs) { return (unsigned)((threads + bs - 1) / bs); }

}  // namespace

struct twx_env {
  Dims d;
  int device;
  uint64_t first_id;
  uint4 *state;
  uint32_t *tables;  // region[2][W] then the initial board in internal chunk order
  // log-structured state: launches since the last compaction (= journal slots in
  // use, identical for every board), compaction period (1..kJournalSlots), and
  // the launch counter that alternates the block order (Infinity Cache reuse)
  int jpos, period;   // journal position (rolling: phase) of the next whole-batch launch
  bool rolling;       // rolling compaction (TWX_ROLLING)
  bool journal_live;  // some journal may hold records
  bool pingpong;
  uint64_t launches;
  int waves;  // persistent grid: resident waves = CUs x waves the LDS image admits per CU
  int init_off;    // word offset of the initial-position chunks inside `tables`
  bool pair;       // whole-batch launches go to step_pair_kernel (two lanes per board; big boards)
  int waves_pair;  // its persistent grid
  // split layout (d.lay = 1): whole-batch launches go to step_window_kernel; persistent grids of its journal
  // and of its rewriting instantiation
  int waves_win, waves_win_full;
  bool win_one_lane;  // journal launches: step_window1_kernel (one lane per board, 64 boards per wave)
  bool small_rewrite;  // small batches skip the journal (TWX_SMALL_BATCH_REWRITE, default 1; 0: tests of the journal path)
  int cus;         // compute units of the device
  // dynamic batch order (TWX_DYNAMIC): device ticket counters (never reset) and the values
  // they will have when the next launch starts
  bool dynamic;
  int dyn_static;  // batches every wave takes in static order before it draws tickets (0 = automatic)
  uint32_t *ticket;
  TicketBases ticket_base;
};

namespace {

// Journal period by board size (launches between two whole-board writes; measured at B = 2^20,
// tools/sweep_small.sh, profiles/r02_small_board_period_sweep.txt): the journal trades write bytes
// for replay work, which pays where the board stream is the bound.  N <= 9 (bound by the integer
// work, at most 112 B of state) rewrite the planes every step, N = 10..12 every second, N = 13..17
// every fourth, larger boards every eighth step.
int default_period(int n) {
  const int W = twx::plane_words(n);
  const int p = W <= 3 ? 1 : W <= 5 ? 2 : W <= 10 ? 4 : 8;
  return p < twx::kJournalSlots ? p : twx::kJournalSlots;
}

int env_int(const char *name, int dflt, int lo, int hi) {
  const char *v = getenv(name);
  if (!v || !*v) return dflt;
  const int x = atoi(v);
  return x < lo ? lo : x > hi ? hi : x;
}

// a handle's work always goes to the handle's device, whatever the caller's current one
// is; the caller's current device is put back when the entry point returns
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  hipError_t err;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      changed = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define TWX_USE(e)                     \
  DeviceGuard device_guard_((e)->device); \
  HIP_TRY(device_guard_.err)

// launch arguments change from call to call on an ordinary handle (journal position, batch
// order, ticket bases): a captured graph would replay stale ones and silently skip or repeat
// boards, or read stale journal slots
int refuse_capture(const twx_env *e, hipStream_t s) {
  if (e->period > 1 || e->pingpong || e->dynamic) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return fail(TWX_E_ARG, "step and status functions cannot be captured into a hipGraph on this handle: create "
                             "it with TWX_GRAPH_SAFE=1");
  }
  return TWX_OK;
}

// one env step (or a flush) on every board, or on the `nsub` boards listed in `ids`;
// maintains the journal position
int run_step(twx_env *e, const int32_t *actions, int32_t *actions_out, uint8_t *status, uint32_t seed,
             int flags, const int32_t *ids, long nsub, hipStream_t s, int rsteps = 1) {
  TWX_USE(e);
  const bool flush = (flags & kFlagFlush) != 0;
  const bool whole_step = !ids && rsteps == 1 && !flush;
  if (int rc = refuse_capture(e, s)) return rc;
  // launches that rewrite their boards in place (no journaling): subset and roll-out launches, period 1, and
  // the small whole-batch steps of the split layout.  They work on materialised planes.
  const bool rewrite = !flush && (!whole_step || e->period == 1 ||
                                  (e->d.lay != 0 && e->small_rewrite && (e->d.bpad / 64) * 4 <= (long)e->waves_win));
  if (rewrite && e->journal_live) {
    if (int rc = run_step(e, nullptr, nullptr, nullptr, 0u, kFlagFlush, nullptr, 0, s)) return rc;
  }
  int jarg = e->jpos;
  if (flush) {
    flags |= kFlagCompact;  // replay what is there, write everything, clear the journals
  } else if (rewrite) {
    // no journaling: planes rewritten in place.  (Split layout, small batches -- a search's leaves: with fewer
    // 64-board batches than a quarter of the resident waves a launch is one wave's latency, not bandwidth, and
    // the pipelined journal kernel's extra iterations cost more than the bytes it saves: 15.3 against 10.5 us
    // per step at N=24, B=16,384.)
    flags |= kFlagCompact;
    jarg = 0;
    if (e->d.lay != 0 && whole_step && e->small_rewrite && (e->d.bpad / 64) * 4 <= (long)e->waves_win) flags |= kFlagWholeTile;
  } else if (!e->rolling && e->jpos == e->period - 1) {
    flags |= kFlagCompact;
  }
  if ((whole_step || flush) && e->rolling && e->period > 1)
    flags |= kFlagRolling | (e->period << 8) | ((e->waves > 0xffff ? 0xffff : e->waves) << 12);
  if (e->pingpong && !ids && (e->launches & 1)) flags |= kFlagReverse;
  {
    // whole-batch launches of the big boards: the two-lanes-per-board kernel
    const bool use_win = e->d.lay != 0 && !ids && rsteps == 1;
    const bool use_pair = use_win || (e->pair && !ids && rsteps == 1 && !(flags & kFlagRolling));
    int bpw = (use_pair && !(use_win && !(flags & kFlagWholeTile))) ? 32 : kWave;  // boards per wave
    const int waves = use_win ? ((flags & kFlagWholeTile) ? e->waves_win_full : e->waves_win) : use_pair ? e->waves_pair : e->waves;
    // a short list of boards is spread over many waves (rows per wave: a power of two, at least 4,
    // while that still fills four waves per CU): see step_body
    if (ids)
      while (bpw > 4 && (nsub + bpw - 1) / bpw < 4L * e->cus) bpw >>= 1;
    // dynamic tail: the groups need equal numbers of waves, and there must be batches beyond
    // the first round; by default the last eighth of the rounds (at least two) is dynamic
    const long nbatch = ids ? (nsub + bpw - 1) / bpw : e->d.bpad / bpw;
    const long grid = nbatch < (long)waves ? nbatch : (long)waves;
    // ... and the draws must stay well below what a counter sustains (~75 M tickets/s): a
    // launch draws about 4 * grid / kTicketGroups tickets per counter in (nbatch / grid) batch
    // times of ~10 us, so the tail is dynamic only while grid^2 <= 300 * nbatch.  Measured at
    // B=2^20: +5..7 % for N >= 12 (2,048 waves and fewer), -12 % / -29 % at N=8 / N=5 (3,072 /
    // 4,096 waves), which this rule keeps static.  (With eight ticket groups instead of four the
    // bound could be twice as high; measured: +2 % at N=6, -5 % at N=9, nothing elsewhere: kept.)
    const bool dyn = e->dynamic && grid % kTicketGroups == 0 && nbatch > grid && grid * grid <= 300 * nbatch;
    const long rounds = nbatch / grid;  // whole rounds
    // (the pair kernel's waves hold half as many boards, so one long flood fill weighs more: the
    // last quarter of its rounds is drawn -- with every batch on its XCD a quarter is as good as a half
    // at N=24 and +1.8 % at N=20)
    const long ndyn = use_pair ? (rounds / 4 > 2 ? rounds / 4 : 2) : (rounds / 8 > 2 ? rounds / 8 : 2);
    long nstatic = e->dyn_static > 0 ? e->dyn_static : rounds - ndyn;
    if (nstatic > rounds) nstatic = rounds;
    if (nstatic < 1) nstatic = 1;
    if (dyn) flags |= kFlagDynamic;
    if (use_win)
      HIP_TRY(step_tables().window[e->d.n](e->state, e->d.bpad, e->d.batch, actions, actions_out, status, seed,
                                           (uint32_t)e->first_id, flags, jarg, waves, e->ticket, e->ticket_base,
                                           (int)nstatic, s));
    else if (use_pair)
      HIP_TRY(step_tables().pair[e->d.n](e->state, e->d.bpad, e->d.batch, actions, actions_out, status, seed,
                                         (uint32_t)e->first_id, flags, jarg, waves, e->ticket, e->ticket_base,
                                         (int)nstatic, s));
    else
      HIP_TRY(step_for(e->d.n)(e->state, e->d.bpad, e->d.batch, actions, actions_out, status, seed,
                               (uint32_t)e->first_id, flags, jarg, e->waves, ids, nsub, rsteps, bpw, e->ticket,
                               e->ticket_base, (int)nstatic, e->d.lay != 0, s));
    if (dyn) {
      // group g: its dynamic batches (idx >= nstatic * grid, idx % groups == g) + two draws per
      // wave that come back past the end (wraps with the counter)
      const long first = nstatic * grid;
      for (int g = 0; g < kTicketGroups; g++) {
        const long nd = nbatch > first + g ? (nbatch - first - g + kTicketGroups - 1) / kTicketGroups : 0;
        e->ticket_base.b[g] += (uint32_t)(nd + 2 * (grid / kTicketGroups));
      }
    }
  }
  if (flush) {
    e->journal_live = false;
    if (!e->rolling) e->jpos = 0;  // (rolling: the phase just keeps running over empty journals)
  } else if (whole_step && e->period > 1) {
    if (e->rolling) {
      e->jpos = (e->jpos + 1) % e->period;
      e->journal_live = true;
    } else {
      e->jpos = (flags & kFlagCompact) ? 0 : e->jpos + 1;
      e->journal_live = e->jpos != 0;
    }
  }
  e->launches++;
  return TWX_OK;
}

// every entry point other than the step functions sees materialised planes
int materialise(const twx_env *ce, hipStream_t s) {
  twx_env *e = const_cast<twx_env *>(ce);
  if (!e->journal_live) return TWX_OK;
  return run_step(e, nullptr, nullptr, nullptr, 0u, kFlagFlush, nullptr, 0, s);
}

}  // namespace

#ifdef TWX_STAMPS
extern "C" int twx_debug_stamps(unsigned long long *out, int reset) {
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16));
  if (reset) {
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z));
  }
  return TWX_OK;
}
#endif

extern "C" {

const char *twx_last_error(void) { return g_err.c_str(); }

// Host-only text helpers (no device work): Board::ToString / ActionToString
int twx_cells_to_string(const uint8_t *cells_host, int n, int ansi, int swapped, int result, char *buf,
                        int cap) {
  if (!cells_host || n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE || (cap > 0 && !buf))
    return fail(TWX_E_ARG, "twx_cells_to_string: bad argument");
  const std::string s = twx::text::BoardToString(cells_host, n, ansi != 0, swapped != 0, result);
  if (cap > 0) {
    const size_t k = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
    memcpy(buf, s.data(), k);
    buf[k] = 0;
  }
  return (int)s.size();
}

int twx_action_to_string(int n, int player, int64_t action, char *buf, int cap) {
  if (!buf || cap < 8 || n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE || action < 0 || action >= n * n)
    return fail(TWX_E_ARG, "twx_action_to_string: bad argument");
  const std::string s = twx::text::ActionToString(n, player, (long)action);
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

int twx_state_words(int n) {
  if (n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE) return TWX_E_SIZE;
  return twx::state_words(n);
}

int twx_mask_words(int n) {
  if (n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE) return TWX_E_SIZE;
  return twx::plane_words(n);
}

int twx_create(int n, int64_t batch, int device, uint64_t first_board_id, twx_env **out) {
  if (!out) return fail(TWX_E_ARG, "twx_create: out is NULL");
  *out = nullptr;
  if (n < TWX_MIN_BOARD_SIZE || n > TWX_MAX_BOARD_SIZE)  // tw.cc:139-144
    return fail(TWX_E_SIZE, "board_size out of range [5..24]: " + std::to_string(n));
  if (batch <= 0) return fail(TWX_E_ARG, "twx_create: batch must be positive");
  // global board ids are 32-bit keys of the policy hash (twixt_geom.h policy_hash): a shard whose
  // ids would wrap is refused rather than silently aliased onto other boards' trajectories
  if (first_board_id > 0xffffffffull || first_board_id + (uint64_t)batch > 0x100000000ull)
    return fail(TWX_E_ARG, "twx_create: first_board_id + batch must not exceed 2^32 (board ids are 32-bit policy keys)");
  DeviceGuard device_guard_(device);
  HIP_TRY(device_guard_.err);
  int cus = 0;
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  twx_env *e = new twx_env();
  e->cus = cus > 0 ? cus : 256;
  e->d.n = n;
  e->d.W = twx::plane_words(n);
  e->d.SW = twx::state_words(n);
  e->d.CH = twx::state_chunks(n);
  e->d.CHT = twx::total_chunks(n);
  e->d.ts = tile_shift(e->d.CHT);
  // the big boards (whole-batch step = two lanes per board) live in the split layout and step through
  // step_window_kernel; TWX_WINDOW=0 keeps them in the chunk layout on step_pair_kernel, TWX_PAIR=0 on the
  // one-lane kernel (both for A/B runs and for the options only that kernel has: rolling compaction)
  e->d.lay = (e->d.CHT > TWX_SPLIT_CHT && env_int("TWX_WINDOW", 1, 0, 1) != 0 && env_int("TWX_PAIR", 1, 0, 1) != 0 &&
              env_int("TWX_ROLLING", 0, 0, 1) == 0 && step_tables().wocc_1[n]() > 0 && step_tables().wocc_f[n]() > 0)
                 ? 1 : 0;
  e->jpos = 0;
  e->period = env_int("TWX_JOURNAL_PERIOD", default_period(n), 1, twx::kJournalSlots);
  e->pingpong = env_int("TWX_PINGPONG", 1, 0, 1) != 0;
  e->rolling = env_int("TWX_ROLLING", 0, 0, 1) != 0;  // measured 10 % slower than compacting all at once
  e->journal_live = false;
  e->launches = 0;
  e->dynamic = env_int("TWX_DYNAMIC", 1, 0, 1) != 0;
  e->dyn_static = env_int("TWX_DYNAMIC_STATIC", 0, 0, 1 << 20);  // 0 = automatic
  if (env_int("TWX_GRAPH_SAFE", 0, 0, 1)) {
    // no host-side launch counters: every step launch has constant arguments, so captured
    // hipGraphs of steps may be replayed (the planes are rewritten in place every step)
    e->period = 1;
    e->pingpong = false;
    e->dynamic = false;
  }
  e->ticket = nullptr;
  e->ticket_base = TicketBases{};
  {
    // as many waves as are resident at once (LDS image and registers; 4 per CU at N=24,
    // 8 at N=12, 12 at N=8): each walks its share of the batches
    int per_cu = step_tables().occ[n]();
    if (per_cu < 1) per_cu = 4;
    if (per_cu > 16) per_cu = 16;
    // N = 12 (13 chunk rows) fits three waves per SIMD by registers and LDS, and runs 22 % SLOWER with
    // them than with two (0.0762 vs 0.0623 ms per 2^20-board step; N = 10, 11 gain 8 % from the third
    // wave, N >= 13 do not fit it): tools/sweep_waves.sh, profiles/r02_waves_per_cu_sweep.txt
    if (e->d.CHT >= 13 && per_cu > 8) per_cu = 8;
    e->waves = (cus > 0 ? cus : 256) * env_int("TWX_WAVES_PER_CU", per_cu, 1, 32);
    const int pocc = step_tables().pocc[n]();
    e->pair = pocc > 0 && env_int("TWX_PAIR", 1, 0, 1) != 0;
    e->waves_pair = (cus > 0 ? cus : 256) * env_int("TWX_PAIR_WAVES_PER_CU", pocc > 0 ? pocc : 1, 1, 32);
    e->waves_win = e->waves_win_full = 0;
    e->win_one_lane = false;
    e->small_rewrite = env_int("TWX_SMALL_BATCH_REWRITE", 1, 0, 1) != 0;
    if (e->d.lay) {
      e->win_one_lane = true;
      e->waves_win = (cus > 0 ? cus : 256) * env_int("TWX_WIN_WAVES_PER_CU", step_tables().wocc_1[n](), 1, 32);
      e->waves_win_full = (cus > 0 ? cus : 256) * env_int("TWX_WIN_FULL_WAVES_PER_CU", step_tables().wocc_f[n](), 1, 32);
    }
  }
  e->d.batch = batch;
  e->d.bpad = (batch + kWave - 1) / kWave * kWave;
  e->device = device;
  e->first_id = first_board_id;
  e->state = nullptr;
  e->tables = nullptr;
  const size_t bytes = e->d.lay ? (size_t)(e->d.bpad / 32) * (size_t)split_tile_bytes(e->d.W)
                                : (size_t)e->d.CHT * (size_t)e->d.bpad * sizeof(uint4);
  if (hipMalloc(&e->state, bytes) != hipSuccess) {
    delete e;
    return fail(TWX_E_NOMEM, "twx_create: hipMalloc of " + std::to_string(bytes) + " bytes failed");
  }
  // tables: region[2][W] (padded to a multiple of 4 words), then the initial position in
  // internal chunk order (CHT uint4)
  e->init_off = (2 * e->d.W + 3) & ~3;
  // (+ zero words for the ticket counters of the dynamic batch order)
  std::vector<uint32_t> canon(2 * e->d.W + e->d.SW), tab(e->init_off + 4 * e->d.CHT + kTicketGroups * kTicketStride, 0u);
  twx::fill_tables(n, canon.data(), canon.data() + 2 * e->d.W);
  for (int i = 0; i < 2 * e->d.W; i++) tab[i] = canon[i];
  for (int p = 0; p < twx::kNumPlanes; p++)
    for (int i = 0; i < e->d.W; i++)
      tab[e->init_off + 4 * (2 * i + (p >> 2)) + (p & 3)] = canon[2 * e->d.W + p * e->d.W + i];
  if (hipMalloc(&e->tables, tab.size() * 4) != hipSuccess ||
      hipMemcpy(e->tables, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(e->state);
    if (e->tables) (void)hipFree(e->tables);
    delete e;
    return fail(TWX_E_NOMEM, "twx_create: table upload failed");
  }
  e->ticket = e->tables + tab.size() - kTicketGroups * kTicketStride;
  if (twx_reset(e, nullptr) != TWX_OK || hipStreamSynchronize(nullptr) != hipSuccess) {
    const std::string why = g_err;
    twx_destroy(e);
    return fail(TWX_E_DEVICE, "twx_create: initial reset failed: " + why);
  }
  *out = e;
  return TWX_OK;
}

int twx_destroy(twx_env *e) {
  if (!e) return TWX_OK;
  DeviceGuard device_guard_(e->device);
  (void)hipFree(e->state);
  (void)hipFree(e->tables);
  delete e;
  return TWX_OK;
}

int twx_board_size(const twx_env *e) { return e ? e->d.n : TWX_E_ARG; }
int64_t twx_batch(const twx_env *e) { return e ? e->d.batch : TWX_E_ARG; }

int twx_reset(twx_env *e, void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_reset: env is NULL");
  TWX_USE(e);
  e->jpos = 0;  // every journal is cleared below
  e->journal_live = false;
  const long threads = (long)e->d.CHT * e->d.bpad;
  hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, e->tables + e->init_off, (const int32_t *)nullptr, 0L);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_reset_boards(twx_env *e, const int32_t *ids, int64_t n, void *stream) {
  if (!e || !ids || n < 0) return fail(TWX_E_ARG, "twx_reset_boards: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  const long threads = (long)e->d.CH * n;  // planes and scalars: the journal rows are shared with the tile's other boards
  hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, e->tables + e->init_off, ids, (long)n);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_step(twx_env *e, const int32_t *actions, uint8_t *status, void *stream) {
  if (!e || !actions) return fail(TWX_E_ARG, "twx_step: env or actions is NULL");
  return run_step(e, actions, nullptr, status, 0u, 0, nullptr, 0, (hipStream_t)stream);
}

int twx_step_boards(twx_env *e, const int32_t *ids, int64_t n, const int32_t *actions, uint8_t *status,
                    void *stream) {
  if (!e || !ids || !actions || n < 0) return fail(TWX_E_ARG, "twx_step_boards: bad argument");
  if (n == 0) return TWX_OK;
  return run_step(e, actions, nullptr, status, 0u, 0, ids, (long)n, (hipStream_t)stream);
}

int twx_step_random_boards(twx_env *e, const int32_t *ids, int64_t n, uint32_t seed, int auto_reset,
                           int32_t *actions_out, uint8_t *status, void *stream) {
  if (!e || !ids || n < 0) return fail(TWX_E_ARG, "twx_step_random_boards: bad argument");
  if (n == 0) return TWX_OK;
  const int flags = kFlagRandom | (auto_reset ? kFlagAutoReset : 0);
  return run_step(e, nullptr, actions_out, status, seed, flags, ids, (long)n, (hipStream_t)stream);
}

int twx_random_rollout(twx_env *e, const int32_t *ids, int64_t n, uint32_t seed, int max_steps, uint8_t *status,
                       void *stream) {
  if (!e || n < 0 || max_steps < 1 || (!ids && n != 0 && n != e->d.batch))
    return fail(TWX_E_ARG, "twx_random_rollout: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (int rc = materialise(e, s)) return rc;  // the rollout works on materialised planes
  if (ids && n == 0) return TWX_OK;
  // one launch: every board plays on, out of LDS, until it is terminal or max_steps moves
  // were made; written back as a compaction (journal stays empty)
  return run_step(e, nullptr, nullptr, status, seed, kFlagRandom | kFlagCompact, ids, ids ? (long)n : 0, s, max_steps);
}

int twx_step_random(twx_env *e, uint32_t seed, int auto_reset, int32_t *actions_out, uint8_t *status,
                    void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_step_random: env is NULL");
  const int flags = kFlagRandom | (auto_reset ? kFlagAutoReset : 0);
  return run_step(e, nullptr, actions_out, status, seed, flags, nullptr, 0, (hipStream_t)stream);
}

int twx_legal_mask_boards(const twx_env *e, const int32_t *ids, int64_t n, uint32_t *mask, int32_t *count,
                          void *stream) {
  if (!e || !mask || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_legal_mask: bad argument");
  if (n == 0) return TWX_OK;
  hipStream_t s = (hipStream_t)stream;
  TWX_USE(e);
  if (int rc = materialise(e, s)) return rc;
  if (count) HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t) * n, s));
  const long threads = (long)e->d.W * n;
  hipLaunchKernelGGL(legal_mask_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, s, e->state, e->d,
                     e->tables, ids, (long)n, mask, count);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_legal_mask(const twx_env *e, uint32_t *mask, int32_t *count, void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_legal_mask: env is NULL");
  return twx_legal_mask_boards(e, nullptr, e->d.batch, mask, count, stream);
}

int twx_status_boards(const twx_env *e, const int32_t *ids, int64_t n, uint8_t *result, int32_t *move_counter,
                      int8_t *current_player, void *stream) {
  if (!e || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_status: bad argument");
  if (n == 0) return TWX_OK;
  // (no materialisation needed: the scalars follow from the stored ones and the journal records)
  TWX_USE(e);
  if (int rc = refuse_capture(e, (hipStream_t)stream)) return rc;
  hipLaunchKernelGGL(status_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, e->state, e->d,
                     e->rolling ? twx::kJournalSlots : e->jpos, ids, (long)n, result, move_counter, current_player);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_status(const twx_env *e, uint8_t *result, int32_t *move_counter, int8_t *current_player,
               void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_status: env is NULL");
  return twx_status_boards(e, nullptr, e->d.batch, result, move_counter, current_player, stream);
}

int twx_observation_boards(const twx_env *e, const int32_t *ids, int64_t n, float *out, void *stream) {
  if (!e || !out || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_observation: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  hipLaunchKernelGGL(observation_kernel, dim3((unsigned)n), dim3(kWave), 0, (hipStream_t)stream, e->state, e->d,
                     ids, out);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_observation(const twx_env *e, float *out, void *stream) {
  if (!e) return fail(TWX_E_ARG, "twx_observation: env is NULL");
  return twx_observation_boards(e, nullptr, e->d.batch, out, stream);
}

int twx_copy_boards(twx_env *dst, const int32_t *dst_ids, const twx_env *src, const int32_t *src_ids,
                    int64_t n, void *stream) {
  if (!dst || !src || !dst_ids || !src_ids || n < 0) return fail(TWX_E_ARG, "twx_copy_boards: bad argument");
  if (dst->d.n != src->d.n) return fail(TWX_E_ARG, "twx_copy_boards: board sizes differ");
  if (n == 0) return TWX_OK;
  if (dst->device != src->device) return fail(TWX_E_ARG, "twx_copy_boards: handles live on different devices");
  TWX_USE(dst);
  if (int rc = materialise(dst, (hipStream_t)stream)) return rc;
  if (int rc = materialise(src, (hipStream_t)stream)) return rc;
  const long threads = (long)dst->d.CH * n;
  hipLaunchKernelGGL(copy_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     dst->state, dst->d, dst_ids, src->state, src->d, src_ids, (long)n);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_export_state(const twx_env *e, const int32_t *ids, int64_t n, uint32_t *out, void *stream) {
  if (!e || !out || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_export_state: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  const long threads = (long)e->d.SW * n;
  hipLaunchKernelGGL(export_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, ids, (long)n, out);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_import_state(twx_env *e, const int32_t *ids, int64_t n, const uint32_t *in, void *stream) {
  if (!e || !in || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_import_state: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  const long threads = (long)e->d.CH * n;
  hipLaunchKernelGGL(import_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, ids, (long)n, in);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

int twx_export_cells(const twx_env *e, const int32_t *ids, int64_t n, uint8_t *out, void *stream) {
  if (!e || !out || n < 0 || (!ids && n > e->d.batch)) return fail(TWX_E_ARG, "twx_export_cells: bad argument");
  if (n == 0) return TWX_OK;
  TWX_USE(e);
  if (int rc = materialise(e, (hipStream_t)stream)) return rc;
  const long threads = (long)e->d.n * e->d.n * n;
  hipLaunchKernelGGL(cells_kernel, dim3(blocks_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                     e->state, e->d, ids, (long)n, out);
  HIP_TRY(hipGetLastError());
  return TWX_OK;
}

}  // extern "C"
