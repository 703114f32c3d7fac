#!/usr/bin/env python3
"""bench.py -- env steps/sec of the batched TwixT step kernel on MI355X.

Metric (BASELINE.json): env steps/sec on random-action rollouts at board_size=24,
batch=2^20 per GPU.  One bench "step" = one twx_step_random launch = one env step
on every board of the batch (terminal boards auto-reset and play move 0 in the
same launch, so a launch is exactly B env steps).  State lives in HBM between
launches (step-granular model).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: boards never interact, so rank r owns the contiguous
global board ids [r*B, (r+1)*B) on GPU r and there is NO collective on the data path and
none inside the timed region.  `--gpus N` without a launcher (WORLD_SIZE unset) makes this
script start the N rank processes itself, before it has touched a GPU; under
torch.distributed.run it is one of the ranks.  Either way the ranks meet on a CPU (gloo)
group for the start / end barriers and for handing their clocks to rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {24: 1320, 12: 348, 8: 168}  # 2*S(N)+8
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0                     # same guide: measured float4-copy ceiling
MIN_TIMED_MS = 50.0                       # the K-step block is repeated until this much GPU time
TRAFFIC_FILE = os.path.join("profiles", "traffic_step_kernel_n24.json")
# Region words a move stores on the big boards, bytes requested per env step (mean over random N=24 games:
# about one link per move -- 16 B when the new peg is a link's west end, 4 B at the other peg otherwise --
# and 8 B of flags for one move in eight): a model constant, the PMC write count is the measurement.
REGION_UPDATE_BYTES = 10.0
# Window bytes a move requests, as a share of the full window of 16 NPC bytes (mean over the oracle's random N=24
# games): 0.698 of the moves have a peg of their colour a knight's move away -- the others fetch nothing -- and
# those request the words of the columns their candidates reach, 102 B on average: 71.4 B per move of 160.
WINDOW_SHARE = 0.446


def journal_period(n):
    """twixt_engine.hip default_period(): launches between two whole-board writes."""
    w = (n * n + 31) // 32
    return 1 if w <= 3 else 2 if w <= 5 else 4 if w <= 10 else 8


def algo_bytes(n):
    return ALGO_BYTES.get(n, 2 * (9 * ((n * n + 7) // 8) + 8) + 8)


def split_layout(n):
    """Big boards (N >= 17, TWX_SPLIT_CHT): split layout, windowed step (twixt_window1_kernel.h)."""
    return 2 * ((n * n + 31) // 32) + 3 > 19


def step_kernel_name(n):
    """The kernel every whole-batch step launch runs at board size n (twixt_engine.hip run_step); on the big
    boards journal launches and compactions are the same kernel (a compaction also stores pegs and scalars)."""
    if split_layout(n):
        return "step_window1_kernel<%d>" % n
    return "step_kernel64_small<%d, false>" % n


def window_geometry(n):
    """WinGeo<N> of twixt_window_kernel.h: words per plane, peg rows, bytes of a board's links + flags
    region, words and 16-byte pieces of a move's window."""
    w = (n * n + 31) // 32
    rp = (w + 1) // 2
    lfb = (24 * w + 15) // 16 * 16
    nw = min((6 * n + 62) // 32, w)
    npc = min((24 * nw + 8 + 15) // 16, lfb // 16)
    return {"W": w, "RP": rp, "LFB": lfb, "NW": nw, "NPC": npc}


def design_bytes_split(n, multi_cell_fills=0.028, region_update=REGION_UPDATE_BYTES, windows=WINDOW_SHARE):
    """Bytes the windowed step requests per env step by construction, big boards.
    A launch with j records to replay reads pegs (16 RP) + scalars (16) + the records (4 j) + the window
    (`windows` x 16 NPC: only a move with a neighbour to link to fetches, and only the columns it can touch)
    + the whole region of the boards with a multi-cell fill, and writes the region words its move
    changes (`region_update`: the cell's link words, flag pairs, the far ends' link words -- a mean over random
    games) plus its record (4 B); the P-th launch of a period, the compaction, writes pegs and scalars instead
    of a record.  Returns (per step averaged over a period, journal launch mean, compaction launch).  The
    memory side moves whole 128-byte lines: a window of 160 B at a 16-byte offset touches 2.1 of them on
    average and a region update dirties 32-byte sectors, so the PMC traffic is above this request count."""
    g, p = window_geometry(n), journal_period(n)
    rows = 16 * g["RP"] + 16
    reads = [rows + 4 * j + windows * 16 * g["NPC"] + multi_cell_fills * g["LFB"] for j in range(p)]
    journal = [r + region_update + 4 for r in reads[:p - 1]]
    compaction = reads[p - 1] + region_update + rows
    jmean = sum(journal) / max(1, len(journal))
    return (sum(journal) + compaction) / p, jmean, compaction


def design_bytes(n):
    """Bytes this design moves per env step by construction, journal period
    P = journal_period(n): every launch reads the board's CH = 2W+1 chunks of 16 B (planes and
    scalars) and the journal records of the running period (4 B per board and record: launch j of a
    period fetches j of them); P-1 launches in P write one journal record (4 B: it carries the
    scalars), the P-th rewrites the CH chunks.  (Flag chunks a multi-cell flood fill dirtied,
    auto-reset boards, and the 5 B of actions + status a step reports add a few bytes.)"""
    if split_layout(n):
        return design_bytes_split(n)[0]
    ch = 2 * ((n * n + 31) // 32) + 1
    p = journal_period(n)
    return 16.0 * ch + 4.0 * sum(range(p)) / p + ((p - 1) * 4.0 + 16.0 * ch) / p


def host_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup quota."""
    cores = len(os.sched_getaffinity(0))
    try:  # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = min(cores, max(1, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = min(cores, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(n, seed, budget_s=18.0):
    """Times the CPU oracle (the scalar port of the reference's algorithm) on all
    host cores with the same policy and workload shape.  Checker code used as a
    *reported baseline only*; its budget does not depend on --steps."""
    lib_path = os.path.join(ROOT, "oracle", "libtwixt_oracle.so")
    if not os.path.exists(lib_path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libtwixt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    lib = C.CDLL(lib_path)
    lib.orc_bench.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    lib.orc_bench.restype = C.c_int64
    cores = int(os.environ.get("TWX_CPU_THREADS", "0")) or host_cores()
    dg = C.c_uint64()
    nsteps = 600
    t0 = time.perf_counter()
    s = lib.orc_bench(n, seed, 16 * cores, nsteps, cores, C.byref(dg))
    dt = time.perf_counter() - t0
    rate = s / dt
    nboards = max(16 * cores, int(rate * budget_s / nsteps) // cores * cores)
    t0 = time.perf_counter()
    s = lib.orc_bench(n, seed, nboards, nsteps, cores, C.byref(dg))
    dt = time.perf_counter() - t0
    return {"value": s / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "%d boards x %d random-policy steps (auto-reset), N=%d, %d OpenMP threads, %.1f s"
                      % (nboards, nsteps, n, cores, dt)}


def active_knobs():
    """TWX_* tuning knobs twx_create would read from this environment (they change what the
    kernel moves, so a committed traffic measurement does not describe such a run)."""
    return sorted(k for k in os.environ if k.startswith("TWX_") and k != "TWX_CPU_THREADS")


def measured_traffic(n, batch):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC passes
    (profiles/traffic_step_kernel_n24.json, written by tools/pmc_traffic.py: FETCH_SIZE x2 per
    the gfx950 correction + WRITE_SIZE).  Counters cannot be collected inside this process, so
    the file is only believed when it was measured on THIS library: it carries the hash of
    the sources (and of the binary) it was taken on, the workload and the knobs.  Returns
    (bytes or None, provenance dict)."""
    from twixt_for_open_spiel_amd import build as b
    src = {"file": TRAFFIC_FILE, "source_stamp_now": b.source_stamp(), "lib_sha256_now": b.lib_stamp()}
    path = os.path.join(ROOT, TRAFFIC_FILE)
    try:
        with open(path) as f:
            js = json.load(f)
    except (OSError, ValueError):
        src["rejected"] = "file missing or unreadable"
        return None, src
    src["source_stamp_measured"] = js.get("source_stamp")
    src["lib_sha256_measured"] = js.get("lib_sha256")
    src["measured_by"] = js.get("method")
    if js.get("board_size") != n or js.get("batch") != batch:
        src["rejected"] = "measured for N=%s, B=%s" % (js.get("board_size"), js.get("batch"))
    elif active_knobs():
        src["rejected"] = "tuning knobs set: " + ",".join(active_knobs())
    elif js.get("lib_sha256") != src["lib_sha256_now"]:
        # the binary that runs decides: a stale .so built from other sources must not borrow the
        # traffic of the build the sources on disk would give (build() skips by mtime)
        src["rejected"] = "measured on another build of libtwixt_hip.so (library hash differs; sources %s)" % (
            "match" if js.get("source_stamp") == src["source_stamp_now"] else "differ too")
    else:
        src["per_kernel"] = js.get("per_kernel")
        return float(js["hbm_bytes_per_launch"]), src
    return None, src


def fused_rollout(env, seed):
    """Secondary, reported separately (never mixed with the step-granular
    metric): one twx_random_rollout launch plays every board from the initial position to
    the end of the game with the state on chip between moves."""
    import torch
    env.reset()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    env.random_rollout(seed, want_status=False)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    _, mc, _ = env.status()
    steps = int(mc.to(torch.int64).sum())
    return {"value": steps / (ms * 1e-3), "unit": "env steps/s", "launch_ms": ms, "env_steps": steps,
            "playouts_per_s": env.batch / (ms * 1e-3),
            "model": "whole random play-outs in one launch; HBM traffic 2*%d B per play-out (state read + "
                     "written once)" % (16 * (2 * ((env.board_size ** 2 + 31) // 32) + 3))}


def observation_rate(env, nboards, reps=10):
    """Secondary: twx_observation on the first `nboards` boards, mid-game
    states.  Write-bound: 12*N*(N-2) floats per board against a state of a few hundred bytes."""
    import torch
    n = env.board_size
    out = env.observation(count=nboards)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        env.observation(count=nboards, out=out)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    per_board = 12 * n * (n - 2) * 4
    gbs = per_board * nboards / (ms * 1e-3) / 1e9
    return {"value": nboards / (ms * 1e-3), "unit": "boards/s", "boards": nboards, "launch_ms": ms,
            "bytes_written_per_board": per_board, "written_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
            "kernel": "observation_kernel", "bound": "hbm (writes)"}


def small_batch_latency(device, seed, cases=((12, 2048), (24, 16384)), steps=300):
    """Secondary: what a search sees (BASELINE config 5's leaf batches): a few thousand boards are
    one round of a few dozen waves, so a step costs the latency of one wave's step, not bandwidth."""
    import torch
    from twixt_for_open_spiel_amd import TwixtBatchEnv
    out = []
    for n, b in cases:
        env = TwixtBatchEnv(n, b, device)
        for _ in range(60):
            env.step_random(seed, True, want_actions=False, want_status=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            env.step_random(seed, True, want_actions=False, want_status=False)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / steps
        out.append({"board_size": n, "batch": b, "us_per_step": us, "env_steps_per_s": b / (us * 1e-6)})
        env.close()
    return out


def per_kernel_launch_ms(launch, sync, period, launches_so_far, periods=8):
    """HIP-event time of every launch over `periods` journal periods, split into the journal launches and
    the compactions (every period-th whole-batch step since the handle was created).  Separate from the
    timed region: an event per launch would perturb it."""
    import torch
    k = period * periods
    sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    ev[0].record()
    for i in range(k):
        launch()
        ev[i + 1].record()
    sync()
    t = [ev[i].elapsed_time(ev[i + 1]) for i in range(k)]
    comp = [t[i] for i in range(k) if (launches_so_far + i) % period == period - 1]
    jour = [t[i] for i in range(k) if (launches_so_far + i) % period != period - 1]
    return (sum(jour) / len(jour) if jour else None), (sum(comp) / len(comp) if comp else None)


def shard_first_board(rank, batch_per_gpu):
    """Rank r owns the contiguous global board ids [r*B, (r+1)*B)."""
    return rank * batch_per_gpu


def aggregate(per_rank, batch_per_gpu, steps, blocks):
    """Whole-job figures from the ranks' own clocks: value = all env steps / the slowest rank's
    elapsed time.  per_rank: list of dicts with rank, first_board_id, elapsed_s, kernel_ms."""
    per_rank = sorted(per_rank, key=lambda r: r["rank"])
    world = len(per_rank)
    assert [r["rank"] for r in per_rank] == list(range(world)), "a rank is missing or doubled"
    for r in per_rank:
        assert r["first_board_id"] == shard_first_board(r["rank"], batch_per_gpu), "shards must tile the id range"
    elapsed = max(r["elapsed_s"] for r in per_rank)
    total = float(batch_per_gpu) * world * steps * blocks
    return {"value": total / elapsed, "elapsed_s": elapsed, "total_env_steps": total,
            "ms_per_step": elapsed / (steps * blocks) * 1e3,
            "per_rank_value": [batch_per_gpu * steps * blocks / r["elapsed_s"] for r in per_rank]}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--board-size", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1 << 20, help="boards per GPU")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--prologue", type=int, default=700,
                    help="untimed setup steps that spread the boards over all game phases")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU and no library: a step is a short sleep (tests the launch, shard and "
                         "aggregation plumbing on CPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the secondary fused-rollout measurement")
    ap.add_argument("--no-observation", action="store_true", help="skip the secondary observation measurement")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """--gpus N without a launcher: start the N rank processes (rank r -> GPU r).  Nothing in
    this process has touched a GPU (device_count() does not initialise one)."""
    n = args.gpus
    if not args.dry_run and not args.share_gpu:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible (use --share-gpu to rehearse "
                             "the ranks on one device)" % (n, have))
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = set(range(n))
    while live:
        time.sleep(0.05)
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc = c
                for o in live:  # the others would wait for it in a barrier
                    procs[o].terminate()
    return rc


def claim_stdout():
    """Keeps the real stdout for the ONE JSON line and points fd 1 at stderr for everything else:
    libraries print there behind Python's back (Gloo's "Rank 0 is connected to 1 peer ranks" banner is
    a C++ write to stdout), and the driver reads stdout as the bench line."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args))
        world, rank, local = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != args.gpus:
            raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    json_out = claim_stdout()  # (a rank process; the parent that only starts the ranks prints nothing)

    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # CPU group: barriers outside the timed region and the ranks' clocks -> rank 0.  The data
        # path has no exchange step, so there is no RCCL communicator to build.
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=600))

    n, B, K = args.board_size, args.batch, args.steps
    first = shard_first_board(rank, B)
    dev = 0 if args.share_gpu else local

    if args.dry_run:
        env = None

        def launch():
            time.sleep(2e-4)

        def sync():
            pass
    else:
        import torch
        from twixt_for_open_spiel_amd import TwixtBatchEnv
        if not torch.cuda.is_available() or torch.cuda.device_count() <= dev:
            raise SystemExit("rank %d: HIP device %d not available (%d visible)" % (rank, dev, torch.cuda.device_count()))
        torch.cuda.set_device(dev)
        env = TwixtBatchEnv(n, B, dev, first_board_id=first)

        def launch():
            env.step_random(args.seed, True, want_actions=False, want_status=False)

        def sync():
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(0 if args.dry_run else args.prologue):
        launch()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.warmup):
        launch()
    sync()
    est_ms = (time.perf_counter() - t0) * 1e3 / max(1, args.warmup)  # per launch, incl. host overhead
    # a K-step block shorter than MIN_TIMED_MS is repeated: the figure must not hang on 3 ms
    blocks = 1
    if args.warmup > 0 and est_ms * K < MIN_TIMED_MS:
        blocks = int(math.ceil(MIN_TIMED_MS / (est_ms * K)))
    if dist is not None:
        import torch
        t = torch.tensor([blocks], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blocks = int(t.item())

    elapsed, kernel_ms_sum = 0.0, 0.0
    barrier()
    for _ in range(blocks):  # each block: EXACTLY K launches between two device synchronisations
        sync()
        if env is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if env is not None:
            ev0.record()
        for _ in range(K):
            launch()
        if env is not None:
            ev1.record()
        sync()
        elapsed += time.perf_counter() - t0
        if env is not None:
            kernel_ms_sum += ev0.elapsed_time(ev1)  # HIP events on the launch stream
    barrier()
    kernel_ms = kernel_ms_sum / (K * blocks) if env is not None else elapsed * 1e3 / (K * blocks)

    journal_ms = compaction_ms = None
    if env is not None and rank == 0 and journal_period(n) > 1 and not active_knobs():
        done = args.prologue + args.warmup + K * blocks
        journal_ms, compaction_ms = per_kernel_launch_ms(launch, sync, journal_period(n), done)
    hist = None
    if env is not None:
        res, _, _ = env.status()
        hist = torch.bincount(res.to(torch.int64), minlength=4).tolist()
    mine = {"rank": rank, "first_board_id": first, "device": dev, "elapsed_s": elapsed, "kernel_ms": kernel_ms,
            "result_hist": hist}
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        agg = aggregate(per_rank, B, K, blocks)
        ab, db = algo_bytes(n), design_bytes(n)
        traffic, traffic_src = (None, {"rejected": "dry run"}) if args.dry_run else measured_traffic(n, B)
        moved = traffic if traffic is not None else db * B  # bytes per launch (mean over a period) really moved
        achieved_all = moved / (kernel_ms * 1e-3) / 1e9
        # one kernel runs every launch (journal launches and compactions alike): its mean bytes, its mean time
        dom = step_kernel_name(n)
        achieved = achieved_all
        out = {
            "metric": "env steps/sec at board_size=%d, batch=%d per GPU, random-action rollouts" % (n, B),
            "value": agg["value"],
            "unit": "env steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": agg["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "timed_blocks": blocks,
            "per_rank_value": agg["per_rank_value"],
            "config": {"workload": "configs[2]: board_size=%d, batch=%d/GPU random rollouts, auto-reset, "
                                   "step-granular (state in HBM between launches)" % (n, B),
                       "board_size": n, "batch_per_gpu": B, "seed": args.seed,
                       "prologue_steps": args.prologue,
                       "sharding": "rank r owns board ids [r*B,(r+1)*B) on GPU r; no collective on the data path "
                                   "or in the timed region (gloo barriers around it)",
                       "first_board_ids": [r["first_board_id"] for r in sorted(per_rank, key=lambda r: r["rank"])],
                       "timing": "%d block(s) of exactly %d launches, each between two device synchronisations; "
                                 "value = all env steps / slowest rank's summed block time" % (blocks, K),
                       "result_hist_rank0": hist, "dry_run": bool(args.dry_run)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "basis": ("bytes moved per launch of this kernel by PMC (traffic) / its mean launch time" if traffic is not None else
                                   "design byte model x boards / kernel time (no PMC measurement of this build and "
                                   "workload is committed: traffic is null)"),
                         "traffic_source": traffic_src,
                         "peak_measured_copy": HBM_COPY_GBS,
                         "frac_of_measured_copy": achieved / HBM_COPY_GBS,
                         "kernel": dom,
                         "kernel_ms": kernel_ms,
                         "kernel_ms_note": "HIP events over the timed region / launches (journal launches and compactions alike)",
                         "launch_classes": ({"journal_launch_ms": journal_ms, "compaction_launch_ms": compaction_ms,
                                             "journal_period": journal_period(n),
                                             "model_bytes_per_board": {"journal_launch": design_bytes_split(n)[1],
                                                                       "compaction": design_bytes_split(n)[2]}
                                             if split_layout(n) else None,
                                             "note": "one HIP event per launch, outside the timed region"}
                                            if journal_ms is not None else None),
                         "design_model_bytes_per_env_step": db,
                         "traffic_over_design_model": (traffic / B / db) if traffic is not None else None,
                         "algorithmic_bytes_per_env_step": ab,
                         "algorithmic_equivalent_GBps": ab * B / (kernel_ms * 1e-3) / 1e9,  # (over all launches)
                         "note": "frac = bytes really moved / kernel time / 8 TB/s, so it cannot exceed 1.  "
                                 "algorithmic_equivalent_GBps prices the same launches at SURVEY 8(d)'s %d B per env "
                                 "step (full state read + write); the log-structured state moves fewer bytes than "
                                 "that, so this model figure may exceed the peak and is NOT a utilisation." % ab},
        }
        if env is not None and world == 1 and not args.no_observation:
            out["observation"] = observation_rate(env, min(B, 1 << 17))
        if env is not None and world == 1 and not args.no_rollout:
            out["fused_rollout"] = fused_rollout(env, args.seed)
            out["small_batch"] = small_batch_latency(dev, args.seed)
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            out["cpu_baseline"] = cpu_baseline(n, args.seed)
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if env is not None:
        env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
