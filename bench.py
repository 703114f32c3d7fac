#!/usr/bin/env python3
"""bench.py -- env steps/sec of the batched TwixT step kernel on MI355X.

Metric (BASELINE.json): env steps/sec on random-action rollouts at board_size=24,
batch=2^20 per GPU.  One bench "step" = one twx_step_random launch = one env step
on every board of the batch (terminal boards auto-reset and play move 0 in the
same launch, so a launch is exactly B env steps).  State lives in HBM between
launches (step-granular model): algorithmic traffic
2*656+8 = 1320 B per env step at N=24.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Boards are independent: every rank owns a contiguous shard of global board ids,
no collective is on the data path (only the timing barrier / max).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {24: 1320, 12: 348, 8: 168}  # 2*S(N)+8
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0                     # same guide: measured float4-copy ceiling


def algo_bytes(n):
    return ALGO_BYTES.get(n, 2 * (9 * ((n * n + 7) // 8) + 8) + 8)


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


def cpu_baseline(n, seed, budget_s=12.0):
    """Times the CPU oracle (the scalar port of the reference's algorithm) on all
    host cores with the same policy and workload shape.  Checker code used as a
    *reported baseline only*."""
    lib_path = os.path.join(ROOT, "oracle", "libtwixt_oracle.so")
    if not os.path.exists(lib_path):
        import subprocess
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


def measured_traffic(n, batch):
    """HBM bytes per launch of step_kernel<N> from the committed rocprofv3 PMC passes
    (profiles/traffic_step_kernel_n24.json: FETCH_SIZE x2 per the gfx950 correction +
    WRITE_SIZE).  Counters cannot be collected inside this process; None if the
    committed measurement does not cover this workload."""
    path = os.path.join(ROOT, "profiles", "traffic_step_kernel_n24.json")
    if n != 24 or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_bytes_per_env_step"]) * batch
    except (OSError, ValueError, KeyError):
        return None


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


def shard_first_board(rank, batch_per_gpu):
    """Rank r owns the contiguous global board ids [r*B, (r+1)*B)."""
    return rank * batch_per_gpu


def max_over_ranks(value, dist, device):
    """Whole-job time = the slowest rank's time."""
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--board-size", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1 << 20, help="boards per GPU")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--prologue", type=int, default=700,
                    help="untimed setup steps that spread the boards over all game phases")
    ap.add_argument("--backend", default="nccl", help="process-group backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the secondary fused-rollout measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from twixt_for_open_spiel_amd import TwixtBatchEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=args.backend)

    n, B = args.board_size, args.batch
    env = TwixtBatchEnv(n, B, local, first_board_id=shard_first_board(rank, B))

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.prologue + args.warmup):
        env.step_random(args.seed, True, want_actions=False, want_status=False)
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        env.step_random(args.seed, True, want_actions=False, want_status=False)
    ev1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    if world > 1:
        elapsed = max_over_ranks(elapsed, dist, "cuda" if args.backend == "nccl" else "cpu")

    res, mc, _ = env.status()
    hist = torch.bincount(res.to(torch.int64), minlength=4).tolist()

    if rank == 0:
        total_steps = float(B) * world * args.steps
        value = total_steps / elapsed
        ab = algo_bytes(n)
        achieved = ab * B / (kernel_ms * 1e-3) / 1e9
        traffic = measured_traffic(n, B)
        out = {
            "metric": "env steps/sec at board_size=%d, batch=%d per GPU, random-action rollouts" % (n, B),
            "value": value,
            "unit": "env steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "configs[2]: board_size=%d, batch=%d/GPU random rollouts, auto-reset, "
                                   "step-granular (state in HBM between launches)" % (n, B),
                       "board_size": n, "batch_per_gpu": B, "seed": args.seed,
                       "prologue_steps": args.prologue, "sharding": "contiguous board ids, no collectives",
                       "result_hist_rank0": hist},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "moved_GBps": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
                         "peak_measured_copy": HBM_COPY_GBS,
                         "frac_of_measured_copy": achieved / HBM_COPY_GBS,
                         "kernel": ("step_kernel64_small<%d, false>" if 2 * ((n * n + 31) // 32) + 3 <= 25
                                    else "step_kernel64<%d, false>") % n,
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_env_step": ab,
                         "note": "achieved = %d B x env steps / kernel time (SURVEY 8(d) model: full state "
                                 "read + write per step).  The log-structured state rewrites the planes only "
                                 "every 8th launch, so the kernel moves fewer bytes per env step than that (traffic, PMC: "
                                 "%s) and the algorithmic figure can exceed the HBM peak."
                                 % (ab, ("%.0f B per env step" % (traffic / B)) if traffic else "not collected for this "
                                    "board size")},
        }
        if world == 1 and not args.no_rollout:
            out["fused_rollout"] = fused_rollout(env, args.seed)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, args.seed)
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
