"""N>1 path on CPU.

* `bench.py --gpus 2` without a launcher starts its two ranks itself; with `--dry-run` the
  ranks do everything but touch a GPU (gloo rendezvous, start/end barriers, the block count
  agreed over ranks, clocks gathered on rank 0), so the launch, shard and aggregation plumbing
  is checked here.  The same under a launcher's environment (RANK / WORLD_SIZE set).
* two gloo ranks, each owning a contiguous shard of global board ids
  (bench.shard_first_board), compose to exactly the single-process run: trajectories depend on
  the global board id only, never on the sharding.  (CPU: the shards are played by the oracle;
  the same composition on the HIP engine is tests/test_two_ranks_gpu.py.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, B, STEPS, SEED = 8, 96, 70, 4


def _run_bench(extra, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=e, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _json_line(out):
    """stdout is the bench line and nothing else: bench.py keeps library chatter (Gloo's connection
    banner is written to stdout by C++) on stderr."""
    lines = out.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out
    return json.loads(lines[0])


def test_bench_parent_starts_its_ranks():
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "7", "--warmup", "2", "--batch", "640"])
    assert r.returncode == 0, r.stderr
    js = _json_line(r.stdout)
    assert js["n_gpus"] == 2 and js["steps"] == 7 and js["scaling"] == "weak"
    assert js["config"]["first_board_ids"] == [0, 640]            # contiguous shards
    assert len(js["per_rank_value"]) == 2
    blocks = js["timed_blocks"]
    assert blocks >= 1
    # whole-job value = all env steps / the slowest rank's time  (<= sum of the per-rank rates)
    assert js["value"] <= sum(js["per_rank_value"]) * (1 + 1e-9)
    assert js["value"] == pytest.approx(2 * min(js["per_rank_value"]), rel=1e-9)
    assert js["ms_per_step"] == pytest.approx(2 * 640 / js["value"] * 1e3, rel=1e-9)
    assert js["roofline"]["traffic"] is None and js["roofline"]["frac"] < 1.0


def test_bench_under_a_launcher_environment():
    import bench
    port = bench.free_port()
    procs = []
    for rank in range(2):
        e = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                                       "--steps", "5", "--warmup", "1", "--batch", "128"], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    js = _json_line(outs[0][0])
    assert outs[1][0] == ""                                       # only rank 0 prints, and only the line
    assert js["n_gpus"] == 2 and js["config"]["first_board_ids"] == [0, 128]


def test_bench_refuses_a_world_that_is_not_gpus():
    r = _run_bench(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr


def test_bench_fails_loudly_without_enough_devices():
    # this container has no GPU: asking for 2 real ones must not quietly bench one
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible")
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0"])
    assert r.returncode != 0 and "HIP device(s) visible" in r.stderr


def test_aggregate_is_total_steps_over_slowest_rank():
    import bench
    ranks = [{"rank": 1, "first_board_id": 100, "elapsed_s": 2.0, "kernel_ms": 1.0},
             {"rank": 0, "first_board_id": 0, "elapsed_s": 1.0, "kernel_ms": 1.0}]
    a = bench.aggregate(ranks, 100, 10, 3)
    assert a["total_env_steps"] == 2 * 100 * 10 * 3
    assert a["value"] == 2 * 100 * 10 * 3 / 2.0
    assert a["per_rank_value"] == [3000.0, 1500.0]
    with pytest.raises(AssertionError):
        bench.aggregate([ranks[0], dict(ranks[1], first_board_id=7)], 100, 10, 3)
    with pytest.raises(AssertionError):
        bench.aggregate([ranks[0], ranks[0]], 100, 10, 3)


def test_roofline_traffic_is_tied_to_the_build(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC measurement and is only reported for the
    library, workload and knobs it was measured on."""
    import bench
    from twixt_for_open_spiel_amd import build as b
    b.build()
    good = {"source_stamp": b.source_stamp(), "lib_sha256": b.lib_stamp(), "board_size": 24, "batch": 1 << 20,
            "hbm_bytes_per_launch": 8.0e8, "method": "m"}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    path = tmp_path / bench.TRAFFIC_FILE
    for k in [k for k in os.environ if k.startswith("TWX_")]:
        monkeypatch.delenv(k)
    path.write_text(json.dumps(good))
    t, src = bench.measured_traffic(24, 1 << 20)
    assert t == 8.0e8 and "rejected" not in src
    assert bench.measured_traffic(12, 1 << 20)[0] is None          # another board size
    assert bench.measured_traffic(24, 1 << 18)[0] is None          # another batch
    monkeypatch.setenv("TWX_JOURNAL_PERIOD", "1")                  # a knob changes what is moved
    t, src = bench.measured_traffic(24, 1 << 20)
    assert t is None and "TWX_JOURNAL_PERIOD" in src["rejected"]
    monkeypatch.delenv("TWX_JOURNAL_PERIOD")
    # the binary that runs decides: a library other than the measured one is refused even when
    # the sources on disk hash to the measured stamp (build() skips by mtime, so a stale .so can sit there)
    path.write_text(json.dumps(dict(good, lib_sha256="0" * 16)))
    t, src = bench.measured_traffic(24, 1 << 20)
    assert t is None and "another build" in src["rejected"] and "sources match" in src["rejected"]
    path.write_text(json.dumps(dict(good, lib_sha256="0" * 16, source_stamp="0" * 16)))
    t, src = bench.measured_traffic(24, 1 << 20)
    assert t is None and "differ too" in src["rejected"]


def test_design_byte_model():
    import bench
    # N=24 (split layout, windowed step), period 8: a launch with j records requests pegs + scalars (160 B), the
    # records (4 j), the window (10 pieces of 16 B), 2.8 % of a whole region (432 B) for the multi-cell fills, and
    # writes the region words of its move (a mean) and its record; the compaction writes pegs and scalars instead
    per_step, journal, compaction = bench.design_bytes_split(24)
    upd = bench.REGION_UPDATE_BYTES
    win = bench.WINDOW_SHARE * 160  # (only a move with a neighbour to link to fetches, and only the columns it can touch)
    assert abs(journal - (160 + 4 * 3 + win + 0.028 * 432 + upd + 4)) < 1e-9
    assert abs(compaction - (160 + 4 * 7 + win + 0.028 * 432 + upd + 160)) < 1e-9
    assert abs(per_step - (7 * journal + compaction) / 8) < 1e-9 and bench.design_bytes(24) == per_step
    assert bench.window_geometry(24) == {"W": 18, "RP": 9, "LFB": 432, "NW": 6, "NPC": 10}
    assert bench.window_geometry(19)["NPC"] == 8 and bench.window_geometry(20)["LFB"] == 320
    assert bench.design_bytes(8) == 2 * 80 and bench.journal_period(9) == 1        # no journal: read + write the board
    assert [bench.journal_period(n) for n in (10, 12, 13, 17, 18, 24)] == [2, 2, 4, 4, 8, 8]
    assert bench.step_kernel_name(24) == "step_window1_kernel<24>" and "small" in bench.step_kernel_name(12)
    assert bench.algo_bytes(24) == 1320 and bench.algo_bytes(12) == 348 and bench.algo_bytes(8) == 168


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import bench
    import oracle_binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first = bench.shard_first_board(rank, B)
    dist.barrier()
    r = ob.rollout_batch(N, SEED, first, B, STEPS, True, threads=1)
    dist.barrier()
    mine = {"rank": rank, "first_board_id": first, "elapsed_s": float(rank + 1), "kernel_ms": 0.0}
    got = [None] * world
    dist.all_gather_object(got, mine)
    agg = bench.aggregate(got, B, STEPS, 1)
    hist = torch.from_numpy(np.bincount(r["planes"][:, -3] & 0xff, minlength=4).astype(np.int64))
    dist.all_reduce(hist)  # host-side aggregation of per-shard counters only
    q.put((rank, first, r["planes"], hist.tolist(), agg["elapsed_s"], r["steps"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_compose():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    ob.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = ob.rollout_batch(N, SEED, 0, 2 * B, STEPS, True, threads=1)
    assert [g[1] for g in got] == [0, B]
    assert np.array_equal(np.concatenate([g[2] for g in got]), whole["planes"])
    want_hist = np.bincount(whole["planes"][:, -3] & 0xff, minlength=4).tolist()
    assert got[0][3] == want_hist and got[1][3] == want_hist
    assert got[0][4] == 2.0 and got[1][4] == 2.0          # the slowest rank's clock
    assert got[0][5] + got[1][5] == whole["steps"] == 2 * B * STEPS
