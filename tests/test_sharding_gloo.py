"""N>1 path on CPU: two gloo ranks, each owning a contiguous shard of global board
ids (bench.shard_first_board), barrier + max-over-ranks timing (bench.max_over_ranks).
The shards, played independently, must compose to exactly the single-process run:
trajectories depend on the global board id only, never on the sharding."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, B, STEPS, SEED = 8, 96, 70, 4


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
    slowest = bench.max_over_ranks(float(rank + 1), dist, "cpu")
    hist = torch.from_numpy(np.bincount(r["planes"][:, -3] & 0xff, minlength=4).astype(np.int64))
    dist.all_reduce(hist)  # host-side aggregation of per-shard counters only
    q.put((rank, first, r["planes"], hist.tolist(), slowest, r["steps"]))
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
    assert got[0][4] == 2.0 and got[1][4] == 2.0          # max over ranks
    assert got[0][5] + got[1][5] == whole["steps"] == 2 * B * STEPS
