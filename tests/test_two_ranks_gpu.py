"""The 2-rank path on the HIP engine (BASELINE config 4's shape in small): two
fresh worker processes share the one visible MI355X, rank r owns the global board ids
[r*B, (r+1)*B) through `first_board_id`, nothing is exchanged between them.  Their exported
states, concatenated, must equal one handle of 2B boards and the CPU oracle bit for bit --
i.e. a board's trajectory depends on its global id only, not on the sharding."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, B, STEPS, SEED = 8, 3000, 90, 5

WORKER = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
rank, n, B, steps, seed, out = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]),
                                int(sys.argv[6]), sys.argv[7])
import torch
from twixt_for_open_spiel_amd import TwixtBatchEnv
env = TwixtBatchEnv(n, B, 0, first_board_id=rank * B)          # every rank on cuda:0 (--share-gpu)
acts = []
for _ in range(steps):
    a, _ = env.step_random(seed, True, want_status=False)
    acts.append(a.cpu().numpy())
np.savez(out, planes=env.export_state().cpu().numpy().view(np.uint32), actions=np.stack(acts))
"""


@pytest.mark.gpu
def test_two_engine_ranks_compose(tmp_path):
    import oracle_binding as ob
    from twixt_for_open_spiel_amd import TwixtBatchEnv
    ob.build()
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, str(r), str(N), str(B), str(STEPS), str(SEED),
                               outs[r]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]                                   # both alive at once on the one GPU
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se
    got = [np.load(o) for o in outs]
    planes = np.concatenate([g["planes"] for g in got])
    actions = np.concatenate([g["actions"] for g in got], axis=1)  # [steps, 2B]

    env = TwixtBatchEnv(N, 2 * B, 0, first_board_id=0)            # one handle, all 2B boards
    acts = []
    for _ in range(STEPS):
        a, _ = env.step_random(SEED, True, want_status=False)
        acts.append(a.cpu().numpy())
    whole = env.export_state().cpu().numpy().view(np.uint32)
    assert np.array_equal(planes, whole)
    assert np.array_equal(actions, np.stack(acts))

    ref = ob.rollout_batch(N, SEED, 0, 2 * B, STEPS, True)["planes"]
    assert np.array_equal(planes[:, :-1], ref[:, :-1])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """`bench.py --gpus 2 --share-gpu` starts its two ranks by itself and reports n_gpus 2."""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "30",
                        "--warmup", "5", "--prologue", "50", "--batch", str(1 << 16)], env=e, timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    js = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert js["n_gpus"] == 2 and js["config"]["first_board_ids"] == [0, 1 << 16]
    assert len(js["per_rank_value"]) == 2 and js["value"] > 0
    assert js["roofline"]["frac"] < 1.0
