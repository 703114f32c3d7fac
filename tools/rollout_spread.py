#!/usr/bin/env python3
"""Small-batch play-outs (the leaf batches of a search): does it pay to give a wave fewer boards?
Emulated through the board-id list: groups of 64 rows with only R valid ids (the other rows are -1 =
padding lanes), so a wave plays R boards.  usage: rollout_spread.py [N] [boards]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twixt_for_open_spiel_amd import TwixtBatchEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
env = TwixtBatchEnv(n, B)
saved = None
for R in (64, 32, 16, 8, 4):
    rows = np.full((B // R, 64), -1, dtype=np.int32)
    rows[:, :R] = np.arange(B, dtype=np.int32).reshape(B // R, R)
    ids = torch.from_numpy(rows.reshape(-1)).cuda()
    ts = []
    for rep in range(6):
        env.reset()
        for _ in range(10):
            env.step_random(7, False, want_actions=False, want_status=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        st = env.random_rollout(3, ids=ids)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    res = st.cpu().numpy().reshape(-1, 64)[:, :R].reshape(-1) & 3
    if saved is None:
        saved = res
    assert np.array_equal(res, saved), "results must not depend on the spread"
    print("N=%d, %d boards, %2d boards per wave (%4d waves): %.3f ms per play-out launch (min of 6: %.3f)"
          % (n, B, R, B // R, sorted(ts)[len(ts) // 2], min(ts)))
