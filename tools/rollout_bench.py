#!/usr/bin/env python3
"""Times twx_random_rollout (whole random play-outs, state on chip between moves)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twixt_for_open_spiel_amd import TwixtBatchEnv

for n, B in ((24, 1 << 20), (12, 1 << 20), (8, 1 << 20)):
    env = TwixtBatchEnv(n, B)
    env.random_rollout(1)
    env.reset()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    env.random_rollout(2, want_status=False)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    res, mc, _ = env.status()
    steps = int(mc.to(torch.int64).sum())
    print("N=%d B=%d: %.2f ms per launch, %d env steps (mean %.1f/board), %.2f G env steps/s, %.1f M play-outs/s"
          % (n, B, ms, steps, steps / B, steps / ms / 1e6, B / ms / 1e3))
    env.close()
