#!/usr/bin/env python3
"""Small whole-batch steps: the whole-batch launch (64 boards per wave, tile-addressed) against the same step
through the listed-boards launch with ids = 0..B-1 (which spreads the boards over many waves)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twixt_for_open_spiel_amd import TwixtBatchEnv

for n, B in ((12, 2048), (8, 4096), (24, 16384), (12, 16384), (12, 65536)):
    env = TwixtBatchEnv(n, B)
    ids = torch.arange(B, dtype=torch.int32, device="cuda")
    for name, kw in (("whole-batch launch", {}), ("listed-boards launch", {"ids": ids})):
        for _ in range(50):
            env.step_random(1, True, want_actions=False, want_status=False, **kw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(300):
            env.step_random(1, True, want_actions=False, want_status=False, **kw)
        b.record()
        torch.cuda.synchronize()
        print("N=%d B=%d %-22s %.2f us per step" % (n, B, name, a.elapsed_time(b) * 1e3 / 300))
    env.close()
