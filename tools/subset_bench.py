#!/usr/bin/env python3
"""How long does a subset step on ~3 % scattered boards take (N=24, B=2^20)?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twixt_for_open_spiel_amd import TwixtBatchEnv
env = TwixtBatchEnv(24, 1 << 20)
for _ in range(300):
    env.step_random(1, True, want_actions=False, want_status=False)
for frac in (0.01, 0.03, 0.06):
    k = int((1 << 20) * frac)
    ids = torch.sort(torch.randperm(1 << 20, device="cuda")[:k]).values.to(torch.int32)
    tot = 0.0
    for rep in range(4):
        env.legal_mask(ids[:8])  # flush: the next 6 subset launches run at journal positions 0..5
        env.step_random(1, True, ids=ids, want_actions=False, want_status=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            env.step_random(1, True, ids=ids, want_actions=False, want_status=False)
        b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b) / 5
    print("subset %.0f%% (%d boards): %.1f us per launch" % (100 * frac, k, tot / 4 * 1e3))
