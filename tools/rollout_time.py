"""tools/rollout_time.py -- fused play-outs (twx_random_rollout) and listed-boards steps timed at one board size.
  [TWIXT_HIP_LIB=path] python tools/rollout_time.py N [BATCH=262144]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twixt_for_open_spiel_amd import TwixtBatchEnv

n = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
env = TwixtBatchEnv(n, B)
best = 1e9
for rep in range(4):
    env.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.random_rollout(7 + rep, want_status=False)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
_, mc, _ = env.status()
steps = int(mc.to(torch.int64).sum())
print("N=%d B=%d: play-outs %.2f ms, %.2f G env steps/s" % (n, B, best * 1e3, steps / best / 1e9))
# listed boards: 4096 of them, one random step each, 200 times
env.reset()
ids = torch.arange(0, 4096 * 8, 8, dtype=torch.int32, device="cuda")
for _ in range(20):
    env.step_random(3, True, want_actions=False, want_status=False, ids=ids)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    env.step_random(3, True, want_actions=False, want_status=False, ids=ids)
torch.cuda.synchronize()
print("   listed boards (4096): %.2f us per step" % ((time.perf_counter() - t0) / 200 * 1e6))
# a small whole batch (a search's leaves)
small = TwixtBatchEnv(n, 4096)
for _ in range(20):
    small.step_random(3, True, want_actions=False, want_status=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    small.step_random(3, True, want_actions=False, want_status=False)
torch.cuda.synchronize()
print("   whole batch of 4096: %.2f us per step" % ((time.perf_counter() - t0) / 200 * 1e6))
