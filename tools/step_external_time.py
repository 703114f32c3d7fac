"""tools/step_external_time.py -- twx_step (actions from the caller) against twx_step_random at one board size:
the same moves, recorded from a random run and replayed as external actions.
  [TWIXT_HIP_LIB=path] python tools/step_external_time.py [N=24] [BATCH=2^20] [STEPS=60]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twixt_for_open_spiel_amd import TwixtBatchEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
env = TwixtBatchEnv(n, B)
for _ in range(200):  # mid-game boards, no game ends within the next `steps` moves at N=24
    env.step_random(5, False, want_actions=False, want_status=False)
rec = []
snap = env.export_state().clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(steps):
    a, _ = env.step_random(5, False, want_status=False)
    rec.append(a)
torch.cuda.synchronize()
t_rand = (time.perf_counter() - t0) / steps
final_random = env.export_state().clone()
env.import_state(snap)
status = torch.empty(B, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(steps):
    env.step(rec[t], status=status)
torch.cuda.synchronize()
t_ext = (time.perf_counter() - t0) / steps
same = bool(torch.equal(env.export_state(), final_random))
print("N=%d B=%d: twx_step_random (actions out) %.1f us per step, twx_step (same moves) %.1f us per step; same final state: %s"
      % (n, B, t_rand * 1e6, t_ext * 1e6, same))
