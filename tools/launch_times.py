#!/usr/bin/env python3
"""Per-launch times across one journal cycle (diagnostic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twixt_for_open_spiel_amd._lib as L
if len(sys.argv) > 1:  # usage: launch_times.py [lib.so]   (TWX_* knobs from the environment)
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from twixt_for_open_spiel_amd import TwixtBatchEnv
print(os.path.basename(L.LIB_PATH), " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("TWX_")))
env = TwixtBatchEnv(24, 1 << 20)
for _ in range(720):
    env.step_random(1, True, want_actions=False, want_status=False)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(65)]
ev[0].record()
for i in range(64):
    env.step_random(1, True, want_actions=False, want_status=False)
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(64)]
for j in range(8):
    print("launch %% 8 == %d: %.1f us" % (j, sum(t[j::8]) / 8))
print("mean %.1f us" % (sum(t) / 64))
