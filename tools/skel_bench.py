#!/usr/bin/env python3
"""Diagnostic: time the memory skeleton of step_kernel<24> (-DTWX_SKELETON build: same
loads / LDS fill / prefetch / stores, no game logic).  Never used by the product."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twixt_for_open_spiel_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "bin", sys.argv[1] if len(sys.argv) > 1 else "libtwixt_hip_skel.so")
from twixt_for_open_spiel_amd import TwixtBatchEnv

env = TwixtBatchEnv(24, 1 << 20)
for _ in range(64):
    env.step_random(1, True, want_actions=False, want_status=False)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 400
a.record()
for _ in range(K):
    env.step_random(1, True, want_actions=False, want_status=False)
b.record()
torch.cuda.synchronize()
print("%s: %.4f ms per launch" % (os.path.basename(L.LIB_PATH), a.elapsed_time(b) / K))
