#!/usr/bin/env python3
"""Times twx_observation (row f1 of SURVEY 8): N=24, B boards -> [B,12,24,22] fp32."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twixt_for_open_spiel_amd._lib as L
if len(sys.argv) > 2:  # usage: obs_bench.py [boards] [lib.so]
    L.LIB_PATH = os.path.abspath(sys.argv[2])
from twixt_for_open_spiel_amd import TwixtBatchEnv

n, B = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
env = TwixtBatchEnv(n, B)
for _ in range(300):
    env.step_random(1, True, want_actions=False, want_status=False)
out = env.observation()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
import ctypes as C
from twixt_for_open_spiel_amd import _lib
a.record()
for _ in range(K):
    _lib.check(env.lib.twx_observation(env._h, C.c_void_p(out.data_ptr()), env._stream()))
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / K
byts = out.numel() * 4
print("twx_observation N=%d B=%d: %.3f ms, %.1f M boards/s, %.0f GB/s written" % (n, B, ms, B / ms / 1e3, byts / ms / 1e6))
# write-only ceiling on the same buffer: a fill of the output tensor
a.record()
for _ in range(K):
    out.fill_(1.0)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / K
print("fill of the same %d MB buffer: %.3f ms, %.0f GB/s written" % (byts >> 20, ms, byts / ms / 1e6))
