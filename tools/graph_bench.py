#!/usr/bin/env python3
"""Diagnostic: launch-bound small batches (search leaves), plain launches against a replayed
hipGraph of the same steps (handle created with TWX_GRAPH_SAFE=1).
usage: graph_bench.py [board_size=12] [batch=2048] [steps=64]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TWX_GRAPH_SAFE"] = "1"
from twixt_for_open_spiel_amd import TwixtBatchEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
K = int(sys.argv[3]) if len(sys.argv) > 3 else 64
env = TwixtBatchEnv(n, B)
side = torch.cuda.Stream()


def run_plain(reps):
    with torch.cuda.stream(side):
        for _ in range(reps * K):
            env.step_random(1, True, want_actions=False, want_status=False)
    side.synchronize()


with torch.cuda.stream(side):
    env.step_random(1, True, want_actions=False, want_status=False)
side.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    for _ in range(K):
        env.step_random(1, True, want_actions=False, want_status=False)
for name, fn in (("plain launches", lambda: run_plain(20)), ("hipGraph replay", lambda: ([g.replay() for _ in range(20)], torch.cuda.synchronize()))):
    fn()
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    print("%-16s N=%d B=%d: %.2f us per step, %.1f M env steps/s" % (name, n, B, dt / (20 * K) * 1e6, 20 * K * B / dt / 1e6))
