#!/usr/bin/env python3
"""BASELINE config 5: MCTS self-play, board_size=12, max_simulations=20000 per move,
batched leaf ApplyAction / IsTerminal / play-outs on one MI355X.  Prints simulations/s and
the split between host tree work and device round trips."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from twixt_for_open_spiel_amd.mcts import DeviceEvaluator, self_play

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 6
L = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
ev = DeviceEvaluator(n, L)
t_dev = [0.0]
orig = ev.evaluate


def timed(paths, seed):
    t0 = time.perf_counter()
    out = orig(paths, seed)
    torch.cuda.synchronize()
    t_dev[0] += time.perf_counter() - t0
    return out


ev.evaluate = timed
self_play(n, 2 * L, L, max_moves=1, evaluator=ev)  # warm up
t_dev[0] = 0.0
t0 = time.perf_counter()
hist, res, total = self_play(n, sims, L, max_moves=moves, evaluator=ev, verbose=True)
dt = time.perf_counter() - t0
print("N=%d: %d moves x %d simulations, leaves/batch %d: %.1f s -> %.0f simulations/s "
      "(device round trips %.1f s = %.0f %%, host tree %.1f s)"
      % (n, len(hist), sims, L, dt, total / dt, t_dev[0], 100 * t_dev[0] / dt, dt - t_dev[0]))
