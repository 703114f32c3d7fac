#!/usr/bin/env python3
"""One-off soak of the final binary: N=24, B=2^20, many auto-reset steps (whole games: wins, draws, resets in the
middle of journal periods), EVERY board compared bit-for-bit with the oracle.  usage: soak_full_size.py [steps] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob
from bench import host_cores
from twixt_for_open_spiel_amd import TwixtBatchEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
n, batch, first = 24, 1 << 20, 123456789
env = TwixtBatchEnv(n, batch, 0, first)
stat = None
t0 = time.time()
for _ in range(steps):
    _, stat = env.step_random(seed, True, want_actions=False)
torch.cuda.synchronize()
print("device: %d steps in %.1f s" % (steps, time.time() - t0), flush=True)
stat = stat.cpu().numpy()
chunk, hist = 1 << 16, np.zeros(4, dtype=np.int64)
for lo in range(0, batch, chunk):
    ref = ob.rollout_batch(n, seed, first + lo, chunk, steps, True, threads=host_cores(), want_trace=True)
    got = env.export_state(torch.arange(lo, lo + chunk, dtype=torch.int32)).cpu().numpy().view(np.uint32)
    sw = got.shape[1]
    assert np.array_equal(got[:, :sw - 1], ref["planes"][:, :sw - 1]), "boards %d..%d" % (lo, lo + chunk)
    assert np.array_equal(stat[lo:lo + chunk], ref["status"][:, -1]), "status %d..%d" % (lo, lo + chunk)
    hist += np.bincount(ref["planes"][:, sw - 3] & 3, minlength=4)
    if lo % (1 << 18) == 0:
        print("compared %d boards (%.0f s)" % (lo + chunk, time.time() - t0), flush=True)
print("OK: %d boards x %d steps bit-exact; results open/red/blue/draw = %s; episodes up to %d"
      % (batch, steps, hist.tolist(), int(env.export_state(torch.arange(0, 4096, dtype=torch.int32))[:, -2].max())))
