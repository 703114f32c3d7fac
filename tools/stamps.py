#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of step_kernel<24> from the -DTWX_STAMPS build
(tools/bin/libtwixt_hip_stamps.so).  Never used by the product or the bench."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twixt_for_open_spiel_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libtwixt_hip_stamps.so")
from twixt_for_open_spiel_amd import TwixtBatchEnv

n, B = 24, 1 << 20
env = TwixtBatchEnv(n, B)
for _ in range(704):
    env.step_random(1, True, want_actions=False, want_status=False)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
names = ["loop/prev-store issue", "LDS fill + prefetch issue", "reset/replay", "legal mask", "select",
         "apply", "flood", "wait prefetch", "status store + loop end", "fixed stores", "dirty chunk stores",
         "reset lanes / compaction stores", "flood: pass setup", "flood: iterations", "flood: write-back"]
if os.environ.get("TWX_WINDOW", "1") != "0" and os.environ.get("TWX_PAIR", "1") != "0":
    # step_window1_kernel's STAMP(i) sites (twixt_window1_kernel.h), 64 boards per wave-step
    names = ["loop end of the previous iteration", "pegs and windows -> LDS", "row prefetch, scalars, reset, replay",
             "legal mask", "move, neighbours, window fetch issue", "(nothing)", "link scan, result, record",
             "wait for rows and windows", "ticket, waiting flood passes", "new fills: notes, requests, passes",
             "record and status stores", "-", "-", "-", "-"]


def run(k):
    env.lib.twx_debug_stamps(buf, 1)
    for _ in range(k):
        env.step_random(1, True, want_actions=False, want_status=False)
    torch.cuda.synchronize()
    env.lib.twx_debug_stamps(buf, 0)
    return [buf[i] for i in range(15)]


# 704 prologue launches = 88 journal periods: the next 7 launches append to the journal, the
# 8th compacts (rewrites every chunk).  The two kinds are reported separately.
normal, compact = [0] * 15, [0] * 15
R = 8
for _ in range(R):
    normal = [x + y for x, y in zip(normal, run(7))]
    compact = [x + y for x, y in zip(compact, run(1))]
for title, acc, launches in (("journal launches", normal, 7 * R), ("compaction launches", compact, R)):
    tot = sum(acc)
    wavesteps = launches * (B // (64 if (os.environ.get('TWX_PAIR', '1') == '0' or os.environ.get('TWX_WINDOW', '1') != '0') else 32))
    print("--- %s ---" % title)
    for i, nm in enumerate(names):
        print("%-32s %9.1f cycles/wave-step  %5.1f %%" % (nm, acc[i] / wavesteps, 100.0 * acc[i] / tot))
    print("total %.1f cycles/wave-step" % (tot / wavesteps))
