#!/usr/bin/env python3
"""Diagnostic A/B timing of step-kernel builds in ONE process (interleaved rounds, same
device), per the guide's rule 24.  usage: ab_bench.py libA.so libB.so [...]"""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twixt_for_open_spiel_amd._lib as L

libs = sys.argv[1:]
n, B = 24, 1 << 20
envs = []
for path in libs:
    lib = C.CDLL(os.path.abspath(path))
    lib.twx_create.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.twx_step_random.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    h = C.c_void_p()
    assert lib.twx_create(n, B, 0, 0, C.byref(h)) == 0
    envs.append((lib, h))
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(i, k):
    lib, h = envs[i]
    for _ in range(k):
        lib.twx_step_random(h, 1, 1, None, None, stream)


for i in range(len(envs)):
    run(i, 720)
torch.cuda.synchronize()
res = [[] for _ in envs]
K = 160
for r in range(7):
    for i in range(len(envs)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run(i, K)
        b.record()
        torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) / K)
for path, r in zip(libs, res):
    print("%-40s median %.4f ms  min %.4f  (%.2f G steps/s)" % (os.path.basename(path), statistics.median(r), min(r), B / statistics.median(r) / 1e6))
