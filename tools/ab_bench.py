#!/usr/bin/env python3
"""Diagnostic A/B timing of step-kernel builds in ONE process (interleaved rounds, same
device), per the guide's rule 24.  usage: ab_bench.py libA.so[:KNOB=V,KNOB=V] libB.so [...]
(knobs are the TWX_* environment variables twx_create reads, set for that handle only)"""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twixt_for_open_spiel_amd._lib as L

specs = sys.argv[1:]
n, B = int(os.environ.get("AB_N", "24")), 1 << 20  # AB_N: board size
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
loaded = {}


def create(spec):
    """(lib, handle) for 'lib.so[:KNOB=V,...]'.  Handles are created and destroyed one at a
    time so that every configuration gets the same allocation: two live 671 MB state buffers
    differ by up to 5 % in speed on MI355X just by where they sit in HBM."""
    path, _, knobs = spec.partition(":")
    knobs = dict(kv.split("=") for kv in knobs.split(",") if kv)
    if path not in loaded:
        lib = C.CDLL(os.path.abspath(path))
        lib.twx_create.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
        lib.twx_step_random.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.twx_destroy.argtypes = [C.c_void_p]
        loaded[path] = lib
    lib = loaded[path]
    os.environ.update(knobs)
    h = C.c_void_p()
    assert lib.twx_create(n, B, 0, 0, C.byref(h)) == 0
    for k in knobs:
        del os.environ[k]
    return lib, h


res = [[] for _ in specs]
K = 160
for r in range(5):
    for i, spec in enumerate(specs):
        lib, h = create(spec)
        for _ in range(720):
            lib.twx_step_random(h, 1, 1, None, None, stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(K):
            lib.twx_step_random(h, 1, 1, None, None, stream)
        b.record()
        torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) / K)
        lib.twx_destroy(h)
for spec, r in zip(specs, res):
    print("%-44s median %.4f ms  min %.4f  max %.4f  (%.2f G steps/s)" % (spec[-44:], statistics.median(r), min(r), max(r), B / statistics.median(r) / 1e6))
