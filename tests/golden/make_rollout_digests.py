#!/usr/bin/env python3
"""Golden vectors of the random-policy rollouts (DESIGN.md "Policy"), produced by the CPU
oracle: for several board sizes and seeds, SHA-256 digests of the per-step (action, status)
trace and of the final exported state of B boards.  The GPU parity tests check the HIP
engine against the live oracle AND against these committed digests, so a silent change of
the oracle (or of the policy hash) cannot go unnoticed.  Run from the repo root."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob

CASES = [(5, 64, 40, 3, 0), (8, 128, 90, 1, 100), (12, 96, 200, 7, 5), (24, 48, 700, 2, 0)]


def digest(n, boards, steps, seed, first, auto_reset):
    r = ob.rollout_batch(n, seed, first, boards, steps, auto_reset, want_trace=True, want_cells=True)
    return {"actions": hashlib.sha256(r["actions"].tobytes()).hexdigest(),
            "status": hashlib.sha256(r["status"].tobytes()).hexdigest(),
            "planes": hashlib.sha256(r["planes"].tobytes()).hexdigest(),
            "cells": hashlib.sha256(r["cells"].tobytes()).hexdigest(),
            "env_steps": int(r["steps"])}


def main():
    out = []
    for n, boards, steps, seed, first in CASES:
        for auto_reset in (True, False):
            d = digest(n, boards, steps, seed, first, auto_reset)
            d.update({"n": n, "boards": boards, "steps": steps, "seed": seed, "first_board": first,
                      "auto_reset": auto_reset})
            out.append(d)
    with open(os.path.join(HERE, "rollout_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
