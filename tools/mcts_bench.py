#!/usr/bin/env python3
"""BASELINE config 5: MCTS self-play, board_size=12, max_simulations=20000 per move,
batched leaf ApplyAction / IsTerminal / play-outs on one MI355X.

Prints simulations/s of
  * the C++ batched-leaf driver on the HIP engine (spiel_adapter/mcts_driver.cc),
  * the SAME C++ search with every leaf evaluated by the CPU oracle on all host cores
    (tests/mcts_cpu_evaluator.h; the checker timed as a CPU baseline, like bench.py's
    cpu_baseline leg) -- "the CPU engine doing the same search" of SURVEY 8(f3), measured,
  * the Python driver (twixt_for_open_spiel_amd/mcts.py) on the HIP engine,
and checks that the two C++ runs chose the same moves with the same visit counts.

usage: mcts_bench.py [board_size=12] [simulations=20000] [moves=6] [leaves=2048] [games=1]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "twixt_for_open_spiel_amd")
SRC = os.path.join(PKG, "spiel_adapter", "mcts_driver.cc")
COMMON = ["g++", "-O2", "-fopenmp", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
          "-I/opt/rocm/include"]


def build_device(out):
    subprocess.check_call(COMMON + [SRC, "-o", out, "-L" + PKG, "-ltwixt_hip", "-L/opt/rocm/lib", "-lamdhip64",
                                    "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def build_cpu(out):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libtwixt_oracle.so"])
    subprocess.check_call(COMMON + ["-DMCTS_EVALUATOR_HEADER=\"mcts_cpu_evaluator.h\"", "-I" + os.path.join(ROOT, "tests"),
                                    "-I" + os.path.join(ROOT, "oracle"), SRC, "-o", out,
                                    "-L" + os.path.join(ROOT, "oracle"), "-ltwixt_oracle",
                                    "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return out


def run(exe, args):
    text = subprocess.run([exe] + [str(a) for a in args], check=True, capture_output=True).stdout.decode()
    rate = float(re.search(r"= (\d+) simulations/s", text).group(1))
    moves = re.findall(r"move (\d+): player (\d) plays (\d+) \((\d+) visits\)", text)
    return rate, moves, text.strip().splitlines()[-2:]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    sims = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    moves = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    L = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
    games = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    tmp = tempfile.mkdtemp()
    dev, cpu = build_device(os.path.join(tmp, "mcts_driver")), build_cpu(os.path.join(tmp, "mcts_cpu"))
    args = [n, sims, moves, L, 1, games]
    run(dev, [n, 2 * L, 1, L, 1, games])  # warm up the device and the page cache
    d_rate, d_moves, d_tail = run(dev, args)
    run(cpu, [n, 2 * L, 1, L, 1, games])
    c_rate, c_moves, c_tail = run(cpu, args)
    print("C++ driver on the HIP engine : %9.0f simulations/s   %s" % (d_rate, d_tail[0]))
    print("    " + d_tail[1])
    print("same search on the CPU oracle: %9.0f simulations/s   %s" % (c_rate, c_tail[0]))
    print("    " + c_tail[1])
    same = d_moves == c_moves
    print("moves and visit counts equal : %s (%d moves)" % (same, len(d_moves)))
    out = {"board_size": n, "max_simulations": sims, "moves": moves, "leaves_per_batch": L, "games": games,
           "device_simulations_per_s": d_rate, "cpu_search_simulations_per_s": c_rate, "same_game": same,
           "cpu_threads": os.cpu_count()}
    if games == 1:
        import torch
        from twixt_for_open_spiel_amd.mcts import DeviceEvaluator, self_play
        ev = DeviceEvaluator(n, L)
        self_play(n, 2 * L, L, max_moves=1, evaluator=ev)  # warm up
        t0 = time.perf_counter()
        hist, _, total = self_play(n, sims, L, max_moves=min(moves, 2), evaluator=ev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["python_driver_simulations_per_s"] = total / dt
        print("Python driver on the HIP engine: %7.0f simulations/s (%d moves)" % (total / dt, len(hist)))
    print(json.dumps(out))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
