#!/usr/bin/env python3
"""BASELINE config 1 plumbing: one TwixT game with a uniform random policy, printed move
by move like open_spiel's `examples/example --game=twixt` (reference README.md:36-42), but
with every LegalActions / ApplyAction / ToString served by the HIP engine.

  python examples/example.py [--board_size 8] [--seed 1] [--no-color]
"""
import argparse
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from twixt_for_open_spiel_amd import TwixtBatchEnv


def to_string(env, ansi, swapped, result):
    cells = env.export_cells([0]).cpu().numpy()[0]
    buf = C.create_string_buffer(1 << 17)
    k = env.lib.twx_cells_to_string(cells.ctypes.data_as(C.c_void_p), env.board_size, int(ansi), int(swapped),
                                    int(result), buf, len(buf))
    return buf.raw[:k].decode("latin-1")


def action_to_string(env, player, a):
    buf = C.create_string_buffer(16)
    env.lib.twx_action_to_string(env.board_size, player, a, buf, 16)
    return buf.value.decode()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board_size", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-color", action="store_true")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    env = TwixtBatchEnv(args.board_size, 1)
    n, swapped, result = args.board_size, 0, 0
    print("Initial state:\n" + to_string(env, not args.no_color, 0, 0))
    while result == 0:
        mask, _ = env.legal_mask()
        _, _, cp = env.status()
        words = mask.cpu().numpy().view(np.uint32)[0]
        legal = [a for a in range(n * n) if (int(words[a >> 5]) >> (a & 31)) & 1]
        a = rng.choice(legal)
        player = int(cp[0])
        print("player %d chose action %d (%s)" % (player, a, action_to_string(env, player, a)))
        st = int(env.step([a])[0])
        assert not st & 4
        result, swapped = st & 3, (st >> 5) & 1
        print("State:\n" + to_string(env, not args.no_color, swapped, result))
    print("Returns:", {1: "1 -1", 2: "-1 1", 3: "0 0"}[result])


if __name__ == "__main__":
    main()
