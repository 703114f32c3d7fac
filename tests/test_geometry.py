"""The compile-time geometry the HIP kernels use (twixt_geom.h: crossing masks
derived from segment intersection, region masks, initial planes) against the
oracle's restated reference tables (twixtboard.cc:38-144, 209-276).  CPU-only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle_binding import Board, load

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("geom") / "geom_dump")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(HERE, "geom_dump.cc")])
    out = subprocess.check_output([exe]).decode().splitlines()
    d = {"cross": {}, "N": {}}
    cur = None
    for ln in out:
        t = ln.split()
        if t[0] == "cross":
            d["cross"][int(t[1])] = [int(v) for v in t[2:]]
        elif t[0] == "N":
            cur = int(t[1])
            d["N"][cur] = {"W": int(t[2]), "SW": int(t[3]), "region": {}}
        elif t[0] == "region":
            d["N"][cur]["region"][int(t[1])] = np.array([int(v) for v in t[2:]], dtype=np.uint32)
        elif t[0] == "init":
            d["N"][cur]["init"] = np.array([int(v) for v in t[1:]], dtype=np.uint32)
    return d


def test_crossing_masks_equal_reference_table(dump):
    lib = load()
    for d in range(8):
        want = [0] * 5
        for i in range(9):
            e = (C.c_int * 3)()
            lib.orc_block_entry(d, i, e)
            fx, fy, ldir = e[0], e[1], e[2]
            assert -3 <= fx <= 1 and -3 <= fy <= 3 and 0 <= ldir <= 3
            want[fx + 3] |= 1 << (8 * ldir + fy + 3)
        assert dump["cross"][d] == want, "direction %d" % d


@pytest.mark.parametrize("n", range(5, 25))
def test_region_and_initial_planes(dump, n):
    g = dump["N"][n]
    b = Board(n)
    assert g["W"] == (n * n + 31) // 32 and g["SW"] == 8 * g["W"] + 4
    assert np.array_equal(g["init"], b.planes())              # tb.cc:209-236 initial position
    for p in range(2):                                          # tb.cc:252-276 initial legal lists
        la = [b.s.legal[p][i] for i in range(b.s.nlegal[p])]
        bits = [a for a in range(n * n) if (int(g["region"][p][a >> 5]) >> (a & 31)) & 1]
        assert bits == la and len(la) == n * (n - 2)
