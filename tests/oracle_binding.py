"""ctypes binding of oracle/libtwixt_oracle.so -- the CPU checker.

Test infrastructure only: nothing under twixt_for_open_spiel_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libtwixt_oracle.so")

MAX_N = 24
MAX_LEGAL = MAX_N * (MAX_N - 2)


class Cell(C.Structure):
    _fields_ = [("color", C.c_int), ("links", C.c_int), ("blocked", C.c_int),
                ("ltb", C.c_ubyte * 2 * 2)]


class State(C.Structure):
    _fields_ = [("n", C.c_int), ("move_counter", C.c_int), ("swapped", C.c_int),
                ("m1x", C.c_int), ("m1y", C.c_int), ("result", C.c_int),
                ("current_player", C.c_int),
                ("cell", Cell * MAX_N * MAX_N),
                ("nlegal", C.c_int * 2),
                ("legal", C.c_int * MAX_LEGAL * 2)]


class Kat(C.Structure):
    _fields_ = [("moves", C.c_int), ("result", C.c_int), ("swapped", C.c_int), ("links", C.c_int),
                ("hist_fnv", C.c_uint64), ("step_fnv", C.c_uint64), ("final_fnv", C.c_uint64)]


def build(force=False):
    src = os.path.join(ORACLE_DIR, "twixt_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libtwixt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(build())
    P = C.POINTER
    lib.orc_init.argtypes = [P(State), C.c_int]
    lib.orc_init.restype = None
    lib.orc_apply.argtypes = [P(State), C.c_int]
    lib.orc_apply.restype = C.c_int
    lib.orc_board_apply.argtypes = [P(State), C.c_int, C.c_int]
    lib.orc_board_apply.restype = None
    lib.orc_is_terminal.argtypes = [P(State)]
    lib.orc_legal_actions.argtypes = [P(State), P(C.c_int)]
    lib.orc_returns.argtypes = [P(State), P(C.c_double)]
    lib.orc_returns.restype = None
    lib.orc_observation.argtypes = [P(State), P(C.c_float)]
    lib.orc_observation.restype = None
    lib.orc_to_string.argtypes = [P(State), C.c_int, C.c_char_p, C.c_int]
    lib.orc_action_to_string.argtypes = [P(State), C.c_int, C.c_int, C.c_char_p]
    lib.orc_export_cells.argtypes = [P(State), P(C.c_uint8)]
    lib.orc_export_cells.restype = None
    lib.orc_export_planes.argtypes = [P(State), P(C.c_uint32)]
    lib.orc_kat_game.argtypes = [C.c_int, C.c_uint64, C.c_int, P(Kat)]
    lib.orc_kat_game.restype = None
    lib.orc_policy_hash.argtypes = [C.c_uint32] * 4
    lib.orc_policy_hash.restype = C.c_uint32
    lib.orc_rollout.argtypes = [P(State), C.c_uint32, C.c_uint32, P(C.c_uint32), C.c_int, C.c_int,
                                P(C.c_int32), P(C.c_uint8)]
    lib.orc_rollout.restype = C.c_int64
    lib.orc_bench.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, P(C.c_uint64)]
    lib.orc_bench.restype = C.c_int64
    lib.orc_size_ok.argtypes = [C.c_int]
    lib.orc_rollout_batch.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                      P(C.c_uint32), P(C.c_uint8), P(C.c_int32), P(C.c_uint8)]
    lib.orc_rollout_batch.restype = C.c_int64
    lib.orc_block_entry.argtypes = [C.c_int, C.c_int, P(C.c_int)]
    lib.orc_block_entry.restype = None
    _lib = lib
    return lib


class Board:
    """Thin object wrapper mirroring the reference's TwixTState surface."""

    def __init__(self, n=8):
        self.lib = load()
        if not self.lib.orc_size_ok(n):
            raise ValueError("board_size out of range [5..24]: %d" % n)
        self.s = State()
        self.lib.orc_init(C.byref(self.s), n)
        self.n = n
        self.episode = C.c_uint32(0)

    def legal_actions(self):
        buf = (C.c_int * MAX_LEGAL)()
        k = self.lib.orc_legal_actions(C.byref(self.s), buf)
        return list(buf[:k])

    def apply(self, a):
        if self.lib.orc_apply(C.byref(self.s), int(a)) != 0:
            raise ValueError("Not a legal action: %d" % a)

    def is_terminal(self):
        return bool(self.lib.orc_is_terminal(C.byref(self.s)))

    def current_player(self):
        return self.s.current_player

    def returns(self):
        r = (C.c_double * 2)()
        self.lib.orc_returns(C.byref(self.s), r)
        return [r[0], r[1]]

    @property
    def result(self):
        return self.s.result

    @property
    def move_counter(self):
        return self.s.move_counter

    def observation(self):
        n = self.n
        out = np.zeros(12 * n * (n - 2), dtype=np.float32)
        self.lib.orc_observation(C.byref(self.s), out.ctypes.data_as(C.POINTER(C.c_float)))
        return out.reshape(12, n, n - 2)

    def to_string(self, ansi=True):
        cap = 1 << 17
        buf = C.create_string_buffer(cap)
        k = self.lib.orc_to_string(C.byref(self.s), int(ansi), buf, cap)
        assert k < cap
        return buf.raw[:k].decode("latin-1")

    def action_to_string(self, player, a):
        buf = C.create_string_buffer(16)
        self.lib.orc_action_to_string(C.byref(self.s), player, a, buf)
        return buf.value.decode()

    def cells(self):
        n = self.n
        out = np.zeros((n, n, 4), dtype=np.uint8)
        self.lib.orc_export_cells(C.byref(self.s), out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def planes(self):
        n = self.n
        W = (n * n + 31) // 32
        out = np.zeros(8 * W + 4, dtype=np.uint32)
        k = self.lib.orc_export_planes(C.byref(self.s), out.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert k == out.size
        return out

    def rollout(self, seed, board, nsteps, auto_reset=True):
        """Returns (actions[int32 nsteps], status[uint8 nsteps], steps played)."""
        acts = np.zeros(nsteps, dtype=np.int32)
        st = np.zeros(nsteps, dtype=np.uint8)
        k = self.lib.orc_rollout(C.byref(self.s), seed, board, C.byref(self.episode), nsteps,
                                 int(auto_reset), acts.ctypes.data_as(C.POINTER(C.c_int32)),
                                 st.ctypes.data_as(C.POINTER(C.c_uint8)))
        return acts, st, k


def rollout_batch(n, seed, first_board, nboards, nsteps, auto_reset=True, threads=0,
                  want_planes=True, want_cells=False, want_trace=False):
    """Oracle rollouts of many boards.  Returns dict(steps, planes, cells, actions, status)."""
    lib = load()
    W = (n * n + 31) // 32
    SW = 8 * W + 4
    planes = np.zeros((nboards, SW), dtype=np.uint32) if want_planes else None
    cells = np.zeros((nboards, n * n, 4), dtype=np.uint8) if want_cells else None
    acts = np.zeros((nboards, nsteps), dtype=np.int32) if want_trace else None
    st = np.zeros((nboards, nsteps), dtype=np.uint8) if want_trace else None

    def p(a, t):
        return a.ctypes.data_as(C.POINTER(t)) if a is not None else None

    steps = lib.orc_rollout_batch(n, seed, first_board, nboards, nsteps, int(auto_reset), threads,
                                  p(planes, C.c_uint32), p(cells, C.c_uint8), p(acts, C.c_int32),
                                  p(st, C.c_uint8))
    return dict(steps=steps, planes=planes, cells=cells, actions=acts, status=st)
