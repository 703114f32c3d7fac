"""ctypes binding of libtwixt_hip.so (include/twixt_hip.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this
module raises, loudly.
"""
import ctypes as C
import os

import torch  # noqa: F401  (first: the process must share ONE libamdhip64 with torch)

PKG = os.path.dirname(os.path.abspath(__file__))
# (TWIXT_HIP_LIB: a diagnostic build of the same library, e.g. tools/bin/*.so -- tools/README.md)
LIB_PATH = os.environ.get("TWIXT_HIP_LIB") or os.path.join(PKG, "libtwixt_hip.so")

TWX_OK = 0
ST_RESULT_MASK, ST_ILLEGAL, ST_RESET, ST_SWAPPED, ST_NOOP = 0x03, 0x04, 0x10, 0x20, 0x40

_lib = None


class TwixtHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TwixtHipError(
            "libtwixt_hip.so not built (%s). Run `python -m twixt_for_open_spiel_amd.build` "
            "or __graft_entry__.build(); there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32p, u8p, u32p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    lib.twx_create.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.twx_destroy.argtypes = [vp]
    lib.twx_board_size.argtypes = [vp]
    lib.twx_batch.argtypes = [vp]
    lib.twx_batch.restype = C.c_int64
    lib.twx_state_words.argtypes = [C.c_int]
    lib.twx_mask_words.argtypes = [C.c_int]
    lib.twx_reset.argtypes = [vp, vp]
    lib.twx_reset_boards.argtypes = [vp, i32p, C.c_int64, vp]
    lib.twx_step.argtypes = [vp, i32p, u8p, vp]
    lib.twx_step_random.argtypes = [vp, C.c_uint32, C.c_int, i32p, u8p, vp]
    lib.twx_step_boards.argtypes = [vp, i32p, C.c_int64, i32p, u8p, vp]
    lib.twx_step_random_boards.argtypes = [vp, i32p, C.c_int64, C.c_uint32, C.c_int, i32p, u8p, vp]
    lib.twx_random_rollout.argtypes = [vp, i32p, C.c_int64, C.c_uint32, C.c_int, u8p, vp]
    lib.twx_legal_mask.argtypes = [vp, u32p, i32p, vp]
    lib.twx_legal_mask_boards.argtypes = [vp, i32p, C.c_int64, u32p, i32p, vp]
    lib.twx_status.argtypes = [vp, u8p, i32p, vp, vp]
    lib.twx_status_boards.argtypes = [vp, i32p, C.c_int64, u8p, i32p, vp, vp]
    lib.twx_observation.argtypes = [vp, vp, vp]
    lib.twx_observation_boards.argtypes = [vp, i32p, C.c_int64, vp, vp]
    lib.twx_copy_boards.argtypes = [vp, i32p, vp, i32p, C.c_int64, vp]
    lib.twx_export_state.argtypes = [vp, i32p, C.c_int64, u32p, vp]
    lib.twx_import_state.argtypes = [vp, i32p, C.c_int64, u32p, vp]
    lib.twx_export_cells.argtypes = [vp, i32p, C.c_int64, u8p, vp]
    lib.twx_cells_to_string.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    lib.twx_action_to_string.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_char_p, C.c_int]
    lib.twx_last_error.argtypes = []
    lib.twx_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc):
    if rc != TWX_OK:
        raise TwixtHipError("twixt_hip error %d: %s" % (rc, load().twx_last_error().decode()))
