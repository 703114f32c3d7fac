"""CPU-only: libtwixt_hip.so loads and exports every symbol include/twixt_hip.h declares."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    with open(os.path.join(ROOT, "include", "twixt_hip.h")) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(twx_[a-z_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = declared()
    for must in ("twx_create", "twx_destroy", "twx_reset", "twx_step", "twx_step_random", "twx_legal_mask",
                 "twx_status", "twx_observation", "twx_copy_boards", "twx_export_state", "twx_import_state",
                 "twx_export_cells", "twx_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from twixt_for_open_spiel_amd import build, _lib
    build.build()
    lib = _lib.load()
    for name in declared():
        assert hasattr(lib, name), name
    # pure host helpers are callable without a GPU
    assert lib.twx_state_words(24) == 8 * 18 + 4 and lib.twx_mask_words(24) == 18
    assert lib.twx_state_words(8) == 20 and lib.twx_state_words(12) == 44
    assert lib.twx_state_words(4) == -2 and lib.twx_state_words(25) == -2
    h = C.c_void_p()
    rc = lib.twx_create(30, 1, 0, 0, C.byref(h))           # tw.cc:139-144, twixt_test.cc:64-71
    assert rc == -2 and lib.twx_last_error().decode() == "board_size out of range [5..24]: 30"
    rc = lib.twx_create(3, 1, 0, 0, C.byref(h))
    assert rc == -2 and lib.twx_last_error().decode() == "board_size out of range [5..24]: 3"


def test_no_cpu_fallback():
    import pytest
    import torch
    from twixt_for_open_spiel_amd import TwixtBatchEnv, TwixtHipError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(TwixtHipError):
        TwixtBatchEnv(8, 4)
