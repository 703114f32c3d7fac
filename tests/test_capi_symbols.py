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


def test_argument_errors_without_a_device():
    """Every entry point rejects a NULL handle (and create its bad arguments) with TWX_E_ARG
    and a message, before touching the device: no exception, no crash crosses the ABI."""
    from twixt_for_open_spiel_amd import build, _lib
    build.build()
    lib = _lib.load()
    E_ARG = -1
    h = C.c_void_p()
    assert lib.twx_create(8, 0, 0, 0, C.byref(h)) == E_ARG and b"batch" in lib.twx_last_error()
    assert lib.twx_create(8, -5, 0, 0, C.byref(h)) == E_ARG
    assert lib.twx_create(8, 4, 0, 0, None) == E_ARG
    # board ids are 32-bit policy keys: a shard that would wrap is refused, not aliased
    assert lib.twx_create(8, 4, 0, 1 << 32, C.byref(h)) == E_ARG and b"2^32" in lib.twx_last_error()
    assert lib.twx_create(8, 4, 0, (1 << 32) - 3, C.byref(h)) == E_ARG
    lib.twx_destroy(None)  # a no-op
    calls = [
        lambda: lib.twx_reset(None, None),
        lambda: lib.twx_reset_boards(None, None, 0, None),
        lambda: lib.twx_step(None, None, None, None),
        lambda: lib.twx_step_random(None, 1, 1, None, None, None),
        lambda: lib.twx_step_boards(None, None, 0, None, None, None),
        lambda: lib.twx_step_random_boards(None, None, 0, 1, 1, None, None, None),
        lambda: lib.twx_random_rollout(None, None, 0, 1, 10, None, None),
        lambda: lib.twx_legal_mask(None, None, None, None),
        lambda: lib.twx_legal_mask_boards(None, None, 0, None, None, None),
        lambda: lib.twx_status(None, None, None, None, None),
        lambda: lib.twx_status_boards(None, None, 0, None, None, None, None),
        lambda: lib.twx_observation(None, None, None),
        lambda: lib.twx_observation_boards(None, None, 0, None, None),
        lambda: lib.twx_copy_boards(None, None, None, None, 0, None),
        lambda: lib.twx_export_state(None, None, 0, None, None),
        lambda: lib.twx_import_state(None, None, 0, None, None),
        lambda: lib.twx_export_cells(None, None, 0, None, None),
    ]
    for i, call in enumerate(calls):
        assert call() == E_ARG, i
        assert lib.twx_last_error().decode().startswith("twx_"), i
    buf = C.create_string_buffer(64)
    assert lib.twx_action_to_string(8, 0, 64, buf, 64) == E_ARG      # action out of range
    assert lib.twx_action_to_string(30, 0, 1, buf, 64) == E_ARG
    assert lib.twx_cells_to_string(None, 8, 0, 0, 0, buf, 64) == E_ARG
    assert lib.twx_board_size(None) < 0 and lib.twx_batch(None) < 0
