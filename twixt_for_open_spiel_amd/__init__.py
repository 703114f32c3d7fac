"""twixt_for_open_spiel_amd: MI355X-native batched TwixT board engine.

The product is libtwixt_hip.so (C ABI in include/twixt_hip.h); this package is
its thin host-side mirror.  Import never falls back to a CPU path.
"""
from ._lib import TwixtHipError, load  # noqa: F401
from .batch_env import TwixtBatchEnv  # noqa: F401

__all__ = ["TwixtBatchEnv", "TwixtHipError", "load"]
