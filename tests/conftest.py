import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # Big boards (split layout, N >= 17): batches too small to fill the chip normally skip the journal (they rewrite their
    # boards every step: twixt_engine.hip run_step).  The parity tests are small batches and are there to test
    # the journal path -- the windowed kernel the benchmark runs -- so they switch that routing off;
    # tests/test_gpu_pair_kernel.py::test_small_batches_rewrite_in_place covers the default.
    os.environ.setdefault("TWX_SMALL_BATCH_REWRITE", "0")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.load()
