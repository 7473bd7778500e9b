import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Tile selection in the tests is the shipped table / its nearest-M fallback: the first-use autotuner
# (diffbir_amd/autotune.py) picks by timing, so two processes could run different (equally valid) kernels for a shape the
# table misses and comparisons between two engine runs would carry that rounding difference.  The autotuner has its own
# tests (test_pipeline_gpu.py::test_golden_under_first_use_autotune, test_abi_cpu.py) which switch it on.
os.environ.setdefault("DBIR_AUTOTUNE", "0")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
