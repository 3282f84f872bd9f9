"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path.

``-m "not gpu"`` tests run in the GPU-less build container (oracle vs golden vectors, host logic,
C-ABI symbol export, gloo ensemble sharding).  ``-m gpu`` tests are the parity tests proper: they
call the HIP path through the C ABI and compare with the oracle / golden vectors.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "default_pack_threshold: keep the library's default kernel selection for "
                                       "narrow worlds (the other GPU tests force the packed kernels on small batches)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(autouse=True)
def _library_test_hooks(monkeypatch):
    """The library honours its DW_TEST_* hooks (shrunk repair queues, injected allocation failures, forced fallback
    paths) only when DW_TEST_HOOKS is set, and reads every switch once, when a handle is created: the tests set it."""
    monkeypatch.setenv("DW_TEST_HOOKS", "1")
