import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    gdir = os.path.join(ROOT, "tests", "golden")

    def load(name):
        return np.load(os.path.join(gdir, name), allow_pickle=False)
    return load


def pytest_collection_finish(session):
    """Tests that hand torch device tensors to the library need torch's HIP runtime up BEFORE the library's own
    (two runtimes share the process; the one that comes second after the other has touched the device reports
    "No HIP GPUs are available").  bench.py and the multi-process tests already start with torch; the single-process
    GPU run does it here, once, when GPU tests were selected and a GPU is there."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
