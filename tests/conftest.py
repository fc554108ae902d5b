import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import __graft_entry__ as g
    g.build()  # compiles only what is out of date; the GPU box receives the prebuilt .so files


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
