import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def uvo():
    """The product package (ctypes over libuvo.so).  GPU tests fail loudly if the HIP library is missing."""
    # Some tests hand torch device buffers to the library.  torch ships its own copy of the HIP runtime; when the system runtime
    # (libuvo's) has already opened the GPU, torch's copy can come up with "No HIP GPUs are available" -- so let torch
    # initialise first, as bench.py does.  CPU-only runs skip this (is_available() is False there).
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except ImportError:
        pass
    return importlib.import_module("u-vip-slam_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("u-vip-slam_amd.synth")
