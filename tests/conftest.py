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
    return importlib.import_module("u-vip-slam_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("u-vip-slam_amd.synth")
