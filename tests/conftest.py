import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def task_const():
    from isaacgymdyros_amd.task_constants import load_task_constants
    return load_task_constants()


@pytest.fixture(scope="session")
def model():
    from isaacgymdyros_amd.model import load_model
    return load_model()
