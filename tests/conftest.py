import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def atlas():
    a = np.load(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_score_atlas.npz"))["atlas"]
    return np.ascontiguousarray(a)


@pytest.fixture(scope="session")
def golden_dyn():
    return np.load(os.path.join(GOLDEN, "pong_dynamics.npz"))


@pytest.fixture(scope="session")
def golden_frames():
    return np.load(os.path.join(GOLDEN, "pong_frames.npz"))
