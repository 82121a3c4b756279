import json
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


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def g2():
    return load_npz("g2_frames.npz"), load_json("g2_index.json")


@pytest.fixture(scope="session")
def g4():
    return load_npz("g4_p1.npz")


@pytest.fixture(scope="session")
def g6():
    return load_npz("g6_p1_more.npz")


@pytest.fixture(scope="session")
def g5():
    return load_npz("g5_edge.npz")
