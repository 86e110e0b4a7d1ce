import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # tests/ may use the oracle (checker only)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def all_golden(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, prefix + "*.npz")))


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_py
    oracle_py.build()
    return oracle_py
