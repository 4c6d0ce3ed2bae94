import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    d = np.load(os.path.join(REPO, "tests", "golden", "reference_golden.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(REPO, "tests", "golden", "reference_golden.json")) as f:
        return json.load(f)
