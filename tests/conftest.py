import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    here = os.path.join(REPO, "tests", "golden")
    with open(os.path.join(here, "meta.json")) as handle:
        meta = json.load(handle)
    return np.load(os.path.join(here, "scenarios.npz")), meta
