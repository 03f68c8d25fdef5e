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
    data = dict(np.load(os.path.join(here, "scenarios.npz")))
    for tag in ("extra", "n4", "a9", "r2", "r3", "r3b"):                            # scenarios added later (make_golden.py --extra / --n4 / --a9)
        extra = os.path.join(here, f"scenarios_{tag}.npz")
        if os.path.isfile(extra):
            data.update(np.load(extra))
            with open(os.path.join(here, f"meta_{tag}.json")) as handle:
                more = json.load(handle)
                meta["scenarios"].update(more.pop("scenarios"))
                meta.update(more)
    return data, meta
