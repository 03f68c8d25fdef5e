import faulthandler
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# No unbounded wait anywhere (round 3's driver run died inside ONE 2-process test that had no timeout and took every later test with it):
#  * every test gets a pytest-timeout limit (signal method: the test FAILS and the run goes on) ...
#  * ... and, for a test stuck inside a C call where no Python signal handler runs (a device sync that never returns), a faulthandler
#    watchdog re-armed per test dumps every thread's traceback and ends the process with a non-zero code instead of hanging the box;
#  * multi-process tests start their ranks through tests/helpers.spawn_bounded (wall-clock cap, children killed, test failed) and create
#    process groups with a 120 s timeout.
# The tests also hold the EXPERIMENTAL kernel forms (built, bit-identical, off by default because they lost their same-box A/B) to the forms that ship: their
# switches act only together with FB_EXPERIMENTAL=1 (csrc/runtime.cpp fb_experimental, engine.py) -- set for every test process and the ranks it spawns.  On its
# own FB_EXPERIMENTAL selects nothing: the default dispatch is what the suite runs wherever a test does not name a switch.
os.environ.setdefault("FB_EXPERIMENTAL", "1")

TEST_TIMEOUT_S = int(os.environ.get("FB_TEST_TIMEOUT_S", "240"))
WATCHDOG_S = int(os.environ.get("FB_TEST_WATCHDOG_S", "420"))

# GPU files in the order of their value as evidence: the reference-pinned scenarios first, the spawn-heavy multi-process file last
FILE_ORDER = ["test_gpu_engine.py", "test_gpu_gradreg.py", "test_gpu_training.py", "test_gpu_bf16_structural.py", "test_gpu_bf16_parity.py",
              "test_gpu_ops.py", "test_gpu_sharded.py"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "soak: long repetition / large-shape GPU cases, run only with FB_SOAK=1 (tools/race_probe.py holds the full soaks)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(FILE_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), -1))          # stable: the order inside a file is kept
    has_timeout = config.pluginmanager.hasplugin("timeout")
    soak = os.environ.get("FB_SOAK") == "1"
    for it in items:
        if has_timeout and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(TEST_TIMEOUT_S))
        if it.get_closest_marker("soak") is not None and not soak:
            it.add_marker(pytest.mark.skip(reason="soak case: FB_SOAK=1 runs it"))


@pytest.fixture(autouse=True)
def _watchdog():
    faulthandler.dump_traceback_later(WATCHDOG_S, exit=True)
    yield
    faulthandler.cancel_dump_traceback_later()


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    here = os.path.join(REPO, "tests", "golden")
    with open(os.path.join(here, "meta.json")) as handle:
        meta = json.load(handle)
    data = dict(np.load(os.path.join(here, "scenarios.npz")))
    for tag in ("extra", "n4", "a9", "r2", "r3", "r3b", "r4"):                      # scenarios added later (make_golden.py --extra / --n4 / --a9 ...)
        extra = os.path.join(here, f"scenarios_{tag}.npz")
        if os.path.isfile(extra):
            data.update(np.load(extra))
            with open(os.path.join(here, f"meta_{tag}.json")) as handle:
                more = json.load(handle)
                meta["scenarios"].update(more.pop("scenarios"))
                meta.update(more)
    return data, meta
