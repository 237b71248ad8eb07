import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ieee_amd  # noqa: E402,F401  (before anything starts the HIP runtime: it picks GPU_MAX_HW_QUEUES, 1 under WORLD_SIZE > 1)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("IEEE_TEST_STACKS"):        # debugging aid: every thread's stack every N seconds into a file
        import faulthandler
        import threading
        import time
        out = open(os.environ.get("IEEE_TEST_STACKS_FILE", "stacks.log"), "w")

        def sample():
            while True:
                time.sleep(float(os.environ["IEEE_TEST_STACKS"]))
                out.write("==== %.1f %s\n" % (time.time(), os.environ.get("PYTEST_CURRENT_TEST", "")))
                out.flush()
                faulthandler.dump_traceback(file=out, all_threads=True)
                out.flush()
        threading.Thread(target=sample, daemon=True).start()


def pytest_collection_modifyitems(config, items):
    """A hung test (a wedged GPU queue, a DataLoader worker that never answers) must fail by itself instead of stalling the
    whole session: 10 minutes per test when pytest-timeout is installed (the slowest test takes 35 s)."""
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
