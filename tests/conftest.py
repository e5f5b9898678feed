import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model():
    from hippopt_amd.robot_model import synthetic_ergocub
    return synthetic_ergocub()


@pytest.fixture(autouse=True)
def _no_page_locked_arrays_outlive_their_test(request):
    """GPU tests hand numpy arrays to handles that page-lock the ones they see twice in a row (hipnlp_set_auto_register).  When a test
    ends its arrays die; a handle that is still alive (or not collected yet) would keep the registration of freed memory, and the HIP
    runtime treats every later host pointer inside such a range as pinned memory of the old registration.  Between
    tests: collect the dead handles (hipnlp_destroy unregisters what a handle registered) and release what live handles still hold —
    the step a caller is told to take before freeing arrays it passed to hipnlp_eval."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    mod = sys.modules.get("hippopt_amd.hipnlp")
    if mod is None:
        return
    import gc
    gc.collect()
    for lib in list(getattr(mod, "_libs", {}).values()):   # (the product library, and the diagnostic build when a test loaded it: each keeps its own table)
        try:
            lib.hipnlp_host_release_auto_ranges()
        except Exception:  # noqa: BLE001
            pass
