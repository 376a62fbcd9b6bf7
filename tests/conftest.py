import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the shared library is a build artefact (git-ignored): always run make -- a no-op when it is up to date, and a
    # library older than its sources must never be what the suite tests
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "safe-bayesian-optimization_amd", "csrc"), "-j4"], check=True,
                   stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def engine():
    """One SweepEngine on device 0 for the GPU tests.  No skip: on a GPU box a missing library or device
    must fail the run (there is no CPU fallback to fall back to)."""
    import safebo_amd
    eng = safebo_amd.SweepEngine(0)
    yield eng
    eng.close()
