import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the shared library is a build artefact (git-ignored): build it when a fresh checkout runs the tests first
    lib = os.path.join(ROOT, "safe-bayesian-optimization_amd", "libsafebo.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "safe-bayesian-optimization_amd", "csrc"), "-j4"], check=True)


@pytest.fixture(scope="session")
def engine():
    """One SweepEngine on device 0 for the GPU tests.  No skip: on a GPU box a missing library or device
    must fail the run (there is no CPU fallback to fall back to)."""
    import safebo_amd
    eng = safebo_amd.SweepEngine(0)
    yield eng
    eng.close()
