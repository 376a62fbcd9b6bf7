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
    eng.set_option("guard_audit_every", 1)       # (production default: one sweep in 16; the suite audits every sweep it runs)
    yield eng
    eng.close()


@pytest.fixture(autouse=True)
def _guard_band_audit(request):
    """Behind every GPU test that swept on the shared engine: the standing audit of the guard band (sbo_profile.guard_audit_*,
    csrc/guard.hip) must not have seen one sampled value outside the band the masks and indices rest on."""
    yield
    if "engine" in request.fixturenames:
        prof = request.getfixturevalue("engine").profile()
        assert prof["guard_audit_violations"] == 0, (prof["guard_audit_violations"], prof["guard_audit_samples"], prof["guard_audit_worst"])
