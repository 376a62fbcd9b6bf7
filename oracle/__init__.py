"""CPU oracle for the SafeOpt / GoOSE candidate sweep -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference algorithm
(dleeim/Safe-Bayesian-Optimization, models/GP_Safe.py, models/SafeOpt.py,
models/GoOSE.py, test/test_SafeOpt.py:324-345).  It exists to *check* the HIP
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package
(``safe-bayesian-optimization_amd``) never does and fails loudly when the HIP
library is missing.

PARITY UNPINNED: the reference's own tests contain no assertion, golden
vector or known answer for this path, and the reference
cannot be executed in the build container (``import jax`` raises
ModuleNotFoundError -- an ordinary error, no wheel, no network).  The oracle is
therefore pinned only by (a) the reference's *adjacent* constants (Benoit
optimum, see tests/test_oracle.py) and (b) an independent implementation of
the same GP posterior (scikit-learn's GaussianProcessRegressor with fixed
hyper-parameters, tests/test_oracle.py).  Every function cites the reference
file:line it restates.
"""
from .gp_oracle import *  # noqa: F401,F403
