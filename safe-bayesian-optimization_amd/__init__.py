"""safebo_amd -- MI355X-native SafeOpt / GoOSE candidate sweep (host side).

Directory name on disk is ``safe-bayesian-optimization_amd``; import it as ``safebo_amd`` through the
shim module at the repository root.  Compute happens only in ``libsafebo.so`` (hand-written gfx950 HIP
kernels behind the C ABI of ``include/safebo.h``); this package is NumPy + ctypes.
"""
from . import _lib
from .engine import SweepEngine
from ._lib import SafeBOError, EmptySafeSetError
from . import GP_Safe, SafeOpt, GoOSE, GP_TR   # mirrors of the reference's models/GP_Safe.py, SafeOpt.py, GoOSE.py, GP_TR.py

__all__ = ["SweepEngine", "SafeBOError", "EmptySafeSetError", "GP_Safe", "SafeOpt", "GoOSE", "GP_TR", "_lib"]
