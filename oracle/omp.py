"""ctypes front end of oracle/c/sweep_omp.c -- the C + OpenMP restatement of the SafeOpt sweep (posterior, bounds, S / U /
M masks, u*, arg-max) used as the multi-threaded CPU column of bench.py and checked against the NumPy oracle in
tests/test_oracle.py.  TEST / BASELINE INFRASTRUCTURE ONLY, like everything under oracle/.  Built on first use with the
image's gcc (``oracle/_build/libsweep_omp.so``; `__graft_entry__.build()` builds it too; -mavx2 rather than -march=native,
because the built file travels from the build container to the GPU box)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "sweep_omp.c")
LIB = os.path.join(HERE, "_build", "libsweep_omp.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-mavx2", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off", SRC, "-o", LIB, "-lm"])
    return LIB


class _Model(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int), ("d", ctypes.c_int), ("q", ctypes.c_int)] + \
               [(k, ctypes.c_void_p) for k in ("X_norm", "Y_norm", "invK", "hyp", "X_mean", "X_std", "Y_mean", "Y_std")]


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.sweep_omp_threads.restype = ctypes.c_int
        _lib.sweep_omp_safeopt.restype = ctypes.c_int
    return _lib


def threads():
    return int(_load().sweep_omp_threads())


def set_threads(n):
    _load().sweep_omp_set_threads(ctypes.c_int(int(n)))


def safeopt_sweep(lo, hi, count, ds, b, first=0, n=None, want_posterior=False, want_masks=False):
    """The sweep over candidates [first, first + n) of the grid.  Returns a dict with count_S, count_M, minimizer_index,
    u_star and, on request, mean / var [N, q] and the S / U / M masks."""
    lib = _load()
    X_norm = np.ascontiguousarray(ds["X_norm"], dtype=np.float64)
    Y_norm = np.ascontiguousarray(ds["Y_norm"], dtype=np.float64)
    nobs, d = X_norm.shape
    q = Y_norm.shape[1]
    invK = np.ascontiguousarray(np.stack([np.asarray(a, dtype=np.float64) for a in ds["invKopt"]]))
    hyp = np.ascontiguousarray(ds["hypopt"], dtype=np.float64)
    vecs = {k: np.ascontiguousarray(ds[k], dtype=np.float64) for k in ("X_mean", "X_std", "Y_mean", "Y_std")}
    keep = [X_norm, Y_norm, invK, hyp] + list(vecs.values())
    m = _Model(nobs, d, q, X_norm.ctypes.data, Y_norm.ctypes.data, invK.ctypes.data, hyp.ctypes.data,
               vecs["X_mean"].ctypes.data, vecs["X_std"].ctypes.data, vecs["Y_mean"].ctypes.data, vecs["Y_std"].ctypes.data)
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    cnt = np.ascontiguousarray(count, dtype=np.int64)
    total = int(np.prod(cnt))
    N = total - first if n is None else int(n)
    mean = np.empty((N, q)) if want_posterior else None
    var = np.empty((N, q)) if want_posterior else None
    masks = [np.empty(N, dtype=np.uint8) if want_masks else None for _ in range(3)]
    res = np.zeros(3, dtype=np.int64)
    ustar = ctypes.c_double(0.0)
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(None)
    rc = lib.sweep_omp_safeopt(ctypes.byref(m), ptr(lo), ptr(hi), ptr(cnt), ctypes.c_longlong(first), ctypes.c_longlong(N),
                               ctypes.c_double(b), ptr(mean), ptr(var), ptr(masks[0]), ptr(masks[1]), ptr(masks[2]), ptr(res),
                               ctypes.byref(ustar))
    del keep
    if rc:
        raise ValueError("sweep_omp_safeopt: unsupported shape")
    out = {"count_S": int(res[0]), "count_M": int(res[1]), "minimizer_index": int(res[2]), "u_star": float(ustar.value)}
    if want_posterior:
        out.update(mean=mean, var=var)
    if want_masks:
        out.update(S=masks[0].astype(bool), U=masks[1].astype(bool), M=masks[2].astype(bool))
    return out
