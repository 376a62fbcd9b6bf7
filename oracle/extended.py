"""Extended-precision evaluation of the reference posterior -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference stores ``invK = inv(K + sn2 I)`` in fp64 (models/GP_Safe.py:231-232) and evaluates
``var = max(0, sf2 - (k^T invK) k)`` in fp64 (:343).  With the noise at the lower end of its search box (log sigma_n = -5,
models/GP_Safe.py:205-206) cond(K) reaches 1e8 and both steps round visibly, so "the oracle's fp64 value" is itself only
one rounding of the formula.  To tell a kernel's error from the formula's own, the same expressions are evaluated here in
``numpy.longdouble`` (x87 80-bit: eps 1.1e-19) -- slow (pure NumPy loops over n), for a few hundred points only:

  * ``posterior_given_invK``: the reference formula with the caller's fp64 ``invK`` taken as exact data (what the device's
    SBO_FACTOR_INVK mode and the fp64 oracle both approximate);
  * ``posterior_true``: the GP posterior itself, K factored in extended precision (what SBO_FACTOR_CHOL approximates; the
    reference's own invK differs from it by ~cond(K) eps).
"""
from __future__ import annotations

import numpy as np

from .gp_oracle import FLOAT32_EPS, mean_prior

LD = np.longdouble


def _kstar(points, ds, i):
    """k[n, N] in extended precision, the expanded distance of models/GP_Safe.py:112-119, 166."""
    d = ds["X_norm"].shape[1]
    h = np.asarray(ds["hypopt"], dtype=LD)[:, i]
    ell, sf2 = np.exp(2 * h[:d]), np.exp(2 * h[d])
    xn = (np.asarray(points, dtype=LD) - np.asarray(ds["X_mean"], dtype=LD)) / np.asarray(ds["X_std"], dtype=LD)
    v = ell ** LD(-0.5)
    A, B = np.asarray(ds["X_norm"], dtype=LD) * v, xn * v
    dist = -2 * (A @ B.T) + np.sum(A * A, axis=1)[:, None] + np.sum(B * B, axis=1)[None, :]
    return sf2 * np.exp(LD(-0.5) * dist), sf2


def _unnormalise(mean_n, var_n, ds, i):
    ys, ym = LD(ds["Y_std"][i]), LD(ds["Y_mean"][i])
    return mean_n * ys + ym, np.maximum(var_n, LD(0)) * ys * ys


def posterior_given_invK(points, ds):
    """(mean[N, q], var[N, q]) of models/GP_Safe.py:341-347 with exact arithmetic on the stored fp64 invK."""
    q = ds["Y_norm"].shape[1]
    mp = mean_prior(ds)
    mean, var = [], []
    for i in range(q):
        k, sf2 = _kstar(points, ds, i)
        invK = np.asarray(ds["invKopt"][i], dtype=LD)
        t = invK @ k                                                        # [n, N]
        rhs = np.asarray(ds["Y_norm"][:, i], dtype=LD) - LD(mp[i])
        m, v = _unnormalise(LD(mp[i]) + t.T @ rhs, sf2 - np.sum(t * k, axis=0), ds, i)
        mean.append(m)
        var.append(v)
    return np.stack(mean, axis=1), np.stack(var, axis=1)


def _cholesky_ld(K):
    n = K.shape[0]
    L = np.zeros_like(K)
    for j in range(n):
        s = K[j, j] - np.dot(L[j, :j], L[j, :j])
        L[j, j] = np.sqrt(s)
        if j + 1 < n:
            L[j + 1:, j] = (K[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def _solve_lower_ld(L, B):
    X = np.array(B, dtype=LD, copy=True)
    for j in range(L.shape[0]):
        X[j] = (X[j] - L[j, :j] @ X[:j]) / L[j, j]
    return X


def posterior_true(points, ds):
    """The GP posterior with K + (sn2 + float32 eps) I (models/GP_Safe.py:227-231) factored in extended precision."""
    d = ds["X_norm"].shape[1]
    q = ds["Y_norm"].shape[1]
    n = ds["X_norm"].shape[0]
    mp = mean_prior(ds)
    mean, var = [], []
    for i in range(q):
        k, sf2 = _kstar(points, ds, i)
        h = np.asarray(ds["hypopt"], dtype=LD)[:, i]
        v = np.exp(2 * h[:d]) ** LD(-0.5)
        A = np.asarray(ds["X_norm"], dtype=LD) * v
        dist = -2 * (A @ A.T) + np.sum(A * A, axis=1)[:, None] + np.sum(A * A, axis=1)[None, :]
        K = sf2 * np.exp(LD(-0.5) * dist) + (np.exp(2 * h[d + 1]) + LD(FLOAT32_EPS)) * np.eye(n, dtype=LD)
        L = _cholesky_ld(K)
        t = _solve_lower_ld(L, k)                                           # L^-1 k
        rhs = _solve_lower_ld(L, (np.asarray(ds["Y_norm"][:, i], dtype=LD) - LD(mp[i]))[:, None])[:, 0]
        m, vv = _unnormalise(LD(mp[i]) + t.T @ rhs, sf2 - np.sum(t * t, axis=0), ds, i)
        mean.append(m)
        var.append(vv)
    return np.stack(mean, axis=1), np.stack(var, axis=1)
