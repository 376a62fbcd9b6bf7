"""Synthetic inputs for the sweep benchmarks and parity tests.

NumPy ``default_rng(20250404 + config_index)`` (PCG64, platform-stable).  Plants are closed-form
restatements of the reference's plant callables, used only to label the synthetic observations:
Benoit ``f = u0^2 + u1^2 + u0 u1`` (problems/Benoit_Problem.py:14-19) with the tight constraint
``g = -(1 - u0 + u1^2 + 2 u1)`` (problems/Benoit_Problem.py:38-44); a build-defined 4-D chained
Rosenbrock (the reference's problems/Rosenbrock_Problem.py:15-18 is 2-D only) with the same constraint on
(x0, x1); the Williams-Otto reactor on its box [[4,7],[70,100]] (problems/WilliamOttoReactor_Problem.py:19-93,
test/test_GoOSE.py:276-280: objective + two constraints, each the steady state of six mass balances -- solved here by a
vectorised Newton iteration in NumPy so that the config can be built without a GPU; ``sbo_plant_wo`` is the device
evaluator of the same plant); and a 6-D sum of sines.
Hyper-parameters are fixed, not fitted: log ell = -0.5, log sigma_f = 0, log sigma_n = -2 for every
output (inside the reference bounds models/GP_Safe.py:205-206).
"""
from __future__ import annotations

import numpy as np

FLOAT32_EPS = float(np.finfo(np.float32).eps)   # models/GP_Safe.py:229
SEED0 = 20250404


def benoit(X):
    f = X[:, 0] ** 2 + X[:, 1] ** 2 + X[:, 0] * X[:, 1]
    g = -(1.0 - X[:, 0] + X[:, 1] ** 2 + 2.0 * X[:, 1])
    return np.stack([f, g], axis=1)


def rosenbrock4(X):
    f = np.zeros(X.shape[0])
    for a in range(3):
        f += 100.0 * (X[:, a + 1] - X[:, a] ** 2) ** 2 + (1.0 - X[:, a]) ** 2
    g = -(1.0 - X[:, 0] + X[:, 1] ** 2 + 2.0 * X[:, 1])
    return np.stack([f, g], axis=1)


def williams_otto(X, iters=80):
    """Noise-free outputs of the reference's William-Otto reactor for input rows X[N, 2] = (Fb, Tr): [N, 3] =
    (get_objective, get_constraint1, get_constraint2) of problems/WilliamOttoReactor_Problem.py:45-93.  The reference finds
    the steady state of the six balances (:19-43) with ``fsolve`` from x0 = 0.1, once per point and output; here all rows
    take Newton steps together (analytic Jacobian; a step is halved until the mass fractions stay non-negative).  Checked
    against the reference's own 100 x 100 table in tests/test_oracle.py."""
    X = np.asarray(X, dtype=np.float64)
    Fb, Tr = X[:, 0], X[:, 1]
    Fa, Vr = 1.8275, 2105.2
    Fr = Fa + Fb
    k1, k2, k3 = (k0 * np.exp(-eta / (Tr + 273)) for k0, eta in ((1.6599e6, 6666.7), (7.2177e8, 8333.3), (2.6745e12, 11111.0)))
    w = np.full((X.shape[0], 6), 0.1)
    a = -Fr / Vr
    for _ in range(iters):
        xa, xb, xc, xp, xe, xg = w.T
        f = np.stack([(Fa - Fr * xa - Vr * xa * xb * k1) / Vr,
                      (Fb - Fr * xb - Vr * xa * xb * k1 - Vr * xb * xc * k2) / Vr,
                      a * xc + 2 * xa * xb * k1 - 2 * xb * xc * k2 - xc * xp * k3,
                      a * xp + xb * xc * k2 - 0.5 * xp * xc * k3,
                      a * xe + 2 * xb * xc * k2,
                      a * xg + 1.5 * xp * xc * k3], axis=1)
        J = np.zeros((X.shape[0], 6, 6))
        J[:, 0, 0], J[:, 0, 1] = a - xb * k1, -xa * k1
        J[:, 1, 0], J[:, 1, 1], J[:, 1, 2] = -xb * k1, a - xa * k1 - xc * k2, -xb * k2
        J[:, 2, 0], J[:, 2, 1], J[:, 2, 2], J[:, 2, 3] = 2 * xb * k1, 2 * xa * k1 - 2 * xc * k2, a - 2 * xb * k2 - xp * k3, -xc * k3
        J[:, 3, 1], J[:, 3, 2], J[:, 3, 3] = xc * k2, xb * k2 - 0.5 * xp * k3, a - 0.5 * xc * k3
        J[:, 4, 1], J[:, 4, 2], J[:, 4, 4] = 2 * xc * k2, 2 * xb * k2, a
        J[:, 5, 2], J[:, 5, 3], J[:, 5, 5] = 1.5 * xp * k3, 1.5 * xc * k3, a
        step = np.linalg.solve(J, -f[:, :, None])[:, :, 0]
        lam = np.ones(X.shape[0])
        for _ in range(30):
            neg = np.any(w + lam[:, None] * step < 0, axis=1)
            if not neg.any():
                break
            lam[neg] *= 0.5
        step *= lam[:, None]
        w = w + step
        if np.max(np.abs(step)) < 1e-15:
            break
    fx = 1043.38 * w[:, 3] * Fr + 20.92 * w[:, 4] * Fr - 79.23 * Fa - 118.34 * Fb
    return np.stack([-fx, 0.12 - w[:, 0], 0.08 - w[:, 5]], axis=1)


def sines6(X, rng):
    y = np.sum(np.sin(3.0 * X), axis=1) + 0.1 * rng.standard_normal(X.shape[0])
    return y[:, None]


CONFIGS = {
    # name: (config_index, bound, plant, q, b)
    "A": dict(index=0, bound=[[-0.6, 1.5], [-1.0, 1.0]], plant="benoit", n=20, count=[50, 50], b=3.0, dtype="f64"),
    "B": dict(index=1, bound=[[-0.6, 1.5], [-1.0, 1.0]], plant="benoit", n=128, count=[2048, 2048], b=3.0, dtype="f64"),
    "C": dict(index=2, bound=[[4.0, 7.0], [70.0, 100.0]], plant="wo", n=256, count=[1024, 1024], b=2.0, dtype="f64"),
    "D": dict(index=3, bound=[[-2.0, 2.0]] * 4, plant="rosenbrock4", n=128, count=[128] * 4, b=3.0, dtype="f64"),
    "E": dict(index=4, bound=[[-1.0, 1.0]] * 6, plant="sines6", n=2048, count=None, N=10_000_000, b=3.0, dtype="f32"),
    "H": dict(index=5, bound=[[-0.6, 1.5], [-1.0, 1.0]], plant="benoit", n=512, count=[4096, 4096], b=3.0, dtype="f64"),
}


def data_normalization(X, Y):
    """models/GP_Safe.py:84-96 (population std)."""
    X_mean, X_std = np.mean(X, axis=0), np.std(X, axis=0)
    Y_mean, Y_std = np.mean(Y, axis=0), np.std(Y, axis=0)
    return (X - X_mean) / X_std, (Y - Y_mean) / Y_std, X_mean, X_std, Y_mean, Y_std


def build_invK(X_norm, hypopt):
    """models/GP_Safe.py:226-232: inv(sf2 exp(-1/2 D) + (sn2 + float32 eps) I), expanded distance (:119)."""
    n, d = X_norm.shape
    out = []
    for i in range(hypopt.shape[1]):
        ell = np.exp(2.0 * hypopt[:d, i])
        sf2 = np.exp(2.0 * hypopt[d, i])
        sn2 = np.exp(2.0 * hypopt[d + 1, i]) + FLOAT32_EPS
        Xa = X_norm * ell ** -0.5
        dist = -2 * np.dot(Xa, Xa.T) + np.sum(Xa ** 2, axis=1)[:, None] + np.sum(Xa ** 2, axis=1)
        out.append(np.linalg.inv(sf2 * np.exp(-0.5 * dist) + sn2 * np.eye(n)))
    return out


def make_dataset(X, Y, hypopt):
    """The ``inference_datasets`` dict (models/GP_Safe.py:236-245) for fixed hyper-parameters."""
    X_norm, Y_norm, X_mean, X_std, Y_mean, Y_std = data_normalization(np.asarray(X, float), np.asarray(Y, float))
    hypopt = np.asarray(hypopt, dtype=np.float64)
    return {"X_mean": X_mean, "X_std": X_std, "Y_mean": Y_mean, "Y_std": Y_std, "X_norm": X_norm,
            "Y_norm": Y_norm, "invKopt": build_invK(X_norm, hypopt), "hypopt": hypopt}


def default_hypopt(d, q, log_ell=-0.5, log_sf=0.0, log_sn=-2.0):
    h = np.empty((d + 2, q))
    h[:d] = log_ell
    h[d] = log_sf
    h[d + 1] = log_sn
    return h


def make_config(name, n=None, seed=None):
    """Observations + model state for a BASELINE.json config.  Returns dict(ds, bound, b, count, dtype, X, Y)."""
    cfg = dict(CONFIGS[name])
    if n is not None:
        cfg["n"] = int(n)
    rng = np.random.default_rng(SEED0 + cfg["index"] if seed is None else seed)
    bound = np.asarray(cfg["bound"], dtype=np.float64)
    d = bound.shape[0]
    X = rng.uniform(bound[:, 0], bound[:, 1], size=(cfg["n"], d))
    if cfg["plant"] == "benoit":
        Y = benoit(X)
    elif cfg["plant"] == "rosenbrock4":
        Y = rosenbrock4(X)
    elif cfg["plant"] == "wo":
        Y = williams_otto(X)
    else:
        Y = sines6(X, rng)
    ds = make_dataset(X, Y, default_hypopt(d, Y.shape[1]))
    cfg.update(ds=ds, bound=bound, X=X, Y=Y, d=d, q=Y.shape[1], rng=rng)
    return cfg


def scattered_points(cfg, N, dtype=np.float32):
    """Config E candidates: points ~ U(bound), [N, d]."""
    bound = cfg["bound"]
    rng = np.random.default_rng(SEED0 + 1000 + cfg["index"])
    return rng.uniform(bound[:, 0], bound[:, 1], size=(N, bound.shape[0])).astype(dtype)
