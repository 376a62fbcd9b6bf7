"""CPU restatement of the reference's William-Otto reactor plant -- TEST INFRASTRUCTURE ONLY,
like the rest of ``oracle/``: it is the checker for ``sbo_plant_wo`` and is never imported by the product package.

Reference: problems/WilliamOttoReactor_Problem.py
  * :19-45  ``odecallback``: six steady-state mass balances of the CSTR (states xa, xb, xc, xp, xe, xg),
            inputs u = (Fb, Tr), Fa = 1.8275, Vr = 2105.2, Arrhenius rates k_i exp(-eta_i / (Tr + 273));
  * :47-64  ``get_objective``: root of the balances from x0 = 0.1 (SciPy ``fsolve``), then
            -(1043.38 xp Fr + 20.92 xe Fr - 79.23 Fa - 118.34 Fb), Fr = Fa + Fb;
  * :67-93  ``get_constraint1 / 2``: 0.12 - xa and 0.08 - xg at the same root.

PINNED: ``tests/golden/wo_contour_reference.npz`` is the reference's own 100 x 100 table of the three outputs over
[4, 7] x [70, 100] (its data/data_contour_WilliamOttoReactor.npz, written by utils/utils_WilliamOttoReactor.py:24-55);
``tests/test_oracle.py`` checks this restatement against every entry.  The root is found here with a damped Newton
iteration on the analytic Jacobian instead of MINPACK's hybrid method: same equations, same start, same root.
"""
import numpy as np

FA = 1.8275
VR = 2105.2
K0 = (1.6599e6, 7.2177e8, 2.6745e12)
ETA = (6666.7, 8333.3, 11111.0)


def wo_residual_jacobian(w, Fb, Tr):
    """Balances f [.., 6] and Jacobian J [.., 6, 6] at states w [.., 6]  (WilliamOttoReactor_Problem.py:19-45)."""
    xa, xb, xc, xp, xe, xg = [w[..., i] for i in range(6)]
    Fr = FA + Fb
    k1 = K0[0] * np.exp(-ETA[0] / (Tr + 273))
    k2 = K0[1] * np.exp(-ETA[1] / (Tr + 273))
    k3 = K0[2] * np.exp(-ETA[2] / (Tr + 273))
    f = np.stack([
        (FA - Fr * xa - VR * xa * xb * k1) / VR,
        (Fb - Fr * xb - VR * xa * xb * k1 - VR * xb * xc * k2) / VR,
        -Fr * xc / VR + 2 * xa * xb * k1 - 2 * xb * xc * k2 - xc * xp * k3,
        -Fr * xp / VR + xb * xc * k2 - 0.5 * xp * xc * k3,
        -Fr * xe / VR + 2 * xb * xc * k2,
        -Fr * xg / VR + 1.5 * xp * xc * k3], axis=-1)
    z = np.zeros_like(xa)
    a = -Fr / VR
    J = np.stack([
        np.stack([a - xb * k1, -xa * k1, z, z, z, z], axis=-1),
        np.stack([-xb * k1, a - xa * k1 - xc * k2, -xb * k2, z, z, z], axis=-1),
        np.stack([2 * xb * k1, 2 * xa * k1 - 2 * xc * k2, a - 2 * xb * k2 - xp * k3, -xc * k3, z, z], axis=-1),
        np.stack([z, xc * k2, xb * k2 - 0.5 * xp * k3, a - 0.5 * xc * k3, z, z], axis=-1),
        np.stack([z, 2 * xc * k2, 2 * xb * k2, z, a + z, z], axis=-1),
        np.stack([z, z, 1.5 * xp * k3, 1.5 * xc * k3, z, a + z], axis=-1)], axis=-2)
    return f, J


def wo_steady_state(U, iters=60):
    """Root of the balances for every input row U [N, 2] = (Fb, Tr), from x0 = 0.1 (fsolve's start, :49, :69, :83)."""
    U = np.asarray(U, dtype=np.float64)
    Fb, Tr = U[:, 0], U[:, 1]
    w = np.full((U.shape[0], 6), 0.1)
    for _ in range(iters):
        f, J = wo_residual_jacobian(w, Fb, Tr)
        step = np.linalg.solve(J, -f[..., None])[..., 0]
        # damping: keep the states non-negative (mass fractions) -- halves the step where it would leave the orthant
        lam = np.ones(U.shape[0])
        for _ in range(30):
            bad = np.any(w + lam[:, None] * step < 0, axis=1)
            if not bad.any():
                break
            lam = np.where(bad, lam * 0.5, lam)
        w = w + lam[:, None] * step
        if np.max(np.abs(lam[:, None] * step)) < 1e-15:
            break
    return w


def wo_outputs(U):
    """[N, 3]: objective (:58-61), constraint 1 (:75), constraint 2 (:89) -- all noise-free."""
    U = np.asarray(U, dtype=np.float64)
    w = wo_steady_state(U)
    Fb = U[:, 0]
    Fr = FA + Fb
    fx = 1043.38 * w[:, 3] * Fr + 20.92 * w[:, 4] * Fr - 79.23 * FA - 118.34 * Fb
    return np.stack([-fx, 0.12 - w[:, 0], 0.08 - w[:, 5]], axis=1)
