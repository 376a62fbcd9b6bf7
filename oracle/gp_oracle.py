"""NumPy restatement of the reference hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

PARITY UNPINNED by the reference's own tests (they assert nothing); see the
package docstring for what pins this file instead.

Reference = /root/reference (dleeim/Safe-Bayesian-Optimization @ 2025-04-04).
Every function names the reference lines it follows.  The reference evaluates
one candidate at a time (``GP_inference`` under ``jit``; batched once with
``vmap`` in test/test_SafeOpt.py:337); here the same arithmetic is written for
a batch of candidates ``points[N, d]`` -- explicit ``invK``, the *expanded*
squared distance, ``(N x n) @ (n x n)`` followed by a row-dot, clipping of the
variance at zero and un-normalisation, in that order.

The set definitions (S, M, G, U, O) are the discretised form of the
reference's constrained-optimisation problems: the
reference poses them over the continuous box and solves them with SciPy
differential evolution; on a candidate set they become masks and arg-max /
arg-min reductions with ties resolved to the lowest flat index.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "data_normalization", "squared_seuclidean", "cov_mat", "calc_cov_mat", "build_invK",
    "make_inference_dataset", "cast_dataset", "mean_prior", "gp_inference", "bounds",
    "grid_axes", "grid_points", "mean_grad", "mean_grad_infnorm", "shifted_norm",
    "safeopt_sweep", "goose_sweep", "tr_sweep", "update_TR", "FLOAT32_EPS", "negative_loglikelihood",
]

FLOAT32_EPS = float(np.finfo(np.float32).eps)  # jitter of the stored inverse, models/GP_Safe.py:229


# --------------------------------------------------------------------------------------
# model state (row a1 of SURVEY.md section 8)
# --------------------------------------------------------------------------------------
def data_normalization(X, Y):
    """models/GP_Safe.py:84-96 -- population mean/std (ddof=0), no zero guard."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    X_mean, X_std = np.mean(X, axis=0), np.std(X, axis=0)
    Y_mean, Y_std = np.mean(Y, axis=0), np.std(Y, axis=0)
    return (X - X_mean) / X_std, (Y - Y_mean) / Y_std, X_mean, X_std, Y_mean, Y_std


def squared_seuclidean(X, Y, V):
    """models/GP_Safe.py:98-120 -- expanded form, V = squared length-scales.

    dist[a, b] = -2 (X V^-1/2)(Y V^-1/2)^T + sum (X V^-1/2)^2 + sum (Y V^-1/2)^2.
    """
    V_sqrt_inv = V ** -0.5                                   # :112
    Xa = X * V_sqrt_inv                                      # :115
    Ya = Y * V_sqrt_inv                                      # :116
    return -2 * np.dot(Xa, Ya.T) + np.sum(Xa ** 2, axis=1)[:, None] + np.sum(Ya ** 2, axis=1)  # :119


def cov_mat(X_norm, Y_norm, W, sf2):
    """models/GP_Safe.py:122-143 (kernel == 'RBF')."""
    if W.shape[0] != X_norm.shape[1]:
        raise ValueError("ERROR W and X_norm dimension should be same")  # :134-135
    return sf2 * np.exp(-0.5 * squared_seuclidean(X_norm, Y_norm, W))      # :140-141


def calc_cov_mat(X_norm, x_norm, ell, sf2):
    """models/GP_Safe.py:146-167 batched: x_norm[N, d] -> k[n, N]."""
    if ell.shape[0] != X_norm.shape[1]:
        raise ValueError("ERROR W and X_norm dimension should be same")  # :159-160
    return sf2 * np.exp(-0.5 * squared_seuclidean(X_norm, x_norm, ell))    # :165-166


def build_invK(X_norm, hypopt):
    """models/GP_Safe.py:226-232 -- K + (sn2 + float32 eps) I, explicit inverse, one per output."""
    n, d = X_norm.shape
    out = []
    for i in range(hypopt.shape[1]):
        ell = np.exp(2.0 * hypopt[:d, i])
        sf2 = np.exp(2.0 * hypopt[d, i])
        sn2 = np.exp(2.0 * hypopt[d + 1, i]) + FLOAT32_EPS
        K = cov_mat(X_norm, X_norm, ell, sf2) + sn2 * np.eye(n)
        out.append(np.linalg.inv(K))
    return out


def negative_loglikelihood(hyper, X_norm, y):
    """models/GP_Safe.py:169-192 -- the objective the reference's DE fit minimises for one output:
    K = sf2 exp(-1/2 D) + (sn2 + 1e-8) I (jitter 1e-8 here, float32 eps in the stored inverse, :184 vs :229), symmetrised
    (:185), L = chol(K) (:186), NLL = y^T K^-1 y + log|K| with K^-1 y by two triangular solves (:187-190) -- no 1/2, no constant.
    ``hyper`` = (log ell_0.., log sigma_f, log sigma_n), consumed as exp(2 h) (:180-182); ``y`` one column of Y_norm, [n] or [n, 1].
    Returns +inf when K is not positive definite in floating point (jnp.linalg.cholesky yields NaN there)."""
    from scipy.linalg import solve_triangular
    X_norm = np.asarray(X_norm, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
    n, d = X_norm.shape
    hyper = np.asarray(hyper, dtype=np.float64)
    W, sf2, sn2 = np.exp(2 * hyper[:d]), np.exp(2 * hyper[d]), np.exp(2 * hyper[d + 1])
    K = cov_mat(X_norm, X_norm, W, sf2) + (sn2 + 1e-8) * np.eye(n)
    K = (K + K.T) * 0.5
    try:
        L = np.linalg.cholesky(K)
    except np.linalg.LinAlgError:
        return np.inf
    logdet = 2 * np.sum(np.log(np.diag(L)))
    alpha = solve_triangular(L.T, solve_triangular(L, y, lower=True), lower=False)
    return float(np.dot(y.T, alpha)[0][0] + logdet)


def make_inference_dataset(X, Y, hypopt):
    """Fill the ``inference_datasets`` dict (models/GP_Safe.py:16-23, 236-245) for fixed hyper-parameters."""
    X_norm, Y_norm, X_mean, X_std, Y_mean, Y_std = data_normalization(X, Y)
    hypopt = np.asarray(hypopt, dtype=np.float64)
    return {
        "X_mean": X_mean, "X_std": X_std, "Y_mean": Y_mean, "Y_std": Y_std,
        "X_norm": X_norm, "Y_norm": Y_norm,
        "invKopt": build_invK(X_norm, hypopt), "hypopt": hypopt,
    }


def cast_dataset(ds, dtype):
    """Round every array of the model state to ``dtype`` (fp32 mode of the oracle)."""
    out = {}
    for key, val in ds.items():
        if key == "invKopt":
            out[key] = [np.asarray(a, dtype=dtype) for a in val]
        else:
            out[key] = np.asarray(val, dtype=dtype)
    return out


def mean_prior(ds):
    """models/GP_Safe.py:331-332 -- pessimistic constraint prior, zero for the objective."""
    mp = (-2 * ds["Y_mean"]) / ds["Y_std"]
    mp = np.array(mp, copy=True)
    mp[0] = 0.0
    return mp


# --------------------------------------------------------------------------------------
# posterior (rows a2-a5)
# --------------------------------------------------------------------------------------
def gp_inference(points, ds, dtype=np.float64, chunk=65536):
    """models/GP_Safe.py:310-352 for a batch: returns (mean[N, q], var[N, q]), un-normalised.

    Per output i: ``mean_i = mp_i + (k^T invK_i)(Y_norm[:, i] - mp_i)`` (:342),
    ``var_i = max(0, sf2 - (k^T invK_i) k)`` (:343), then ``mean*Y_std + Y_mean``,
    ``var*Y_std**2`` (:346-347).
    """
    dtype = np.dtype(dtype)
    if dtype != np.float64:
        ds = cast_dataset(ds, dtype)
    points = np.asarray(points, dtype=dtype)
    N = points.shape[0]
    n, d = ds["X_norm"].shape
    q = ds["Y_norm"].shape[1]
    mp = mean_prior(ds).astype(dtype)
    mean = np.empty((N, q), dtype=dtype)
    var = np.empty((N, q), dtype=dtype)
    for s in range(0, N, chunk):
        x = points[s:s + chunk]
        xnorm = (x - ds["X_mean"]) / ds["X_std"]                                  # :326
        for i in range(q):
            hyper = ds["hypopt"][:, i]
            ell, sf2 = np.exp(2 * hyper[:d]), np.exp(2 * hyper[d])                # :338
            k = calc_cov_mat(ds["X_norm"], xnorm, ell, sf2)                       # :341  [n, Nc]
            kinv = np.matmul(k.T, ds["invKopt"][i])                               # k^T invK  [Nc, n]
            m = mp[i] + np.matmul(kinv, ds["Y_norm"][:, i] - mp[i])               # :342
            v = np.maximum(0, sf2 - np.sum(kinv * k.T, axis=1))                   # :343
            mean[s:s + chunk, i] = m * ds["Y_std"][i] + ds["Y_mean"][i]           # :346
            var[s:s + chunk, i] = v * ds["Y_std"][i] ** 2                         # :347
    return mean, var


def bounds(mean, var, b):
    """models/SafeOpt.py:34-45 -- ucb = mean + b sqrt(var), lcb = mean - b sqrt(var)."""
    b = mean.dtype.type(b)
    s = np.sqrt(var)
    return mean - b * s, mean + b * s


# --------------------------------------------------------------------------------------
# candidate grids (test/test_SafeOpt.py:324-334: linspace + meshgrid 'xy' + ravel => axis 0 fastest)
# --------------------------------------------------------------------------------------
def grid_axes(lo, hi, count):
    """Per-axis coordinates: x_a(i) = lo_a + i * (hi_a - lo_a)/(count_a - 1), last point = hi_a exactly."""
    axes = []
    for l, h, c in zip(lo, hi, count):
        c = int(c)
        if c == 1:
            axes.append(np.array([float(l)]))
            continue
        step = (float(h) - float(l)) / (c - 1)
        ax = float(l) + np.arange(c, dtype=np.float64) * step
        ax[-1] = float(h)
        axes.append(ax)
    return axes


def grid_points(lo, hi, count, first=0, n=None):
    """Flat candidate list [n, d]; flat index g = sum_a i_a * prod_{b<a} count_b (axis 0 fastest)."""
    axes = grid_axes(lo, hi, count)
    total = int(np.prod([len(a) for a in axes]))
    if n is None:
        n = total - first
    g = np.arange(first, first + n, dtype=np.int64)
    pts = np.empty((n, len(axes)), dtype=np.float64)
    for a, ax in enumerate(axes):
        pts[:, a] = ax[g % len(ax)]
        g = g // len(ax)
    return pts


# --------------------------------------------------------------------------------------
# Lipschitz bound (row a9)
# --------------------------------------------------------------------------------------
def mean_grad(points, ds, chunk=65536):
    """Analytic d MEAN_i / d x_a -- what ``jax.grad(self.mean)`` (models/SafeOpt.py:68-71) evaluates.

    d/dx_a = Y_std[i] * sum_j alpha_ij k_j (X_norm[j,a] - xn_a) / ell_a / X_std[a],
    alpha_i = invK_i (Y_norm[:, i] - mp_i).  Returns grad[N, q, d].
    """
    points = np.asarray(points, dtype=np.float64)
    N = points.shape[0]
    n, d = ds["X_norm"].shape
    q = ds["Y_norm"].shape[1]
    mp = mean_prior(ds)
    out = np.empty((N, q, d))
    for s in range(0, N, chunk):
        xnorm = (points[s:s + chunk] - ds["X_mean"]) / ds["X_std"]
        for i in range(q):
            hyper = ds["hypopt"][:, i]
            ell, sf2 = np.exp(2 * hyper[:d]), np.exp(2 * hyper[d])
            k = calc_cov_mat(ds["X_norm"], xnorm, ell, sf2)              # [n, Nc]
            alpha = ds["invKopt"][i] @ (ds["Y_norm"][:, i] - mp[i])      # [n]
            w = k * alpha[:, None]                                       # [n, Nc]
            for a in range(d):
                diff = ds["X_norm"][:, a][:, None] - xnorm[:, a][None, :]
                out[s:s + chunk, i, a] = ds["Y_std"][i] * np.sum(w * diff, axis=0) / ell[a] / ds["X_std"][a]
    return out


def mean_grad_infnorm(points, ds):
    """models/SafeOpt.py:68-71 -- max_a |d MEAN_i / d x_a| per candidate and output: [N, q]."""
    return np.max(np.abs(mean_grad(points, ds)), axis=2)


def shifted_norm(xg, xh):
    """models/SafeOpt.py:87 -- ||x - x' + 1e-8||_2 with the 1e-8 added to every component."""
    diff = xg - xh + 1e-8
    return np.sqrt(np.sum(diff * diff, axis=-1))


def _lipschitz(points, ds, quirk_L_index):
    """L_c used for constraint c = 1..q-1.  models/SafeOpt.py:79-83 gives L_i = max ||grad MEAN_i||_inf;
    with ``quirk_L_index`` the reference's loop-leaked index is reproduced: every constraint uses
    L_{q-1} (models/SafeOpt.py:110, models/GoOSE.py:100)."""
    gn = mean_grad_infnorm(points, ds)
    L = np.max(gn, axis=0)                       # [q]
    q = L.shape[0]
    Lc = np.array([L[q - 1] if quirk_L_index else L[c] for c in range(q)])
    return L, Lc


# --------------------------------------------------------------------------------------
# SafeOpt sweep (rows a6-a10, a12)
# --------------------------------------------------------------------------------------
def safeopt_sweep(points, ds, b, quirk_L_index=True, mean_var=None):
    """Discretised SafeOpt iteration on a candidate list.

    S  = {g : lcb_i(g) >= 0 for all i >= 1}                         models/SafeOpt.py:57-59
    u* = min_S ucb_0                                                 :47-51, 61
    M  = {g in S : lcb_0(g) <= u*}                                   :62
    U  = {h : lcb_i(h) <= 0 for all i >= 1}  (max_i lcb_i <= 0)      :73-77, 109
    G_c= {g in S : exists h in U, ucb_c(g) - L_c ||x_g - x_h + 1e-8|| >= 0}   :85-88, 111
    minimiser = argmax_M var_0, expander_c = argmax_{G_c} var_0; report sqrt(var_0)   :55, 65-66, 92, 117-124
    """
    points = np.asarray(points, dtype=np.float64)
    mean, var = mean_var if mean_var is not None else gp_inference(points, ds)
    lcb, ucb = bounds(mean, var, b)
    q = mean.shape[1]
    S = np.all(lcb[:, 1:] >= 0, axis=1)
    U = np.all(lcb[:, 1:] <= 0, axis=1)
    L, Lc = _lipschitz(points, ds, quirk_L_index)
    res = {"mean": mean, "var": var, "lcb": lcb, "ucb": ucb, "S": S, "U": U, "L": L, "L_used": Lc,
           "empty_safe_set": not bool(S.any())}
    if not S.any():
        return res
    u_star = np.min(ucb[S, 0])
    M = S & (lcb[:, 0] <= u_star)
    var0 = var[:, 0]
    mi = int(np.argmax(np.where(M, var0, -np.inf)))
    res.update({"u_star": u_star, "M": M, "minimizer_index": mi, "minimizer_std": float(np.sqrt(var0[mi]))})
    G = np.zeros((q - 1, points.shape[0]), dtype=bool)
    exp_idx = np.full(q - 1, -1, dtype=np.int64)
    exp_std = np.zeros(q - 1)
    for c in range(1, q):
        # ucb_c - L ||.|| >= 0 is evaluated exactly as written (no division by L)
        G[c - 1] = _exists_within_lipschitz(points, S, U, ucb[:, c], Lc[c])
        if G[c - 1].any():
            e = int(np.argmax(np.where(G[c - 1], var0, -np.inf)))
            exp_idx[c - 1], exp_std[c - 1] = e, float(np.sqrt(var0[e]))
    res.update({"G": G, "expander_index": exp_idx, "expander_std": exp_std})
    # Expander() keeps the most uncertain expander, first on ties (models/SafeOpt.py:119-122)
    found = exp_idx >= 0
    if found.any():
        best = int(np.argmax(np.where(found, exp_std, -np.inf)))
        res.update({"expander_best": best + 1, "expander_best_index": int(exp_idx[best]),
                    "expander_best_std": float(exp_std[best])})
    else:
        res.update({"expander_best": 0, "expander_best_index": -1, "expander_best_std": 0.0})
    # loop decision rule, test/test_SafeOpt.py:153-158: minimiser iff std_min > std_exp
    res["choose_minimizer"] = bool(res["minimizer_std"] > res["expander_best_std"])
    return res


def _exists_within_lipschitz(points, src_mask, dst_mask, ucb_c, L, chunk=2048):
    """g in src: exists h in dst with ucb_c[g] - L * ||x_g - x_h + 1e-8|| >= 0 (models/SafeOpt.py:85-88)."""
    out = np.zeros(points.shape[0], dtype=bool)
    gi = np.nonzero(src_mask)[0]
    hi = np.nonzero(dst_mask)[0]
    if gi.size == 0 or hi.size == 0:
        return out
    xh = points[hi]
    for s in range(0, gi.size, chunk):
        idx = gi[s:s + chunk]
        dist = shifted_norm(points[idx][:, None, :], xh[None, :, :])
        out[idx] = np.any(ucb_c[idx][:, None] - L * dist >= 0, axis=1)
    return out


# --------------------------------------------------------------------------------------
# GoOSE sweep (row a11, a12)
# --------------------------------------------------------------------------------------
def goose_sweep(points, ds, b, quirk_L_index=True, mean_var=None):
    """Discretised GoOSE iteration (models/GoOSE.py:63-119, test/test_GoOSE.py:151-162).

    x_safe = argmin_S lcb_0                                                     GoOSE.py:63-67
    O_c    = {h in U : exists g in S, ucb_c(g) - L_c ||x_g - x_h + 1e-8|| >= 0}  :93-101
    target = argmin over c of min_{O_c} lcb_0 (first c on ties)                  :82, 106-112
    x_obs  = argmin_S ||x_g - target||_2                                         :116-119
    """
    points = np.asarray(points, dtype=np.float64)
    mean, var = mean_var if mean_var is not None else gp_inference(points, ds)
    lcb, ucb = bounds(mean, var, b)
    q = mean.shape[1]
    S = np.all(lcb[:, 1:] >= 0, axis=1)
    U = np.all(lcb[:, 1:] <= 0, axis=1)
    L, Lc = _lipschitz(points, ds, quirk_L_index)
    res = {"mean": mean, "var": var, "lcb": lcb, "ucb": ucb, "S": S, "U": U, "L": L, "L_used": Lc,
           "empty_safe_set": not bool(S.any())}
    if not S.any():
        return res
    si = int(np.argmin(np.where(S, lcb[:, 0], np.inf)))
    res.update({"safe_min_index": si, "safe_min_lcb": float(lcb[si, 0])})
    O = np.zeros((q - 1, points.shape[0]), dtype=bool)
    t_idx = np.full(q - 1, -1, dtype=np.int64)
    t_lcb = np.full(q - 1, np.inf)
    gi = np.nonzero(S)[0]
    for c in range(1, q):
        hi = np.nonzero(U)[0]
        cov = np.zeros(hi.size, dtype=bool)
        for s in range(0, hi.size, 2048):
            hh = hi[s:s + 2048]
            dist = shifted_norm(points[gi][None, :, :], points[hh][:, None, :])      # [c, |S|]  (x_g - x_h + 1e-8)
            cov[s:s + 2048] = np.any(ucb[gi, c][None, :] - Lc[c] * dist >= 0, axis=1)
        O[c - 1, hi[cov]] = True
        if O[c - 1].any():
            t = int(np.argmin(np.where(O[c - 1], lcb[:, 0], np.inf)))
            t_idx[c - 1], t_lcb[c - 1] = t, float(lcb[t, 0])
    res.update({"O": O, "target_index_c": t_idx, "target_lcb_c": t_lcb})
    if (t_idx >= 0).any():
        best = int(np.argmin(t_lcb))
        ti = int(t_idx[best])
        res.update({"target_best": best + 1, "target_index": ti, "target_lcb": float(t_lcb[best])})
        choose_safe_min = bool(res["safe_min_lcb"] <= res["target_lcb"])          # test/test_GoOSE.py:158
        d2 = np.sum((points - points[ti]) ** 2, axis=1)                            # cdist (Euclidean), GoOSE.py:117
        oi = int(np.argmin(np.where(S, np.sqrt(d2), np.inf)))
        res.update({"choose_safe_min": choose_safe_min, "explore_index": oi})
    else:
        res.update({"target_best": 0, "target_index": -1, "target_lcb": np.inf, "choose_safe_min": True,
                    "explore_index": -1})
    return res


# --------------------------------------------------------------------------------------
# Trust-region acquisition: models/GP_TR.py:43-51, 56-91
# --------------------------------------------------------------------------------------
def tr_sweep(points, ds, b, x_0, r, mean_var=None):
    """argmin of lcb_0 over {g : lcb_i(g) >= 0 for all i >= 1 and ||x_g - x_0||_2 <= r} (models/GP_TR.py:43-51;
    the ball constraint there is NonlinearConstraint(norm(x - x_0), 0, r), no 1e-8 shift)."""
    points = np.asarray(points, dtype=np.float64)
    mean, var = mean_var if mean_var is not None else gp_inference(points, ds)
    lcb, ucb = bounds(mean, var, b)
    S = np.all(lcb[:, 1:] >= 0, axis=1)
    diff = points - np.asarray(x_0, dtype=np.float64)
    dist = np.sqrt(np.sum(diff * diff, axis=1))
    T = S & (dist <= r)
    res = {"S": S, "T": T, "lcb": lcb, "empty": not bool(T.any())}
    if T.any():
        i = int(np.argmin(np.where(T, lcb[:, 0], np.inf)))
        res.update({"index": i, "lcb_min": float(lcb[i, 0])})
    return res


def update_TR(TR_parameters, x_initial, x_new, radius, plant_old, plant_new, gp_obj_old, gp_obj_new):
    """Ratio test of models/GP_TR.py:56-91: returns (centre, radius)."""
    r = radius
    for i in range(1, len(plant_new)):
        if plant_new[i] < 0.:
            return x_initial, r * TR_parameters["radius_red"]                       # :73-75
    rho = (plant_new[0] - plant_old[0]) / (gp_obj_new - gp_obj_old)                  # :79
    if plant_old[0] < plant_new[0]:
        return x_initial, r * TR_parameters["radius_red"]                           # :81-82
    if rho < TR_parameters["rho_lb"]:
        return x_initial, r * TR_parameters["radius_red"]                           # :86-87
    if rho < TR_parameters["rho_ub"]:
        return x_new, r                                                               # :89-90
    return x_new, min(r * TR_parameters["radius_inc"], TR_parameters["radius_max"])  # :92-93
