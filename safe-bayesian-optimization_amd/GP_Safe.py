"""Host-side mirror of the reference's GP model class (models/GP_Safe.py), NumPy + libsafebo.so.

Same method names, argument meaning and error behaviour as the reference ``GP`` so its drivers keep working
(``from models.GP_Safe import GP`` -> ``from safebo_amd.GP_Safe import GP``); what changes is where the
arithmetic runs: the posterior (``GP_inference``) is evaluated by the HIP kernels behind the C ABI, batched
when the caller passes many points.  Model fitting (normalisation, hyper-parameter search) is host NumPy/SciPy by
default; with ``fit_on_device = True`` the differential-evolution search evaluates its whole population per
generation with the batched device objective ``sbo_nll_batch``.

Differences that are deliberate and documented:
  * no JAX: ``key`` arguments are ``numpy.random.Generator`` objects (or seeds); the reference's threefry
    stream cannot be reproduced without JAX, parity is defined on given X, Y;
  * ``sobol_seq`` multistart vectors are generated with ``scipy.stats.qmc.Sobol`` (same role, unused by the
    reference's DE path as well: models/GP_Safe.py:209 builds them, :224 never reads them);
  * ``fixed_hyper=`` lets a caller skip the DE fit and install given hyper-parameters (benchmarks, tests).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import differential_evolution

from .engine import SweepEngine

FLOAT32_EPS = float(np.finfo(np.float32).eps)


class GP:
    def __init__(self, plant_system, device: int = 0, dtype: str = "f64", seed: int = 42) -> None:
        self.plant_system = plant_system
        self.n_fun = len(plant_system)
        self.key = np.random.default_rng(seed)              # reference: jax.random.PRNGKey(42), GP_Safe.py:15
        self.inference_datasets = {"X_mean": [], "X_std": [], "Y_mean": [], "Y_std": [],
                                   "X_norm": [], "Y_norm": [], "invKopt": [], "hypopt": []}   # GP_Safe.py:16-23
        self.dtype = dtype
        self.device = device
        self._engine = None          # created lazily: constructing a GP must not require a GPU
        self._model_version = 0      # bumped by update_inference_dataset
        self._uploaded_version = -1
        self.fixed_hyper = None
        self.de_options = {}         # forwarded to scipy DE (e.g. {"seed": 0, "maxiter": 50})
        self.fit_on_device = False   # True: DE evaluates whole populations with the batched device NLL (sbo_nll_batch);
                                     # "de": the whole DE search runs on the device (sbo_fit_de), polish on the host
        self.var_out = True

    # ---- engine plumbing -----------------------------------------------------------------------------------
    @property
    def engine(self) -> SweepEngine:
        if self._engine is None:
            self._engine = SweepEngine(self.device)
        return self._engine

    def _sync_model(self):
        """Upload ``inference_datasets`` when it changed since the last upload."""
        if self._uploaded_version != self._model_version:
            self.engine.set_model(self.inference_datasets, dtype=self.dtype, kernel=self.kernel)
            self._uploaded_version = self._model_version
            self._cand_token = None

    # ---- data sampling (models/GP_Safe.py:30-78) -------------------------------------------------------------
    def Ball_sampling(self, x_dim, n_sample, r_i, key=None):
        """Uniform samples in the ball of radius r_i around the origin: normal direction times U^(1/d) radius."""
        rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key) if key is not None else self.key
        xi = rng.standard_normal((n_sample, x_dim))
        unit = xi / np.linalg.norm(xi, axis=1, keepdims=True)
        radius = rng.uniform(size=(n_sample, 1)) ** (1.0 / x_dim)
        return r_i * radius * unit

    def Data_sampling(self, n_sample, x_0, r, noise=0.):
        x_0 = np.asarray(x_0, dtype=np.float64)
        X = self.Ball_sampling(x_0.shape[0], n_sample, r, self.key) + x_0
        Y = np.zeros((n_sample, self.n_fun))
        for i in range(n_sample):
            for j in range(self.n_fun):
                Y[i, j] = self.plant_system[j](X[i], noise)
        return X, Y

    # ---- GP operations (models/GP_Safe.py:84-245) ------------------------------------------------------------
    def data_normalization(self):
        self.X_mean, self.X_std = np.mean(self.X, axis=0), np.std(self.X, axis=0)     # population std, :92
        self.Y_mean, self.Y_std = np.mean(self.Y, axis=0), np.std(self.Y, axis=0)
        return (self.X - self.X_mean) / self.X_std, (self.Y - self.Y_mean) / self.Y_std

    def squared_seuclidean_jax(self, X, Y, V):
        """Expanded standardised squared distance; V holds the *squared* length-scales (models/GP_Safe.py:98-120)."""
        v = V ** -0.5
        Xa, Ya = X * v, Y * v
        return -2 * np.dot(Xa, Ya.T) + np.sum(Xa ** 2, axis=1)[:, None] + np.sum(Ya ** 2, axis=1)

    def Cov_mat(self, kernel, X_norm, Y_norm, W, sf2):
        if W.shape[0] != X_norm.shape[1]:
            raise ValueError("ERROR W and X_norm dimension should be same")
        elif kernel != "RBF":
            raise ValueError("ERROR no kernel with name ", kernel)
        return sf2 * np.exp(-0.5 * self.squared_seuclidean_jax(X_norm, Y_norm, W))

    def calc_Cov_mat(self, kernel, X_norm, x_norm, ell, sf2):
        x_norm = np.asarray(x_norm).reshape(1, self.nx_dim)
        if ell.shape[0] != X_norm.shape[1]:
            raise ValueError("ERROR W and X_norm dimension should be same")
        elif kernel != "RBF":
            raise ValueError("ERROR no kernel with name ", kernel)
        return sf2 * np.exp(-0.5 * self.squared_seuclidean_jax(X_norm, x_norm, ell))

    def negative_loglikelihood(self, hyper, X, Y):
        """y^T K^-1 y + log|K| with jitter 1e-8 -- no 1/2 and no constant (models/GP_Safe.py:169-192)."""
        d = self.nx_dim
        W, sf2, sn2 = np.exp(2 * hyper[:d]), np.exp(2 * hyper[d]), np.exp(2 * hyper[d + 1])
        K = self.Cov_mat(self.kernel, X, X, W, sf2) + (sn2 + 1e-8) * np.eye(self.n_point)
        K = (K + K.T) * 0.5
        try:
            L = np.linalg.cholesky(K)
        except np.linalg.LinAlgError:
            return np.inf
        alpha = np.linalg.solve(L.T, np.linalg.solve(L, Y))
        return float(np.dot(Y.T, alpha)[0][0] + 2 * np.sum(np.log(np.diag(L))))

    def determine_hyperparameters(self, X_norm, Y_norm):
        """One GP per output; bounds [-1.5, 1.5]^(d+1) x [-5, -2] searched by SciPy DE (models/GP_Safe.py:194-234),
        then invK = inv(K + (sn2 + float32 eps) I)."""
        d = self.nx_dim
        bounds = np.array([[-1.5, 1.5]] * (d + 1) + [[-5.0, -2.0]])
        hypopt = np.zeros((d + 2, self.ny_dim))
        invKopt = []
        for i in range(self.ny_dim):
            if self.fixed_hyper is not None:
                hypopt[:, i] = np.asarray(self.fixed_hyper, dtype=np.float64)[:, i]
            elif self.fit_on_device == "de":
                # the whole DE search on the device (sbo_fit_de): SciPy's defaults as the reference uses them (popsize 15 per
                # dimension, Latin-hypercube start, best1bin, mutation (0.5, 1), recombination 0.7, tol 0.01), deferred
                # updating, then SciPy's own polish step (L-BFGS-B from the best member) on the host objective
                from scipy.optimize import minimize
                from scipy.stats import qmc
                opts = dict(self.de_options)
                seed = int(opts.get("seed", 0) or 0)
                P = int(opts.get("popsize", 15)) * (d + 2)
                pop = qmc.scale(qmc.LatinHypercube(d + 2, seed=seed).random(P), bounds[:, 0], bounds[:, 1])
                y_i = np.ascontiguousarray(Y_norm[:, i])
                best, energy, _ = self.engine.fit_de(X_norm, y_i, bounds, pop, seed=seed + i, maxiter=int(opts.get("maxiter", 1000)),
                                                     tol=float(opts.get("tol", 0.01)), atol=float(opts.get("atol", 0.0)))
                if opts.get("polish", True):
                    res = minimize(self.negative_loglikelihood, best, args=(X_norm, Y_norm[:, i:i + 1]), method="L-BFGS-B",
                                   bounds=bounds)
                    if res.fun < energy:
                        best = res.x
                hypopt[:, i] = best
            elif self.fit_on_device:
                # same objective and bounds; SciPy hands over the whole trial population (vectorized, deferred
                # updating) and the device returns one NLL per member
                y_i = np.ascontiguousarray(Y_norm[:, i])
                res = differential_evolution(lambda H: self.engine.nll_batch(X_norm, y_i, H.T), bounds=bounds,
                                             vectorized=True, updating="deferred", **self.de_options)
                hypopt[:, i] = res.x
            else:
                res = differential_evolution(self.negative_loglikelihood, args=(X_norm, Y_norm[:, i:i + 1]), bounds=bounds,
                                             **self.de_options)
                hypopt[:, i] = res.x
            ell = np.exp(2.0 * hypopt[:d, i])
            sf2 = np.exp(2.0 * hypopt[d, i])
            sn2 = np.exp(2.0 * hypopt[d + 1, i]) + FLOAT32_EPS
            K = self.Cov_mat(self.kernel, X_norm, X_norm, ell, sf2) + sn2 * np.eye(self.n_point)
            invKopt.append(np.linalg.inv(K))
        return hypopt, invKopt

    def update_inference_dataset(self):
        ds = self.inference_datasets
        ds["X_mean"], ds["X_std"], ds["Y_mean"], ds["Y_std"] = self.X_mean, self.X_std, self.Y_mean, self.Y_std
        ds["X_norm"], ds["Y_norm"], ds["invKopt"], ds["hypopt"] = self.X_norm, self.Y_norm, self.invKopt, self.hypopt
        self._model_version += 1

    # ---- initialisation / update (models/GP_Safe.py:251-304) ---------------------------------------------------
    def GP_initialization(self, X, Y, kernel, multi_hyper, var_out=True):
        self.X, self.Y, self.kernel = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64), kernel
        self.n_point, self.nx_dim = self.X.shape
        self.ny_dim = self.Y.shape[1]
        self.multi_hyper = multi_hyper
        self.var_out = var_out
        self.X_norm, self.Y_norm = self.data_normalization()
        self.hypopt, self.invKopt = self.determine_hyperparameters(self.X_norm, self.Y_norm)
        self.update_inference_dataset()

    def add_sample(self, x_new, y_new, incremental: bool = False):
        """Reference behaviour (models/GP_Safe.py:283-304): re-normalise, refit, rebuild.  ``incremental=True`` is the
        opt-in fast path of SURVEY.md 8(f) rank 2: normalisation constants and hyper-parameters stay frozen and the device
        model gains one row in O(n^2) (``sbo_model_append``); the Python-side ``inference_datasets`` follows with the
        bordered inverse, also O(n^2)."""
        self.X = np.vstack([self.X, np.asarray(x_new, dtype=np.float64)])
        self.Y = np.vstack([self.Y, np.asarray(y_new, dtype=np.float64)])
        self.n_point = self.X.shape[0]
        if incremental:
            self._sync_model()
            xn = (np.asarray(x_new, dtype=np.float64).reshape(-1) - self.X_mean) / self.X_std
            yn = (np.asarray(y_new, dtype=np.float64).reshape(-1) - self.Y_mean) / self.Y_std
            self.engine.append_sample(xn, yn)
            # the host copy of the model follows: bordered inverse of K + sn2 I under the frozen hyper-parameters, so that
            # ``inference_datasets`` stays a complete, uploadable state (X_norm [n+1, d] next to invKopt [n+1, n+1])
            d = self.nx_dim
            for i in range(self.ny_dim):
                ell, sf2 = np.exp(2.0 * self.hypopt[:d, i]), np.exp(2.0 * self.hypopt[d, i])
                sn2 = np.exp(2.0 * self.hypopt[d + 1, i]) + FLOAT32_EPS
                k = self.Cov_mat(self.kernel, self.X_norm, xn[None, :], ell, sf2)[:, 0]
                u = self.invKopt[i] @ k
                s = (sf2 + sn2) - float(k @ u)
                self.invKopt[i] = np.block([[self.invKopt[i] + np.outer(u, u) / s, -u[:, None] / s],
                                            [-u[None, :] / s, np.array([[1.0 / s]])]])
            self.X_norm = np.vstack([self.X_norm, xn])
            self.Y_norm = np.vstack([self.Y_norm, yn])
            self.update_inference_dataset()              # bumps _model_version: the cached sweeps of SafeOpt / GoOSE.BO are stale
            self._uploaded_version = self._model_version  # ... but the device model is already current: no re-upload
            self._cand_token = None
            return
        self.X_norm, self.Y_norm = self.data_normalization()
        self.hypopt, self.invKopt = self.determine_hyperparameters(self.X_norm, self.Y_norm)
        self.update_inference_dataset()

    # ---- inference (models/GP_Safe.py:310-352) -----------------------------------------------------------------
    def GP_inference(self, x, inference_dataset=None):
        """Posterior of every output at ``x``.  ``x`` of shape [d] returns (mean[q], var[q]) like the reference
        (or the scalar objective mean when ``var_out`` is False, models/GP_Safe.py:349-352); ``x`` of shape [N, d]
        is the batched form the reference reaches with ``vmap`` and returns (mean[N, q], var[N, q])."""
        if inference_dataset is not None and inference_dataset is not self.inference_datasets:
            eng = self.engine
            eng.set_model(inference_dataset, dtype=self.dtype, kernel=getattr(self, "kernel", "RBF"))
            self._uploaded_version = -1
        else:
            self._sync_model()
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        pts = x.reshape(1, -1) if single else x
        self.engine.set_points(pts)
        self._cand_token = None
        mean, var = self.engine.posterior()
        if single:
            if self.var_out:
                return mean[0], var[0]
            return mean[0].flatten()[0]
        return mean, var
