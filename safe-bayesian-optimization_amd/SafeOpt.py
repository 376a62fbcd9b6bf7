"""Host-side mirror of the reference's SafeOpt class (models/SafeOpt.py) on the MI355X sweep engine.

The reference poses S_t / M_t / G_t as continuous constrained problems and solves each with SciPy differential
evolution, one posterior evaluation per Python callback (models/SafeOpt.py:47-124).  Here the same definitions are
evaluated on a candidate grid over ``bound`` by one device sweep; the class keeps the
reference's method names and return conventions so ``test/test_SafeOpt.py``-style drivers run unchanged:

    GP_m = SafeOpt.BO(plant_system, bound, b)                 # + grid=(n0, n1, ..) candidates per axis
    minimizer, std_minimizer = GP_m.Minimizer()
    expander, std_expander = GP_m.Expander()

``mean/ucb/lcb(x, i)`` accept one point [d] (returns a scalar, as the reference) or many [N, d] (the
``vmap(GP_m.lcb, in_axes=(0, None))(points, 1)`` call of test/test_SafeOpt.py:337 becomes ``GP_m.lcb(points, 1)``).
"""
from __future__ import annotations

import numpy as np

from .GP_Safe import GP


class BO(GP):
    def __init__(self, plant_system, bound, b, grid=None, device: int = 0, dtype: str = "f64",
                 reference_quirk_L_index: bool = True, seed: int = 42):
        GP.__init__(self, plant_system, device=device, dtype=dtype, seed=seed)
        self.bound = np.asarray(bound, dtype=np.float64)
        self.b = b
        d = self.bound.shape[0]
        # candidate grid over the box; default = the 400 points per axis of create_data_for_plot (test_SafeOpt.py:325)
        self.grid = tuple(int(g) for g in (grid if grid is not None else (400,) * d))
        self.reference_quirk_L_index = reference_quirk_L_index
        self._sweep_cache = None      # (model_version, result dict)
        self._cand_token = None

    # ---- plant ------------------------------------------------------------------------------------------------
    def calculate_plant_outputs(self, x, noise=0):
        return np.array([plant(x, noise) for plant in self.plant_system])

    # ---- bounds (models/SafeOpt.py:29-45) ------------------------------------------------------------------------
    def _bound_value(self, x, i, kind):
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        self._sync_model()
        self.engine.set_points(x.reshape(1, -1) if single else x)
        self._cand_token = None
        out = self.engine.bounds(self.b, i, kind)
        return out[0] if single else out

    def mean(self, x, i):
        return self._bound_value(x, i, "mean")

    def ucb(self, x, i):
        return self._bound_value(x, i, "ucb")

    def lcb(self, x, i):
        return self._bound_value(x, i, "lcb")

    def lcb_constraint_min(self, x):
        """max over the constraints of lcb_i(x) -- '<= 0' therefore means every constraint LCB is <= 0
        (models/SafeOpt.py:73-77)."""
        return max(self.lcb(x, i) for i in range(1, self.n_fun))

    # ---- the sweep ----------------------------------------------------------------------------------------------
    def _grid_resident(self):
        self._sync_model()
        token = (self._model_version, self.grid)
        if self._cand_token != token:
            self.engine.set_grid(self.bound[:, 0], self.bound[:, 1], self.grid)
            self._cand_token = token

    def sweep(self, want_masks: bool = False) -> dict:
        """One SafeOpt iteration on the candidate grid (cached until the model changes)."""
        if self._sweep_cache is not None and self._sweep_cache[0] == (self._model_version, self.grid) and not want_masks:
            return self._sweep_cache[1]
        self._grid_resident()
        res = self.engine.sweep_safeopt(self.b, quirk_L_index=self.reference_quirk_L_index, want_masks=want_masks)
        self._sweep_cache = ((self._model_version, self.grid), res)
        return res

    def masks(self) -> dict:
        """S / U / M / G_c masks of the current model, reshaped to the grid (axis 0 fastest -> last array axis)."""
        self.sweep(want_masks=True)
        shape = self.grid[::-1]
        out = {k: self.engine.mask(k).reshape(shape) for k in ("S", "U", "M")}
        for c in range(1, self.n_fun):
            out[f"G{c}"] = self.engine.mask("G", c).reshape(shape)
        return out

    # ---- acquisition (models/SafeOpt.py:47-124) --------------------------------------------------------------------
    def minimize_obj_ucb(self, safe_set_cons=None):
        """min over S_t of ucb_0 -> (x, value); ``safe_set_cons`` is accepted for signature parity and ignored:
        the safe-set constraints are the sweep's S mask."""
        res = self.sweep(want_masks=True)
        S = self.engine.mask("S")
        ucb0 = self.engine.bounds(self.b, 0, "ucb")
        g = int(np.argmin(np.where(S, ucb0, np.inf)))
        return self._grid_point(g), res["u_star"]

    def Minimizer(self):
        res = self.sweep()
        return res["minimizer_x"], res["minimizer_std"]

    def infnorm_mean_grad(self, x, i):
        """max_a |d MEAN_i / d x_a| at one point (analytic form of jax.grad(self.mean), models/SafeOpt.py:68-71)."""
        ds = self.inference_datasets
        d = self.nx_dim
        x = np.asarray(x, dtype=np.float64)
        hyper = ds["hypopt"][:, i]
        ell, sf2 = np.exp(2 * hyper[:d]), np.exp(2 * hyper[d])
        mp = 0.0 if i == 0 else -2 * ds["Y_mean"][i] / ds["Y_std"][i]
        xn = (x - ds["X_mean"]) / ds["X_std"]
        k = self.calc_Cov_mat(self.kernel, ds["X_norm"], xn, ell, sf2)[:, 0]
        w = k * (ds["invKopt"][i] @ (ds["Y_norm"][:, i] - mp))
        grad = ds["Y_std"][i] * ((ds["X_norm"] - xn).T @ w) / ell / ds["X_std"]
        return float(np.max(np.abs(grad)))

    def maximize_infnorm_mean_grad(self, i):
        """L_i = max over the candidate grid of ||grad MEAN_i||_inf (the reference maximises over the box with DE)."""
        return float(self.sweep()["L"][i])

    def Lipschitz_continuity_constraint(self, x, i, max_infnorm_mean_grad):
        """x = [x, x'] stacked: ucb_i(x) - L ||x - x' + 1e-8|| (models/SafeOpt.py:85-88)."""
        x = np.asarray(x, dtype=np.float64)
        d = self.nx_dim
        return self.ucb(x[:d], i) - max_infnorm_mean_grad * np.linalg.norm(x[:d] - x[d:] + 1e-8)

    def Expander(self):
        res = self.sweep()
        return res["expander_x"], res["expander_std"]

    # ---- helpers ----------------------------------------------------------------------------------------------------
    def _grid_point(self, g: int):
        x = np.empty(len(self.grid))
        for a, cnt in enumerate(self.grid):
            i = g % cnt
            g //= cnt
            lo, hi = self.bound[a]
            x[a] = hi if (i == cnt - 1 and cnt > 1) else lo + i * ((hi - lo) / (cnt - 1) if cnt > 1 else 0.0)
        return x
