"""Host-side mirror of the reference's trust-region class (models/GP_TR.py) on the MI355X sweep engine
(the same posterior + safe-set sweep with one extra ball mask).

    x_new, obj = GP_m.minimize_obj_lcb(r, x_0)       # argmin lcb_0 over S_t and ||x - x_0|| <= r   models/GP_TR.py:43-51
    x_0, r = GP_m.update_TR(x_0, x_new, r, plant_old, plant_new)                                   # :56-91
"""
from __future__ import annotations

import numpy as np

from .SafeOpt import BO as _SafeOptBO


class BO(_SafeOptBO):
    def __init__(self, plant_system, bound, b, TR_parameters, grid=None, device: int = 0, dtype: str = "f64", seed: int = 42):
        _SafeOptBO.__init__(self, plant_system, bound, b, grid=grid, device=device, dtype=dtype, seed=seed)
        self.TR_parameters = TR_parameters

    def minimize_obj_lcb(self, r, x_0):
        self._grid_resident()
        res = self.engine.sweep_tr(self.b, x_0, r)
        self._sweep_cache = None
        if res["index"] < 0:            # nothing safe inside the ball on this grid (the reference's DE would return an
            return np.asarray(x_0, dtype=np.float64).copy(), np.inf   # infeasible point): stay at the centre
        return res["x"], res["lcb"]

    def TR_constraint(self, x, x_0, r):
        return r - np.linalg.norm(np.asarray(x) - np.asarray(x_0) + 1e-8)       # models/GP_TR.py:53-54

    def update_TR(self, x_initial, x_new, radius, plant_oldoutput, plant_newoutput):
        """Ratio test of models/GP_TR.py:56-91: shrink on a constraint violation, on an increase of the plant objective or
        on rho < rho_lb; keep the radius for rho in [rho_lb, rho_ub); grow it (capped) above."""
        p = self.TR_parameters
        r = radius
        gp_old = self.GP_inference(np.asarray(x_initial, dtype=np.float64), self.inference_datasets)[0][0]
        gp_new = self.GP_inference(np.asarray(x_new, dtype=np.float64), self.inference_datasets)[0][0]
        for i in range(1, self.n_fun):
            if plant_newoutput[i] < 0.:
                return x_initial, r * p["radius_red"]
        rho = (plant_newoutput[0] - plant_oldoutput[0]) / (gp_new - gp_old)
        if plant_oldoutput[0] < plant_newoutput[0]:
            return x_initial, r * p["radius_red"]
        if rho < p["rho_lb"]:
            return x_initial, r * p["radius_red"]
        elif rho < p["rho_ub"]:
            return x_new, r
        return x_new, min(r * p["radius_inc"], p["radius_max"])
