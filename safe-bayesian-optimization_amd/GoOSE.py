"""Host-side mirror of the reference's GoOSE class (models/GoOSE.py) on the MI355X sweep engine.

    x_safe_min, min_safe_lcb = GP_m.minimize_obj_lcb()         # argmin_{S_t} lcb_0          models/GoOSE.py:63-67
    x_target, target_lcb     = GP_m.Target()                   # argmin over the optimistic sets   :80-114
    x_new                    = GP_m.explore_safeset(x_target)  # argmin_{S_t} ||x - target||_2     :116-119

all three read one device sweep of the candidate grid (cached until the model changes).
"""
from __future__ import annotations

import numpy as np

from .SafeOpt import BO as _SafeOptBO


class BO(_SafeOptBO):
    def __init__(self, plant_system, bound, b, grid=None, device: int = 0, dtype: str = "f64",
                 reference_quirk_L_index: bool = True, seed: int = 42):
        _SafeOptBO.__init__(self, plant_system, bound, b, grid=grid, device=device, dtype=dtype,
                            reference_quirk_L_index=reference_quirk_L_index, seed=seed)
        self._goose_cache = None

    def goose_sweep(self, want_masks: bool = False) -> dict:
        key = (self._model_version, self.grid)
        if self._goose_cache is not None and self._goose_cache[0] == key and not want_masks:
            return self._goose_cache[1]
        self._grid_resident()
        res = self.engine.sweep_goose(self.b, quirk_L_index=self.reference_quirk_L_index, want_masks=want_masks)
        self._goose_cache = (key, res)
        return res

    def minimize_obj_lcb(self):
        res = self.goose_sweep()
        return res["safe_min_x"], res["safe_min_lcb"]

    def Target(self):
        res = self.goose_sweep()
        if res["target_index"] < 0:      # no optimistic point on this grid
            return np.full(len(self.grid), np.nan), np.inf
        return res["target_x"], res["target_lcb"]

    def explore_safeset(self, target):
        """Closest safe candidate to ``target`` (cdist Euclidean, models/GoOSE.py:117)."""
        res = self.goose_sweep()
        target = np.asarray(target, dtype=np.float64)
        if res["target_index"] >= 0 and np.array_equal(target, res["target_x"]):
            return res["explore_x"]
        # (a target of the caller's own: the same arg-min on the device, over the safe set the sweep left resident)
        return self.engine.explore_safeset(target)[1]
