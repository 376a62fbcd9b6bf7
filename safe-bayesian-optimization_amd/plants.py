"""Host-side mirror of the reference's William-Otto reactor problem class (problems/WilliamOttoReactor_Problem.py), with
the steady-state solve batched on the device (``sbo_plant_wo``).

Same method names as the reference (``get_objective``, ``get_constraint1``, ``get_constraint2``) so its drivers keep
working; ``evaluate(U)`` is the batched form (the reference loops over points in Python, e.g. 10 000 ``fsolve`` calls for
its contour table).  Differences: the disturbance noise comes from a NumPy generator (no JAX PRNG), clipped and scaled as
in the reference (:50-52), and is added to Fb before the solve."""
from __future__ import annotations

import numpy as np

from .engine import SweepEngine


class WilliamOttoReactor:
    def __init__(self, measure_disturbance: bool = False, device: int = 0, seed: int = 42, engine: SweepEngine | None = None):
        self.measure_disturbance = measure_disturbance
        self.key = np.random.default_rng(seed)
        self._normal = 0.0
        self._engine = engine
        self._device = device
        self.noise_generator()

    @property
    def engine(self) -> SweepEngine:
        if self._engine is None:
            self._engine = SweepEngine(self._device)
        return self._engine

    def noise_generator(self):
        """Advance the disturbance stream (reference: split the PRNG key, :16-17); all three outputs of one call share it."""
        self._normal = float(np.clip(self.key.standard_normal(), -2.05, 2.05))

    def evaluate(self, U, noise: float = 0.0) -> np.ndarray:
        """[N, 3] = (objective, constraint 1, constraint 2) for U[N, 2] = (Fb, Tr)."""
        U = np.array(np.atleast_2d(U), dtype=np.float64)
        U[:, 0] += self._normal * np.sqrt(noise)
        return self.engine.plant_wo(U)

    def _one(self, u, noise, col):
        val = float(self.evaluate(np.asarray(u, dtype=np.float64)[None, :2], noise)[0, col])
        if self.measure_disturbance:
            return val, self._normal * np.sqrt(noise)
        return val

    def get_objective(self, u, noise=0.):
        return self._one(u, noise, 0)

    def get_constraint1(self, u, noise=0.):
        return self._one(u, noise, 1)

    def get_constraint2(self, u, noise=0.):
        return self._one(u, noise, 2)
