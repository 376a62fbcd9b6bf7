"""SweepEngine -- thin NumPy/ctypes host object over libsafebo.so (one per process and per GPU).

Holds what crosses the seam of SURVEY.md section 8(b): the ``inference_datasets`` dict
(models/GP_Safe.py:236-245), ``b`` (models/SafeOpt.py:13) and a candidate set, and exposes the
batched calls that replace ``vmap(BO.lcb)`` (test/test_SafeOpt.py:337), ``Minimizer``/``Expander``
(models/SafeOpt.py:53-124) and ``minimize_obj_lcb``/``Target``/``explore_safeset``
(models/GoOSE.py:63-119).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

_DTYPES = {"f64": (L.SBO_F64, np.float64), "f32": (L.SBO_F32, np.float32),
           np.float64: (L.SBO_F64, np.float64), np.float32: (L.SBO_F32, np.float32)}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class SweepEngine:
    def __init__(self, device: int = 0):
        self._lib = L.load()
        self._ctx = C.c_void_p()
        L.check(self._lib.sbo_init(int(device), C.byref(self._ctx)))
        self.device = int(device)
        self.tag, self.np_dtype = _DTYPES["f64"]
        self.n = self.d = self.q = 0
        self.n_local = 0
        self.first = 0
        self.world, self.rank = 1, 0

    # ---- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.sbo_shutdown(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self):
        L.check(self._lib.sbo_synchronize(self._ctx))

    def set_option(self, key: str, value: int):
        L.check(self._lib.sbo_set_option(self._ctx, key.encode(), int(value)))

    # ---- multi-GPU ---------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(L.SBO_COMM_ID_BYTES)
        L.check(L.load().sbo_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, world: int, rank: int, uid: bytes | None):
        buf = C.create_string_buffer(uid, L.SBO_COMM_ID_BYTES) if uid is not None else None
        L.check(self._lib.sbo_comm_init(self._ctx, int(world), int(rank), buf))
        self.world, self.rank = int(world), int(rank)

    def comm_barrier(self):
        L.check(self._lib.sbo_comm_barrier(self._ctx))

    # ---- model -------------------------------------------------------------------------------
    def set_model(self, ds: dict, dtype="f64", kernel: str = "RBF", use_invK: bool = True):
        """Upload the ``inference_datasets`` dict.  ``use_invK=False`` lets the library factor
        K + (sn2 + float32 eps) I itself (Cholesky, contraction with L^-1)."""
        self.tag, self.np_dtype = _DTYPES[dtype]
        X_norm = _f64(ds["X_norm"])
        Y_norm = _f64(ds["Y_norm"])
        if X_norm.ndim != 2 or Y_norm.ndim != 2 or X_norm.shape[0] != Y_norm.shape[0]:
            raise ValueError("X_norm / Y_norm must be [n, d] and [n, q]")
        n, d = X_norm.shape
        q = Y_norm.shape[1]
        hyp = _f64(ds["hypopt"])
        if hyp.shape != (d + 2, q):
            raise ValueError("ERROR W and X_norm dimension should be same")   # models/GP_Safe.py:134-135
        arrs = [_f64(ds[k]) for k in ("X_mean", "X_std", "Y_mean", "Y_std")]
        if arrs[0].shape != (d,) or arrs[1].shape != (d,) or arrs[2].shape != (q,) or arrs[3].shape != (q,):
            raise ValueError("X_mean/X_std must be [d], Y_mean/Y_std must be [q]")
        args = [self._ctx, self.tag, kernel.encode(), n, d, q, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(arrs[3]),
                _ptr(X_norm), _ptr(Y_norm), _ptr(hyp)]
        if use_invK:
            # `invKopt` is a list of q separate [n, n] arrays in the reference (models/GP_Safe.py:231-232): handed over as such
            parts = [_f64(a) for a in ds["invKopt"]]
            if len(parts) != q or any(a.shape != (n, n) for a in parts):
                raise ValueError("invKopt must hold q matrices of shape [n, n]")
            ptrs = (C.c_void_p * q)(*[a.ctypes.data for a in parts])
            L.check(self._lib.sbo_model_set_list(*args, ptrs))
        else:
            L.check(self._lib.sbo_model_set(*args, None))
        self.n, self.d, self.q = n, d, q

    def append_sample(self, x_norm_new, y_norm_new):
        """One more observation under frozen hyper-parameters and normalisation: O(n^2) on the
        device instead of a rebuild.  ``x_norm_new`` [d] and ``y_norm_new`` [q] are normalised with the constants of the
        last ``set_model``."""
        x = _f64(x_norm_new).reshape(-1)
        y = _f64(y_norm_new).reshape(-1)
        if x.shape != (self.d,) or y.shape != (self.q,):
            raise ValueError("x_norm_new must be [d] and y_norm_new [q]")
        L.check(self._lib.sbo_model_append(self._ctx, _ptr(x), _ptr(y)))
        self.n += 1

    # ---- candidates --------------------------------------------------------------------------
    def set_points(self, points, first: int = 0):
        pts = np.asarray(points)
        if pts.ndim != 2:
            raise ValueError("points must be [N, d]")
        if pts.dtype == np.float32:
            tag = L.SBO_F32
            pts = np.ascontiguousarray(pts)
        else:
            tag = L.SBO_F64
            pts = _f64(pts)
        L.check(self._lib.sbo_candidates_points(self._ctx, _ptr(pts), tag, pts.shape[0], pts.shape[1], int(first)))
        self.n_local, self.first = pts.shape[0], int(first)

    def set_grid(self, lo, hi, count, first: int = 0, n_local: int | None = None):
        lo, hi = _f64(lo), _f64(hi)
        cnt = np.ascontiguousarray(np.asarray(count, dtype=np.int64))
        if not (lo.shape == hi.shape == cnt.shape) or lo.ndim != 1:
            raise ValueError("lo, hi, count must be 1-D of equal length")
        total = int(np.prod([int(c) for c in cnt]))
        if n_local is None:
            n_local = total - int(first)
        L.check(self._lib.sbo_candidates_grid(self._ctx, lo.shape[0], _ptr(lo), _ptr(hi), _ptr(cnt), int(first),
                                              int(n_local)))
        self.n_local, self.first = int(n_local), int(first)

    def set_grid_sharded(self, lo, hi, count):
        """Grid sharded over the communicator's ranks by hyper-planes of the slowest axis."""
        lo, hi = _f64(lo), _f64(hi)
        cnt = np.ascontiguousarray(np.asarray(count, dtype=np.int64))
        if not (lo.shape == hi.shape == cnt.shape) or lo.ndim != 1:
            raise ValueError("lo, hi, count must be 1-D of equal length")
        first, nloc = C.c_int64(), C.c_int64()
        L.check(self._lib.sbo_candidates_grid_sharded(self._ctx, lo.shape[0], _ptr(lo), _ptr(hi), _ptr(cnt),
                                                      C.byref(first), C.byref(nloc)))
        self.n_local, self.first = int(nloc.value), int(first.value)

    # ---- hot path ----------------------------------------------------------------------------
    def posterior_run(self):
        L.check(self._lib.sbo_posterior_run(self._ctx))

    def posterior(self):
        """Batched ``GP_inference`` (models/GP_Safe.py:310-352): (mean[N, q], var[N, q])."""
        mean = np.empty((self.n_local, self.q), dtype=self.np_dtype)
        var = np.empty((self.n_local, self.q), dtype=self.np_dtype)
        L.check(self._lib.sbo_posterior_get(self._ctx, _ptr(mean), _ptr(var)))
        return mean, var

    def bounds(self, b: float, index: int, kind: str):
        """Batched ``BO.mean/ucb/lcb(points, index)`` (models/SafeOpt.py:29-45)."""
        k = {"mean": L.SBO_MEAN, "ucb": L.SBO_UCB, "lcb": L.SBO_LCB, "var": L.SBO_VAR}[kind]
        out = np.empty(self.n_local, dtype=self.np_dtype)
        L.check(self._lib.sbo_bounds(self._ctx, float(b), int(index), k, _ptr(out)))
        return out

    def _opts(self, b, quirk, want_masks, posterior_ready, lean=False):
        return L.SweepOpts(float(b), int(bool(quirk)), int(bool(want_masks)), int(bool(posterior_ready)), int(lean))

    def sweep_safeopt(self, b: float, quirk_L_index: bool = True, want_masks: bool = False,
                      posterior_ready: bool = False, lean: bool = False) -> dict:
        """``lean``: the caller wants the sweep's result only.  1 / True: mean / var may stay unwritten where no stage of the sweep
        reads them; 2: they need not be evaluated there at all (``posterior()`` behind a lean sweep runs K1 again).  A lean sweep of
        a model with constraints reports ``L[0] = 0``: no sweep reads the objective's Lipschitz key (models/SafeOpt.py:110)."""
        res = L.SafeOptResult()
        opts = self._opts(b, quirk_L_index, want_masks, posterior_ready, lean)
        L.check(self._lib.sbo_sweep_safeopt(self._ctx, C.byref(opts), C.byref(res)))
        d, q = self.d, self.q
        return {
            "minimizer_index": int(res.minimizer_index), "minimizer_x": np.array(res.minimizer_x[:d]),
            "minimizer_std": float(res.minimizer_std),
            "expander_index_c": np.array(res.expander_index_c[:q - 1], dtype=np.int64),
            "expander_std_c": np.array(res.expander_std_c[:q - 1]),
            "expander_best_c": int(res.expander_best_c), "expander_index": int(res.expander_index),
            "expander_x": np.array(res.expander_x[:d]), "expander_std": float(res.expander_std),
            "choose_minimizer": bool(res.choose_minimizer), "u_star": float(res.u_star),
            "L": np.array(res.L[:q]), "count_S": int(res.count_S), "count_U": int(res.count_U),
            "count_M": int(res.count_M), "count_G": np.array(res.count_G[:q - 1], dtype=np.int64),
            "n_exact_rechecks": int(res.n_exact_rechecks), "guard_band": int(res.guard_band),
            "guard_rechecks": int(res.guard_rechecks), "guard_passes": int(res.guard_passes),
        }

    def sweep_goose(self, b: float, quirk_L_index: bool = True, want_masks: bool = False,
                    posterior_ready: bool = False) -> dict:
        res = L.GooseResult()
        opts = self._opts(b, quirk_L_index, want_masks, posterior_ready)
        L.check(self._lib.sbo_sweep_goose(self._ctx, C.byref(opts), C.byref(res)))
        d, q = self.d, self.q
        return {
            "safe_min_index": int(res.safe_min_index), "safe_min_x": np.array(res.safe_min_x[:d]),
            "safe_min_lcb": float(res.safe_min_lcb),
            "target_index_c": np.array(res.target_index_c[:q - 1], dtype=np.int64),
            "target_lcb_c": np.array(res.target_lcb_c[:q - 1]), "target_best_c": int(res.target_best_c),
            "target_index": int(res.target_index), "target_x": np.array(res.target_x[:d]),
            "target_lcb": float(res.target_lcb), "explore_index": int(res.explore_index),
            "explore_x": np.array(res.explore_x[:d]), "choose_safe_min": bool(res.choose_safe_min),
            "L": np.array(res.L[:q]), "count_S": int(res.count_S), "count_U": int(res.count_U),
            "count_O": np.array(res.count_O[:q - 1], dtype=np.int64), "n_exact_rechecks": int(res.n_exact_rechecks),
            "guard_band": int(res.guard_band), "guard_rechecks": int(res.guard_rechecks), "guard_passes": int(res.guard_passes),
        }

    def sweep_tr(self, b: float, x_0, r: float, posterior_ready: bool = False) -> dict:
        """Trust-region acquisition of models/GP_TR.py:43-51: argmin lcb_0 over S_t and the ball ||x - x_0|| <= r."""
        res = L.TRResult()
        opts = self._opts(b, True, True, posterior_ready)
        x0 = _f64(x_0)
        if x0.shape != (self.d,):
            raise ValueError("x_0 must have shape [d]")
        L.check(self._lib.sbo_sweep_tr(self._ctx, C.byref(opts), _ptr(x0), float(r), C.byref(res)))
        return {"index": int(res.index), "x": np.array(res.x[:self.d]), "lcb": float(res.lcb),
                "count_S": int(res.count_S), "count_T": int(res.count_T), "guard_band": int(res.guard_band),
                "guard_rechecks": int(res.guard_rechecks), "guard_passes": int(res.guard_passes)}

    def explore_safeset(self, target):
        """``BO.explore_safeset(target)`` (models/GoOSE.py:116-119) for a caller's own target: (flat index, x) of the candidate of the
        last sweep's safe set closest to ``target`` -- arg-min on the device."""
        t = _f64(target)
        if t.shape != (self.d,):
            raise ValueError("target must have shape [d]")
        idx = C.c_int64()
        x = np.zeros(L.SBO_MAX_D)
        L.check(self._lib.sbo_explore_safeset(self._ctx, _ptr(t), C.byref(idx), _ptr(x)))
        return int(idx.value), x[:self.d].copy()

    def mask(self, which: str, c: int = 0) -> np.ndarray:
        w = {"S": L.SBO_MASK_S, "U": L.SBO_MASK_U, "M": L.SBO_MASK_M, "G": L.SBO_MASK_G, "O": L.SBO_MASK_O}[which]
        out = np.empty(self.n_local, dtype=np.uint8)
        L.check(self._lib.sbo_masks_get(self._ctx, w, int(c), _ptr(out)))
        return out.astype(bool)

    def nll_batch(self, X_norm, y, hypers):
        """``negative_loglikelihood`` (models/GP_Safe.py:169-192) for a population: hypers[P, d+2] -> NLL[P]."""
        X = _f64(X_norm)
        yv = _f64(np.asarray(y).reshape(-1))
        H = _f64(hypers)
        if X.ndim != 2 or H.ndim != 2 or H.shape[1] != X.shape[1] + 2 or yv.shape[0] != X.shape[0]:
            raise ValueError("X_norm [n, d], y [n], hypers [P, d + 2]")
        out = np.empty(H.shape[0], dtype=np.float64)
        L.check(self._lib.sbo_nll_batch(self._ctx, X.shape[0], X.shape[1], _ptr(X), _ptr(yv), H.shape[0], _ptr(H), _ptr(out)))
        return out

    def fit_de(self, X_norm, y, bounds, init_pop, seed: int = 0, maxiter: int = 1000, tol: float = 0.01, atol: float = 0.0):
        """Differential evolution of ``negative_loglikelihood`` on the device (``sbo_fit_de``).  bounds[d+2, 2],
        init_pop[P, d+2].  Returns (best hyper-parameters [d+2], NLL, generations run)."""
        X = _f64(X_norm)
        yv = _f64(np.asarray(y).reshape(-1))
        B = _f64(bounds)
        pop = _f64(init_pop)
        D = X.shape[1] + 2
        if X.ndim != 2 or B.shape != (D, 2) or pop.ndim != 2 or pop.shape[1] != D or yv.shape[0] != X.shape[0]:
            raise ValueError("X_norm [n, d], y [n], bounds [d+2, 2], init_pop [P, d+2]")
        lo, hi = _f64(B[:, 0]), _f64(B[:, 1])
        best, energy, gens = np.empty(D), C.c_double(), C.c_int()
        L.check(self._lib.sbo_fit_de(self._ctx, X.shape[0], X.shape[1], _ptr(X), _ptr(yv), pop.shape[0], _ptr(lo), _ptr(hi), _ptr(pop),
                                     int(seed), int(maxiter), float(tol), float(atol), _ptr(best), C.byref(energy), C.byref(gens)))
        return best, float(energy.value), int(gens.value)

    def plant_wo(self, U) -> np.ndarray:
        """William-Otto reactor outputs (objective, constraint 1, constraint 2) for input rows U[N, 2] = (Fb, Tr)."""
        u = _f64(np.atleast_2d(U))
        if u.ndim != 2 or u.shape[1] != 2:
            raise ValueError("U must be [N, 2]")
        out = np.empty((u.shape[0], 3), dtype=np.float64)
        L.check(self._lib.sbo_plant_wo(self._ctx, u.shape[0], _ptr(u), _ptr(out)))
        return out

    def profile_struct(self):
        """The raw ``sbo_profile`` of the last sweep (a timing loop keeps these and converts them with ``profile_dict`` afterwards:
        building the dict costs several microseconds of a 0.2 ms sweep)."""
        p = L.Profile()
        L.check(self._lib.sbo_profile_get(self._ctx, C.byref(p)))
        return p

    @staticmethod
    def profile_dict(p) -> dict:
        return {name: (list(getattr(p, name)) if name.startswith("guard_") and not name.startswith("guard_audit") and name != "guard_ms" else getattr(p, name))
                for name, _ in L.Profile._fields_}

    def profile(self) -> dict:
        return self.profile_dict(self.profile_struct())
