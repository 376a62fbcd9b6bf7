"""A short SafeOpt campaign with *fitted* hyper-parameters (reference loop test/test_SafeOpt.py:135-186: refit by DE after
every sample, noise-free Benoit plant -> the fit drives log sigma_n to its lower bound -5): per iteration the device
sweep against the oracle on the same model.  Dev tool behind tests/test_gpu_parity.py::test_fitted_campaign_models_*.
    python tools/dev_fitted_campaign.py > gpurun_out/fitted_campaign.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                      # noqa: E402
from oracle import extended        # noqa: E402
from safebo_amd import SafeOpt     # noqa: E402


def benoit_f(u, noise=0):
    return u[0] ** 2 + u[1] ** 2 + u[0] * u[1]


def benoit_g(u, noise=0):
    return -(1. - u[0] + u[1] ** 2 + 2. * u[1])


def main():
    bound = np.array([[-.6, 1.5], [-1., 1.]])
    grid = (72, 70)
    m = SafeOpt.BO([benoit_f, benoit_g], bound, 3.0, grid=grid, seed=7)
    m.de_options = {"seed": 3, "maxiter": 40, "tol": 1e-3}
    X, Y = m.Data_sampling(4, np.array([1.4, -.8]), 0.3)        # test/test_SafeOpt.py:28-31
    m.GP_initialization(X, Y, "RBF", multi_hyper=5, var_out=True)
    pts = oracle.grid_points(bound[:, 0], bound[:, 1], list(grid))
    for it in range(16):
        ds = m.inference_datasets
        res = m.sweep(want_masks=True)
        masks = {k: m.engine.mask(k) for k in ("S", "U", "M")}
        masks["G"] = m.engine.mask("G", 1)
        mean, var = m.engine.posterior()
        ref = oracle.safeopt_sweep(pts, ds, 3.0)
        sub = np.arange(0, pts.shape[0], 29)
        gm, gv = extended.posterior_given_invK(pts[sub], ds)
        ys = np.maximum(1.0, ds["Y_std"])
        dm, dv = np.abs(mean - ref["mean"]) / ys, np.abs(var - ref["var"]) / ys ** 2
        rec = dict(it=it, n=int(m.n_point), hyp=np.round(ds["hypopt"], 3).tolist(),
                   cond=[float(np.linalg.cond(np.linalg.inv(k))) for k in ds["invKopt"]],
                   kernel=m.engine.profile()["posterior_kernel"],
                   dmean=float(dm.max()), dvar=float(dv.max()),
                   dmean_abs=float(np.abs(mean - ref["mean"]).max()), dvar_abs=float(np.abs(var - ref["var"]).max()),
                   oracle_vs_ext=[float(np.max(np.abs(ref["mean"][sub] - gm) / ys)), float(np.max(np.abs(ref["var"][sub] - gv) / ys ** 2))],
                   device_vs_ext=[float(np.max(np.abs(mean[sub] - gm) / ys)), float(np.max(np.abs(var[sub] - gv) / ys ** 2))],
                   mask_diff={k: int((masks[k] != (ref[k] if k != "G" else ref["G"][0])).sum()) for k in masks},
                   counts={k: int(v.sum()) for k, v in masks.items()},
                   min_abs_lcb1=float(np.min(np.abs(ref["lcb"][:, 1]))),
                   idx=[res["minimizer_index"], int(ref["minimizer_index"]), int(res["expander_index"]), int(ref["expander_best_index"])],
                   L=[list(map(float, res["L"])), list(map(float, ref["L"]))])
        print(json.dumps(rec), flush=True)
        x_new = res["minimizer_x"] if res["choose_minimizer"] else res["expander_x"]
        m.add_sample(x_new, m.calculate_plant_outputs(x_new))


if __name__ == "__main__":
    main()
