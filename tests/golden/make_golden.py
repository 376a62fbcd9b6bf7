"""Generates tests/golden/*.npz from the NumPy oracle (run from the repo root: python tests/golden/make_golden.py).

The reference cannot be executed in the build container (no jax) and its tests hold no golden vectors for
this path, so these fixtures are produced by the oracle itself: they pin the oracle against regressions and
give the GPU tests fixed inputs/outputs that do not depend on the oracle running on the GPU box.
Each file holds the inputs (X, Y, hypopt, bound, count, b, quirk) and the outputs of one small case.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from safebo_amd import synthetic  # noqa: E402

CASES = [
    # name, config, n, count, b, quirk
    ("benoit_n20_50x50", "A", 20, [50, 50], 3.0, True),
    ("benoit_n4_40x40", "A", 4, [40, 40], 3.0, True),          # the reference's own starting size (test_SafeOpt.py:28)
    ("benoit_n128_64x48", "B", 128, [64, 48], 3.0, True),
    ("benoit_n100_33x31_noquirk", "B", 100, [33, 31], 2.0, False),
    ("wo3_n64_48x40", "C", 64, [48, 40], 2.0, True),
    ("wo3_n64_48x40_noquirk", "C", 64, [48, 40], 2.0, False),
    ("rosen4_n128_9x8x7x6", "D", 128, [9, 8, 7, 6], 0.5, True),
    ("benoit_n512_32x24", "H", 512, [32, 24], 3.0, True),
]


def nll_case(n, d, P=40):
    """inputs / outputs of the hyper-parameter objective (oracle.negative_loglikelihood, models/GP_Safe.py:169-192): a
    population inside the reference's search box (:205-206) on a seeded data set"""
    rng = np.random.default_rng(n)
    X = rng.uniform(-1, 1, size=(n, d))
    Xn = (X - X.mean(0)) / X.std(0)
    y = np.sin(Xn.sum(1))
    y = (y - y.mean()) / y.std()
    H = np.column_stack([rng.uniform(-1.5, 1.5, size=(P, d + 1)), rng.uniform(-5.0, -2.0, size=P)])
    return Xn, y, H, np.array([oracle.negative_loglikelihood(h, Xn, y) for h in H])


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    rec = {}
    for n, d in ((4, 2), (20, 2), (45, 2), (128, 4), (300, 3)):
        Xn, y, H, f = nll_case(n, d)
        rec.update({f"X_{n}_{d}": Xn, f"y_{n}_{d}": y, f"H_{n}_{d}": H, f"nll_{n}_{d}": f})
    np.savez_compressed(os.path.join(out_dir, "nll_population.npz"), **rec)
    print("nll_population", {k: v.shape for k, v in rec.items() if k.startswith("nll")})
    for name, cfg_name, n, count, b, quirk in CASES:
        cfg = synthetic.make_config(cfg_name, n=n)
        lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
        pts = oracle.grid_points(lo, hi, count)
        ds = oracle.make_inference_dataset(cfg["X"], cfg["Y"], cfg["ds"]["hypopt"])
        for k in ("X_norm", "Y_norm", "X_mean", "X_std", "Y_mean", "Y_std"):
            assert np.array_equal(ds[k], cfg["ds"][k]), k      # host generator == oracle restatement
        r = oracle.safeopt_sweep(pts, ds, b, quirk_L_index=quirk)
        g = oracle.goose_sweep(pts, ds, b, quirk_L_index=quirk, mean_var=(r["mean"], r["var"]))
        rec = dict(X=cfg["X"], Y=cfg["Y"], hypopt=ds["hypopt"], bound=cfg["bound"], count=np.array(count), b=b,
                   quirk=quirk, mean=r["mean"], var=r["var"], lcb=r["lcb"], ucb=r["ucb"], S=r["S"], U=r["U"], L=r["L"],
                   empty=r["empty_safe_set"])
        if not r["empty_safe_set"]:
            rec.update(u_star=r["u_star"], M=r["M"], G=r["G"], minimizer_index=r["minimizer_index"],
                       minimizer_std=r["minimizer_std"], expander_index=r["expander_index"], expander_std=r["expander_std"],
                       expander_best=r["expander_best"], choose_minimizer=r["choose_minimizer"],
                       O=g["O"], safe_min_index=g["safe_min_index"], safe_min_lcb=g["safe_min_lcb"],
                       target_index_c=g["target_index_c"], target_lcb_c=g["target_lcb_c"], target_index=g["target_index"],
                       target_best=g["target_best"], explore_index=g["explore_index"], choose_safe_min=g["choose_safe_min"])
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, "N =", pts.shape[0], "S", int(r["S"].sum()), "empty", r["empty_safe_set"],
              "" if r["empty_safe_set"] else ("M %d G %s O %s" % (r["M"].sum(), r["G"].sum(1), g["O"].sum(1))))


if __name__ == "__main__":
    main()
