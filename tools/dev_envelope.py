"""Posterior error of every grid kernel over the reference's whole hyper-parameter box (models/GP_Safe.py:205-206:
log ell, log sigma_f in [-1.5, 1.5], log sigma_n in [-5, -2]) -- dev tool behind DESIGN.md's parity table.

For each regime: |device - oracle|, and both against the extended-precision evaluation of the same formula
(oracle/extended.py), for K1b / K1g / the generic kernel, caller's invK and library Cholesky.  Run on the GPU box:
    python tools/dev_envelope.py > gpurun_out/envelope.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                      # noqa: E402
import safebo_amd                  # noqa: E402
from oracle import extended        # noqa: E402
from safebo_amd import synthetic   # noqa: E402


def nerr(a, b, ystd, p):
    return float(np.max(np.abs(np.asarray(a, dtype=np.longdouble) - b) / np.maximum(1.0, ystd) ** p))


def main():
    eng = safebo_amd.SweepEngine(0)
    rows = []
    regimes = [(-2.0, -0.5, 0.0), (-3.5, -0.5, 0.0), (-5.0, -0.5, 0.0), (-5.0, 0.5, 0.0), (-5.0, 1.5, 0.0), (-5.0, -1.5, 0.0),
               (-5.0, -0.5, 1.5), (-5.0, 0.5, 1.5), (-5.0, -1.0, -1.5), (-4.0, 0.0, 1.0)]
    for cfg_name, n in (("B", 20), ("B", 128), ("C", 256), ("H", 512)):
        for log_sn, log_ell, log_sf in regimes:
            cfg = synthetic.make_config(cfg_name, n=n)
            d, q = cfg["d"], cfg["q"]
            ds = synthetic.make_dataset(cfg["X"], cfg["Y"], synthetic.default_hypopt(d, q, log_ell=log_ell, log_sf=log_sf, log_sn=log_sn))
            lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [96, 80]
            pts = oracle.grid_points(lo, hi, count)
            sub = np.arange(0, pts.shape[0], 41)
            om, ov = oracle.gp_inference(pts, ds)
            gm, gv = extended.posterior_given_invK(pts[sub], ds)
            tm, tv = extended.posterior_true(pts[sub], ds)
            ys = ds["Y_std"]
            rec = dict(config=cfg_name, n=n, log_sn=log_sn, log_ell=log_ell, log_sf=log_sf,
                       cond=float(np.linalg.cond(np.linalg.inv(ds["invKopt"][0]))),
                       oracle_vs_given=[nerr(om[sub], gm, ys, 1), nerr(ov[sub], gv, ys, 2)],
                       oracle_vs_true=[nerr(om[sub], tm, ys, 1), nerr(ov[sub], tv, ys, 2)])
            for use_invK in (True, False):
                for bil in (1, 0):
                    eng.set_option("bilinear", bil)
                    eng.set_model(ds, use_invK=use_invK)
                    eng.set_grid(lo, hi, count)
                    mean, var = eng.posterior()
                    kern = eng.profile()["posterior_kernel"]
                    if bil == 1 and kern != 4:
                        continue
                    ref_m, ref_v = (gm, gv) if use_invK else (tm, tv)
                    key = f"{'invK' if use_invK else 'chol'}_k{kern}"
                    rec[key] = dict(vs_oracle=[nerr(mean, om, ys, 1), nerr(var, ov, ys, 2)],
                                    vs_oracle_abs=[float(np.max(np.abs(mean - om))), float(np.max(np.abs(var - ov)))],
                                    vs_extended=[nerr(mean[sub], ref_m, ys, 1), nerr(var[sub], ref_v, ys, 2)])
            eng.set_option("bilinear", 1)
            rows.append(rec)
            print(json.dumps(rec), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
