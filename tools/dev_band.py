#!/usr/bin/env python3
"""development: what the guard band of K1i / K1b is made of, per config and length scale: tools/dev_band.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
eng.set_option("guard_audit_every", 1)
eng.set_option("guard_audit", 16384)
def show(tag, ds, lo, hi, count, b):
    eng.set_grid(lo, hi, count)
    eng.set_model(ds, dtype="f64")
    for sweep in range(2):
        res = eng.sweep_safeopt(b)
        eng.synchronize()
        p = eng.profile()
        q = 2
        f = lambda k: " ".join(f"{v:.2e}" for v in p[k][:q])
        print(f"{tag} sweep {sweep} kernel {p['posterior_kernel']} open {res['guard_band']} passes {res['guard_passes']} | dm {f('guard_dm')} an {f('guard_analytic_dm')} probe {f('guard_probe_dm')} | dv {f('guard_dv')} an {f('guard_analytic_dv')} probe {f('guard_probe_dv')} | rl {f('guard_rl')} | audit {p['guard_audit_samples']} viol {p['guard_audit_violations']} worst {p['guard_audit_worst']:.3g}", flush=True)
for name in ("B", "H"):
    cfg = synthetic.make_config(name)
    show(name, cfg["ds"], cfg["bound"][:, 0], cfg["bound"][:, 1], list(cfg["count"]), cfg["b"])
base = synthetic.make_config("B", n=128)
for le in (0.0, -0.5, -0.85, -1.0, -1.25, -1.5):
    ds = synthetic.make_dataset(base["X"], base["Y"], synthetic.default_hypopt(2, 2, log_ell=le))
    show(f"B128 logell {le}", ds, base["bound"][:, 0], base["bound"][:, 1], [512, 512], 3.0)
