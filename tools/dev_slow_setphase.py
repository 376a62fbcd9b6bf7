"""Developer driver: set-phase time of config B / H models with other length-scales, K1b vs K1g posterior (phase events on)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
eng.set_option("phase_events", 1)
for name, lls in (("B", (-1.5, -1.0, 0.0)), ("H", (0.0,))):
    cfg = synthetic.make_config(name)
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
    for ll in lls:
        ds = synthetic.make_dataset(cfg["X"], cfg["Y"], synthetic.default_hypopt(2, 2, log_ell=ll))
        for forced in (0, 1):
            if forced and name == "H":
                continue
            eng.set_option("bilinear", 0 if forced else 1)
            eng.set_model(ds)
            eng.sweep_safeopt(cfg["b"])
            r = eng.sweep_safeopt(cfg["b"])
            p = eng.profile()
            print(name, ll, "K1g" if forced else "auto", {k: round(v, 3) for k, v in p.items() if k.endswith("_ms")}, "L", r["L"], "u*", r["u_star"],
                  "S", r["count_S"], "U", r["count_U"], "M", r["count_M"], "G", r["count_G"], "rechecks", r["n_exact_rechecks"], flush=True)
        eng.set_option("bilinear", 1)
eng.close()
