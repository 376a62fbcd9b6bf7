#!/usr/bin/env python3
"""Guard band of the first sweep (K1i) and of the second (K1b) of a model on the BASELINE 2-D configs: widths and counts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import safebo_amd
from safebo_amd import synthetic
for name in sys.argv[1:] or ["B", "H", "C"]:
    cfg = synthetic.make_config(name)
    q = cfg["q"]
    eng = safebo_amd.SweepEngine(0)
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], list(cfg["count"]))
    eng.set_model(cfg["ds"], dtype="f64", use_invK=True)
    for k in range(2):
        r = eng.sweep_safeopt(cfg["b"])
        p = eng.profile()
        g = eng.sweep_goose(cfg["b"], posterior_ready=True) if q > 1 else None
        ys = np.maximum(1.0, cfg["ds"]["Y_std"])
        print(f"config {name} sweep {k + 1}: kernel {p['posterior_kernel']}  dm/ys {np.array(p['guard_dm'][:q]) / ys}  dv/ys^2 {np.array(p['guard_dv'][:q]) / ys ** 2}"
              f"  rl {np.array(p['guard_rl'][:q])}  safeopt guard {r['guard_band']}/{r['guard_rechecks']}/{r['guard_passes']}"
              + (f"  goose guard {g['guard_band']}/{g['guard_rechecks']}/{g['guard_passes']}" if g else "") + f"  L {r['L'][:q]}")
    eng.close()
