#!/usr/bin/env python3
"""development: how many 64 x 128 posterior tiles hold a safe candidate / a G member / an M member"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import safebo_amd
from safebo_amd import synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "H"
cfg = synthetic.make_config(name)
eng = safebo_amd.SweepEngine(0)
eng.set_model(cfg["ds"], dtype="f64")
W, H = cfg["count"]
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], [W, H])
res = eng.sweep_safeopt(cfg["b"], want_masks=True)
for k in ("S", "U", "M", "G"):
    m = eng.mask(k, 1) if k == "G" else eng.mask(k)
    t = m.reshape(H // 64, 64, W // 128, 128).any(axis=(1, 3))
    print(f"{name} {k}: {m.sum()} candidates ({100.0 * m.mean():.1f} %), {t.sum()} of {t.size} tiles")
print({k: res[k] for k in ("count_S", "count_U", "count_M", "count_G", "n_exact_rechecks")})
