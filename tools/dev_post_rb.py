"""Developer driver: resident-model sweeps of configs H / B / C with the tiled (post_rb 1, 2) and the resident (3) posterior kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name in ("H", "B", "C"):
    cfg = synthetic.make_config(name)
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
    for rb in (1, 2, 3):
        eng.set_option("post_rb", rb)
        eng.set_model(cfg["ds"], dtype="f64")
        dev, k1, wall = [], [], []
        for it in range(60):
            t0 = time.perf_counter()
            eng.sweep_safeopt(cfg["b"])
            wall.append(time.perf_counter() - t0)
            p = eng.profile()
            dev.append(p["total_ms"]); k1.append(p["posterior_ms"])
        print(f"{name} post_rb={rb}: wall {1e3 * np.median(wall[10:]):.3f} ms, device {np.median(dev[10:]):.3f}, K1 {np.median(k1[10:]):.3f}", flush=True)
eng.close()
