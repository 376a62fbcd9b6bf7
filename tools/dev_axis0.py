"""Developer driver: resident-model sweeps of configs H / B / C with the fine axis-0 pass as a workgroup per line (0) / a wave per line (1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name, kind in (("H", "safeopt"), ("B", "safeopt"), ("C", "safeopt"), ("C", "goose")):
    cfg = synthetic.make_config(name)
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
    eng.set_model(cfg["ds"], dtype="f64")
    for w in (0, 1, 0, 1):
        eng.set_option("axis0_waves", w)
        dev, wall = [], []
        for it in range(80):
            t0 = time.perf_counter()
            (eng.sweep_safeopt if kind == "safeopt" else eng.sweep_goose)(cfg["b"])
            wall.append(time.perf_counter() - t0)
            dev.append(eng.profile()["total_ms"])
        print(f"{name} {kind} axis0_waves={w}: wall {1e3 * np.median(wall[10:]):.3f} ms, device {np.median(dev[10:]):.3f}", flush=True)
eng.close()
