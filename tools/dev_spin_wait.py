"""Developer driver: wall clock per sweep / per model change with the host polling the stream (option spin_wait = 1) or
sleeping in hipStreamSynchronize (0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
OPT = sys.argv[1] if len(sys.argv) > 1 else "spin_wait"
eng = safebo_amd.SweepEngine(0)
for name in ("B", "H", "C"):
    cfg = synthetic.make_config(name)
    eng.set_model(cfg["ds"])
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
    for spin in (0, 1, 0, 1):
        eng.set_option(OPT, spin)
        for _ in range(20):
            eng.sweep_safeopt(cfg["b"])
        t = []
        for _ in range(200):
            t0 = time.perf_counter()
            eng.sweep_safeopt(cfg["b"])
            t.append(time.perf_counter() - t0)
        dev = eng.profile()["total_ms"]
        tm = []
        for _ in range(30):
            t0 = time.perf_counter()
            eng.set_model(cfg["ds"])
            tm.append(time.perf_counter() - t0)
        print(name, OPT, spin, "sweep wall ms median", round(1e3 * float(np.median(t)), 4), "mean", round(1e3 * float(np.mean(t)), 4), "device", round(dev, 4),
              "set_model ms", round(1e3 * float(np.median(tm)), 4), flush=True)
eng.close()
