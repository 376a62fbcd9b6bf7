"""Developer driver: 300 model-change iterations of config B, per-step wall / device times, outliers listed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
cfg = synthetic.make_config("B")
alt = synthetic.make_config("B", seed=synthetic.SEED0 + 100 + cfg["index"])
eng.set_model(cfg["ds"])
eng.set_grid_sharded(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0):
    eng.sweep_safeopt(cfg["b"])
rows = []
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    t0 = time.perf_counter()
    eng.set_model((alt if it % 2 else cfg)["ds"])
    t1 = time.perf_counter()
    eng.sweep_safeopt(cfg["b"])
    t2 = time.perf_counter()
    p = eng.profile()
    rows.append((it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, p["posterior_ms"], p["total_ms"], p["posterior_setup_ms"], p["posterior_kernel"]))
a = np.array([r[1:6] for r in rows])
print("median set_model %.3f sweep %.3f k1 %.3f device %.3f build %.3f" % tuple(np.median(a, axis=0)))
print("mean   set_model %.3f sweep %.3f k1 %.3f device %.3f build %.3f" % tuple(np.mean(a, axis=0)))
for r in rows:
    if r[1] > 1.0 or r[2] > 1.0:
        print("outlier step %d: set_model %.3f sweep %.3f k1 %.3f device %.3f build %.3f kernel %d" % r)
eng.close()
