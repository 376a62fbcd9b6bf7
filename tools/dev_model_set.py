"""Developer driver: sbo_model_set time (device factorisation) for a few n, both factor modes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name, n in (("B", 64), ("B", 100), ("B", 128), ("B", 200), ("C", 256), ("H", 512), ("H", 1024), ("E", 2048)):
    cfg = synthetic.make_config(name, n=n)
    for use_invK in (True, False):
        ts = []
        for it in range(3):
            t = time.perf_counter(); eng.set_model(cfg["ds"], dtype="f64", use_invK=use_invK); ts.append(time.perf_counter() - t)
        print(f"{name} n={n} q={cfg['q']} use_invK={use_invK}: set_model {min(ts)*1e3:.1f} ms", flush=True)
# append timing
cfg = synthetic.make_config("H", n=500)
ds = cfg["ds"]
eng.set_model(ds, dtype="f64")
xn = (np.array([0.3, -0.2]) - ds["X_mean"]) / ds["X_std"]
ts = []
for it in range(10):
    t = time.perf_counter(); eng.append_sample(xn + 0.01 * it, np.zeros(cfg["q"])); ts.append(time.perf_counter() - t)
print(f"append_sample at n=500..510: {min(ts)*1e3:.2f} ms (first {ts[0]*1e3:.2f} ms)")
