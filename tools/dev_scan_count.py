"""Developer driver: how many candidates of a SafeOpt sweep the coarse expander decision leaves open (SBO_DEBUG_SCAN=1 prints them)."""
import os, sys
os.environ["SBO_DEBUG_SCAN"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name in ("H", "B"):
    cfg = synthetic.make_config(name)
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
    eng.set_model(cfg["ds"], dtype="f64")
    r = eng.sweep_safeopt(cfg["b"])
    print(name, {k: r[k] for k in ("count_S", "count_U", "count_M", "n_exact_rechecks")}, "G", r["count_G"].tolist(), "L", r["L"].tolist(), flush=True)
eng.close()
