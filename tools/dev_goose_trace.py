"""Developer driver: a few GoOSE sweeps of config B (posterior reused) -- run under rocprofv3 --kernel-trace for a timeline."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "B"
eng = safebo_amd.SweepEngine(0)
cfg = synthetic.make_config(name)
eng.set_model(cfg["ds"], dtype="f64")
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
eng.posterior_run()
for it in range(6):
    r = eng.sweep_goose(cfg["b"], posterior_ready=True)
    p = eng.profile()
    print(it, round(p["total_ms"], 3), "S", r["count_S"], "U", r["count_U"], "O", r["count_O"].tolist(), r["target_index"], r["explore_index"], flush=True)
print(name, {k: round(v, 3) for k, v in p.items() if k.endswith("_ms")}, "O", r["count_O"].tolist(), "rechecks", r["n_exact_rechecks"])
S = eng.mask("S")
for c in range(1, cfg["q"]):
    u = eng.bounds(cfg["b"], c, "ucb")
    Lc = r["L"][cfg["q"] - 1]
    h = (cfg["bound"][:, 1] - cfg["bound"][:, 0]) / (np.array(cfg["count"]) - 1)
    print(f"   c={c}: L={Lc:.4g} rmax={u[S].max() / Lc:.4g}  steps per axis: {np.round(u[S].max() / Lc / h, 1).tolist()}")
