"""Developer driver: SafeOpt sweeps of config H on a READY posterior (set phase only) -- for kernel stats without the posterior kernel in front."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
cfg = synthetic.make_config(sys.argv[1] if len(sys.argv) > 1 else "H")
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
eng.set_model(cfg["ds"], dtype="f64")
eng.posterior_run()
dev = []
for it in range(60):
    eng.sweep_safeopt(cfg["b"], posterior_ready=True)
    dev.append(eng.profile()["total_ms"])
print(f"set phase only: device {np.median(dev[10:]):.3f} ms")
eng.close()
