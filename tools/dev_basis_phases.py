import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name in ("B", "H"):
    cfg = synthetic.make_config(name)
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
    for i in range(3):
        eng.set_model(cfg["ds"])
    eng.sweep_safeopt(cfg["b"])
