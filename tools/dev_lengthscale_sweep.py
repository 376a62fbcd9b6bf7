"""Developer driver: resident-model SafeOpt sweep time against the length-scale, full-size configs B and H -- where the GEMM
posterior (K1b) holds over the reference's hyper-parameter box (log ell in [-1.5, 1.5], models/GP_Safe.py:205) and what the
fall-back to the O(n^2) kernel (K1g) costs.  Prints JSON lines."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name in ("B", "H"):
    cfg = synthetic.make_config(name)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    eng.set_grid(lo, hi, cfg["count"])
    N = int(np.prod(cfg["count"]))
    for ll in (-1.5, -1.25, -1.0, -0.75, -0.5, 0.0, 0.5, 1.5):
        ds = synthetic.make_dataset(cfg["X"], cfg["Y"], synthetic.default_hypopt(2, 2, log_ell=ll))
        rec = {"config": name, "n": cfg["n"], "log_ell": ll}
        for forced in (0, 1):
            if forced and name == "H" and ll not in (-1.5, -0.5):
                continue
            eng.set_option("bilinear", 0 if forced else 1)
            eng.set_model(ds)
            try:
                eng.sweep_safeopt(cfg["b"])
            except safebo_amd.EmptySafeSetError:
                rec["empty"] = True
                continue
            reps = 3 if forced else 20
            t = time.perf_counter()
            for _ in range(reps):
                eng.sweep_safeopt(cfg["b"])
            dt = (time.perf_counter() - t) / reps
            p = eng.profile()
            rec["K1g" if forced else "auto"] = {"kernel": p["posterior_kernel"], "sweep_ms": dt * 1e3, "k1_ms": p["posterior_ms"], "cand_per_s": N / dt}
        eng.set_option("bilinear", 1)
        print(json.dumps(rec), flush=True)
eng.close()
