#!/usr/bin/env python3
"""development: wall time of model-change iterations (set_model + first sweep) under an option's values:
    tools/dev_iter_ab.py CONFIG OPTION V0,V1[,..] [ITERS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import safebo_amd
from safebo_amd import synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "H"
opt = sys.argv[2] if len(sys.argv) > 2 else "grad_defer"
vals = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 60
cfg = synthetic.make_config(name)
alt = synthetic.make_config(name, seed=synthetic.SEED0 + 100 + cfg["index"])
eng = safebo_amd.SweepEngine(0)
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
for rep in range(3):
    for v in vals:
        eng.set_option(opt, v)
        ts, tm = [], []
        for it in range(iters + 6):
            t0 = time.perf_counter()
            eng.set_model((alt if it % 2 else cfg)["ds"], dtype="f64")
            t1 = time.perf_counter()
            eng.sweep_safeopt(cfg["b"])
            t2 = time.perf_counter()
            if it >= 6:
                ts.append(t2 - t0); tm.append(t1 - t0)
        p = eng.profile()
        print(f"{name} {opt}={v}: iteration {1e3 * np.median(ts):.4f} ms (mean {1e3 * np.mean(ts):.4f}), set_model {1e3 * np.median(tm):.4f}, kernel {p['posterior_kernel']} path {p['set_path']}", flush=True)
eng.close()
