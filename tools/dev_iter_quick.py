"""Developer driver: wall-clock cost of model-change iterations (set_model + tables + sweep) on configs B / H / C, optionally with options:
    python tools/dev_iter_quick.py [opt=val,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):
    k, v = kv.split("=")
    eng.set_option(k, int(v))
for name, kind in (("B", "safeopt"), ("H", "safeopt"), ("C", "goose"), ("C", "safeopt")):
    cfg = synthetic.make_config(name)
    alt = synthetic.make_config(name, seed=synthetic.SEED0 + 100 + cfg["index"])
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
    step = (lambda: eng.sweep_safeopt(cfg["b"])) if kind == "safeopt" else (lambda: eng.sweep_goose(cfg["b"]))
    ts, tw = [], []
    for it in range(24):
        eng.synchronize()
        t0 = time.perf_counter()
        eng.set_model((alt if it % 2 else cfg)["ds"], dtype="f64")
        t1 = time.perf_counter()
        step()
        t2 = time.perf_counter()
        if it >= 4:
            ts.append(t1 - t0); tw.append(t2 - t1)
    p = eng.profile()
    print(f"{name} {kind}: iteration {1e3 * (np.mean(ts) + np.mean(tw)):.3f} ms = set_model {1e3 * np.mean(ts):.3f} + sweep call {1e3 * np.mean(tw):.3f} "
          f"(device {p['total_ms']:.3f}, K1 {p['posterior_ms']:.3f}, table enqueue {p['posterior_setup_ms']:.3f}); median {1e3 * np.median(np.add(ts, tw)):.3f}", flush=True)
eng.close()
