"""Developer driver: cost of the fp64 recheck of fp32 sweeps at scale (config B grid in fp32, config E scattered points)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name, npts in (("B", None), ("E", 1_000_000), ("E", 4_000_000)):
    cfg = synthetic.make_config(name)
    for recheck in (1, 0):
        eng.set_option("fp64_recheck", recheck)
        t = time.perf_counter(); eng.set_model(cfg["ds"], dtype="f32", use_invK=False); tm = time.perf_counter() - t
        if npts is None:
            eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
            N = int(np.prod(cfg["count"]))
        else:
            eng.set_points(synthetic.scattered_points(cfg, npts))
            N = npts
        eng.sweep_safeopt(cfg["b"])
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r = eng.sweep_safeopt(cfg["b"]); ts.append(time.perf_counter() - t)
        p = eng.profile()
        print(f"{name} N={N} recheck={recheck}: set_model {tm*1e3:.2f} ms, sweep {min(ts)*1e3:.3f} ms (K1 {p['posterior_ms']:.3f}, recheck {p['recheck_ms']:.3f} ms, "
              f"{p['fp64_rechecks']} = {100.0*p['fp64_rechecks']/N:.2f} % re-evaluated, device total {p['total_ms']:.3f}) minimizer {r['minimizer_index']} |M| {r['count_M']} |S| {r['count_S']}", flush=True)
eng.set_option("fp64_recheck", 1)
eng.close()
