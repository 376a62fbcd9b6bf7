"""Developer driver: K1t (Chebyshev-node interpolation on 3-D / 4-D grids) against K1g -- values on a 64^4 grid, sweep results and
times on config D (128^4):   SBO_DEBUG_TENSOR=1 python tools/dev_tensor.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
cfg = synthetic.make_config("D")
lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
eng.set_model(cfg["ds"], dtype="f64")
only = os.environ.get('DEV_TENSOR_ONLY')
for count in (([128] * 4,) if only == '128' else ([64] * 4, [128] * 4)):
    out = {}
    l, h = lo[:len(count)], hi[:len(count)]
    for opt in ((1,) if only else (0, 1)):
        eng.set_option("tensor_cheb", opt)
        eng.set_grid(l, h, count)
        t0 = time.perf_counter()
        res = eng.sweep_safeopt(cfg["b"])
        t1 = time.perf_counter()
        res = eng.sweep_safeopt(cfg["b"])
        t2 = time.perf_counter()
        p = eng.profile()
        print(f"count {count} tensor_cheb={opt}: kernel {p['posterior_kernel']} first call {1e3*(t1-t0):.2f} ms, second {1e3*(t2-t1):.2f} ms, "
              f"device {p['total_ms']:.2f} (K1 {p['posterior_ms']:.2f})", flush=True)
        keep = {k: res[k] for k in ("minimizer_index", "minimizer_std", "expander_index_c", "expander_std_c", "L", "u_star", "count_S", "count_U", "count_M", "count_G", "n_exact_rechecks") if k in res}
        if np.prod(count) <= 1 << 24:
            keep["mean"], keep["var"] = eng.posterior()
        out[opt] = keep
    for k in (out[0] if 0 in out else ()):
        a, b = np.asarray(out[0][k], dtype=np.float64), np.asarray(out[1][k], dtype=np.float64)
        print(f"   {k}: max |diff| {np.max(np.abs(a - b)):.3e}" + ("" if a.size > 8 else f"   {a.ravel()} vs {b.ravel()}"), flush=True)
eng.close()
