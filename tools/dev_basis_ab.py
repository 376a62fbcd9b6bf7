"""Developer driver: device bases (pivoted Gram-Schmidt, k_bl_basis) against the host SVD bases: ranks, posterior difference,
error against the oracle, build time.  SBO_BL_TIMING=1 prints the ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name, n, ll, count in (("B", 128, -0.5, [160, 96]), ("B", 128, 0.5, [160, 96]), ("H", 512, -0.5, [128, 96]), ("H", 512, -1.0, [128, 96]),
                           ("C", 256, -0.5, [96, 130]), ("B", 64, 1.5, [100, 90]), ("B", 128, -0.5, [2048, 2048]), ("H", 512, -0.5, [4096, 4096])):
    cfg = synthetic.make_config(name, n=n)
    ds = synthetic.make_dataset(cfg["X"], cfg["Y"], synthetic.default_hypopt(2, cfg["q"], log_ell=ll))
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    out = {}
    for host in (1, 0):
        eng.set_option("bl_host_bases", host)
        eng.set_model(ds)
        eng.set_grid(lo, hi, count)
        print(f"--- {name} n={n} ll={ll} {count} host_bases={host}", flush=True)
        eng.posterior_run()
        p = eng.profile()
        ts = []
        for rep in range(3):
            eng.set_model(ds)
            t = time.perf_counter(); eng.posterior_run(); eng.synchronize(); ts.append(time.perf_counter() - t)
        p2 = eng.profile()
        N = count[0] * count[1]
        if N <= 1 << 16:
            out[host] = eng.posterior()
        print(f"    kernel {p['posterior_kernel']} first build {p['posterior_setup_ms']:.3f} ms, repeat posterior call {min(ts)*1e3:.3f} ms (build {p2['posterior_setup_ms']:.3f}, K1 {p2['posterior_ms']:.3f})", flush=True)
    if out:
        pts = oracle.grid_points(lo, hi, count)
        om, ov = oracle.gp_inference(pts, ds)
        ys = np.maximum(1.0, ds["Y_std"])
        for host in (1, 0):
            m, v = out[host]
            print(f"    host={host}: vs oracle mean {np.max(np.abs(m - om) / ys):.2e} var {np.max(np.abs(v - ov) / ys**2):.2e}")
        print(f"    device vs host bases: mean {np.max(np.abs(out[0][0] - out[1][0]) / ys):.2e} var {np.max(np.abs(out[0][1] - out[1][1]) / ys**2):.2e}")
eng.close()
