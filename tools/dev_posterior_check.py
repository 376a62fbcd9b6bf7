"""Developer check (not part of the test-suite): posterior parity + timing on the GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import safebo_amd
from safebo_amd import synthetic
import oracle

eng = safebo_amd.SweepEngine(0)
for name, n, count, dtype in [("A", 20, [50, 50], "f64"), ("B", 128, [96, 64], "f64"), ("B", 100, [70, 33], "f64"),
                              ("H", 512, [64, 32], "f64"), ("C", 256, [64, 48], "f64"), ("D", 128, [9, 8, 7, 6], "f64"),
                              ("B", 128, [96, 64], "f32")]:
    cfg = synthetic.make_config(name, n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    for use_invK in (True, False):
        eng.set_model(cfg["ds"], dtype=dtype, use_invK=use_invK)
        eng.set_grid(lo, hi, count)
        mean, var = eng.posterior()
        pts = oracle.grid_points(lo, hi, count)
        om, ov = oracle.gp_inference(pts, cfg["ds"], dtype=np.float64)
        print(f"{name} n={n} grid={count} {dtype} invK={use_invK}: max|dmean|={np.max(np.abs(mean-om)):.3e} max|dvar|={np.max(np.abs(var-ov)):.3e}"
              f"  (|mean|max={np.max(np.abs(om)):.2f}, var max={ov.max():.3f})")
    # explicit points path
    eng.set_model(cfg["ds"], dtype=dtype)
    eng.set_points(pts)
    m2, v2 = eng.posterior()
    print("   explicit-points vs grid: ", np.max(np.abs(m2 - mean)), np.max(np.abs(v2 - var)))
    lcb = eng.bounds(cfg["b"], 1 if cfg["q"] > 1 else 0, "lcb")
    ol, ou = oracle.bounds(om, ov, cfg["b"])
    print("   lcb diff", np.max(np.abs(lcb - ol[:, 1 if cfg['q'] > 1 else 0])))

# timing at config B
cfg = synthetic.make_config("B")
eng.set_model(cfg["ds"], dtype="f64")
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
for it in range(3):
    eng.posterior_run()
    p = eng.profile()
    print(f"B posterior: {p['posterior_ms']:.3f} ms, {p['candidates']/p['posterior_ms']*1e3:.3e} pts/s, {p['posterior_flops']/p['posterior_ms']/1e9:.2f} TFLOP/s (algorithmic)")
cfg = synthetic.make_config("H")
eng.set_model(cfg["ds"], dtype="f64")
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], [2048, 2048])
for it in range(2):
    eng.posterior_run()
    p = eng.profile()
    print(f"H(n=512, 2048^2) posterior: {p['posterior_ms']:.3f} ms, {p['candidates']/p['posterior_ms']*1e3:.3e} pts/s, {p['posterior_flops']/p['posterior_ms']/1e9:.2f} TFLOP/s (algorithmic)")
