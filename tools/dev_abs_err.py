"""Developer driver: max ABSOLUTE and normalised |device - oracle| of the posterior per BASELINE config (sample grids / lists)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name, count, dtype in (("A", [50, 50], "f64"), ("B", [192, 160], "f64"), ("C", [160, 144], "f64"), ("H", [128, 96], "f64"),
                           ("D", [12, 11, 10, 9], "f64"), ("E", None, "f32")):
    cfg = synthetic.make_config(name)
    ds = cfg["ds"]
    eng.set_model(ds, dtype=dtype, use_invK=(dtype == "f64"))
    if count is None:
        pts = synthetic.scattered_points(cfg, 4096).astype(np.float64)
        eng.set_points(pts)
    else:
        lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
        pts = oracle.grid_points(lo, hi, count)
        eng.set_grid(lo, hi, count)
    m, v = eng.posterior()
    om, ov = oracle.gp_inference(pts, ds)
    ys = np.maximum(1.0, ds["Y_std"])
    print(f"{name} ({dtype}, kernel {eng.profile()['posterior_kernel']}, Y_std {np.round(ds['Y_std'], 3).tolist()}): "
          f"abs mean {np.max(np.abs(m - om)):.2e} var {np.max(np.abs(v - ov)):.2e} | normalised mean {np.max(np.abs(m - om) / ys):.2e} "
          f"var {np.max(np.abs(v - ov) / ys ** 2):.2e}", flush=True)
eng.close()
