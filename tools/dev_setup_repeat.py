"""Developer driver: steady-state cost of a model change on a resident grid (set_model + K1b table build + posterior)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name in ("B", "H"):
    cfg = synthetic.make_config(name)
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
    for it in range(5):
        t0 = time.perf_counter()
        eng.set_model(cfg["ds"], dtype="f64")
        t1 = time.perf_counter()
        eng.posterior_run()
        eng.synchronize()
        t2 = time.perf_counter()
        p = eng.profile()
        print(f"{name} #{it}: set_model {1e3 * (t1 - t0):.2f} ms, posterior incl. table build {1e3 * (t2 - t1):.2f} ms (build {p['posterior_setup_ms']:.2f}, kernels {p['posterior_ms']:.3f})", flush=True)
