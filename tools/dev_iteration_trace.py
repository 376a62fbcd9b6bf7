"""Developer driver: a few model-change iterations (set_model + table build + sweep) of config B / H for a kernel + copy trace.
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/dev_iteration_trace.py B
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safebo_amd
from safebo_amd import synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "B"
eng = safebo_amd.SweepEngine(0)
cfg = synthetic.make_config(name)
alt = synthetic.make_config(name, seed=synthetic.SEED0 + 100 + cfg["index"])
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
for it in range(8):
    t0 = time.perf_counter()
    eng.set_model((alt if it % 2 else cfg)["ds"], dtype="f64")
    t1 = time.perf_counter()
    eng.sweep_safeopt(cfg["b"])
    t2 = time.perf_counter()
    p = eng.profile()
    print(f"{name} #{it}: set_model {1e3 * (t1 - t0):.3f} ms, sweep call {1e3 * (t2 - t1):.3f} ms (table build {p['posterior_setup_ms']:.3f}, device {p['total_ms']:.3f})", flush=True)
eng.close()
