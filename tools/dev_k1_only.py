"""Developer driver: K1 only, config B (n=128) and n=512 on 2048^2, for counter collection."""
import sys
sys.path.insert(0, "/root/repo" if "/root/repo" not in sys.path else ".")
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if len(sys.argv) > 2: eng.set_option("k1_wgs_per_cu", int(sys.argv[2]))
if len(sys.argv) > 3: eng.set_option("k1_strips", int(sys.argv[3]))
for name, n in (("B", 128), ("H", 512)):
    cfg = synthetic.make_config(name, n=n)
    eng.set_model(cfg["ds"], dtype="f64")
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], [2048, 2048])
    best = 1e9
    for it in range(reps):
        eng.posterior_run()
        p = eng.profile()
        best = min(best, p["posterior_ms"])
    print(f"n={n}: best {best:.3f} ms, {p['posterior_flops']/best/1e9:.2f} TFLOP/s algorithmic")
