"""Developer driver: K1c only on config E (scattered 6-D, n=2048, fp32) for counter collection."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = synthetic.make_config("E")
eng.set_model(cfg["ds"], dtype="f32", use_invK=False)
eng.set_points(synthetic.scattered_points(cfg, N))
best = 1e9
for it in range(reps):
    eng.posterior_run(); p = eng.profile(); best = min(best, p["posterior_ms"])
print(f"E N={N}: best {best:.2f} ms, {p['posterior_flops']/best/1e9:.1f} TFLOP/s f32 algorithmic")
