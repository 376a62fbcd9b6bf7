"""Developer driver: config E (scattered 6-D points, n=2048, fp32) and config C/D timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
cfg = synthetic.make_config("E")
t = time.perf_counter(); eng.set_model(cfg["ds"], dtype="f32", use_invK=False); print("model_set E: %.2f s" % (time.perf_counter() - t))
pts = synthetic.scattered_points(cfg, N)
eng.set_points(pts)
for it in range(3):
    eng.posterior_run(); p = eng.profile()
    print(f"E N={N}: K1 {p['posterior_ms']:.2f} ms, {N/p['posterior_ms']*1e3:.3e} pts/s, {p['posterior_flops']/p['posterior_ms']/1e9:.1f} TFLOP/s (f32 algorithmic)")
r = eng.sweep_safeopt(3.0); p = eng.profile()
print("E sweep:", {k: round(v, 3) for k, v in p.items() if k.endswith("_ms")}, r["count_M"], r["minimizer_index"])
for name in ("C", "D"):
    cfg = synthetic.make_config(name)
    cnt = cfg["count"] if name == "C" else [128, 128, 64, 16]
    eng.set_model(cfg["ds"], dtype="f64")
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cnt)
    for it in range(2):
        try:
            r = eng.sweep_safeopt(cfg["b"]); p = eng.profile()
            print(name, cnt, {k: round(v, 3) for k, v in p.items() if k.endswith("_ms")}, f"{p['candidates']/p['total_ms']*1e3:.3e} pts/s, K1 {p['posterior_flops']/p['posterior_ms']/1e9:.1f} TF", r["count_S"], r["count_G"], r["n_exact_rechecks"])
        except safebo_amd.EmptySafeSetError as e:
            p = eng.profile(); print(name, cnt, "empty safe set")
    if name == "C":
        for it in range(2):
            r = eng.sweep_goose(cfg["b"]); p = eng.profile()
            print("C goose", {k: round(v, 3) for k, v in p.items() if k.endswith("_ms")}, r["count_O"], r["target_index"], r["explore_index"])
