"""Developer driver: config D at full size (128^4 = 268 M candidates, n = 128) on one GPU -- large-N path check."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
import oracle
eng = safebo_amd.SweepEngine(0)
cfg = synthetic.make_config("D")
eng.set_model(cfg["ds"], dtype="f64")
cnt = [128, 128, 128, 128]
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cnt)
for it in range(2):
    t = time.perf_counter(); r = eng.sweep_safeopt(0.5); dt = time.perf_counter() - t
    p = eng.profile()
    print(f"D 128^4: wall {dt*1e3:.1f} ms", {k: round(v, 2) for k, v in p.items() if k.endswith("_ms")}, f"{p['candidates']/p['total_ms']*1e3:.3e} pts/s; K1 {p['posterior_flops']/p['posterior_ms']/1e9:.1f} TF")
    print("   S", r["count_S"], "U", r["count_U"], "M", r["count_M"], "G", r["count_G"], "minimizer", r["minimizer_index"], r["minimizer_x"], "expander", r["expander_index_c"], "rechecks", r["n_exact_rechecks"])
# spot-check the posterior and the masks on random candidates against the oracle
rng = np.random.default_rng(0)
idx = np.sort(rng.choice(128 ** 4, size=2000, replace=False))
pts = np.stack([cfg["bound"][a, 0] + ((idx // 128 ** a) % 128) * ((cfg["bound"][a, 1] - cfg["bound"][a, 0]) / 127) for a in range(4)], axis=1)
for a in range(4):
    pts[(idx // 128 ** a) % 128 == 127, a] = cfg["bound"][a, 1]
om, ov = oracle.gp_inference(pts, cfg["ds"])
lcb = eng.bounds(0.5, 1, "lcb")[idx]
ol, ou = oracle.bounds(om, ov, 0.5)
S = eng.mask("S")[idx]
print("spot check: max |lcb_1 - oracle| / Y_std =", np.max(np.abs(lcb - ol[:, 1])) / cfg["ds"]["Y_std"][1], " S mismatches:", int(np.sum(S != (ol[:, 1] >= 0))))
