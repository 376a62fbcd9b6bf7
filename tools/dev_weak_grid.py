"""Developer driver: the weak-scaling grids of bench.py (slowest axis x N) swept by ONE rank -- per-phase times show how the
set kernels behave when the grid gets finer along one axis only (per-rank cost at N ranks ~ 1/N of these, plus halo)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
cfg = synthetic.make_config("B")
eng.set_model(cfg["ds"], dtype="f64")
for mult in (1, 2, 4, 8):
    cnt = [2048, 2048 * mult]
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cnt)
    eng.posterior_run()
    keep = {}
    for blocks in (0, 1):
        eng.set_option("scan_blocks", blocks)
        for it in range(3):
            r = eng.sweep_safeopt(cfg["b"], posterior_ready=True); p = eng.profile()
        G = eng.mask("G", 1)
        print(cnt, "blocked" if blocks else "stepwise", {k: round(v, 3) for k, v in p.items() if k.endswith("_ms")}, "S", r["count_S"], "G", r["count_G"].tolist(), "rechecks", r["n_exact_rechecks"], flush=True)
        for it in range(3):
            r = eng.sweep_goose(cfg["b"], posterior_ready=True); p = eng.profile()
        print("   goose", {k: round(v, 3) for k, v in p.items() if k.endswith("_ms")}, "O", r["count_O"].tolist(), flush=True)
        keep[blocks] = (G, eng.mask("O", 1))
    print("   G equal", bool(np.array_equal(keep[0][0], keep[1][0])), "O equal", bool(np.array_equal(keep[0][1], keep[1][1])), flush=True)
