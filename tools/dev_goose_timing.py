"""Developer driver: GoOSE sweep phase timings (posterior reused) on configs C, B and a slice of D; transform vs pairs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
for name, cnt in (("A", [50, 50]), ("C", None), ("B", None), ("D", [128, 128, 64, 16])):
    cfg = synthetic.make_config(name)
    cnt = cnt or cfg["count"]
    eng.set_model(cfg["ds"], dtype="f64")
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cnt)
    eng.posterior_run()
    masks = {}
    for pairs in (1, 0):
        eng.set_option("goose_pairs", pairs)
        for it in range(3):
            try:
                r = eng.sweep_goose(cfg["b"], posterior_ready=True); p = eng.profile()
                print(name, cnt, "pairs" if pairs else "pdt  ", {k: round(v, 3) for k, v in p.items() if k.endswith("_ms")}, "S", r["count_S"], "U", r["count_U"],
                      "O", r["count_O"].tolist(), r["target_index"], r["explore_index"], "rechecks", r["n_exact_rechecks"], flush=True)
            except safebo_amd.EmptySafeSetError:
                print(name, cnt, "empty safe set")
                break
        else:
            masks[pairs] = [eng.mask("O", c) for c in range(1, cfg["q"])]
    S = eng.mask("S")
    for c in range(1, cfg["q"]):
        u = eng.bounds(cfg["b"], c, "ucb")
        Lc = r["L"][cfg["q"] - 1]
        h = (cfg["bound"][:, 1] - cfg["bound"][:, 0]) / (np.array(cnt) - 1)
        print(f"   c={c}: L={Lc:.4g} rmax={u[S].max() / Lc:.4g}  cells per axis: {np.round(u[S].max() / Lc / h, 1).tolist()}")
    if len(masks) == 2:
        print("   O masks equal:", [bool(np.array_equal(a, b)) for a, b in zip(masks[0], masks[1])], flush=True)
