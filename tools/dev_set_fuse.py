"""Developer driver: an A/B option of the set phase (default "set_fuse") on against off -- every mask, index and count
must be identical; device times of the sweep with phase events off.
    python tools/dev_set_fuse.py [option] > gpurun_out/set_fuse.log
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic

OPT = sys.argv[1] if len(sys.argv) > 1 else "set_fuse"
eng = safebo_amd.SweepEngine(0)
bad = 0


def run(tag, ds, lo, hi, count, b, quirk=True):
    global bad
    eng.set_model(ds)
    eng.set_grid(lo, hi, count)
    q = len(ds["invKopt"])
    out = {}
    for v in (0, 1):
        eng.set_option(OPT, v)
        for _ in range(3):
            eng.sweep_safeopt(b, quirk_L_index=quirk)
        ts = []
        for _ in range(5):
            eng.sweep_safeopt(b, quirk_L_index=quirk)
            ts.append(eng.profile()["total_ms"] - eng.profile()["posterior_ms"])
        r = eng.sweep_safeopt(b, quirk_L_index=quirk, want_masks=True)
        masks = [eng.mask(k) for k in ("S", "U", "M")] + [eng.mask("G", c) for c in range(1, q)]
        out[v] = (r, masks, float(np.median(ts)))
    r0, m0, t0 = out[0]
    r1, m1, t1 = out[1]
    diff = [int((a != b_).sum()) for a, b_ in zip(m0, m1)]
    keys = ("expander_index", "minimizer_index", "count_S", "count_U", "count_M", "u_star", "expander_std", "minimizer_std")
    same = all(v == 0 for v in diff) and all(r0[k] == r1[k] for k in keys) and list(r0["count_G"]) == list(r1["count_G"]) \
        and list(r0["L"]) == list(r1["L"])
    bad += 0 if same else 1
    print(tag, "OK" if same else "MISMATCH", "diff", diff, "G", list(r0["count_G"]), "M", r0["count_M"], "set phase ms", round(t0, 4), "->",
          round(t1, 4), flush=True)


for name in os.environ.get("AB_CONFIGS", "B,H,C").split(","):
    cfg = synthetic.make_config(name)
    run(name, cfg["ds"], cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"], cfg["b"])
    run(name + "-noquirk", cfg["ds"], cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"], cfg["b"], quirk=False)

rng = np.random.default_rng(5)
for trial, count in enumerate(() if os.environ.get("AB_NORAND") else ([264, 256], [517, 301], [2044, 1999], [333, 1111], [1000, 257], [256, 4099])):
    cfg = synthetic.make_config("B" if trial % 3 else "C", n=int(rng.integers(12, 90)), seed=100 + trial)
    q = cfg["q"]
    hyp = synthetic.default_hypopt(2, q, log_ell=float(rng.uniform(-1.2, 0.3)), log_sf=float(rng.uniform(-0.5, 0.5)),
                                   log_sn=float(rng.uniform(-4, -2)))
    ds = synthetic.make_dataset(cfg["X"], cfg["Y"], hyp)
    for b in (1.0, 2.0, 3.5):
        try:
            run(f"rand{trial}-{count}-q{q}-b{b}", ds, cfg["bound"][:, 0], cfg["bound"][:, 1], count, b, quirk=bool(trial & 1))
        except safebo_amd.SafeBOError as e:
            print(f"rand{trial}-{count}-b{b}", "skipped:", str(e)[:80])
            eng.set_option(OPT, 1)
print("MISMATCHES", bad)
eng.close()
sys.exit(1 if bad else 0)
