"""Developer check: SafeOpt sweep parity (masks bit-exact, indices, scalars) + timing on the GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import safebo_amd
from safebo_amd import synthetic
import oracle

eng = safebo_amd.SweepEngine(0)
def check(name, n, count, quirk=True, explicit=False, b=None):
    cfg = synthetic.make_config(name, n=n)
    if b is not None: cfg["b"] = b
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    eng.set_model(cfg["ds"], dtype="f64")
    pts = oracle.grid_points(lo, hi, count)
    if explicit: eng.set_points(pts)
    else: eng.set_grid(lo, hi, count)
    o = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"], quirk_L_index=quirk)
    try:
        r = eng.sweep_safeopt(cfg["b"], quirk_L_index=quirk, want_masks=True)
    except safebo_amd.EmptySafeSetError as e:
        print(f"{name} n={n} {count}: empty safe set; oracle says empty={o['empty_safe_set']}")
        return
    ok = True
    for w, key in (("S", "S"), ("U", "U"), ("M", "M")):
        m = eng.mask(w)
        nd = int(np.sum(m != o[key])); ok &= nd == 0
        print(f"   mask {w}: {m.sum()} set, {nd} mismatches")
    for c in range(1, cfg["q"]):
        m = eng.mask("G", c)
        nd = int(np.sum(m != o["G"][c - 1])); ok &= nd == 0
        print(f"   mask G_{c}: {m.sum()} set, {nd} mismatches; rechecks {r['n_exact_rechecks']}")
    print(f"   u* {r['u_star']:.15g} vs {o['u_star']:.15g}; L {r['L']} vs {o['L']}")
    print(f"   minimizer {r['minimizer_index']} {r['minimizer_std']:.12g} vs {o['minimizer_index']} {o['minimizer_std']:.12g}; x={r['minimizer_x']} vs {pts[o['minimizer_index']]}")
    print(f"   expander {r['expander_index_c']} {r['expander_std_c']} vs {o['expander_index']} {o['expander_std']}; best c {r['expander_best_c']} vs {o['expander_best']}; choose_min {r['choose_minimizer']} vs {o['choose_minimizer']}")
    ok &= r['minimizer_index'] == o['minimizer_index'] and np.array_equal(r['expander_index_c'], o['expander_index'])
    print(f"{name} n={n} {count} quirk={quirk} explicit={explicit}: {'OK' if ok else 'MISMATCH'}")

check("A", 20, [50, 50])
check("A", 20, [50, 50], quirk=False)
check("A", 20, [50, 50], explicit=True)
check("B", 128, [96, 80])
check("C", 64, [72, 64])
check("C", 64, [72, 64], quirk=False)
check("D", 40, [12, 11, 10, 9])
check("D", 128, [12, 11, 10, 9], b=0.5)
check("D", 128, [12, 11, 10, 9], b=1.0, quirk=False)
check("A", 12, [37, 1])

cfg = synthetic.make_config("B")
eng.set_model(cfg["ds"], dtype="f64")
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
for it in range(3):
    t = time.perf_counter(); r = eng.sweep_safeopt(cfg["b"]); dt = time.perf_counter() - t
    p = eng.profile()
    print(f"B sweep: wall {dt*1e3:.2f} ms; " + ", ".join(f"{k}={v:.3f}" for k, v in p.items() if k.endswith('_ms')), f"S={r['count_S']} U={r['count_U']} M={r['count_M']} G={r['count_G']} rechecks={r['n_exact_rechecks']}")
