"""Developer driver: K1b (bilinear reduced-basis posterior) against K1g (separable tables) and the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import safebo_amd
from safebo_amd import synthetic
eng = safebo_amd.SweepEngine(0)
which = sys.argv[1:] or ["B", "C", "H"]
for name in which:
    cfg = synthetic.make_config(name)
    cnt = cfg["count"]
    t = time.perf_counter(); eng.set_model(cfg["ds"], dtype="f64"); t_model = time.perf_counter() - t
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cnt)
    out = {}
    for bil in (0, 1):
        eng.set_option("bilinear", bil)
        t = time.perf_counter(); eng.posterior_run(); t_first = time.perf_counter() - t
        ms = []
        for it in range(3):
            eng.posterior_run(); ms.append(eng.profile()["posterior_ms"])
        mean, var = eng.posterior()
        r = eng.sweep_safeopt(cfg["b"]); p = eng.profile()
        out[bil] = (mean, var, r)
        print(f"{name} {cnt} n={cfg['n']} {'K1b' if bil else 'K1g'}: posterior {min(ms):.3f} ms (first call {t_first*1e3:.1f} ms, model_set {t_model*1e3:.0f} ms); "
              f"sweep {p['total_ms']:.3f} ms = {p['candidates']/p['total_ms']*1e3:.3e} cand/s; S {r['count_S']} M {r['count_M']} G {r['count_G'].tolist()} L {r['L']}", flush=True)
    ystd = cfg["ds"]["Y_std"]
    dm = np.max(np.abs(out[0][0] - out[1][0]) / np.maximum(1, ystd)); dv = np.max(np.abs(out[0][1] - out[1][1]) / np.maximum(1, ystd) ** 2)
    print(f"   K1b vs K1g: mean {dm:.2e} var {dv:.2e}; minimizer {out[0][2]['minimizer_index']} {out[1][2]['minimizer_index']}; expander {out[0][2]['expander_index']} {out[1][2]['expander_index']}")
    rng = np.random.default_rng(0)
    N = out[1][0].shape[0]
    sub = np.sort(rng.choice(N, size=2048, replace=False))
    pts = oracle.grid_points(cfg["bound"][:, 0], cfg["bound"][:, 1], cnt)[sub] if N <= 5_000_000 else None
    if pts is None:
        ax = oracle.grid_axes(cfg["bound"][:, 0], cfg["bound"][:, 1], cnt)
        pts = np.stack([ax[0][sub % cnt[0]], ax[1][sub // cnt[0]]], axis=1)
    om, ov = oracle.gp_inference(pts, cfg["ds"])
    for bil in (0, 1):
        em = np.max(np.abs(out[bil][0][sub] - om) / np.maximum(1, ystd)); ev = np.max(np.abs(out[bil][1][sub] - ov) / np.maximum(1, ystd) ** 2)
        print(f"   {'K1b' if bil else 'K1g'} vs oracle (2048 random points): mean {em:.2e} var {ev:.2e}")
